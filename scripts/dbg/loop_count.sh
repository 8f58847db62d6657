#!/bin/bash
# Static instruction mix of the raster kernel's evaluation loop (compile only, no GPU):
#   bash scripts/dbg/loop_count.sh [extra -D flags]   -> VALU / LDS / VMEM counts of the main path (clipped-pair path compiled out)
cd "$(dirname "$0")/../.."
mkdir -p build/r3 && cd build/r3
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I../../include -I../../occlusionenv_amd/csrc -fno-slp-vectorize -DOCC_DBG2_NO_PAIR "$@" \
  -save-temps -o /tmp/x_lc.so ../../occlusionenv_amd/csrc/occ_kernels.hip 2>/dev/null
grep -A12 "\.name:.*occ_raster2_kernelILb1ELb1ELb1" occ_kernels-hip-amdgcn-amd-amdhsa-gfx950.s | grep -E "spill|vgpr_count|group_seg" | tr '\n' ' '; echo
grep -B8 "\.name:.*occ_raster2_kernelILb1ELb1ELb1" occ_kernels-hip-amdgcn-amd-amdhsa-gfx950.s | grep group_segment
awk '/^_ZN3occ18occ_raster2_kernelILb1ELb1ELb1EEEvNS_12RasterParamsE:/{p=1} p{print} /s_endpgm/{if(p){exit}}' occ_kernels-hip-amdgcn-amd-amdhsa-gfx950.s > r2lc.s
python3 - r2lc.s <<'PY'
import sys,re
lines=open(sys.argv[1]).read().split('\n')
hdr=[i for i,l in enumerate(lines) if re.search(r'This Loop Header: Depth=3',l)]
exps=[i for i,l in enumerate(lines) if 'v_exp_f32' in l]
h=next(h for h in hdr if any(h<e<h+450 for e in exps))
k=h
while not lines[k].startswith('.LBB'): k-=1
lab=lines[k].split(':')[0][1:]          # e.g. LBB22_519 -> "BB22_519" appears in the comments of its blocks
lab=lab[1:] if lab.startswith('L') else lab
from collections import Counter
c=Counter(); inloop=True; last=k
QUARTER=('v_rcp','v_exp','v_log','v_sqrt','v_rsq','v_mul_lo_u32','v_mul_hi','v_mad_u64','v_mad_i64','v_sin','v_cos')
j=k
while j < len(lines):
    l=lines[j]
    if (l.startswith('.LBB') or l.lstrip().startswith('; %bb.')) and j>k:
        ctx=' '.join(lines[j:j+4])
        inloop = lab in ctx
        if not inloop and j > h+100: break
    if inloop:
        t=l.strip().split()
        if t and not t[0].startswith(('.',';')) and not t[0].endswith(':'):
            op=t[0]
            kk='valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_','scratch_','buffer_')) else 'other'
            c[kk]+=1
            if op.startswith(QUARTER): c['quarter_rate']+=1
            if op.startswith('v_cndmask'): c['cndmask']+=1
            if op.startswith('v_mov'): c['v_mov']+=1
        last=j
    j+=1
print('loop lines',k,last,dict(c),'valu cycles ~',4*c['valu']+12*c['quarter_rate'])
PY

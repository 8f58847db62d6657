// occ_setup.hpp -- setup + sort kernels: projection, z-clipping, culling, face records, scan rows.
// Part of the single translation unit occ_kernels.hip (included inside namespace occ; not a stand-alone header).

// ------------------------------------------------------------------------------------------
// setup: projection, z-clipping, culling, record build with ordered compaction
// ------------------------------------------------------------------------------------------
struct PVert {      // one projected vertex
    float x, y, z;  // x_ndc, y_ndc, z_view
    float t[4];     // d x/d el, d y/d el, d x/d az, d y/d az
};

struct VVert {  // view-space vertex with tangents
    float v[3];
    float de[3], da[3];
};

// NB a surviving unclipped face is projected twice: before the block-wide compaction barrier, where finish_tri()
// takes the ONE visibility decision and derives the conservative pixel bbox (both cross the barrier: the count in a
// register, the bbox through LDS), and after it, for the record's DATA only (the world corners and the three 1 / z_view,
// not the projections, are what a face keeps in registers).  The two inlined copies may round differently in the last
// bit (FMA contraction); nothing is decided from the second one, and the bbox carries 1e-3 px of slack.
// DIVISIONS (round 4): an IEEE f32 division is ten instructions, one of them quarter-rate, and a 512-face round used to
// hold seventeen: 1 / z per corner three times over (culling, record, tangents), four pixel centres / S in finish_tri,
// 1 / area and 3 x 1 / |edge|^2 of the record.  Now seven (1 / z once per corner, carried along; the centres multiply
// by 1 / S, computed once per block): step -24 us on the bench, records bit-identical (scripts/dbg/ab_outputs.py).
__device__ __forceinline__ void project_pos(const VVert& q, PVert& p, float& iz) {
    // [P3D] x_ndc = x_view * s / z_view  
    iz = 1.0f / q.v[2];
    p.x = q.v[0] * kProjScale * iz;
    p.y = q.v[1] * kProjScale * iz;
    p.z = q.v[2];
}
// (iz_in: 1 / z_view of this vertex if the caller has it already - an IEEE division costs ten instructions, one of them
// quarter-rate, and the setup kernel used to evaluate it three times per corner; nullptr = divide here)
template <bool GRAD>
__device__ __forceinline__ PVert project(const VVert& q, const float* iz_in = nullptr, float* iz_out = nullptr) {
    PVert p;
    float iz;
    if (iz_in) {
        iz = *iz_in;
        p.x = q.v[0] * kProjScale * iz;
        p.y = q.v[1] * kProjScale * iz;
        p.z = q.v[2];
    } else {
        project_pos(q, p, iz);
    }
    if (iz_out) *iz_out = iz;
    if (GRAD) {
        p.t[0] = (kProjScale * q.de[0] - p.x * q.de[2]) * iz;
        p.t[1] = (kProjScale * q.de[1] - p.y * q.de[2]) * iz;
        p.t[2] = (kProjScale * q.da[0] - p.x * q.da[2]) * iz;
        p.t[3] = (kProjScale * q.da[1] - p.y * q.da[2]) * iz;
    } else {
        p.t[0] = p.t[1] = p.t[2] = p.t[3] = 0.f;
    }
    return p;
}

// [P3D] clip_faces: intersection of edge (a -> b) with z = kZClip, weight detached
template <bool GRAD>
__device__ __forceinline__ PVert cut_edge(const VVert& a, const VVert& b) {
    PVert p;
    const float w = (a.v[2] - kZClip) / (a.v[2] - b.v[2]);
    const float iw = 1.0f - w;
    const float ic = 1.0f / kZClip;
    p.z = a.v[2] * iw + b.v[2] * w;
    p.x = (kProjScale * a.v[0] * iw + kProjScale * b.v[0] * w) * ic;
    p.y = (kProjScale * a.v[1] * iw + kProjScale * b.v[1] * w) * ic;
    if (GRAD) {
        p.t[0] = (kProjScale * a.de[0] * iw + kProjScale * b.de[0] * w) * ic;
        p.t[1] = (kProjScale * a.de[1] * iw + kProjScale * b.de[1] * w) * ic;
        p.t[2] = (kProjScale * a.da[0] * iw + kProjScale * b.da[0] * w) * ic;
        p.t[3] = (kProjScale * a.da[1] * iw + kProjScale * b.da[1] * w) * ic;
    } else {
        p.t[0] = p.t[1] = p.t[2] = p.t[3] = 0.f;
    }
    return p;
}

struct Tri {
    PVert v[3];
    uint4 bbox;  // conservative pixel bbox x = xl | yl << 16, y = xh | yh << 16 | corner-cut bits << 28 (finish_tri; pixel
                 // coordinates are < 2048); z = key of the smallest vertex depth
    int tx0, ty0, tx1, ty1;
};

__device__ __forceinline__ float tri_lim(int S) { return 1.0f - 1.0f / (float)S; }  // outermost pixel centre

// Returns false if the triangle can never be matched to a pixel (culled / degenerate / off screen).  EVERY field
// is filled with in-range values either way (clamped pixel bbox), whatever the verdict.  occ_setup_kernel calls this
// ONCE per unclipped face, before the compaction barrier; the verdict and the pixel bbox cross the barrier (count
// in a register, bbox through LDS) and are not derived again.
__device__ __forceinline__ bool finish_tri(Tri& t, int S, float lim, float ifS) {  // lim = tri_lim(S), ifS = 1 / S
    const float x0 = t.v[0].x, y0 = t.v[0].y, x1 = t.v[1].x, y1 = t.v[1].y, x2 = t.v[2].x, y2 = t.v[2].y;
    // [P3D] face_area = EdgeFunction(v0; v1, v2); back faces are culled (environment.py:253,271)
    const float area = (x0 - x1) * (y2 - y1) - (y0 - y1) * (x2 - x1);
    bool vis = area > kEpsilon;  // false for a back face, zero area or NaN
    vis = vis && !(fmax3(t.v[0].z, t.v[1].z, t.v[2].z) < 0.0f);
    const float bx0 = fmin3(x0, x1, x2) - kSqrtBlur, bx1 = fmax3(x0, x1, x2) + kSqrtBlur;
    const float by0 = fmin3(y0, y1, y2) - kSqrtBlur, by1 = fmax3(y0, y1, y2) + kSqrtBlur;
    vis = vis && !(bx1 < -lim || bx0 > lim || by1 < -lim || by0 > lim);
    // pixel index of an NDC coordinate: u(f) = (S-1) - ((f+1)*S - 1)/2   (decreasing)
    const float fS = (float)S;
    auto u = [&](float f) { return (fS - 1.0f) - ((f + 1.0f) * fS - 1.0f) * 0.5f; };
    // pixels whose centre can pass the exact float test bx0 <= xf <= bx1 (u is decreasing); 1e-3 px of slack
    // covers the rounding of u() - the per-pixel float test in eval_face stays the authority
    int xl = (int)ceilf(u(bx1) - 1e-3f), xh = (int)floorf(u(bx0) + 1e-3f);
    int yl = (int)ceilf(u(by1) - 1e-3f), yh = (int)floorf(u(by0) + 1e-3f);
    vis = vis && (max(xl, 0) <= min(xh, S - 1)) && (max(yl, 0) <= min(yh, S - 1));
    xl = min(max(xl, 0), S - 1);
    yl = min(max(yl, 0), S - 1);
    xh = min(max(xh, xl), S - 1);
    yh = min(max(yh, yl), S - 1);
    t.tx0 = xl / OCC_BLOCK;
    t.tx1 = xh / OCC_BLOCK;
    t.ty0 = yl / OCC_BLOCK;
    t.ty1 = yh / OCC_BLOCK;
    // CORNER CUT: the pixel box of a small face is the face's own box grown by sqrt(blur) on every side - a rounded
    // square, of which the blur disc leaves corner pixels out.  A corner pixel whose centre is farther than sqrt(blur)
    // from the face's OWN box is farther than that from the triangle inside it: it can neither be inside nor within
    // the blur radius (relative margin 1e-3 on the squared distance against the rounding of either side).  Bits 28..31
    // of bbox.y: corner (xl, yl), (xh, yl), (xl, yh), (xh, yh) can be skipped; occ_raster2_kernel leaves those (face,
    // pixel) pairs out of its rounds (6 % of the rounds of the bench's sub-pixel faces; the exact point-triangle
    // distance per corner finds 2 % more and costs the setup kernel 0.6 ms: measured, not kept).
    uint32_t corner = 0u;
#ifndef OCC_NO_CORNER_CUT  // (the A/B build of tests/test_gpu_parity.py: results must not change by a bit)
    {
        // pixel centre (the raster kernel divides by S; 1 / S is exact for a power of two, and the 1e-3 margin below covers an ulp)
        auto ctr = [&](int i) { return -1.0f + (2.0f * (float)(S - 1 - i) + 1.0f) * ifS; };
        auto gap = [](float lo, float hi, float c) { return fmaxf(fmaxf(lo - c, c - hi), 0.0f); };
        const float fx0 = fmin3(x0, x1, x2), fx1 = fmax3(x0, x1, x2), fy0 = fmin3(y0, y1, y2), fy1 = fmax3(y0, y1, y2);
        const float gxl = gap(fx0, fx1, ctr(xl)), gxh = gap(fx0, fx1, ctr(xh));
        const float gyl = gap(fy0, fy1, ctr(yl)), gyh = gap(fy0, fy1, ctr(yh));
        const float lim2 = kBlurRadius * 1.001f;
        corner = (gxl * gxl + gyl * gyl > lim2 ? 1u : 0u) | (gxh * gxh + gyl * gyl > lim2 ? 2u : 0u) |
                 (gxl * gxl + gyh * gyh > lim2 ? 4u : 0u) | (gxh * gxh + gyh * gyh > lim2 ? 8u : 0u);
    }
#endif
    const uint32_t zb = __float_as_uint(fmin3(t.v[0].z, t.v[1].z, t.v[2].z));
    t.bbox = make_uint4((uint32_t)xl | ((uint32_t)yl << 16), (uint32_t)xh | ((uint32_t)yh << 16) | (corner << 28),
                        (zb & 0x80000000u) ? ~zb : (zb | 0x80000000u), 0u);
    return vis;
}

// [P3D] HardFlatShader terms of ONE face: flat shading uses the face normal and the face centre only,
// so (ambient + diffuse) and the specular term are per-face constants of the current camera.  w0..w2 = the
// ORIGINAL face's world-space corners (also for z-clipped pieces), cpos = camera centre.  Computed once per
// visible face by the setup kernel; the combine kernel then shades a pixel with one gather.
struct Shade {
    float amb_diff, spec;
};
__device__ __forceinline__ Shade flat_shade(const float* w0, const float* w1, const float* w2, float cx, float cy, float cz) {
    // hardware sqrt / rcp (1 ulp) instead of the IEEE sequences: ~1e-7 relative on a colour in [0.5, 1]
    auto inv_len = [](float x, float y, float z) { return frcp(fmaxf(__builtin_amdgcn_sqrtf(x * x + y * y + z * z), kShadeEps)); };
    const float ax = w1[0] - w0[0], ay = w1[1] - w0[1], az = w1[2] - w0[2];
    const float bx = w2[0] - w0[0], by = w2[1] - w0[1], bz = w2[2] - w0[2];
    float nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
    float in_ = inv_len(nx, ny, nz);
    nx *= in_; ny *= in_; nz *= in_;
    in_ = inv_len(nx, ny, nz);  // F.normalize again in diffuse()/specular()
    nx *= in_; ny *= in_; nz *= in_;
    const float third = 1.0f / 3.0f;
    const float ccx = (w0[0] + w1[0] + w2[0]) * third, ccy = (w0[1] + w1[1] + w2[1]) * third,
                ccz = (w0[2] + w1[2] + w2[2]) * third;
    float lx = kLightX - ccx, ly = kLightY - ccy, lz = kLightZ - ccz;
    const float il = inv_len(lx, ly, lz);
    lx *= il; ly *= il; lz *= il;
    const float cosang = nx * lx + ny * ly + nz * lz;
    const float diffuse = kDiffuse * fmaxf(cosang, 0.f);
    float vx = cx - ccx, vy = cy - ccy, vz = cz - ccz;
    const float iv = inv_len(vx, vy, vz);
    vx *= iv; vy *= iv; vz *= iv;
    const float rx = -lx + 2.f * (cosang * nx), ry = -ly + 2.f * (cosang * ny), rz = -lz + 2.f * (cosang * nz);
    float sa = fmaxf(vx * rx + vy * ry + vz * rz, 0.f) * (cosang > 0.f ? 1.f : 0.f);
    sa *= sa; sa *= sa; sa *= sa; sa *= sa; sa *= sa; sa *= sa;  // ^64
    Shade sh;
    sh.amb_diff = kAmbient + diffuse;
    sh.spec = kSpecular * sa;
    return sh;
}

template <bool GRAD>
__device__ __forceinline__ void write_record(float* __restrict__ r, uint4* __restrict__ scan_row,
                                             int pos, const Tri& t, int face_id, int flags, Shade sh) {
    const float x0 = t.v[0].x, y0 = t.v[0].y, x1 = t.v[1].x, y1 = t.v[1].y, x2 = t.v[2].x, y2 = t.v[2].y;
    // [P3D] BarycentricCoordsForward: area = EdgeFunction(v2; v0, v1) + kEpsilon
    const float area = (x2 - x0) * (y1 - y0) - (y2 - y0) * (x1 - x0) + kEpsilon;
    const float l01 = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
    const float l02 = (x2 - x0) * (x2 - x0) + (y2 - y0) * (y2 - y0);
    const float l12 = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1);
    float4* r4 = reinterpret_cast<float4*>(r);
    // slot map: occ_constants.h (R_X0 .. R_TAN)
    r4[0] = make_float4(x0, y0, t.v[0].z, x1);
    r4[1] = make_float4(y1, t.v[1].z, x2, y2);
    r4[2] = make_float4(t.v[2].z, __int_as_float(face_id), __int_as_float(flags), 1.0f / area);
    // part 3: the face's ambient + diffuse shading term.  (Until round 4 this part held [P3D]'s float bbox +- sqrt(blur),
    // which no kernel read any more, and the term travelled in a 16-byte row of its own per record - rec_bbox, a tenth of
    // the bytes the kernel writes.)
    r4[3] = make_float4(sh.amb_diff, 0.f, 0.f, 0.f);
    r4[4] = make_float4(l01 <= kEpsilon ? -1.0f : 1.0f / l01, l02 <= kEpsilon ? -1.0f : 1.0f / l02,
                        l12 <= kEpsilon ? -1.0f : 1.0f / l12, sh.spec);
    if (GRAD) {
        r4[5] = make_float4(t.v[0].t[0], t.v[0].t[1], t.v[0].t[2], t.v[0].t[3]);
        r4[6] = make_float4(t.v[1].t[0], t.v[1].t[1], t.v[1].t[2], t.v[1].t[3]);
        r4[7] = make_float4(t.v[2].t[0], t.v[2].t[1], t.v[2].t[2], t.v[2].t[3]);
    }
    // scan row in face order (occ_sort_kernel re-orders dense objects): (pixel bbox, nearest depth key, record index)
    *scan_row = make_uint4(t.bbox.x, t.bbox.y, t.bbox.z, (uint32_t)pos);
}

// Parts 0..4 of a record (positions, id, flags, 1 / area, float bbox, 1 / |edge|^2 x 3, specular term) and its bbox and
// scan rows: write_record without the tangents.  The setup kernel stages a record in two pieces - these five parts,
// then the three tangent parts - through a five-part LDS slot (see there).
__device__ __forceinline__ void write_record_lo(float4* __restrict__ r4, uint4* __restrict__ scan_row,
                                                int pos, const Tri& t, int face_id, int flags, Shade sh) {
    write_record<false>(reinterpret_cast<float*>(r4), scan_row, pos, t, face_id, flags, sh);
}

// union pixel bbox and smallest depth key of every 64-entry chunk of the scan order (two-level scan)
__device__ __forceinline__ void chunk_boxes(const uint4* __restrict__ scan, uint4* __restrict__ cbx, int nr, int wave,
                                            int lane, int nwaves = 4) {
    const int nch = (nr + 63) >> 6;
    for (int c = wave; c < nch; c += nwaves) {
        const int j = c * 64 + lane;
        uint4 bb = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
        if (j < nr) bb = scan[j];
        int xl = bb.x & 0xFFFF, yl = bb.x >> 16, xh = bb.y & 0xFFFF, yh = (bb.y >> 16) & 0x0FFF;  // (corner-cut bits above)
        uint32_t zk = bb.z;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            xl = min(xl, __shfl_xor(xl, m, 64));
            yl = min(yl, __shfl_xor(yl, m, 64));
            xh = max(xh, __shfl_xor(xh, m, 64));
            yh = max(yh, __shfl_xor(yh, m, 64));
            zk = min(zk, (uint32_t)__shfl_xor((int)zk, m, 64));
        }
        if (lane == 0)
            cbx[c] = make_uint4((uint32_t)xl | ((uint32_t)yl << 16), (uint32_t)xh | ((uint32_t)yh << 16), zk, 0u);
    }
}

// Record span of (env, object) eo: fixed stride rec_cap, or the variable layout of OccWorkspace.rec_off
struct RecSpan {
    size_t base;  // first record
    int cap;      // records reserved
    size_t cbox;  // first chunk box
};
template <class WS>  // (OccWorkspace in any address space)
__device__ __forceinline__ RecSpan rec_span(const WS& ws, int rec_cap, int eo) {
    RecSpan r;
    if (ws.rec_off) {
        const long long b = ws.rec_off[eo];
        r.base = (size_t)b;
        r.cap = (int)(ws.rec_off[eo + 1] - b);
        r.cbox = (size_t)(b >> 6);  // spans are multiples of 64 records
    } else {
        r.base = (size_t)eo * rec_cap;
        r.cap = rec_cap;
        r.cbox = (size_t)eo * ((rec_cap + 63) >> 6);
    }
    return r;
}

// Variable record layout: every (env, object) gets room for 2 x the faces of ITS mesh (a z-clipped face can split in
// two), rounded up to 64; skipped scene rows get none.  One block: prefix sum over 3*n_env entries.
// Objects that no longer fit into rec_total get an empty span and raise OCC_STATUS_REC_OVERFLOW.
// The launch's prologue as well: the eight work-queue heads and the header of the work-item order are zeroed here
// (two memset launches less in a sequence of ~25 dependent small launches at ~5 us each).
// A STEP launch (occ_step) folds the camera update in: blocks 1.. take 1024 envs each (OccCameraArgs; cam_args.n = 0: none).
__global__ __launch_bounds__(1024) void occ_recoff_kernel(OccScene sc, long long* __restrict__ rec_off, long long rec_total,
                                                          int* __restrict__ status, uint32_t* __restrict__ queue,
                                                          uint32_t* __restrict__ order_hdr, OccCameraArgs cam_args,
                                                          float* __restrict__ cam) {
    __shared__ long long s_part[16];
    __shared__ long long s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x > 0) {
        const int n = (int)(blockIdx.x - 1) * 1024 + tid;
        if (n < cam_args.n)
            camera_one(cam_args.mode, cam_args.action, cam_args.el, cam_args.az, cam_args.radius, cam, cam_args.cam_pos_out,
                       cam_args.cam_pos_out2, n);
        return;
    }
    if (tid < 8) queue[tid * kQueueStride] = 0u;
    if (order_hdr && tid < kOrdBlk) order_hdr[tid] = 0u;
    if (!rec_off) return;  // fixed record layout: nothing to lay out
    const int M = 3 * sc.n_env;
    // ONE round (round 5; until then ceil(M / 1024) dependent rounds of scan + two barriers each): every thread takes
    // `per` CONSECUTIVE entries, the block scans the threads' totals once, the entries' offsets follow from the thread's.
    const int per = (M + 1023) / 1024;
    const int i0 = tid * per;
    long long tsum = 0;
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i < M && !(sc.skip && sc.skip[i / 3])) {
            const int mesh = sc.scene_mesh[i];
            const int nF = sc.mesh_face_off[mesh + 1] - sc.mesh_face_off[mesh];
            tsum += ((2ll * nF + 63) >> 6) << 6;
        }
    }
    long long incl = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    long long woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const long long v = s_part[w];
        if (w < wave) woff += v;
        tot += v;
    }
    long long off = woff + incl - tsum;  // offset of this thread's first entry
    for (int k = 0; k < per; ++k) {
        const int i = i0 + k;
        if (i >= M) break;
        long long c = 0;
        if (!(sc.skip && sc.skip[i / 3])) {
            const int mesh = sc.scene_mesh[i];
            const int nF = sc.mesh_face_off[mesh + 1] - sc.mesh_face_off[mesh];
            c = ((2ll * nF + 63) >> 6) << 6;
        }
        if (off + c > rec_total) {  // does not fit: empty span at the end of the buffer
            rec_off[i] = rec_total;
            if (c) atomicOr(&status[i / 3], OCC_STATUS_REC_OVERFLOW);
        } else {
            rec_off[i] = off;
        }
        off += c;
    }
    if (tid == 0) rec_off[M] = min(tot, rec_total);
    (void)s_carry;
}

// Objects with many visible faces (>= kSortMin records: a pixel then collects far more than K candidates) get
// their scan order sorted front to back - bitonic sort of (depth key, record index) in LDS - so that the raster
// kernel reaches "every pixel of the block holds its K nearest" after the nearest faces and skips the rest.
#ifndef OCC_SORT_MIN
#define OCC_SORT_MIN 4096
#endif
constexpr int kSortMin = OCC_SORT_MIN;
constexpr int kSortCap = 8192;  // records of one object that the sort kernel's LDS holds (64 KiB of keys)
// an object whose scan order occ_sort_kernel has re-sorted: its sorted rows are in OccWorkspace.rec_bbox
__host__ __device__ __forceinline__ bool scan_is_sorted(int nrec) { return nrec >= kSortMin && nrec <= kSortCap; }
// (256 threads of one block; every thread of the block returns together before the first barrier when there is nothing to sort)
__device__ __forceinline__ void sort_object(const OccScene& sc, const OccWorkspace& ws, int sort_cap, int eo) {
    extern __shared__ unsigned long long s_keys[];  // sort_cap keys: depth key << 32 | record index
    const int nr = ws.nrec[eo];
    if (!scan_is_sorted(nr) || sort_cap < kSortCap) return;
    int p2 = 1;
    while (p2 < nr) p2 <<= 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RecSpan span = rec_span(ws, sc.rec_cap, eo);
    // the rows in face order (written by the setup kernel) -> the same rows front to back, in the object's span of rec_bbox
    const uint4* __restrict__ rows = reinterpret_cast<const uint4*>(ws.scan) + span.base;
    uint4* __restrict__ sorted = reinterpret_cast<uint4*>(ws.rec_bbox) + span.base;
    for (int i = tid; i < p2; i += 256) s_keys[i] = i < nr ? (((unsigned long long)rows[i].z << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= p2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < p2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_keys[i], b = s_keys[ixj];
                    const bool asc = (i & k) == 0;
                    if ((a > b) == asc) {
                        s_keys[i] = b;
                        s_keys[ixj] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < nr; i += 256) {
        const int j = (int)(s_keys[i] & 0xFFFFFFFFu);
        const uint4 bb = rows[j];
        sorted[i] = make_uint4(bb.x, bb.y, bb.z, (uint32_t)j);
    }
    __syncthreads();
    chunk_boxes(sorted, reinterpret_cast<uint4*>(ws.rec_cbox) + span.cbox, nr, wave, lane);
}
__global__ __launch_bounds__(256) void occ_sort_kernel(OccScene sc, OccWorkspace ws, int sort_cap) {
    sort_object(sc, ws, sort_cap, blockIdx.x);
}

constexpr int kSetupVcapMax = 4096;  // vertices of one object that the setup kernel stages in LDS (48 KB) at most

struct CamRT {
    float R[9], T[3], dRe[9], dTe[3], dRa[9], dTa[3];
};

// world-space vertex k of face f: pool vertex + object offset in f32 (environment.py:148,171)
__device__ __forceinline__ void world_vertex(const int* __restrict__ pool_faces, const float* __restrict__ pool_verts,
                                             int vo, int fo, int f, int k, float ox, float oy, float oz, float* w) {
    const int vi = pool_faces[(size_t)(fo + f) * 3 + k];
    const float* pv = pool_verts + (size_t)(vo + vi) * 3;
    w[0] = pv[0] + ox;
    w[1] = pv[1] + oy;
    w[2] = pv[2] + oz;
}

__device__ __forceinline__ void world_corner(const float* __restrict__ pool_verts, int vo, int vi, float ox, float oy,
                                             float oz, float* w) {
    const float* pv = pool_verts + (size_t)(vo + vi) * 3;
    w[0] = pv[0] + ox;
    w[1] = pv[1] + oy;
    w[2] = pv[2] + oz;
}

__device__ __forceinline__ void view_pos(const CamRT& c, const float* w, VVert& q) {
#pragma unroll
    for (int j = 0; j < 3; ++j) q.v[j] = w[0] * c.R[j] + w[1] * c.R[3 + j] + w[2] * c.R[6 + j] + c.T[j];
}
template <bool GRAD>
__device__ __forceinline__ void view_from_world(const CamRT& c, const float* w, VVert& q) {
    view_pos(c, w, q);
    if (GRAD) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            q.de[j] = w[0] * c.dRe[j] + w[1] * c.dRe[3 + j] + w[2] * c.dRe[6 + j] + c.dTe[j];
            q.da[j] = w[0] * c.dRa[j] + w[1] * c.dRa[3 + j] + w[2] * c.dRa[6 + j] + c.dTa[j];
        }
    }
}

template <bool GRAD>
__device__ __forceinline__ void load_camera(const float* __restrict__ c, CamRT& C) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        C.R[i] = c[C_R + i];
        C.dRe[i] = GRAD ? c[C_DR_EL + i] : 0.f;
        C.dRa[i] = GRAD ? c[C_DR_AZ + i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        C.T[i] = c[C_T + i];
        C.dTe[i] = GRAD ? c[C_DT_EL + i] : 0.f;
        C.dTa[i] = GRAD ? c[C_DT_AZ + i] : 0.f;
    }
}

// Faces that straddle z = kZClip ([P3D] clip_faces cases 3 and 4).  Rare (the camera must be within
// ~0.5 of the geometry), so this lives out of line: it re-derives everything from the face index, both when the
// face is counted and when its records are written, and keeps its dynamically indexed arrays off the hot path.
// Everything arrives by value (the camera is re-read from memory) so that nothing of the caller's state has its
// address taken - that would pin the kernel arguments and the camera in scratch for the fast path as well.
template <bool GRAD>
__device__ __attribute__((noinline)) int clip_face_slow(const int* __restrict__ pool_faces, const float* __restrict__ pool_verts,
                                                        const float* __restrict__ camp, int S, int vo, int fo, int f,
                                                        float ox, float oy, float oz, Tri* out, int* flags) {
    CamRT c;
    load_camera<GRAD>(camp, c);
    VVert q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float w[3];
        world_vertex(pool_faces, pool_verts, vo, fo, f, k, ox, oy, oz, w);
        view_from_world<GRAD>(c, w, q[k]);
    }
    const bool b0 = q[0].v[2] < kZClip, b1 = q[1].v[2] < kZClip, b2 = q[2].v[2] < kZClip;
    const int nb = (int)b0 + (int)b1 + (int)b2;
    flags[0] = flags[1] = FLAG_CLIPPED;
    if (nb == 2) {
        // case 3: p1 = the vertex in front; new triangle (p4, p5, p1)
        const int i1 = !b0 ? 0 : (!b1 ? 1 : 2);
        const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
        out[0].v[0] = cut_edge<GRAD>(q[i1], q[i2]);
        out[0].v[1] = cut_edge<GRAD>(q[i1], q[i3]);
        out[0].v[2] = project<GRAD>(q[i1]);
        return finish_tri(out[0], S, tri_lim(S), 1.0f / (float)S) ? 1 : 0;
    }
    if (nb == 1) {
        // case 4: p1 = the vertex behind; quad -> (p4, p2, p5), (p5, p2, p3)
        const int i1 = b0 ? 0 : (b1 ? 1 : 2);
        const int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
        const PVert p4 = cut_edge<GRAD>(q[i1], q[i2]);
        const PVert p5 = cut_edge<GRAD>(q[i1], q[i3]);
        const PVert p2 = project<GRAD>(q[i2]);
        const PVert p3 = project<GRAD>(q[i3]);
        Tri ta, tb;
        ta.v[0] = p4; ta.v[1] = p2; ta.v[2] = p5;
        tb.v[0] = p5; tb.v[1] = p2; tb.v[2] = p3;
        const bool oka = finish_tri(ta, S, tri_lim(S), 1.0f / (float)S), okb = finish_tri(tb, S, tri_lim(S), 1.0f / (float)S);
        if (oka && okb) {
            out[0] = ta; out[1] = tb;
            flags[0] = FLAG_PAIR_FIRST | FLAG_CLIPPED; flags[1] = FLAG_PAIR_SECOND | FLAG_CLIPPED;
            return 2;
        }
        if (oka) { out[0] = ta; return 1; }
        if (okb) { out[0] = tb; return 1; }
    }
    return 0;  // nb == 3: the whole face is behind the clip plane
}

// VERTICES ONCE (round 4): a mesh vertex is shared by ~6 faces, and every face used to gather its three corners from
// global memory (17.7 M faces x 3 scattered 12-byte reads per step of the bench).  With `vcap` > 0 the block first
// copies the object's WORLD-space vertices (pool vertex + offset, the very expression world_corner evaluates) into
// dynamic LDS - coalesced - and the faces gather from there; an object with more than vcap vertices keeps the global
// gathers (block-uniform choice).  Same values either way: the records do not change by a bit.
#ifndef OCC_SETUP_TB
#define OCC_SETUP_TB 512
#endif
constexpr int kSetupTB = OCC_SETUP_TB;  // threads per block of the setup kernel (one block per (env, object))
constexpr int kHalfPad = 5;             // 16-byte parts of a staging slot: parts 0..4 of a record, then its three tangent parts (an odd
                                        // stride: the lanes' 16-byte stores fall on different banks)
// BLOCK SIZE AND LDS (round 4).  Sixteen waves per CU either way (<= 128 VGPRs); the block size decides how many waves
// share ONE copy of the object's vertices.  Round 3: 256-thread blocks, four to a CU with 40 KB each - whole records
// staged (9 KB per wave), no room for the vertices (31 KB for a 2 562-vertex mesh: with them only two blocks fitted and
// the kernel got 18 % slower, although at EQUAL residency the LDS vertices beat the global gathers by 13 %).  Now a
// record is staged in two pieces through a five-part slot (5 KB per wave), and two 512-thread blocks per CU hold their
// vertices in 78 KB each.  Measured on one box (step minus raster, us; scripts/ab_toggle.py): 256 threads / global
// gathers 800, 256 / LDS (two blocks per CU) 880, 512 / global 818, 512 / LDS 786, 1024 / global 914, 1024 / LDS 880.
template <bool GRAD, int TB>
__global__ __launch_bounds__(TB, 4) void occ_setup_kernel(OccScene sc, const float* __restrict__ cam, OccWorkspace ws, int vcap) {
    constexpr int W = TB / 64;  // waves per block
    extern __shared__ float s_wv[];  // 3 x vcap floats: world x | y | z of the object's vertices
    __shared__ int s_wcnt[2][W];  // double-buffered: one barrier per TB-face round
    __shared__ int s_rect[4];
    __shared__ uint2 s_box[TB];  // pixel bbox of this thread's face as its ONE visibility evaluation found it
    __shared__ float4 s_rec[W * 64 * kHalfPad];  // per wave: HALF the records of one round, staged for coalesced stores
    // work-item order (ws.order): faces per cell of a <= 16 x 16 grid over the image (cell = 8x8 tile, or a square of
    // tiles when the image has more than 16 tiles a side), faces too large to count cell by cell, tiles per cost class
    __shared__ uint32_t s_tcost[256];
    __shared__ uint32_t s_cls[kOrdClasses];
    __shared__ uint32_t s_big;
    const int eo = blockIdx.x;  // env*3 + object
    const int env = eo / 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (sc.skip && sc.skip[env]) {  // scene row not rendered in this launch: no records, no work items
        if (tid == 0) {
            ws.nrec[eo] = 0;
            ws.objrect[eo * 4 + 0] = 1 << 20;
            ws.objrect[eo * 4 + 1] = 1 << 20;
            ws.objrect[eo * 4 + 2] = -1;
            ws.objrect[eo * 4 + 3] = -1;
        }
        return;
    }
    const int mesh = sc.scene_mesh[eo];
    const int vo = sc.mesh_vert_off[mesh];
    const int fo = sc.mesh_face_off[mesh];
    const int nF = sc.mesh_face_off[mesh + 1] - fo;
    const float ox = sc.scene_offset[eo * 3], oy = sc.scene_offset[eo * 3 + 1], oz = sc.scene_offset[eo * 3 + 2];
    const float* __restrict__ c = cam + (size_t)env * OCC_CAM_STRIDE;
    CamRT C;
    load_camera<GRAD>(c, C);
    const int* __restrict__ pool_faces = sc.pool_faces;
    const float* __restrict__ pool_verts = sc.pool_verts;
    const int nV = sc.mesh_vert_off[mesh + 1] - vo;
    const bool vlds = vcap > 0 && nV <= vcap;  // block-uniform
    if (vlds) {
        for (int v = tid; v < nV; v += TB) {
            const float* pv = pool_verts + (size_t)(vo + v) * 3;
            s_wv[v] = pv[0] + ox;
            s_wv[vcap + v] = pv[1] + oy;
            s_wv[2 * vcap + v] = pv[2] + oz;
        }
    }
    auto corner = [&](int vi, float* w) {
        if (vlds) {
            w[0] = s_wv[vi];
            w[1] = s_wv[vcap + vi];
            w[2] = s_wv[2 * vcap + vi];
        } else {
            world_corner(pool_verts, vo, vi, ox, oy, oz, w);
        }
    };
    const RecSpan span = rec_span(ws, sc.rec_cap, eo);
    const int S = sc.img, rec_cap = span.cap;
    const float lim = tri_lim(S), ifS = 1.0f / (float)S;  // (once per block, not once per round)
    const bool ordered = ws.order != nullptr;
    const int tiles_side = S / 8;
    int cs = 0;  // tiles per cell side = 1 << cs
    while ((tiles_side >> cs) > 16) ++cs;
    if (ordered) {
        if (tid < 256) s_tcost[tid] = 0u;
        if (tid < kOrdClasses) s_cls[tid] = 0u;
        if (tid == 0) s_big = 0u;
    }
    if (tid == 0) {
        s_rect[0] = 1 << 20;
        s_rect[1] = 1 << 20;
        s_rect[2] = -1;
        s_rect[3] = -1;
    }
    __syncthreads();
    float* __restrict__ rec = ws.rec + span.base * OCC_REC_STRIDE;
    uint4* __restrict__ scan = reinterpret_cast<uint4*>(ws.scan) + span.base;
    int total = 0;
    bool overflow = false;
    int rx0 = 1 << 20, ry0 = 1 << 20, rx1 = -1, ry1 = -1;  // this thread's share of the object's block rect
    int round = 0;
    // vertex indices of the NEXT round's face are fetched one round ahead: the index -> vertex -> projection chain
    // of a round then starts at the vertex gather
    int vi0 = 0, vi1 = 0, vi2 = 0;
    if (tid < nF) {
        const int* pf = pool_faces + (size_t)(fo + tid) * 3;
        vi0 = pf[0]; vi1 = pf[1]; vi2 = pf[2];
    }
    for (int base = 0; base < nF; base += TB, round ^= 1) {
        const int f = base + tid;
        int cnt = 0;
        bool slow = false;
        float w0[3], w1[3], w2[3];  // world-space corners and 1 / z_view: all that a surviving face carries across the barrier
        float iz0 = 0.f, iz1 = 0.f, iz2 = 0.f;
        const int c0 = vi0, c1 = vi1, c2 = vi2;
        if (f + TB < nF) {
            const int* pf = pool_faces + (size_t)(fo + f + TB) * 3;
            vi0 = pf[0]; vi1 = pf[1]; vi2 = pf[2];
        }
        if (f < nF) {
            Tri tri;  // fast path: the unclipped face, positions only (recomputed for the survivors below)
            VVert q0, q1, q2;
            corner(c0, w0);
            corner(c1, w1);
            corner(c2, w2);
            view_from_world<false>(C, w0, q0);
            view_from_world<false>(C, w1, q1);
            view_from_world<false>(C, w2, q2);
            slow = (q0.v[2] < kZClip) || (q1.v[2] < kZClip) || (q2.v[2] < kZClip);
            if (!slow) {
                tri.v[0] = project<false>(q0, nullptr, &iz0);
                tri.v[1] = project<false>(q1, nullptr, &iz1);
                tri.v[2] = project<false>(q2, nullptr, &iz2);
                cnt = finish_tri(tri, S, lim, ifS) ? 1 : 0;
                s_box[tid] = make_uint2(tri.bbox.x, tri.bbox.y);  // read back by this thread only, after the barrier
            }
        }
        if (__ballot(slow)) {
            if (slow) {
                Tri tmp[2];
                int fl[2];
                cnt = clip_face_slow<GRAD>(pool_faces, pool_verts, c, S, vo, fo, f, ox, oy, oz, tmp, fl);
            }
        }
        // ordered compaction: exclusive prefix of cnt in {0,1,2} over the block
        const unsigned long long m1 = __ballot(cnt >= 1), m2 = __ballot(cnt == 2);
        const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const int pre = __popcll(m1 & lt) + __popcll(m2 & lt);
        if (lane == 0) s_wcnt[round][wave] = __popcll(m1) + __popcll(m2);
#ifdef OCC_SETUP_FULL_BARRIER  // (A/B build)
        __syncthreads();
#else
        // (only the wave counts cross this barrier, through LDS: no need to wait for the round's record stores to land)
        block_lds_barrier();
#endif
        int woff = 0, itot = 0;
        if (W <= 4) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const int cw = s_wcnt[round][w];
                if (w < wave) woff += cw;
                itot += cw;
            }
        } else {  // sixteen counts held in sixteen registers cost the loop its register budget
            int incl = lane < W ? s_wcnt[round][lane] : 0;  // inclusive scan over the first W lanes, read back as scalars
#pragma unroll
            for (int d = 1; d < W; d <<= 1) {
                const int t = __shfl_up(incl, d, 64);
                if (lane >= d) incl += t;
            }
            itot = __builtin_amdgcn_readlane(incl, W - 1);
            woff = wave ? __builtin_amdgcn_readlane(incl, __builtin_amdgcn_readfirstlane(wave) - 1) : 0;
        }
        const int pos = total + woff + pre;
        // Records of a wave are consecutive (ordered compaction): the survivors put theirs into LDS and the wave
        // copies the block out with full-width 16-byte stores (a lane writing its own 128-byte record straight to
        // memory issues eight partial-line stores).  Waves with a z-clipped face, or at the
        // capacity limit, store directly.
        const int wstart = total + woff, nw = __popcll(m1) + __popcll(m2);
        const bool staged = (__ballot(slow) == 0ull) && (wstart + nw <= rec_cap);
        bool rec_fast = false;  // this lane holds an unclipped survivor whose record goes through the wave's staging
        // tangents of the three corners (parts 5..7): only for the faces that survived culling, vertex by vertex
        auto tangents = [&](float4* __restrict__ r5) {
            VVert q;
            PVert pk;
            view_from_world<true>(C, w0, q);
            pk = project<true>(q, &iz0);
            r5[0] = make_float4(pk.t[0], pk.t[1], pk.t[2], pk.t[3]);
            view_from_world<true>(C, w1, q);
            pk = project<true>(q, &iz1);
            r5[1] = make_float4(pk.t[0], pk.t[1], pk.t[2], pk.t[3]);
            view_from_world<true>(C, w2, q);
            pk = project<true>(q, &iz2);
            r5[2] = make_float4(pk.t[0], pk.t[1], pk.t[2], pk.t[3]);
        };
        if (cnt >= 1) {
            if (pos + cnt <= rec_cap) {
                int x0, y0, x1, y1;
                const Shade sh = flat_shade(w0, w1, w2, c[C_C], c[C_C + 1], c[C_C + 2]);
                if (!slow) {
                    // positions again (data only - the verdict and the pixel bbox are the ones found above)
                    Tri tri;
                    {
                        VVert q;
                        view_from_world<false>(C, w0, q);
                        tri.v[0] = project<false>(q, &iz0);
                        view_from_world<false>(C, w1, q);
                        tri.v[1] = project<false>(q, &iz1);
                        view_from_world<false>(C, w2, q);
                        tri.v[2] = project<false>(q, &iz2);
                        const uint2 bx = s_box[tid];
                        const uint32_t zb = __float_as_uint(fmin3(tri.v[0].z, tri.v[1].z, tri.v[2].z));
                        tri.bbox = make_uint4(bx.x, bx.y, (zb & 0x80000000u) ? ~zb : (zb | 0x80000000u), 0u);
                        tri.tx0 = (int)(bx.x & 0xFFFFu) / OCC_BLOCK;
                        tri.ty0 = (int)(bx.x >> 16) / OCC_BLOCK;
                        tri.tx1 = (int)(bx.y & 0xFFFFu) / OCC_BLOCK;
                        tri.ty1 = (int)((bx.y >> 16) & 0x0FFFu) / OCC_BLOCK;
                    }
                    if (staged) {  // parts 0..4 now; the tangents follow once the wave has copied these out
                        rec_fast = true;
                        write_record_lo(&s_rec[(wave * 64 + pre) * kHalfPad], scan + pos, pos, tri, f, 0, sh);
                    } else {
                        float4* __restrict__ r4 = reinterpret_cast<float4*>(rec + (size_t)pos * OCC_REC_STRIDE);
                        write_record_lo(r4, scan + pos, pos, tri, f, 0, sh);
                        if (GRAD) tangents(r4 + 5);
                    }
                    x0 = tri.tx0; y0 = tri.ty0; x1 = tri.tx1; y1 = tri.ty1;
                } else {
                    Tri tmp[2];
                    int fl[2];
                    clip_face_slow<GRAD>(pool_faces, pool_verts, c, S, vo, fo, f, ox, oy, oz, tmp, fl);
                    write_record<GRAD>(rec + (size_t)pos * OCC_REC_STRIDE, scan + pos, pos, tmp[0], f, fl[0], sh);
                    x0 = tmp[0].tx0; y0 = tmp[0].ty0; x1 = tmp[0].tx1; y1 = tmp[0].ty1;
                    if (cnt == 2) {
                        write_record<GRAD>(rec + (size_t)(pos + 1) * OCC_REC_STRIDE, scan + pos + 1, pos + 1,
                                           tmp[1], f, fl[1], sh);
                        x0 = min(x0, tmp[1].tx0); y0 = min(y0, tmp[1].ty0);
                        x1 = max(x1, tmp[1].tx1); y1 = max(y1, tmp[1].ty1);
                    }
                }
                rx0 = min(rx0, x0); ry0 = min(ry0, y0);
                rx1 = max(rx1, x1); ry1 = max(ry1, y1);
                if (ordered) {  // this face in the cost of every cell its bbox touches (a handful)
                    const int cx0 = max(x0 >> (1 + cs), 0), cx1 = min(x1 >> (1 + cs), 15);
                    const int cy0 = max(y0 >> (1 + cs), 0), cy1 = min(y1 >> (1 + cs), 15);
                    if ((cx1 - cx0 + 1) * (cy1 - cy0 + 1) <= 16) {
                        for (int cy = cy0; cy <= cy1; ++cy)
                            for (int cx = cx0; cx <= cx1; ++cx) atomicAdd(&s_tcost[cy * 16 + cx], (uint32_t)cnt);
                    } else {
                        atomicAdd(&s_big, (uint32_t)cnt);  // counted for every tile
                    }
                }
            } else {
                overflow = true;
            }
        }
        if (staged) {  // wave-uniform: parts 0..4 out, then the tangents (parts 5..7) through the same LDS slots
            auto wave_sync = [] {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            };
            float4* __restrict__ dst = reinterpret_cast<float4*>(rec + (size_t)wstart * OCC_REC_STRIDE);
            const float4* src = &s_rec[wave * 64 * kHalfPad];
            wave_sync();
            for (int i = lane; i < nw * 5; i += 64) {
                const int rj = i / 5, part = i - rj * 5;
                dst[rj * kRecParts + part] = src[i];  // (slot stride = 5 parts: the staged parts are contiguous)
            }
            if (GRAD) {
                wave_sync();
                if (rec_fast) tangents(&s_rec[(wave * 64 + pre) * kHalfPad]);
                wave_sync();
                for (int i = lane; i < nw * 3; i += 64) {
                    const int rj = i / 3, part = i - rj * 3;
                    dst[rj * kRecParts + 5 + part] = src[rj * kHalfPad + part];
                }
            }
            wave_sync();
        }
        total += itot;
    }
    if (overflow || total > rec_cap) atomicOr(&ws.status[env], OCC_STATUS_REC_OVERFLOW);
    // object block rect: wave reduction, then one LDS atomic per wave
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        rx0 = min(rx0, __shfl_xor(rx0, m, 64));
        ry0 = min(ry0, __shfl_xor(ry0, m, 64));
        rx1 = max(rx1, __shfl_xor(rx1, m, 64));
        ry1 = max(ry1, __shfl_xor(ry1, m, 64));
    }
    if (lane == 0) {
        atomicMin(&s_rect[0], rx0);
        atomicMin(&s_rect[1], ry0);
        atomicMax(&s_rect[2], rx1);
        atomicMax(&s_rect[3], ry1);
    }
    {
        // SCAN ORDER of the raster kernel: (pixel bbox, key of the nearest vertex depth, record index) rows, written
        // with the records in face order (mesh order is spatially coherent, which makes the chunk boxes selective);
        // occ_sort_kernel re-orders dense objects front to back.  The depth keys make the raster kernel's
        // pruning exact in ANY order; the order only decides how early it bites.
        const int nr = min(total, rec_cap);
        __syncthreads();  // scan[] of the whole object written (and s_rect complete)
        chunk_boxes(scan, reinterpret_cast<uint4*>(ws.rec_cbox) + span.cbox, nr, wave, lane, W);
    }
    if (ordered && total > 0 && total <= rec_cap && s_rect[2] >= s_rect[0] && s_rect[3] >= s_rect[1] && s_rect[0] >= 0 && s_rect[1] >= 0) {
        // every tile of the object's rect (= one work item of occ_raster2_kernel) gets its cost class and a rank
        // among the object's tiles of that class; the class totals of the XCD queue grow by this object's counts and
        // tell where its share of each class starts (occ_order_kernel turns this into the item list)
        const int tx0 = s_rect[0] >> 1, ty0 = s_rect[1] >> 1;
        const int tw = (s_rect[2] >> 1) - tx0 + 1, th = (s_rect[3] >> 1) - ty0 + 1;
        uint32_t* __restrict__ tord = ws.order + ord_tiles_word(sc.n_env) + (size_t)eo * tiles_side * tiles_side;
        const uint32_t big = s_big;
        for (int local = tid; local < tw * th; local += TB) {
            const int tx = tx0 + local % tw, ty = ty0 + local / tw;
            const int cls = ord_class(s_tcost[min(ty >> cs, 15) * 16 + min(tx >> cs, 15)] + big);
            tord[local] = (atomicAdd(&s_cls[cls], 1u) << 5) | (uint32_t)cls;
        }
        __syncthreads();
        if (tid < kOrdClasses) {
            const uint32_t cnt = s_cls[tid];
            ws.order[kOrdBlk + (size_t)eo * kOrdClasses + tid] =
                cnt ? atomicAdd(&ws.order[kOrdCounts + (env & 7) * kOrdClasses + tid], cnt) : 0u;
        }
    }
    if (tid == 0) {
        // more records than the span holds (OCC_STATUS_REC_OVERFLOW is set): the object is left out altogether
        // rather than rendered from a truncated list whose last slot may never have been written
        const bool fits = total <= rec_cap;
        ws.nrec[eo] = fits ? total : 0;
        ws.objrect[eo * 4 + 0] = fits ? s_rect[0] : (1 << 20);
        ws.objrect[eo * 4 + 1] = fits ? s_rect[1] : (1 << 20);
        ws.objrect[eo * 4 + 2] = fits ? s_rect[2] : -1;
        ws.objrect[eo * 4 + 3] = fits ? s_rect[3] : -1;
    }
}

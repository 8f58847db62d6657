"""Diagnostic (GPU box): run one parity case through the engine and dump what its action gradient is made of -
the full face records (positions, tangents), the camera buffer, the per-object alpha / d alpha planes - to an .npz
that scripts/dbg/fwd_grad_emul.py --engine reads back in the authoring container.

    python scripts/dbg/dump_grad_case.py seed:mesh:img:az_range:radius out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import parity_utils as PU  # noqa: E402

parts = sys.argv[1].split(":")
seed, mesh, img, azr, radius = int(parts[0]), parts[1], int(parts[2]), float(parts[3]), float(parts[4])
case = PU.make_case(2, seed, mesh, azr)
got = PU.run_engine(case, img, radius=radius)
eng = got["engine"]
torch.cuda.synchronize()
nrec = eng._ws_tensors["nrec"].cpu().numpy()[: eng.NT * 3]
rec_off = eng._rec_tensors["rec_off"].cpu().numpy().view(np.int64)
rec = eng._rec_tensors["rec"].view(torch.float32)
out = dict(grad=got["grad"].numpy(), obj_grad=got["obj_grad"].numpy(), alphas=got["alphas"].numpy(), jac=got["jac"].numpy(),
           object_mass=got["object_mass"].numpy(), cam=eng.cam.cpu().numpy(), loss=got["loss"].numpy())
for eo in range(eng.N * 3):
    n, base = int(nrec[eo]), int(rec_off[eo])
    out["rec%d" % eo] = rec[base * 32: (base + n) * 32].cpu().numpy().reshape(n, 32).copy()
np.savez_compressed(sys.argv[2], **out)
print("dumped", sys.argv[2], {k: v.shape for k, v in out.items()})

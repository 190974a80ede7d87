"""MLP / conv / edge kernel time against batch size (run under rocprofv3 --kernel-trace; ARREAU_MLP_NB=1|2 forces the
tile geometry): predict_scores at N = 20 * B atoms for a range of B, three evaluations each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd import build
build.build(verbose=False)
from arreau_amd.checkpoint import make_synthetic_model
from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
from tests.helpers import random_state

dev = torch.device("cuda", 0)
m = make_synthetic_model(S=90, seed=1234).to(dev)
eng = m.engine()
for B in [int(v) for v in os.environ.get("SIZES", "52,103,154,205,256,308,359,410").split(",")]:
    frac, types, lengths, angles, na = random_state(90, [20] * B, 5, sampler_like=True)
    f, ty, le, an = (frac.to(dev), types.to(dev, torch.int32), lengths.to(dev), angles.to(dev))
    off = crystal_offsets(na, dev)
    t = torch.full((B,), 500, device=dev, dtype=torch.int32)
    for _ in range(3):
        eng.predict_scores(f, ty, le, an, t, off)
    torch.cuda.synchronize()
print("done")

"""Phase timing of the fp16x3 edge kernel (GPU box).  Rebuilds the library with -DARREAU_EDGE_TIMING, runs a few
sampler steps of the bench workload and prints the share of shader-clock ticks wave 0 spends per phase.

    ARREAU_EXTRA_HIPCC_FLAGS=-DARREAU_EDGE_TIMING python tools/edge_timing.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "ARREAU_EDGE_TIMING" in os.environ.get("ARREAU_EXTRA_HIPCC_FLAGS", ""), "set ARREAU_EXTRA_HIPCC_FLAGS=-DARREAU_EDGE_TIMING"
import torch  # noqa: E402

from arreau_amd import build as _build  # noqa: E402

_build.build(force=True, verbose=False)
import numpy as np  # noqa: E402

from arreau_amd.checkpoint import make_synthetic_model  # noqa: E402
from arreau_amd.diffusion.diffusion_helpers import crystal_offsets  # noqa: E402

lib = ctypes.CDLL(_build.LIB)
ticks = (ctypes.c_ulonglong * 8)()
dev = torch.device("cuda:0")
B, n, S = int(os.environ.get("B", 256)), int(os.environ.get("NATOMS", 20)), 90
N = B * n
model = make_synthetic_model(S=S, seed=1234).to(dev)
eng = model.engine()
torch.manual_seed(1000)
rng = np.random.RandomState(1000)
f32 = dict(device=dev, dtype=torch.float32)
ang = torch.tensor(np.stack([np.full(B, 90.0), rng.uniform(90, 180, B), np.full(B, 90.0)], 1), dtype=torch.float32).to(**f32)
lengths = torch.randn(B, 3).to(**f32)
frac = torch.randn(N, 3).to(**f32)
types = torch.full((N,), S - 1, device=dev, dtype=torch.int32)
off = crystal_offsets(torch.full((B,), n), dev)
t = torch.full((B,), 999, device=dev, dtype=torch.int32)


def step():
    eng.predict_scores(frac, types, lengths, ang, t, off)


for _ in range(3):
    step()
torch.cuda.synchronize()
lib.arreau_debug_edge_ticks(ticks, 1)
reps = 10
for _ in range(reps):
    step()
torch.cuda.synchronize()
lib.arreau_debug_edge_ticks(ticks, 0)
tot = sum(ticks[:5])
names = ["set-up", "layer 1 (4 chunks)", "layer 2 (8 chunks)", "projections (20 chunks)", "tail store"]
wgs = N // 2
for i, nm in enumerate(names):
    print("%-26s %6.1f%%   %9.0f ticks / workgroup" % (nm, 100.0 * ticks[i] / tot, ticks[i] / reps / wgs))
print("total ticks / workgroup %.0f" % (tot / reps / wgs))
if ticks[6]:
    print("shader clock while the kernel runs: %.0f MHz (clock64 / wall_clock64 x 100 MHz)" % (100.0 * ticks[5] / ticks[6]))

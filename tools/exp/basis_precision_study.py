"""How many bits does the STASHED BASIS need?  (CPU, oracle only; round 3.)

Since round 3 the only large intermediate of a sampling step is the windowed basis [E * 16][256] that the edge kernel
writes once and every layer's message kernel reads back (conv_proj.hip).  The kernels hold it as the two fp16 planes of
the split-precision scheme (hi = f16(b), lo = f16((b - hi) * 2^11): 4 bytes per value).  This script replaces the basis
inside the fp32 oracle's FiberBundleConv by cheaper encodings and reports the change of the network outputs against the
unmodified fp32 run (full-size architecture, trained-like synthetic weights):

    hi fp16 + lo as OCP fp8 e4m3 / e5m2 (3 bytes per value), hi alone (2 bytes), and plain significand rounding.

    python tools/exp/basis_precision_study.py        (results: profiles/r03_basis_precision_study.txt)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd.checkpoint import make_synthetic_model  # noqa: E402
from oracle import ponita as OP, sampler as OS  # noqa: E402
from tests.helpers import oracle_from_module, random_state  # noqa: E402
from tools.exp.k_precision_study import round_significand  # noqa: E402

MODE = ["exact"]
_orig = OP.fiber_bundle_conv


def encode(b):
    m = MODE[0]
    if m == "exact":
        return b
    if m.startswith("bits"):
        return round_significand(b, int(m[4:]))
    hi = b.to(torch.float16).to(torch.float32)
    if m == "hi":
        return hi
    if m == "f16f16":
        return hi + ((b - hi) * 2048.0).to(torch.float16).to(torch.float32) / 2048.0
    dt = {"f16e4m3": torch.float8_e4m3fn, "f16e5m2": torch.float8_e5m2}[m]
    return hi + ((b - hi) * 2048.0).to(dt).to(torch.float32) / 2048.0


def conv(sd, prefix, x, edge_index, kernel_basis, fiber_kernel_basis, stats=None):
    return _orig(sd, prefix, x, edge_index, encode(kernel_basis), fiber_kernel_basis, stats=stats)


def main():
    S = 90
    model = make_synthetic_model(S=S, seed=1234, trained_like=True)
    om32 = oracle_from_module(model, torch.float32)
    OP.fiber_bundle_conv = conv
    for name, counts, kw in (("64 x 2, cells 6-9 A", [64] * 2, dict(cell=(6.0, 9.0))), ("20 x 8, cells 4-8 A", [20] * 8, dict(cell=(4.0, 8.0)))):
        frac, types, lengths, angles, na = random_state(S, counts, 7, **kw)
        N, B = int(na.sum()), len(counts)
        batch = torch.arange(B).repeat_interleave(na)
        args = (frac, F.one_hot(types, S), torch.full((N,), 500), na, lengths, angles, batch)
        MODE[0] = "exact"
        base = OS.predict_scores(om32, *args)
        print(name, ": max |eps|, |logits|, |len0| =", " ".join("%.3g" % float(a.abs().max()) for a in base[:3]))
        for mode in ("f16f16", "f16e4m3", "f16e5m2", "hi", "bits18", "bits16", "bits14", "bits12"):
            MODE[0] = mode
            q = OS.predict_scores(om32, *args)
            print("   basis as %-8s: max change of eps / logits / len0 = " % mode +
                  " / ".join("%.2e" % float((a - b).abs().max()) for a, b in zip(q[:3], base[:3])))
        MODE[0] = "exact"


if __name__ == "__main__":
    main()

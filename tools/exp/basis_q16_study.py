"""What does the BLOCK-QUANTISED basis stash cost?  (CPU, oracle only; round 5.)

Since round 5 the basis form stores the windowed basis as 16-bit mantissas with one power-of-two exponent per block of eight
values (arreau_amd/csrc/f16x3.h: bq_encode8 -- the eight values one lane holds of a k-block; 2.125 bytes per value where the
planes of the split scheme took 3: fp16 + e4m3 residual, round 3), and every fp16x3 edge kernel rounds its basis values to that
grid.  This script replaces the kernel projection inside the fp32 oracle's FiberBundleConv by an emulation of the kernels'
arithmetic (products exact, accumulation in fp64: only the operand formats differ) for each stash format, with three fp16
products ("f16x3") and with the cross products on e4m3 operands ("e4m3hw", the default of conv_proj.hip), and reports the
network outputs' distance to the fp64 oracle, to the fp32 oracle and to the two-fp16-planes form (ARREAU_BASIS_Q16=0) -- the
figure tests/test_gpu_parity.py::test_launch_geometry_switches_and_counted_waits_are_bitwise_neutral bounds on the GPU.

    python tools/exp/basis_q16_study.py        (results: profiles/r05_basis_q16_study.txt)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tools.exp.cross_precision_study as CS  # noqa: E402
from arreau_amd.checkpoint import make_synthetic_model  # noqa: E402
from oracle import ponita as OP, sampler as OS  # noqa: E402
from tests.helpers import oracle_from_module, random_state  # noqa: E402


def q_block(b, bits=16, grp=8):
    """bq_encode8 + decode: signed `bits`-bit mantissas, one exponent per `grp` consecutive values; amax (1 + 2^-10) < 2^E, E >= -17"""
    R = b.shape[0]
    v = b.reshape(R, -1, grp).double()
    am = v.abs().amax(-1, keepdim=True) * (1 + 2.0 ** -10)
    E = (torch.floor(torch.log2(am.clamp_min(2.0 ** -80))) + 1).clamp(min=-17)
    s = torch.exp2((bits - 1) - E)
    return (torch.round(v * s) / s).float().reshape(b.shape)


FMT = [None]   # None: the planes alone; ("planes", "e4m3") = round 3/4's stash; (bits, group) = block-quantised


def project(basis, W):
    m = CS.MODE[0]
    if m == "exact":
        return F.linear(basis, W)
    shp = basis.shape
    b = basis.reshape(-1, shp[-1]).float()
    fmt = FMT[0]
    q = b if fmt in (None, "f16e4m3") else q_block(b, *fmt)
    b1 = q.to(torch.float16).float()
    r = (q - b1) * 2048.0
    b2 = r.to(torch.float16).float() if fmt is None else CS.q_e4m3(r)   # (block-quantised: the e4m3 residual is exact)
    a1 = W.to(torch.float16).float()
    a2 = ((W - a1) * 2048.0).to(torch.float16).float()
    main = b1.double() @ a1.double().T
    if m == "f16x3":
        cross = b2.double() @ a1.double().T + b1.double() @ a2.double().T
    else:  # "e4m3hw": conv_proj.hip's X8 form
        cross = (CS.q_e4m3(b2).double() @ CS.q_e4m3(a1 * 64).double().T + CS.q_e4m3(b1).double() @ CS.q_e4m3(a2 * 64).double().T) / 64
    return (main + cross / 2048.0).float().reshape(shp[:-1] + (W.shape[0],))


def main():
    S = 90
    CS.project = project
    OP.fiber_bundle_conv = CS.conv
    for wtag, heavy in (("Gaussian-initialised weights (trained_like)", False), ("heavy-tailed kernel / basis weights", True)):
        model = make_synthetic_model(S=S, seed=1234, trained_like=True)
        if heavy:
            from tests.helpers import make_heavy_tailed
            make_heavy_tailed(model, seed=5)
        om32 = oracle_from_module(model, torch.float32)
        om64 = oracle_from_module(model, torch.float64)
        print("==", wtag)
        for name, counts, kw in (("64 x 2, cells 6-9 A", [64] * 2, dict(cell=(6.0, 9.0))), ("20 x 8, cells 4-8 A", [20] * 8, dict(cell=(4.0, 8.0)))):
            frac, types, lengths, angles, na = random_state(S, counts, 7, **kw)
            N, B = int(na.sum()), len(counts)
            batch = torch.arange(B).repeat_interleave(na)
            args = (frac, F.one_hot(types, S), torch.full((N,), 500), na, lengths, angles, batch)
            CS.MODE[0] = "exact"
            base = OS.predict_scores(om32, *args)
            ref64 = OS.predict_scores(om64, frac.double(), F.one_hot(types, S), torch.full((N,), 500), na, lengths.double(), angles.double(), batch)
            d = lambda q, r: " / ".join("%.2e" % float((a.double() - b.double()).abs().max()) for a, b in zip(q[:3], r[:3]))
            print(name, ": max |eps|, |logits|, |len0| =", " ".join("%.3g" % float(a.abs().max()) for a in base[:3]),
                  "; fp32 oracle to fp64:", d(base, ref64))
            for mode in ("f16x3", "e4m3hw"):
                CS.MODE[0], FMT[0] = mode, None
                planes = OS.predict_scores(om32, *args)
                for fmt in (None, "f16e4m3", (16, 8), (16, 4), (16, 32), (15, 8), (14, 8)):
                    CS.MODE[0], FMT[0] = mode, fmt
                    q = OS.predict_scores(om32, *args)
                    print("   %-7s stash %-10s: to fp64 = %s ; to the fp32 oracle = %s ; to two fp16 planes = %s" % (
                        mode, {None: "f16 + f16"}.get(fmt, fmt), d(q, ref64), d(q, base), d(q, planes)))
            CS.MODE[0], FMT[0] = "exact", None


if __name__ == "__main__":
    main()

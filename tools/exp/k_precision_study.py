"""How many significand bits do the per-layer edge kernels K need?  (CPU, oracle only.)

The K stash -- [L][E * 16][C] fp32, written by the edge kernel and read back by the conv kernels -- is the only large
intermediate of the sampling step (1.68 GB at 256 x 20: DESIGN.md sections 2 and 8).  This script rounds K to fewer
significand bits inside the fp32 oracle's FiberBundleConv (round to nearest on the magnitude) and reports the change of
the network outputs against the unmodified fp32 run, for the full-size architecture (S = 90, C = 128, L = 5) with
trained-like synthetic weights.

    python tools/exp/k_precision_study.py        (results: profiles/r02g_k_precision_study.txt)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd.checkpoint import make_synthetic_model  # noqa: E402
from oracle import ponita as OP, sampler as OS  # noqa: E402
from tests.helpers import oracle_from_module, random_state  # noqa: E402

BITS = [24]


def round_significand(k, bits):
    if bits >= 24:
        return k
    drop = 24 - bits
    a = k.abs().contiguous().view(torch.int32)
    a = ((a + (1 << (drop - 1))) >> drop) << drop
    return a.view(torch.float32) * torch.sign(k)


BLOCK = [None]  # (group size, signed mantissa bits) of a block-scaled format, or None


def block_quant(k, group, mant_bits):
    """`group` consecutive channels share one power-of-two scale; signed integer mantissas of `mant_bits` bits."""
    g = k.reshape(-1, group)
    m = g.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    step = torch.exp2(torch.floor(torch.log2(m)) + 1 - (mant_bits - 1))
    lim = 2 ** (mant_bits - 1) - 1
    return (torch.round(g / step).clamp(-lim, lim) * step).reshape(k.shape)


def conv_with_rounded_k(sd, prefix, x, edge_index, kernel_basis, fiber_kernel_basis, stats=None):
    kernel = round_significand(F.linear(kernel_basis, sd[prefix + ".kernel.weight"]), BITS[0])
    if BLOCK[0] is not None:
        kernel = block_quant(kernel, *BLOCK[0])
    messages = kernel * x[edge_index[0]]
    x_1 = torch.zeros_like(x).index_add_(0, edge_index[1], messages)
    fiber_kernel = F.linear(fiber_kernel_basis, sd[prefix + ".fiber_kernel.weight"])
    x_2 = torch.einsum("boc,opc->bpc", x_1, fiber_kernel) / fiber_kernel.shape[-2]
    if stats is not None:
        stats.append((x, x_1, x_2))
    return x_2 + sd[prefix + ".bias"], messages


def main():
    S = 90
    model = make_synthetic_model(S=S, seed=1234, trained_like=True)
    om32 = oracle_from_module(model, torch.float32)
    OP.fiber_bundle_conv = conv_with_rounded_k
    for name, counts, kw in (("64 x 2, cells 6-9 A", [64] * 2, dict(cell=(6.0, 9.0))), ("20 x 8, cells 4-8 A", [20] * 8, dict(cell=(4.0, 8.0)))):
        frac, types, lengths, angles, na = random_state(S, counts, 7, **kw)
        N, B = int(na.sum()), len(counts)
        batch = torch.arange(B).repeat_interleave(na)
        args = (frac, F.one_hot(types, S), torch.full((N,), 500), na, lengths, angles, batch)
        BITS[0] = 24
        base = OS.predict_scores(om32, *args)
        print(name, ": max |eps|, |logits|, |len0| =", " ".join("%.3g" % float(a.abs().max()) for a in base[:3]))
        for bits in (20, 18, 16, 14, 12, 11, 10, 8):
            BITS[0] = bits
            q = OS.predict_scores(om32, *args)
            print("   K with %2d significand bits: max change of eps / logits / len0 = " % bits +
                  " / ".join("%.2e" % float((a - b).abs().max()) for a, b in zip(q[:3], base[:3])))
        BITS[0] = 24
        for cfg in ((16, 16), (4, 16), (8, 15), (4, 15), (4, 14)):  # 2.06 / 2.25 / 2.0 / 2.0 / 2.0 bytes per value with an 8-bit scale
            BLOCK[0] = cfg
            q = OS.predict_scores(om32, *args)
            print("   K block-scaled, groups of %2d channels, %2d-bit signed mantissas: " % cfg +
                  " / ".join("%.2e" % float((a - b).abs().max()) for a, b in zip(q[:3], base[:3])))
        BLOCK[0] = None


if __name__ == "__main__":
    main()

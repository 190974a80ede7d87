"""How many bits do the CROSS products of the split-precision kernel projection need?  (CPU, oracle only; round 4.)

conv_proj.hip evaluates K = basis . Wk^T as three fp16 matrix products,
    K ~= b1 a1 + (b2 a1 + b1 a2) / 2^11,   b1 = f16(b), b2 = e4m3((b - b1) 2^11) (as stashed), a1 = f16(W), a2 = f16((W - a1) 2^11).
The two cross products sit 2^-11 below the main one, so their OPERANDS need only a few bits.  On gfx950 the
block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 runs e4m3 operands at twice and e2m3 (fp6) operands at four times the
fp16 rate: both cross products of two 32-wide k-blocks are ONE K = 128 instruction ([a1 | a2 | a1' | a2'] . [b2 | b1 | b2' | b1']).
This script replaces the projection inside the fp32 oracle's FiberBundleConv by an emulation of each candidate (products
exact, accumulation in fp64: only the operand formats differ) and reports the change of the network outputs against
the unmodified fp32 oracle AND against the emulation of today's kernel (full-size architecture, trained-like weights).

    python tools/exp/cross_precision_study.py       (results: profiles/r04_cross_precision_study.txt)
"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arreau_amd.checkpoint import make_synthetic_model  # noqa: E402
from oracle import ponita as OP, sampler as OS  # noqa: E402
from tests.helpers import oracle_from_module, random_state  # noqa: E402

MODE = ["exact"]
_orig = OP.fiber_bundle_conv
E4 = torch.float8_e4m3fn


def q_e4m3(v):
    return v.clamp(-448.0, 448.0).to(E4).to(torch.float32)


def q_e2m3(v):
    """Round to the OCP fp6 e2m3 grid: step 0.125 below 2, 0.25 below 4, 0.5 up to 7.5 (saturating)."""
    a = v.abs().clamp(max=7.5)
    step = torch.where(a < 2.0, 0.125, torch.where(a < 4.0, 0.25, 0.5))
    return torch.sign(v) * torch.clamp(torch.round(a / step) * step, max=7.5)


def lane_blocks(p_first, p_second):
    """[R, 256] x 2 -> [R, 4 kb-pairs, 4 g, 32]: what one lane (row, g) of the K = 128 instruction holds for a pair of
    k-blocks: [first(kb) 8 | second(kb) 8 | first(kb+1) 8 | second(kb+1) 8], k = 32 kb + 8 g + e."""
    R = p_first.shape[0]
    f = p_first.reshape(R, 4, 2, 4, 8)   # [R, kb-pair, kb in pair, g, e]
    s = p_second.reshape(R, 4, 2, 4, 8)
    return torch.stack([f[:, :, 0], s[:, :, 0], f[:, :, 1], s[:, :, 1]], dim=3).reshape(R, 4, 4, 32)


def block_scale(blk, fixed=None):
    """power-of-two scale per lane block so that the block maximum lands in (3.75, 7.5]"""
    if fixed is not None:
        return torch.full(blk.shape[:-1] + (1,), float(fixed))
    m = blk.abs().amax(dim=-1, keepdim=True).clamp_min(2.0 ** -40)
    return torch.exp2(torch.ceil(torch.log2(m / 7.5)))


def cross_fp6(b1, b2, a1, a2, fixed_b=None, fixed_a=None):
    """sum over k of q6(a1) q6(b2) + q6(a2) q6(b1) with one scale per lane block of 32 values (E8M0, as the instruction takes them)"""
    B = lane_blocks(b2, b1)     # [R, 4, 4, 32]
    A = lane_blocks(a1, a2)     # [C, 4, 4, 32]
    sb, sa = block_scale(B, fixed_b), block_scale(A, fixed_a)
    Bq = (q_e2m3(B / sb) * sb).double()
    Aq = (q_e2m3(A / sa) * sa).double()
    return torch.einsum("rpgk,cpgk->rc", Bq, Aq)


def project(basis, W):
    m = MODE[0]
    if m == "exact":
        return F.linear(basis, W)
    shp = basis.shape
    b = basis.reshape(-1, shp[-1]).float()
    b1 = b.to(torch.float16).float()
    b2 = q_e4m3((b - b1) * 2048.0)  # as stashed today
    a1 = W.to(torch.float16).float()
    a2 = ((W - a1) * 2048.0).to(torch.float16).float()
    main = b1.double() @ a1.double().T
    if m == "f16x3":     # today's kernel
        cross = b2.double() @ a1.double().T + b1.double() @ a2.double().T
    elif m == "e4m3":    # every cross operand rounded to e4m3
        cross = q_e4m3(b2).double() @ q_e4m3(a1).double().T + q_e4m3(b1).double() @ q_e4m3(a2).double().T
    elif m == "e4m3s":   # e4m3 with the weight planes scaled into the format's range (x 64: power of two, exact)
        cross = (q_e4m3(b2).double() @ q_e4m3(a1 * 64).double().T + q_e4m3(b1).double() @ q_e4m3(a2 * 64).double().T) / 64
    elif m == "e4m3hw":  # what conv_proj.hip's X8 form computes: ONE fp8 product per pair of k-blocks, so the four operand scales
        # are tied -- a1_8 = e4m3(64 a1), b2_8 = e4m3(b2) (as stashed), a2_8 = e4m3(64 a2), b1_8 = e4m3(b1): X = 64 (a1 b2 + a2 b1)
        cross = (q_e4m3(b2).double() @ q_e4m3(a1 * 64).double().T + q_e4m3(b1).double() @ q_e4m3(a2 * 64).double().T) / 64
    elif m == "e2m3blk":  # fp6, data-dependent scale per lane block on both sides
        cross = cross_fp6(b1, b2, a1, a2)
    elif m == "e2m3fixB":  # fp6, weights block-scaled (host), basis with ONE fixed scale (no maximum to find in the kernel)
        cross = cross_fp6(b1, b2, a1, a2, fixed_b=FIXED_B[0])
    elif m == "nocross_b2":  # without b2 a1 (what "hi alone" costs)
        cross = b1.double() @ a2.double().T
    elif m == "nocross":
        cross = torch.zeros_like(main)
    else:
        raise ValueError(m)
    return (main + cross / 2048.0).float().reshape(shp[:-1] + (W.shape[0],))


FIXED_B = [1.0]


def conv(sd, prefix, x, edge_index, kernel_basis, fiber_kernel_basis, stats=None):
    kernel = project(kernel_basis, sd[prefix + ".kernel.weight"])
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
    om64 = oracle_from_module(model, torch.float64)
    OP.fiber_bundle_conv = conv
    for name, counts, kw in (("64 x 2, cells 6-9 A", [64] * 2, dict(cell=(6.0, 9.0))), ("20 x 8, cells 4-8 A", [20] * 8, dict(cell=(4.0, 8.0)))):
        frac, types, lengths, angles, na = random_state(S, counts, 7, **kw)
        N, B = int(na.sum()), len(counts)
        batch = torch.arange(B).repeat_interleave(na)
        args = (frac, F.one_hot(types, S), torch.full((N,), 500), na, lengths, angles, batch)
        MODE[0] = "exact"
        base = OS.predict_scores(om32, *args)
        ref64 = OS.predict_scores(om64, frac.double(), F.one_hot(types, S), torch.full((N,), 500), na, lengths.double(), angles.double(), batch)
        MODE[0] = "f16x3"
        cur = OS.predict_scores(om32, *args)
        print(name, ": max |eps|, |logits|, |len0| =", " ".join("%.3g" % float(a.abs().max()) for a in base[:3]))
        d64 = lambda q: " / ".join("%.2e" % float((a.double() - b).abs().max()) for a, b in zip(q[:3], ref64[:3]))
        print("   %-12s: distance to the fp64 oracle (eps / logits / len0) = %s" % ("fp32 oracle", d64(base)))
        modes = ["f16x3", "e4m3hw", "e4m3", "e4m3s", "e2m3blk", "e2m3fixB:1", "e2m3fixB:0.5", "e2m3fixB:0.25", "nocross_b2", "nocross"]
        for mode in modes:
            if ":" in mode:
                MODE[0], FIXED_B[0] = mode.split(":")[0], float(mode.split(":")[1])
            else:
                MODE[0] = mode
            q = OS.predict_scores(om32, *args)
            print("   %-12s: to fp64 = %s ; change against today's arithmetic = %s" % (
                mode, d64(q), " / ".join("%.2e" % float((a - b).abs().max()) for a, b in zip(q[:3], cur[:3]))))
        MODE[0] = "exact"


if __name__ == "__main__":
    main()

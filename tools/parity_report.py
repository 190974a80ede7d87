#!/usr/bin/env python3
"""Measured deviations of the HIP score network from the CPU oracle (GPU box).

    python tools/parity_report.py [--out profiles/parity_r02.json]

For the S=90 / T=1000 architecture (the 1.1M-parameter model) and three batches -- [20]x8 sampler-like,
[64]x2 dense cells, and the 420-crystal ragged batch -- at t in {999, 500, 2, 1}, with the oracle's edge
list teacher-forced (network parity, independent of neighbour tie-breaking), records

    max |delta| of eps / logits / len0   HIP vs the fp32 oracle   and   HIP vs the fp64 oracle

for the default kernels (fp16x3 split precision) and for the exact fp32-MFMA kernels
(arreau_model_set_variant(0, 0)), plus the fp32 oracle's own distance to fp64.  The parity tests assert the
bounds this report shows (tests/test_gpu_parity.py reads nothing from here; the numbers are evidence).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "parity_r05.json"))
    ap.add_argument("--quick", action="store_true", help="only the [20]x8 case")
    ap.add_argument("--heavy-tailed", action="store_true", help="Student-t kernel / basis weights (tests/helpers.py: make_heavy_tailed)")
    args = ap.parse_args()
    from arreau_amd.checkpoint import make_synthetic_model
    from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
    from oracle import sampler as OS
    from tests.helpers import oracle_from_module, random_state, slots_from_edges

    dev = torch.device("cuda", 0)
    S = 90
    m = make_synthetic_model(S=S, seed=1234)
    if args.heavy_tailed:
        from tests.helpers import make_heavy_tailed
        make_heavy_tailed(m, seed=5)
    m = m.to(dev)
    om32, om64 = oracle_from_module(m, torch.float32), oracle_from_module(m, torch.float64)
    eng = m.engine()
    rng = np.random.RandomState(5)
    cases = [("20x8_sampler_like", [20] * 8, dict(sampler_like=True), 4)]
    if not args.quick:
        cases += [("20x40_sampler_like", [20] * 40, dict(sampler_like=True), 6), ("64x2_dense", [64, 64], dict(cell=(6.0, 9.0)), 41),
                  ("ragged_420", [int(v) for v in rng.randint(1, 7, size=420)], dict(cell=(3.5, 9.0)), 91),
                  ("20x256_bench_size", [20] * 256, dict(cell=(4.0, 8.0)), 17)]
    # (edge variant, mlp variant): the default kernels, the exact fp32-MFMA kernels, the small-batch ConvNext kernel
    # (hidden dimension split over eight waves) and the shape-general fp32 GEMM network
    # Round 4: "basis_form_*" = the message path of launches above 2,000 receivers (what bench.py's `value` runs: stashed basis,
    # per-layer projection in conv_proj.hip), forced at these oracle-sized batches with ARREAU_BASIS_MIN_RECEIVERS=240 (read per
    # launch) -- with the two cross products on the fp8 matrix instruction (the default), with three fp16 products
    # (ARREAU_CROSS_FP8=0); ARREAU_BASIS_FP8=0 in the environment of the whole run gives the same rows on two fp16 planes in the stash instead of
    # the block-quantised values (that switch is read once per process).  Batches of at most 240 atoms
    # cannot take the basis form at all: their "basis_form_*" rows are SKIPPED (round 4 recorded them under a name they did not
    # run).
    variants = [("default_fp16x3", 4, 3, {}), ("fp32_mfma", 0, 0, {}), ("small_launch_mlp_form", 4, 4, {}), ("general_fp32_gemm", 5, 3, {}),
                ("basis_form_fp8_cross", 4, 3, {"ARREAU_BASIS_MIN_RECEIVERS": "240"}),
                ("basis_form_fp16_cross", 4, 3, {"ARREAU_BASIS_MIN_RECEIVERS": "240", "ARREAU_CROSS_FP8": "0"})]
    report = {"model": "synthetic S=90 T=1000 C=128 D=256 L=5 (make_synthetic_model seed 1234, trained_like)" +
                       (" with heavy-tailed kernel / basis weights (make_heavy_tailed seed 5)" if args.heavy_tailed else ""),
              "edges": "oracle's radius_graph_pbc, teacher-forced", "device": torch.cuda.get_device_name(0),
              "cases": {}}
    worst = {v[0]: {"eps": 0.0, "logits": 0.0, "len0": 0.0, "eps_vs_f64": 0.0, "logits_vs_f64": 0.0,
                    "len0_vs_f64": 0.0} for v in variants}
    for name, num_atoms, kw, seed in cases:
        frac, types, lengths, angles, na = state = random_state(S, num_atoms, seed, **kw)
        B, N = len(num_atoms), int(sum(num_atoms))
        batch = torch.arange(B).repeat_interleave(na)
        d = lambda v: v.to(dev).contiguous()
        f, ty, le, an, off = d(frac), d(types.to(torch.int32)), d(lengths), d(angles), crystal_offsets(na, dev)
        report["cases"][name] = {"crystals": B, "atoms": N, "max_atoms_per_crystal": int(max(num_atoms)), "t": {}}
        for t in (999, 500, 2, 1):
            t0 = time.time()
            tt = torch.full((N,), t)
            eps32, log32, len32, (ei, dists, direction, _c, _l) = OS.predict_scores(
                om32, frac, F.one_hot(types, S), tt, na, lengths, angles, batch, return_graph=True)
            eps64, log64, len64 = OS.predict_scores(
                om64, frac.double(), F.one_hot(types, S), tt, na, lengths.double(), angles.double(), batch,
                edges=(ei, dists.double(), direction.double()))
            deg, src, sdir, sdist = slots_from_edges(ei, dists, direction, N, 8)
            edges = tuple(x.to(dev).contiguous() for x in (deg, src, sdir, sdist))
            t_c = torch.full((B,), t, device=dev, dtype=torch.int32)
            rec = {"edges": int(ei.shape[1]),
                   "magnitudes": {"eps": float(eps32.abs().max()), "logits": float(log32.abs().max()),
                                  "len0": float(len32.abs().max())},
                   "oracle_f32_vs_f64": {"eps": float((eps32.double() - eps64).abs().max()),
                                         "logits": float((log32.double() - log64).abs().max()),
                                         "len0": float((len32.double() - len64).abs().max())}}
            for vname, ev, mv, env in variants:
                eng.set_variant(ev, mv)
                os.environ.update(env)
                try:
                    eps, logits, len0 = eng.predict_scores(f, ty, le, an, t_c, off, edges=edges)
                    st = eng.check_status()
                finally:
                    for k in env:
                        del os.environ[k]
                if vname.startswith("basis_form") and (st["conv_variant"] != 2 or st["conv_cross_fp8"] != (1 if "fp8" in vname else 0)):
                    rec[vname] = {"skipped": f"this batch ran message path {st['conv_variant']} with conv_cross_fp8 = {st['conv_cross_fp8']}"}
                    continue
                e, l, g = eps.cpu(), logits.cpu(), len0.cpu()
                r = {"kernels": [st["edge_kernel"], st["mlp_kernel"]], "message_path": st["conv_variant"], "fp8_cross": st["conv_cross_fp8"], "basis_row_bytes": st["basis_row_bytes"],
                     "eps": float((e - eps32).abs().max()), "logits": float((l - log32).abs().max()),
                     "len0": float((g - len32).abs().max()),
                     "eps_vs_f64": float((e.double() - eps64).abs().max()),
                     "logits_vs_f64": float((l.double() - log64).abs().max()),
                     "len0_vs_f64": float((g.double() - len64).abs().max())}
                rec[vname] = r
                for k in worst[vname]:
                    worst[vname][k] = max(worst[vname][k], r[k])
            eng.set_variant(4, 3)
            report["cases"][name]["t"][str(t)] = rec
            print(f"{name} t={t}: fp16x3 d(eps,logits,len0) = {rec['default_fp16x3']['eps']:.2e} "
                  f"{rec['default_fp16x3']['logits']:.2e} {rec['default_fp16x3']['len0']:.2e} | fp32-mfma "
                  f"{rec['fp32_mfma']['eps']:.2e} {rec['fp32_mfma']['logits']:.2e} {rec['fp32_mfma']['len0']:.2e} "
                  f"| oracle f32 vs f64 logits {rec['oracle_f32_vs_f64']['logits']:.2e}  ({time.time() - t0:.1f} s)",
                  flush=True)
    report["worst"] = worst
    report["ratio_fp16x3_over_fp32mfma_vs_f64"] = {
        k: worst["default_fp16x3"][k + "_vs_f64"] / max(worst["fp32_mfma"][k + "_vs_f64"], 1e-30)
        for k in ("eps", "logits", "len0")}
    report["ratio_basis_form_fp8_cross_over_fp32mfma_vs_f64"] = {
        k: worst["basis_form_fp8_cross"][k + "_vs_f64"] / max(worst["fp32_mfma"][k + "_vs_f64"], 1e-30)
        for k in ("eps", "logits", "len0")}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(report, fh, indent=1)
    print("worst:", json.dumps(worst))
    print("wrote", args.out)


if __name__ == "__main__":
    main()

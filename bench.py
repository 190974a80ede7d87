#!/usr/bin/env python3
"""Benchmark of the reverse-diffusion sampling step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one iteration of the sampling loop (diffusion/diffusion_loss.py:318-347 of the reference)
over one batch: predict_scores (PBC neighbour list + Ponita score network) + the three noise draws +
the reverse updates.  Workload at every N: BASELINE.json configs[1] per GPU -- 256 crystals x 20 atoms,
T = 1000, fp32, synthetic 1.17M-parameter checkpoint (C=128, O=16, D=256, L=5, S=90), state drawn like
the sampler's start.  Crystals are independent, so ranks hold disjoint sub-batches and there is no
data-path collective (weak scaling); RCCL is used only for the timing barrier / max-over-ranks.

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (spec)
# The edge kernel evaluates every fp32 product as six bf16 MFMA products (exact 8+8+8-bit operand splits,
# fp32 accumulation), so its matrix-pipe roof in fp32-equivalent FLOP/s is the bf16 peak / 6.
BF16X6_EQUIV_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
# default kernels: three fp16 MFMA products per fp32 product (two 11-bit operand planes) -> fp16 peak / 3
F16X3_EQUIV_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 3.0


def edge_kernel_flops_per_row(C=128, D=256, L=5):
    """Algorithmic FLOPs the edge kernel replaces per (edge, orientation) row:
    basis MLP 2*(258*C + C*D) + per-layer kernel projection 2*L*D*C."""
    return 2 * (258 * C + C * D) + 2 * L * D * C


def step_flops_per_atom(k=8, S=90, C=128, D=256, L=5, O=16, W=4):
    """F = E*O*[2(258C + CD) + 2L*DC + 2LC] + N*O*[L(2OC + 4W C^2) + 2(S+78)C + 2LC(S+4)]
    per atom with E = k edges per atom  (= 83.0 MFLOP at the defaults)."""
    per_row = 2 * (258 * C + C * D) + 2 * L * D * C + 2 * L * C
    per_node_ori = L * (2 * O * C + 4 * W * C * C) + 2 * (S + 78) * C + 2 * L * C * (S + 4)
    return k * O * per_row + O * per_node_ori


def config_label(B, n):
    """BASELINE.json config the (crystals per GPU, atoms per crystal) pair corresponds to."""
    return {(1, 8): "BASELINE configs[0]", (256, 20): "BASELINE configs[1]", (1024, 20): "BASELINE configs[2] (8192 / 8 GPUs)",
            (1024, 64): "BASELINE configs[3]"}.get((B, n), "custom size")


def measured_traffic(B, n):
    """HBM bytes per launch of the edge kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 FETCH correction applied) -- only valid for the workload they were taken on."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic_pmc.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        if d.get("crystals_per_gpu") == B and d.get("atoms_per_crystal") == n:
            return d["edge_kernel_hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def log(msg):
    sys.stderr.write(f"[bench] {msg}\n")
    sys.stderr.flush()


def host_cores():
    """CPU threads this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return int(os.environ.get("ARREAU_CPU_THREADS", min(n, 32)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=256, help="crystals per GPU (configs[1]: 256)")
    ap.add_argument("--atoms", type=int, default=20, help="atoms per crystal (configs[1]: 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=2)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # ARREAU_BENCH_BACKEND=gloo + ARREAU_BENCH_ONE_DEVICE=1 rehearse the multi-rank path on a one-GPU box
    backend = os.environ.get("ARREAU_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("ARREAU_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from arreau_amd import _hip, build
    build.build(verbose=False)
    from arreau_amd.checkpoint import make_synthetic_model
    from arreau_amd.diffusion.diffusion_helpers import crystal_offsets

    S, T = 90, 1000
    B, n = args.batch_per_gpu, args.atoms
    N = B * n
    model = make_synthetic_model(S=S, seed=1234).to(dev)  # same weights on every rank
    eng = model.engine()

    # sampler-start state (diffusion_loss.py:294-316), different crystals per rank
    torch.manual_seed(1000 + rank)
    rng = np.random.RandomState(1000 + rank)
    angles = torch.tensor(np.stack([np.full(B, 90.0), rng.uniform(90, 180, B), np.full(B, 90.0)], 1), dtype=torch.float32)
    lengths = torch.randn(B, 3)
    frac = torch.randn(N, 3)
    f32 = dict(device=dev, dtype=torch.float32)
    frac_d, len_d, ang_d = frac.to(**f32), lengths.to(**f32), angles.to(**f32)
    types_d = torch.full((N,), S - 1, device=dev, dtype=torch.int32)
    off_d = crystal_offsets(torch.full((B,), n), dev)
    lat_d = torch.zeros(B, 3, 3, **f32)
    t_d = torch.empty(B, device=dev, dtype=torch.int32)
    gen = torch.Generator(device=dev).manual_seed(77 + rank)

    # Random-init weights predict unphysical cell lengths, so a free-running state drifts to huge, sparse
    # cells within a few steps (E/N falls from 8 to ~4), which would shrink the timed work.  A trained model
    # keeps the cell compact (E/N = 8.00 in the sampler regime, SURVEY.md symbol table), so every step re-imposes
    # the sampler-start lengths; coordinates and atom types evolve freely.  edges_per_atom_* reports the result.
    len_start = len_d.clone()
    timestep = [T - 1]

    def one_step():
        t = timestep[0]
        t_d.fill_(t)
        len_d.copy_(len_start)
        eps, logits, len0 = eng.predict_scores(frac_d, types_d, len_d, ang_d, t_d, off_d)
        z_l = torch.randn((B, 3), generator=gen, **f32)
        z_f = torch.randn((N, 3), generator=gen, **f32)
        u_t = torch.rand((N, S), generator=gen, **f32)
        eng.reverse_step(frac_d, types_d, len_d, ang_d, t_d, off_d, eps, logits, len0, z_l, z_f, u_t, lat_d)
        timestep[0] = t - 1 if t > 1 else T - 1

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def degree_sum():
        _, _, _, edges = eng.predict_scores(frac_d, types_d, len_d, ang_d, t_d, off_d, return_edges=True)
        return int(edges[0].sum().item())

    log(f"rank {rank}: model packed, state ready (B={B}, n={n}); warm-up {args.warmup} steps")
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    log("warm-up done; timing")
    t_d.fill_(timestep[0])
    e_start = degree_sum()
    _hip.check(_hip.lib().arreau_profile_edge_kernel(1), "profile on")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    import ctypes
    mean_ms, launches = ctypes.c_double(), ctypes.c_int64()
    _hip.check(_hip.lib().arreau_edge_kernel_time_ms(ctypes.byref(mean_ms), ctypes.byref(launches)), "edge time")
    _hip.check(_hip.lib().arreau_profile_edge_kernel(0), "profile off")
    e_end = degree_sum()

    if dist is not None:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
        ms_per_step = 1e3 * elapsed / args.steps
        crystal_steps_per_s = world * B * args.steps / elapsed
        e_mean = 0.5 * (e_start + e_end)
        edge_flops = e_mean * 16 * edge_kernel_flops_per_row()
        edge_tflops = edge_flops / (mean_ms.value * 1e-3) / 1e12 if mean_ms.value > 0 else 0.0
        step_flops = step_flops_per_atom() * N * (e_mean / (8.0 * N))  # scaled by the edge density actually seen
        variant = int(os.environ.get("ARREAU_EDGE_VARIANT", "4"))
        if variant == 4:
            edge_kernel_name = ("edge_kernel_f16x3<128,256> (pair invariants + basis MLP + 5 kernel projections; "
                                "fp32 products as 3 fp16 MFMA products, fp32 accumulate)")
            edge_peak = F16X3_EQUIV_PEAK_TFLOPS
            edge_peak_note = "fp32-equivalent roof of the split scheme: dense fp16 MFMA 2500 TFLOP/s / 3 products"
        elif variant == 3:
            edge_kernel_name = ("edge_kernel_bf16x6<128,256> (pair invariants + basis MLP + 5 kernel projections; "
                                "fp32 products as 6 bf16 MFMA products, fp32 accumulate)")
            edge_peak = BF16X6_EQUIV_PEAK_TFLOPS
            edge_peak_note = "fp32-equivalent roof of the split scheme: dense bf16 MFMA 2500 TFLOP/s / 6 products"
        else:
            edge_kernel_name = "edge_kernel<128,256> (v_mfma_f32_32x32x2_f32)"
            edge_peak = MFMA_F32_PEAK_TFLOPS
            edge_peak_note = "fp32-input MFMA peak"
        edge_traffic = measured_traffic(B, n)
        out = {
            "metric": "denoising steps/sec (crystal-steps, whole node)",
            "value": crystal_steps_per_s,
            "unit": "crystal-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 inputs/outputs/accumulation; each fp32 product of the dense layers is evaluated as 3 fp16 "
                          "MFMA products (two 11-bit operand planes, f16x3.h) -- fp32-grade accuracy, parity 1e-5 vs the "
                          "fp32 CPU path (tests/test_gpu_parity.py); ARREAU_EDGE_VARIANT=0 ARREAU_MLP_VARIANT=0 selects "
                          "the plain fp32-MFMA kernels",
            "data": "synthetic",
            "config": {
                "workload": f"{config_label(B, n)} per GPU: batch={B} crystals x {n} atoms, 1000-step sampler "
                            f"(T=1000, 999 network evaluations per crystal), fp32",
                "crystals_per_gpu": B, "atoms_per_crystal": n, "num_timesteps": T,
                "checkpoint": "synthetic 1.17M-param (S=90,C=128,O=16,D=256,L=5,k=8,R=5), seed 1234",
                "edges_per_atom_start": e_start / N, "edges_per_atom_end": e_end / N,
                "parallelism": f"replicas x{world}, disjoint sub-batches, no data-path collective",
            },
            "batch_steps_per_sec": world * args.steps / elapsed,
            "crystals_per_min": 60.0 * crystal_steps_per_s / (T - 1),
            "step_tflops_algorithmic": world * step_flops / (ms_per_step * 1e-3) / 1e12,
            "roofline": {
                "kernel": edge_kernel_name,
                "bound": "mfma", "achieved": edge_tflops, "peak": edge_peak, "unit": "TFLOP/s",
                "frac": edge_tflops / edge_peak, "traffic": edge_traffic,
                "peak_note": edge_peak_note, "frac_of_fp32_mfma_peak": edge_tflops / MFMA_F32_PEAK_TFLOPS,
                "avg_launch_ms": mean_ms.value, "launches_timed": int(launches.value),
                "algorithmic_flops_per_launch": edge_flops,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, B, n, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(model, B, n, steps):
    """The oracle (CPU restatement of the reference step, pure PyTorch, all host cores) timed on a bounded
    sample of the SAME workload: `steps` full steps of the B x n batch after one warm-up step."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import sampler as OS
    from tests.helpers import oracle_from_module

    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads")
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        om = oracle_from_module(model, dtype)
        S, T = om.hp["S"], om.hp["T"]
        torch.manual_seed(0)
        np.random.seed(0)
        frac, types, lengths, angles, num_atoms = OS.init_state(om, n, B, dtype)
        batch = torch.arange(B).repeat_interleave(n)
        N = B * n
        times = []
        t = T - 1
        for it in range(steps + 1):
            t0 = time.perf_counter()
            scores = OS.predict_scores(om, frac, F.one_hot(types, S), torch.full((N,), t), num_atoms, lengths,
                                       angles, batch)
            noise = OS.StepNoise(torch.randn(B, 3, dtype=dtype), torch.randn(N, 3, dtype=dtype),
                                 torch.rand(N, S, dtype=dtype))
            frac, types, lengths, _ = OS.reverse_step(om, frac, types, lengths, angles, num_atoms, scores, t, noise)
            times.append(time.perf_counter() - t0)
            log(f"cpu baseline {tag} step {it}: {times[-1]:.2f} s")
            t -= 1
        per_step = float(np.mean(times[1:]))
        res[tag] = B / per_step
        if tag == "f32" and per_step * (steps + 1) > 40:
            break  # keep the default run bounded on slow hosts
    return {
        "value": res["f32"], "unit": "crystal-steps/s", "cores": cores, "kind": "port",
        "sample": f"{steps} full steps of batch={B} x {n} atoms after 1 warm-up (oracle, torch CPU, {cores} threads)",
        "value_f64": res.get("f64"),
    }


if __name__ == "__main__":
    main()

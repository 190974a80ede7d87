#!/usr/bin/env python3
"""Benchmark of the reverse-diffusion sampling step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Both forms run N ranks, one per GPU: with `--gpus N > 1` and no WORLD_SIZE in the environment bench.py starts the N
rank processes itself (before anything touches a GPU), otherwise it is one rank of an existing launch.

A "step" is one iteration of the sampling loop (diffusion/diffusion_loss.py:318-347 of the reference) over one batch:
predict_scores (PBC neighbour list + Ponita score network) + the three noise draws + the reverse updates.  Workload at
every N (default): BASELINE.json configs[1] per GPU -- 256 crystals x 20 atoms, T = 1000, fp32, synthetic
1.17M-parameter checkpoint (C=128, O=16, D=256, L=5, S=90), state drawn like the sampler's start.  `--config c1|c3|c4`
selects the other BASELINE configurations (c1: 1 crystal x 8 atoms, T = 100; c3: 1024 x 20 per GPU; c4: 1024 x 64).
Crystals are independent, so ranks hold disjoint sub-batches and there is no data-path collective (weak scaling); the
process group is used only for the timing barrier / max-over-ranks.

The K steps are ONE arreau_sample_loop call per stretch of timesteps (Philox noise inside the update kernels, timestep on
the device, nothing on the host between steps), timed twice: as eager launches -- hipEvents bracket every edge-kernel
launch there: the roofline figure -- and as hipGraph replay of the captured step (bit-identical results).  `value` is the loop PONITA_DIFFUSION.sample runs by default for the
configuration (graph replay for samplers of at least 200 steps, i.e. every 1000-step config; eager for the 100-step
single-crystal config): `loop_mode` names it, `eager_loop` / `graph_loop` hold both.

Prints ONE JSON line on rank 0 (see the keys at the bottom).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA = vector peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (spec)
# bf16x6: every fp32 product as six bf16 MFMA products -> roof in fp32-equivalent FLOP/s = 16-bit peak / 6
BF16X6_EQUIV_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0
# default kernels: three fp16 MFMA products per fp32 product (two 11-bit operand planes) -> fp16 peak / 3
F16X3_EQUIV_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 3.0

CONFIGS = {  # name -> (crystals per GPU, atoms per crystal, num_timesteps); c5 = the training step (run_rank_c5)
    "c1": (1, 8, 100), "c2": (256, 20, 1000), "c3": (1024, 20, 1000), "c4": (1024, 64, 1000), "c5": (64, 0, 1000)}


def edge_kernel_flops_per_row(C=128, D=256, L=5):
    """Algorithmic FLOPs the edge kernel replaces per (edge, orientation) row:
    basis MLP 2*(258*C + C*D) + per-layer kernel projection 2*L*D*C."""
    return 2 * (258 * C + C * D) + 2 * L * D * C


def conv_pass_bytes(N, C=128, O=16):
    """ALGORITHMIC HBM bytes of one conv pass ("each layer = one conv pass (read x, write x_1) + one node
    pass"; edge kernels never materialised): 2 * N * O * C * 4.  What the message kernel moves beyond that -- the stashed
    basis it streams, the sender rows it gathers -- is the design's own traffic and is reported as such, not as work."""
    return 2 * N * O * C * 4


def step_bytes(N, E, B, S=90, C=128, L=5, O=16):
    """algorithmic bytes of one step = 4 * [(5L + 1) N O C + 2 E 6 + N (S + 4) + 4 N + 6 B]
    (213.8 KB per atom-step at E = 8 N: 1.095 GB at 256 x 20)."""
    return 4 * ((5 * L + 1) * N * O * C + 2 * E * 6 + N * (S + 4) + 4 * N + 6 * B)


def stash_bytes_per_launch(N, row_bytes, k=8, O=16):
    """Bytes of the stashed basis one launch of the message kernel streams: ALL k slots of every receiver (the kernel
    copies a receiver's whole block; unused slots included), row_bytes per (slot, orientation) row as the library reports
    it (arreau_model_status: 768 = fp16 + fp8 e4m3 plane, 1024 = two fp16 planes)."""
    return N * k * O * row_bytes


def conv_proj_flops_per_row(C=128, D=256):
    """Algorithmic FLOPs of the kernel projection of one layer per (edge, orientation) row (conv.py:110)."""
    return 2 * D * C


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


def measured_traffic(B, n, key="edge_kernel_hbm_bytes_per_launch"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs,
    gfx950 FETCH correction applied) -- only valid for the workload they were taken on: profiles/hbm_traffic_pmc.json
    holds one entry per (crystals per GPU, atoms per crystal)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic_pmc.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        for entry in (d if isinstance(d, list) else [d]):
            if entry.get("crystals_per_gpu") == B and entry.get("atoms_per_crystal") == n:
                return entry.get(key)
    except (OSError, ValueError, KeyError):
        pass
    return None


def log(msg):
    sys.stderr.write(f"[bench] {msg}\n")
    sys.stderr.flush()


_RESULT_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries print there too (gloo announces its connections, ROCm warns
    about missing files), so a rank process keeps the real stdout for the result line and points file descriptor 1 at
    stderr for everything else, C++ writers included."""
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _RESULT_OUT


def emit(result):
    out = claim_stdout()
    out.write(json.dumps(result) + "\n")
    out.flush()


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None, help="BASELINE configuration (default c2)")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="crystals per GPU (configs[1]: 256)")
    ap.add_argument("--atoms", type=int, default=None, help="atoms per crystal (configs[1]: 20)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-variant", action="store_true", help="skip the second timed loop on the fp32-MFMA kernels")
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--hidden-dim", type=int, default=128, help="c5 only: hidden_dim of the trained network (the reference's "
                    "`make train` preset is 200; shapes other than 128 run on the shape-general kernels)")
    ap.add_argument("--no-full-sampler", action="store_true", help="skip the measured run of the whole T-1 step sampler")
    ap.add_argument("--mlp-variant", type=int, default=3, help="ConvNext kernel set (3: default, picks its small-launch form by "
                    "size; 4: always the small-launch form; 0-2: cross-check arithmetic)")
    ap.add_argument("--no-graph-loop", action="store_true", help="time only the eager loop (profiling runs: per-kernel "
                    "statistics and PMC counters of whole-batch launches only); implies --eager-value")
    ap.add_argument("--eager-value", action="store_true", help="report the eager loop as `value` even where the product "
                    "defaults to graph replay")
    ap.add_argument("--no-other-configs", action="store_true", help="default run only: skip the short legs of the other BASELINE "
                    "configurations (c1, c4, c5) that the one-GPU default run appends as `other_configs`")
    ap.add_argument("--groups", type=int, default=0, help="experiment, needs ARREAU_ALLOW_MULTISTREAM=1: slices of the batch on separate streams "
                    "(0 = the product: one stream; 2 was 3 %% faster at 256 x 20 but is not run-to-run reproducible)")
    args = ap.parse_args(argv)
    B, n, T = CONFIGS[args.config or "c2"]
    args.batch_per_gpu = args.batch_per_gpu or B
    args.atoms = args.atoms or n
    args.T = T
    return args


# --------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` (no torchrun) starts N rank processes, one per GPU
# --------------------------------------------------------------------------------------------------------------
def visible_gpu_count():
    """GPUs this process could use, counted WITHOUT touching HIP (the launcher parent must stay a process that never
    initialised the GPU runtime): KFD topology nodes with SIMDs (CPU nodes have simd_count 0), narrowed by the
    *_VISIBLE_DEVICES lists a launcher may have set."""
    import glob
    n = 0
    props = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not props:  # no KFD topology in this mount namespace: one DRM render node per GPU
        n = len(glob.glob("/dev/dri/renderD*"))
    for prop in props:
        try:
            with open(prop) as fh:
                for line in fh:
                    key, _, val = line.partition(" ")
                    if key == "simd_count":
                        n += int(val) > 0
                        break
        except OSError:
            pass
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        lst = os.environ.get(var)
        if lst is not None:
            n = min(n, len([x for x in lst.split(",") if x.strip() != ""]))
    return n


def launch_ranks(n, argv, stub=False):
    """Start `n` copies of this script as ranks 0..n-1 (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment) BEFORE this process touches a GPU; rank 0's stdout (the JSON line) is passed through.  Returns the
    exit code (non-zero if any rank failed; the others are then stopped by PID)."""
    if not stub and os.environ.get("ARREAU_BENCH_ONE_DEVICE", "0") != "1":  # (rehearsal mode shares one GPU)
        have = visible_gpu_count()  # sysfs + environment only: this parent never calls into HIP
        if have < n:
            log(f"--gpus {n} requested but only {have} GPU(s) are visible")
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ARREAU_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:  # a failed rank leaves the others waiting at a barrier: stop them (exact PIDs)
                    q.terminate()
    return rc


def main():
    args = parse_args()
    stub = os.environ.get("ARREAU_BENCH_STUB", "0") == "1"  # CPU rehearsal of the launcher / timing protocol (tests)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], stub=stub))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    claim_stdout()
    if stub:
        return run_stub(args, rank, world)
    if args.config == "c5":
        return run_rank_c5(args, rank, local_rank, world)
    return run_rank(args, rank, local_rank, world)


def run_stub(args, rank, world):
    """The launcher / barrier / max-over-ranks protocol with a stand-in step on the CPU (gloo): what the world_size-2
    CPU test exercises; prints the same JSON skeleton."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    for _ in range(args.warmup):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))  # rank 1 is slower: the reported time must be the max
    local = time.perf_counter() - t0  # this rank's own work, before it waits for the others
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [local]
    if world > 1:
        tt = torch.tensor([local, elapsed], dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank = [float(g[0]) for g in gathered]
        elapsed = max(float(g[1]) for g in gathered)
    ranks = {"backend": dist.get_backend() if world > 1 else None, "world_size": dist.get_world_size() if world > 1 else 1,
             "data_path_collectives": 0,
             "per_rank": gather_json(dist if world > 1 else None, "cpu", world, {"rank": rank, "pid": os.getpid(),
                                                                               "hostname": socket.gethostname()})}
    if rank == 0:
        emit({"metric": "stub", "value": world * args.batch_per_gpu * args.steps / elapsed, "n_gpus": world,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
              "per_rank_ms": [1e3 * e / args.steps for e in per_rank], "ranks": ranks})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def device_identity(torch, dev):
    """Name, UUID and PCI address of this rank's GPU (what distinguishes N MI355X of one node in the record)."""
    p = torch.cuda.get_device_properties(dev)
    out = {"device": p.name, "device_index": dev.index, "hostname": socket.gethostname(),
           "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}
    for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id", "gcnArchName", "multi_processor_count"):
        v = getattr(p, key, None)
        if v is not None:
            out[key] = str(v) if key == "uuid" else v
    return out


def gather_json(dist, where, world, obj, cap=2048):
    """every rank's small record on every rank (fixed-size byte tensors through all_gather: works on RCCL and gloo alike)"""
    if dist is None:
        return [obj]
    import torch
    raw = json.dumps(obj).encode()[:cap]
    buf = torch.zeros(cap + 4, dtype=torch.uint8)
    buf[:4] = torch.tensor(list(len(raw).to_bytes(4, "little")), dtype=torch.uint8)
    buf[4:4 + len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    buf = buf.to(where)
    got = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(got, buf)
    out = []
    for g in got:
        g = g.cpu()
        n = int.from_bytes(bytes(g[:4].tolist()), "little")
        out.append(json.loads(bytes(g[4:4 + n].tolist()).decode()))
    return out


def other_config_legs():
    """Short legs of the other BASELINE configurations, each in a child process of its own (a fresh library state, the same
    `python bench.py --config ...` a user would run; the parent has finished its GPU work), so that ONE default run -- the
    command the driver times -- carries a measured figure for every configuration: c1 (configs[0]: 99 eager steps = one whole
    sampler), c4 (configs[3]: graph replay at 1024 x 64), c5 (configs[4]: training steps).  No CPU baseline in the legs;
    `steps` / `warmup` of the headline are untouched.  Each record: ms_per_step, value, unit, loop_mode, the leg's own
    `roofline` block and config.workload."""
    legs = (("c1", ["--steps", "99", "--warmup", "20"]), ("c4", ["--steps", "5", "--warmup", "2"]), ("c5", ["--steps", "30", "--warmup", "8"]))
    out = {}
    for name, extra in legs:
        cmd = [sys.executable, os.path.abspath(__file__), "--config", name, "--no-cpu-baseline", "--no-fp32-variant",
               "--no-full-sampler", "--no-other-configs"] + extra
        t0 = time.perf_counter()
        try:
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ARREAU_BENCH_CHILD")}
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, timeout=240, text=True)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out[name] = {k: d.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "loop_mode", "forward_backward_ms",
                                               "crystals_per_min", "roofline")}
            out[name]["config"] = {"workload": d.get("config", {}).get("workload")}
            out[name]["leg_wall_s"] = time.perf_counter() - t0
        except Exception as exc:  # a failed leg must not cost the headline
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        log(f"other_configs {name}: {out[name].get('ms_per_step')} ms per step ({time.perf_counter() - t0:.1f} s)")
    return out


def run_rank(args, rank, local_rank, world):
    import ctypes

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # ARREAU_BENCH_BACKEND=gloo + ARREAU_BENCH_ONE_DEVICE=1 rehearse the multi-rank path on a one-GPU box
    backend = os.environ.get("ARREAU_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("ARREAU_BENCH_ONE_DEVICE", "0") == "1"
    if not one_device and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} GPU(s) visible")
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    # ARREAU_BENCH_FORCE_DIST=1: a ONE-rank run also creates the process group and goes through every collective of the N-rank path
    # (barriers, the timing all-gather, the rank records; c5: the gradient all-reduce) -- the only way to execute the RCCL branch on
    # a one-GPU box (two ranks cannot share a device under RCCL)
    force_dist = world == 1 and os.environ.get("ARREAU_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from arreau_amd import _hip, build
    build.build(verbose=False)
    from arreau_amd.checkpoint import make_synthetic_model
    from arreau_amd.diffusion.diffusion_helpers import crystal_offsets

    S, T = 90, args.T
    B, n = args.batch_per_gpu, args.atoms
    N = B * n
    model = make_synthetic_model(S=S, seed=1234, num_timesteps=T).to(dev)  # same weights on every rank
    eng = model.engine()
    eng.set_variant(mlp=args.mlp_variant)

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

    # Random-init weights predict unphysical cell lengths, so a free-running state drifts to huge, sparse cells within a
    # few steps (E/N falls from 8 to ~4), which would shrink the timed work.  A trained model keeps the cell compact
    # (E/N = 8.00 in the sampler regime, SURVEY.md symbol table), so the bench samples at FIXED cell lengths (the
    # sampler-start ones; arreau_sample_loop's d_fixed_lengths); coordinates and atom types evolve freely.
    # edges_per_atom_* reports the density actually processed.
    len_start = len_d.clone()
    seed = 77 + rank
    timestep = [T - 1]
    # crystal-aligned slices on separate streams for the graph loop (0: what PONITA_DIFFUSION.sample chooses: 2 from 4096 atoms)
    groups = args.groups if args.groups > 0 else 1  # (the product's default: one stream; see DiffusionLoss.sample)
    eng.set_batch_layout(torch.full((B,), n), groups=groups)

    def run_steps(k, use_graph=False):
        """k iterations of the product's sampling loop (score network, in-kernel Philox noise, reverse updates), enqueued
        by ONE arreau_sample_loop call per stretch of consecutive timesteps (the loop wraps from t = 1 back to T - 1)."""
        while k > 0:
            t = timestep[0]
            n = min(k, t)
            eng.sample_loop(frac_d, types_d, len_d, ang_d, off_d, t, n, seed, None, lat_d, use_graph=use_graph,
                            fixed_lengths=len_start)
            k -= n
            timestep[0] = t - n if t - n >= 1 else T - 1

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def degree_sum():
        t_d.fill_(timestep[0])
        _, _, _, edges = eng.predict_scores(frac_d, types_d, len_d, ang_d, t_d, off_d, return_edges=True)
        return int(edges[0].sum().item())

    local_elapsed = [0.0]

    def timed_loop(steps, use_graph=False):
        sync()
        t0 = time.perf_counter()
        run_steps(steps, use_graph)
        torch.cuda.synchronize(dev)
        local_elapsed[0] = time.perf_counter() - t0  # this rank's own work, before it waits for the others
        sync()
        return time.perf_counter() - t0

    log(f"rank {rank}: model packed, state ready (B={B}, n={n}, T={T}); warm-up {args.warmup} steps")
    run_steps(args.warmup)
    torch.cuda.synchronize(dev)
    log("warm-up done; timing")
    def gather(local, total):
        """per-rank own times and the barrier-to-barrier time, MAX over ranks"""
        if dist is None:
            return [local], total
        tt = torch.tensor([local, total], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        return [float(g[0]) for g in gathered], max(float(g[1]) for g in gathered)

    # Loop 1, eager: one launch per kernel and step, hipEvents around every edge-kernel launch (the roofline figure).
    e_start = degree_sum()
    _hip.check(_hip.lib().arreau_profile_edge_kernel(1), "profile on")
    el_eager = timed_loop(args.steps)
    mean_ms, launches = ctypes.c_double(), ctypes.c_int64()
    _hip.check(_hip.lib().arreau_edge_kernel_time_ms(ctypes.byref(mean_ms), ctypes.byref(launches)), "edge time")
    conv_ms, conv_launches = ctypes.c_double(), ctypes.c_int64()
    _hip.check(_hip.lib().arreau_conv_kernel_time_ms(ctypes.byref(conv_ms), ctypes.byref(conv_launches)), "conv time")
    _hip.check(_hip.lib().arreau_profile_edge_kernel(0), "profile off")
    e_end = degree_sum()
    # (ARREAU_BENCH_TIMING_ONLY=1: builder experiments with libraries that compute wrong numbers on purpose -- tools/exp --;
    # the status word is read but not enforced and the line says so: such a line is not a result)
    timing_only = os.environ.get("ARREAU_BENCH_TIMING_ONLY", "0") == "1"
    check = (lambda: eng.status(reset=True)) if timing_only else eng.check_status
    status = check()  # raises on non-finite outputs / clamped indices; names the kernels that really ran
    per_rank_eager, el_eager = gather(local_elapsed[0], el_eager)

    # Loop 2, hipGraph replay of the captured step (arreau_sample_loop use_graph = 1) on one stream: bit-identical results,
    # what PONITA_DIFFUSION.sample runs by default for samplers of at least 200 steps (DiffusionLoss.sample).
    # (--groups 2: the opt-in pipelined slices on two streams.)
    if args.no_graph_loop:
        el_graph, per_rank_graph = el_eager, per_rank_eager
    else:
        run_steps(3, use_graph=True)  # warm-up of the graph path (capture, instantiation, stream / event creation)
        el_graph = timed_loop(args.steps, use_graph=True)
        per_rank_graph, el_graph = gather(local_elapsed[0], el_graph)
        check()

    # `value` is the loop the product runs for THIS configuration: the same policy as DiffusionLoss.sample's default
    # (graph replay from 200 sampler steps on; the 100-step single-crystal config runs eagerly).
    production_graph = (T - 1) >= 200 and not args.eager_value and not args.no_graph_loop
    elapsed, per_rank = (el_graph, per_rank_graph) if production_graph else (el_eager, per_rank_eager)
    loop_mode = ("hipGraph replay of the captured step" + (f", {groups} pipelined slices per GPU" if groups > 1 else "")
                 if production_graph
                 else "eager launches")
    eager_loop = {"steps": args.steps, "ms_per_step": 1e3 * el_eager / args.steps,
                  "note": "one launch per kernel and step; the edge-kernel hipEvents (roofline) are taken here"}
    graph_loop = None if args.no_graph_loop else {"steps": args.steps, "ms_per_step": 1e3 * el_graph / args.steps, "slices": groups,
                  "note": "hipGraph replay of the captured step on one stream (bit-identical to the eager loop): "
                          "PONITA_DIFFUSION.sample's default for >= 200 sampler steps"}

    # The headline metric measured rather than extrapolated: ONE call of PONITA_DIFFUSION.sample for the whole sampler
    # (T - 1 network evaluations of this batch; host-side initial draws, the library loop in its default mode, the
    # SampleResult copied back to the host), wall clock.  fixed_cell keeps the synthetic checkpoint's cells at the
    # sampler-start density, as in the timed loop above.
    full_sampler = None
    if world == 1 and not args.no_full_sampler and B * n <= 256 * 20:
        from arreau_amd.diffusion.inference.visualize_crystal import VisualizationSetting
        torch.manual_seed(5)
        np.random.seed(5)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        res = model.sample(n, B, VisualizationSetting.NONE, False, fixed_cell=True)
        wall = time.perf_counter() - t0
        assert res.frac_x.shape == (N, 3) and np.isfinite(res.frac_x).all()
        full_sampler = {"wall_s": wall, "steps": T - 1, "crystals": B, "crystals_per_min": 60.0 * B / wall,
                        "ms_per_step_incl_everything": 1e3 * wall / (T - 1)}
        eng.set_batch_layout(torch.full((B,), n), groups=groups)

    # second, short timed loop on the exact fp32-MFMA kernels (v_mfma_f32_32x32x2_f32): what the same step costs
    # without the split-precision scheme, priced against the 157.3 TFLOP/s fp32 matrix peak
    fp32_variant = None
    if not args.no_fp32_variant and world == 1:
        eng.set_variant(0, 0)
        k32 = max(3, min(10, args.steps))
        run_steps(2)
        _hip.check(_hip.lib().arreau_profile_edge_kernel(1), "profile on")
        el32 = timed_loop(k32)
        m32, l32 = ctypes.c_double(), ctypes.c_int64()
        _hip.check(_hip.lib().arreau_edge_kernel_time_ms(ctypes.byref(m32), ctypes.byref(l32)), "edge time")
        _hip.check(_hip.lib().arreau_profile_edge_kernel(0), "profile off")
        st32 = check()
        eng.set_variant(4, args.mlp_variant)
        e32 = degree_sum()
        fl32 = e32 * 16 * edge_kernel_flops_per_row()
        fp32_variant = {"kernels": [st32["edge_kernel"], st32["mlp_kernel"]], "steps": k32,
                        "ms_per_step": 1e3 * el32 / k32, "edge_avg_launch_ms": m32.value,
                        "edge_tflops": fl32 / (m32.value * 1e-3) / 1e12 if m32.value > 0 else 0.0,
                        "frac_of_157.3": (fl32 / (m32.value * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if m32.value > 0 else 0.0}

    # What lets the record prove N ranks on N distinct devices (and which library carried the barrier): every rank reports
    # its device identity and the work it really processed; no collective touches the data path.
    ranks_info = {"backend": (dist.get_backend() if dist is not None else None),
                  "world_size": (dist.get_world_size() if dist is not None else 1),
                  "data_path_collectives": 0,
                  "note": "the process group carries only the timing barriers and the gathers of these records",
                  "per_rank": gather_json(dist, dev if backend == "nccl" else "cpu", world,
                                          dict(device_identity(torch, dev), rank=rank, local_rank=local_rank, pid=os.getpid(),
                                               degree_sum_start=e_start, degree_sum_end=e_end, crystals=B, atoms=N))}
    if rank == 0:
        log(f"timed region: {args.steps} steps in {elapsed:.3f} s")
        ms_per_step = 1e3 * elapsed / args.steps
        crystal_steps_per_s = world * B * args.steps / elapsed
        e_mean = 0.5 * (e_start + e_end)
        # one edge-kernel launch per slice of the batch and step: the algorithmic FLOPs of a launch are its share
        launches_per_step = max(1, round(launches.value / args.steps))
        edge_flops = e_mean * 16 * edge_kernel_flops_per_row() / launches_per_step
        edge_tflops = edge_flops / (mean_ms.value * 1e-3) / 1e12 if mean_ms.value > 0 else 0.0
        step_flops = step_flops_per_atom() * N * (e_mean / (8.0 * N))  # scaled by the edge density actually seen
        ran = status["edge_kernel"]
        if ran == "fp16x3":
            edge_kernel_name = ("edge_kernel_f16x3<128,256> (pair invariants + basis MLP + 5 kernel projections; "
                                "fp32 products as 3 fp16 MFMA products, fp32 accumulate)")
            edge_peak = F16X3_EQUIV_PEAK_TFLOPS
            edge_peak_note = "fp32-equivalent roof of the split scheme: dense fp16 MFMA 2500 TFLOP/s / 3 products"
        elif ran == "bf16x6":
            edge_kernel_name = ("edge_kernel_bf16x6<128,256> (pair invariants + basis MLP + 5 kernel projections; "
                                "fp32 products as 6 bf16 MFMA products, fp32 accumulate)")
            edge_peak = BF16X6_EQUIV_PEAK_TFLOPS
            edge_peak_note = "fp32-equivalent roof of the split scheme: dense bf16 MFMA 2500 TFLOP/s / 6 products"
        else:
            edge_kernel_name = "edge_kernel<128,256> (v_mfma_f32_32x32x2_f32)"
            edge_peak = MFMA_F32_PEAK_TFLOPS
            edge_peak_note = "fp32-input MFMA peak"
        edge_traffic = measured_traffic(B, n)
        edge_roof = {
            "kernel": edge_kernel_name,
            "bound": "mfma", "achieved": edge_tflops, "peak": edge_peak, "unit": "TFLOP/s",
            "frac": edge_tflops / edge_peak, "traffic": edge_traffic,
            "peak_note": edge_peak_note, "frac_of_fp32_mfma_peak": edge_tflops / MFMA_F32_PEAK_TFLOPS,
            "avg_launch_ms": mean_ms.value, "launches_timed": int(launches.value),
            "algorithmic_flops_per_launch": edge_flops,
        }
        roofline = edge_roof
        Nl = N / launches_per_step
        if status.get("conv_variant") == 2 and conv_launches.value > 0:
            # Default path since round 3: the edge kernel stops after the basis and every layer's message kernel projects it
            # itself.  That kernel, L launches per step, is where the step's time is.  It is a dense contraction (2 D C FLOP per
            # (edge, orientation) row, SURVEY.md 8(d) "kernel projection 50.5 %"), so the headline fraction is the MFMA one;
            # its ALGORITHMIC bytes are those of 8(d)'s conv pass (read x, write x_1).  The basis stash it streams is traffic the
            # design adds: reported beside the PMC figure, never as achieved work.
            rows = e_mean * 16 / launches_per_step
            row_bytes = int(status.get("basis_row_bytes") or 768)
            # the roof of the arithmetic this kernel runs: three fp16 products per fp32 product (2500 / 3), or -- round 4 default --
            # one fp16 product + the two cross products as ONE fp8 product at twice the fp16 rate = two fp16-equivalents (2500 / 2)
            x8 = bool(status.get("conv_cross_fp8"))
            proj_peak = MFMA_BF16_PEAK_TFLOPS / 2.0 if x8 else edge_peak
            proj_peak_note = ("fp32-equivalent roof of the projection's scheme: main product on fp16 MFMAs + both cross products as one fp8 "
                              "(e4m3) MFMA product at twice the fp16 rate = 2 fp16-equivalents: 2500 TFLOP/s / 2" if x8 else edge_peak_note)
            cbytes = conv_pass_bytes(Nl)
            cflops = rows * conv_proj_flops_per_row()
            t = conv_ms.value * 1e-3
            front_flops = rows * 2 * (258 * 128 + 128 * 256)
            edge_roof = dict(edge_roof, kernel="edge_kernel_f16x3<128,256,PROJ=false> (pair invariants + basis MLP; stores the basis "
                             f"planes, {row_bytes} B per row)", achieved=front_flops / (mean_ms.value * 1e-3) / 1e12,
                             algorithmic_flops_per_launch=front_flops)
            edge_roof["frac"] = edge_roof["achieved"] / edge_peak
            edge_roof["frac_of_fp32_mfma_peak"] = edge_roof["achieved"] / MFMA_F32_PEAK_TFLOPS
            edge_roof["traffic"] = measured_traffic(B, n, "edge_kernel_hbm_bytes_per_launch")
            pmc = measured_traffic(B, n, "conv_proj_hbm_bytes_per_launch")
            roofline = {
                "kernel": "conv_proj_kernel<128,256> x L per step (kernel projection of the layer on fp16x3 MFMAs from the stashed "
                          "basis + message passing + spherical convolution; conv.py:110-127)",
                "bound": "mfma", "achieved": cflops / t / 1e12, "peak": proj_peak, "unit": "TFLOP/s",
                "frac": cflops / t / 1e12 / proj_peak, "traffic": pmc,
                "peak_note": proj_peak_note, "frac_of_fp32_mfma_peak": cflops / t / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "frac_of_f16x3_roof": cflops / t / 1e12 / F16X3_EQUIV_PEAK_TFLOPS, "cross_products": "fp8 e4m3" if x8 else "fp16",
                "avg_launch_ms": conv_ms.value, "launches_timed": int(conv_launches.value),
                "launches_per_step": int(round(conv_launches.value / args.steps)),
                "algorithmic_flops_per_launch": cflops,
                "flops_note": "2 D C FLOP per (edge, orientation) row, rows = (sum of in-degrees measured on the device) x 16",
                "algorithmic_bytes_per_launch": cbytes,
                "bytes_note": "SURVEY.md 8(d): one conv pass = read x + write x_1 = 2 N O C 4 bytes; the stash is NOT in it",
                "hbm": {"achieved": cbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cbytes / t / 1e9 / HBM_PEAK_GBS},
                "stash_bytes_per_launch": stash_bytes_per_launch(Nl, row_bytes),
                "stash_note": f"design-added stream: {row_bytes} B per (slot, orientation) row, all 8 slots of every receiver "
                              "(written once by the edge kernel, read once per layer); counted in `traffic`, not in `achieved`",
                "traffic_over_algorithmic_bytes": (pmc / cbytes) if pmc else None,
                "share_of_step": conv_ms.value * round(conv_launches.value / args.steps) / (1e3 * el_eager / args.steps),
                "edge_kernel": edge_roof,
            }
        # both step-level fractions, from the algorithmic figures
        sb = step_bytes(N, e_mean, B)
        step_traffic = measured_traffic(B, n, "hbm_bytes_per_step_sampling_kernels")
        step_roof = {
            "algorithmic_flops": step_flops, "algorithmic_bytes": sb,
            "achieved_flops": step_flops / (ms_per_step * 1e-3) / (MFMA_F32_PEAK_TFLOPS * 1e12),
            "achieved_flops_of_split_scheme_roof": step_flops / (ms_per_step * 1e-3) / (edge_peak * 1e12),
            "achieved_hbm": sb / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
            "traffic": step_traffic, "traffic_over_algorithmic_bytes": (step_traffic / sb) if step_traffic else None,
            "note": "achieved_flops = F / (t x 157.3e12) (can exceed 1: the dense layers run as 3 fp16 MFMA products per fp32 "
                    f"product; against that scheme's roof, {edge_peak:.0f} TFLOP/s: achieved_flops_of_split_scheme_roof); achieved_hbm = "
                    "Bytes / (t x 8e12); per GPU; traffic = PMC bytes per step of the committed rocprofv3 passes",
        }
        out = {
            "metric": "denoising steps/sec (crystal-steps, whole node)",
            "value": crystal_steps_per_s,
            "unit": "crystal-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "per_rank_ms": [1e3 * e / args.steps for e in per_rank],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "dtype_note": "fp32 inputs/outputs/accumulation; kernels that ran (reported by the library, "
                          f"arreau_model_status): edge={status['edge_kernel']}, mlp={status['mlp_kernel']}.  fp16x3 = each "
                          "fp32 product of the dense layers as 3 fp16 MFMA products (two 11-bit operand planes, f16x3.h); "
                          f"message path (conv kernel variant {status.get('conv_variant')}): 2 = no K stash -- the edge kernel "
                          "stores the windowed basis once (fp16 plane + residual plane rounded to fp8 e4m3: 3 bytes per value, 11 + 4 "
                          f"significand bits, basis_row_bytes={status.get('basis_row_bytes')}; ARREAU_BASIS_FP8=0 or a model whose calibration "
                          "dropped the fp8 plane: both planes fp16) and every layer's message kernel "
                          f"projects it, its two cross products as one fp8 (e4m3) MFMA product: conv_cross_fp8={status.get('conv_cross_fp8')} "
                          "(round 4; ARREAU_CROSS_FP8=0: three fp16 products); 1 = the round-2 pair with a K stash of 3-byte floats; environment: "
                          f"ARREAU_BASIS_FP8={os.environ.get('ARREAU_BASIS_FP8', 'calibrated')} ARREAU_CONV_VARIANT={os.environ.get('ARREAU_CONV_VARIANT', '2')}; "
                          "measured deviations from the fp32 / fp64 CPU oracle: profiles/parity_r05.json; "
                          "roofline.fp32_mfma_variant = the same step on the plain fp32-MFMA kernels",
            "data": "synthetic",
            "config": {
                "workload": f"{config_label(B, n)} per GPU: batch={B} crystals x {n} atoms, {T}-step sampler "
                            f"(T={T}, {T - 1} network evaluations per crystal), fp32",
                "crystals_per_gpu": B, "atoms_per_crystal": n, "num_timesteps": T,
                "checkpoint": "synthetic 1.17M-param (S=90,C=128,O=16,D=256,L=5,k=8,R=5), seed 1234",
                "edges_per_atom_start": e_start / N, "edges_per_atom_end": e_end / N,
                "parallelism": f"replicas x{world}, disjoint sub-batches, no data-path collective",
                "slices_per_gpu": launches_per_step,
            },
            "loop_mode": loop_mode,
            "eager_loop": eager_loop,
            "graph_loop": graph_loop,
            "full_sampler_measured": full_sampler,
            "batch_steps_per_sec": world * args.steps / elapsed,
            "crystals_per_min": 60.0 * crystal_steps_per_s / (T - 1),
            "step_tflops_algorithmic": world * step_flops / (ms_per_step * 1e-3) / 1e12,
            "roofline": dict(roofline, step=step_roof, fp32_mfma_variant=fp32_variant),
            "ranks": ranks_info,
        }
        if timing_only:
            out["INVALID_timing_only_experiment"] = True
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(model, B, n, T, args.cpu_steps)
        if args.config is None and world == 1 and not args.no_other_configs and not args.no_graph_loop and (B, n) == CONFIGS["c2"][:2]:
            del eng, model
            torch.cuda.empty_cache()
            out["other_configs"] = other_config_legs()
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_rank_c5(args, rank, local_rank, world):
    """BASELINE configs[4]: the score-matching training step, data parallel (global batch = 64 x n_gpus; the reference's
    512 at 8 GPUs).  A step = forward noising + score network forward + losses + backward (all in libarreau_hip.so) +
    ONE all-reduce of the flat fp32 gradient bucket + clip + Adam + refresh of the library's training weights.  Data:
    synthetic crystals with Alexandria-PBE's statistics (arreau_amd.diffusion.lattice_dataset.synthetic_alexandria_like)."""
    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    backend = os.environ.get("ARREAU_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("ARREAU_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    # ARREAU_BENCH_FORCE_DIST=1: a ONE-rank run also creates the process group and goes through every collective of the N-rank path
    # (barriers, the timing all-gather, the rank records; c5: the gradient all-reduce) -- the only way to execute the RCCL branch on
    # a one-GPU box (two ranks cannot share a device under RCCL)
    force_dist = world == 1 and os.environ.get("ARREAU_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from arreau_amd import build
    build.build(verbose=False)
    from arreau_amd.checkpoint import default_args
    from arreau_amd.diffusion.lattice_dataset import CrystalDataset, collate, synthetic_alexandria_like
    from arreau_amd.lightning_wrappers.diffusion import PONITA_DIFFUSION
    from arreau_amd.train import optimizer_step

    B = args.batch_per_gpu
    ds = CrystalDataset(configs=synthetic_alexandria_like(4096, seed=0))  # same table (S = 90) on every rank
    torch.manual_seed(1234)
    model = PONITA_DIFFUSION(default_args(lr=3e-4, epochs=10, hidden_dim=args.hidden_dim), ds.z_table).to(dev)
    optimizer = model.configure_optimizers(max_epochs=10)["optimizer"]
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    rng = np.random.RandomState(100 + rank)
    batches = [collate([ds[int(i)] for i in rng.choice(len(ds), B, replace=False)]) for _ in range(8)]
    torch.manual_seed(2000 + rank)
    n_atoms = float(np.mean([int(b.num_atoms.sum()) for b in batches]))

    def one_step(i):
        loss = model.training_step(batches[i % len(batches)])
        optimizer_step(model, optimizer, world, always_reduce=force_dist)
        return loss

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # set-up, like packing the weights: size the library's activation buffers for the largest batch of the run (a steady
    # training loop has met it long ago; regrowing them is a device synchronisation + allocation of gigabytes)
    model.diffusion_loss(model, max(batches, key=lambda b: int(b.num_atoms.sum())), None, training=True)
    log(f"rank {rank}: training bench, {B} crystals / GPU, mean {n_atoms:.0f} atoms per batch; warm-up {args.warmup} steps")
    for i in range(max(args.warmup, 2)):  # the first step also callibrates the conv weights
        one_step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = one_step(i)
    torch.cuda.synchronize(dev)
    local = time.perf_counter() - t0
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = [local]
    if dist is not None:
        tt = torch.tensor([local, elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        per_rank = [float(g[0]) for g in gathered]
        elapsed = max(float(g[1]) for g in gathered)
    # device-only part (forward + backward, no optimizer / collective), for the roofline line
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for i in range(args.steps):
        model.training_step(batches[i % len(batches)])
    torch.cuda.synchronize(dev)
    fb_ms = 1e3 * (time.perf_counter() - t1) / args.steps
    ranks_info = {"backend": (dist.get_backend() if dist is not None else None),
                  "world_size": (dist.get_world_size() if dist is not None else 1),
                  "data_path_collectives": 1 if (world > 1 or force_dist) else 0,
                  "note": "one all-reduce of the flat fp32 gradient bucket per step (arreau_amd.train.optimizer_step); RCCL when the backend is nccl",
                  "per_rank": gather_json(dist, dev if backend == "nccl" else "cpu", world,
                                          dict(device_identity(torch, dev), rank=rank, local_rank=local_rank, pid=os.getpid(),
                                               mean_atoms_per_batch=n_atoms))}
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        flops = 3.0 * step_flops_per_atom(C=args.hidden_dim) * n_atoms  # forward + backward ~ 3 x forward
        out = {
            "metric": "training steps/sec (crystal-steps, whole node) -- BASELINE configs[4]",
            "value": world * B * args.steps / elapsed, "unit": "crystal-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "per_rank_ms": [1e3 * e / args.steps for e in per_rank],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: score-matching training step, global batch {B * world} = {B} crystals x "
                                   f"{world} GPUs, Alexandria-like synthetic crystals (mean {n_atoms / B:.1f} atoms), T=1000, fp32, "
                                   f"hidden_dim {args.hidden_dim}" + (" (the reference's `make train` preset)" if args.hidden_dim == 200 else ""),
                       "crystals_per_gpu": B, "mean_atoms_per_batch": n_atoms, "hidden_dim": args.hidden_dim,
                       "parallelism": f"data parallel x{world}: one all-reduce of the {n_params / 1e6:.2f}M-parameter fp32 gradient per step"},
            "last_loss": float(loss),
            "ranks": ranks_info,
            "forward_backward_ms": fb_ms,
            "roofline": training_roofline(flops, fb_ms, args.hidden_dim, B, alg_bytes=3.0 * step_bytes(n_atoms, 8 * n_atoms, B, C=args.hidden_dim)),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_training(model, batches[0], args.cpu_steps)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def training_roofline(flops, fb_ms, hidden_dim=128, crystals=64, alg_bytes=None):
    """The training step's matrix work against the roof of the arithmetic it runs in.  ARREAU_TRAIN_GEMM (train_net.hip) selects it:
    default `split` = forward products as fp16x3 (3 fp16 MFMA products per fp32 product: 2500 / 3 TFLOP/s), products with a gradient
    operand as bf16x6 (2500 / 6); one third of the algorithmic FLOPs are forward, two thirds backward, so the step's roof is the
    harmonic mix 1 / (1/3 / 833 + 2/3 / 417) = 500 TFLOP/s.  `exact`: every product on v_mfma_f32_32x32x2_f32 (157.3)."""
    mode = os.environ.get("ARREAU_TRAIN_GEMM", "split")
    if mode == "exact":
        peak, note = MFMA_F32_PEAK_TFLOPS, "exact fp32 MFMA products (ARREAU_TRAIN_GEMM=exact): dense fp32-input MFMA peak"
    elif mode == "fp16":
        peak, note = F16X3_EQUIV_PEAK_TFLOPS, "every product as fp16x3 (ARREAU_TRAIN_GEMM=fp16): 2500 / 3"
    else:
        peak = 1.0 / ((1.0 / 3.0) / F16X3_EQUIV_PEAK_TFLOPS + (2.0 / 3.0) / BF16X6_EQUIV_PEAK_TFLOPS)
        note = ("split-precision products on the 16-bit matrix pipe (sgemm_split_kernel): forward fp16x3 (2500 / 3 TFLOP/s of fp32-equivalent "
                "work), gradient products bf16x6 (2500 / 6); harmonic mix over 1/3 forward + 2/3 backward FLOPs")
    ach = flops / (fb_ms * 1e-3) / 1e12
    # bytes through the L2s per forward + backward step from the committed PMC passes (tools/hbm_traffic_c5.sh): valid for the default
    # workload only (64 crystals per GPU, hidden_dim 128, split-precision products)
    traffic = measured_traffic(64, 0, "hbm_bytes_per_step_forward_backward") if (mode == "split" and hidden_dim == 128 and crystals == 64) else None
    out = {"kernel": "training step, forward + backward (sgemm_split_kernel products + element-wise / gather kernels of train_net.hip)",
           "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "peak_note": note,
           "frac_of_fp32_mfma_peak": ach / MFMA_F32_PEAK_TFLOPS, "gemm_mode": mode}
    if alg_bytes:
        # the same rule as the sampling line: `achieved` from ALGORITHMIC bytes -- forward + backward ~ three passes
        # over the tensors of SURVEY 8(d)'s step --, the PMC bytes (every intermediate of the step is materialised) beside it as traffic
        out["hbm"] = {"achieved": alg_bytes / (fb_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": alg_bytes / (fb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg_bytes,
                      "note": "3 x SURVEY 8(d)'s algorithmic bytes of one forward step over the forward + backward time",
                      "traffic": traffic, "traffic_over_algorithmic_bytes": (traffic / alg_bytes) if traffic else None}
    return out


def cpu_baseline_training(model, batch, steps):
    """The oracle's training step on the host cores: DiffusionLoss.__call__ restated in plain torch (oracle/training.py),
    forward + torch autograd backward, fp32, on one of the bench's own batches."""
    import torch
    from oracle import training as TR
    from tests.helpers import oracle_from_module

    cores = host_cores()
    torch.set_num_threads(cores)
    om = oracle_from_module(model, torch.float32)
    for v in om.sd.values():
        if v.is_floating_point() and v.numel() > 0:
            v.requires_grad_(True)
    B, N, S = int(batch.num_atoms.numel()), int(batch.num_atoms.sum()), om.hp["S"]
    lattice0 = batch.L0.reshape(-1, 3, 3).float()
    times = []
    for it in range(steps + 1):
        t = torch.randint(1, om.hp["T"] + 1, (B,))
        noise = (torch.randn(N, 3), torch.rand(N, S), torch.randn(B, 3))
        t0 = time.perf_counter()
        loss = TR.diffusion_loss(om, batch.X0.float(), batch.A0, lattice0, batch.num_atoms, t, *noise)
        loss.backward()
        times.append(time.perf_counter() - t0)
        for v in om.sd.values():
            v.grad = None
        log(f"cpu baseline (training) step {it}: {times[-1]:.2f} s")
    per_step = sum(times[1:]) / max(1, len(times) - 1)
    return {"value": B / per_step, "unit": "crystal-steps/s", "cores": cores, "kind": "port",
            "sample": f"{steps} forward + backward steps of one {B}-crystal batch ({N} atoms) after 1 warm-up "
                      f"(oracle + torch autograd, fp32, {cores} threads)"}


def cpu_baseline(model, B, n, T, steps):
    """The oracle (CPU restatement of the reference step, pure PyTorch, all host cores) timed on a bounded sample of
    the SAME workload: for the default batch `steps` full steps after one warm-up; for config 1 (1 crystal x 8 atoms,
    T = 100) the whole 99-step sampler, as SURVEY 8(d) asks; for batches too large for a few seconds per CPU step
    (config 3/4) a slice of the crystals at the same atoms per crystal, which says so in `sample`."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from oracle import sampler as OS
    from tests.helpers import oracle_from_module

    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads")
    Bc = B
    note = ""
    while Bc * n > 256 * 20 and Bc > 1:  # bound one CPU step to the cost of the default batch
        Bc //= 2
    if Bc != B:
        note = f" -- a {Bc}-crystal slice of the {B}-crystal batch, same atoms per crystal"
    full_sampler = B * n <= 64  # config 1: run every timestep
    res = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        om = oracle_from_module(model, dtype)
        S = om.hp["S"]
        torch.manual_seed(0)
        np.random.seed(0)
        frac, types, lengths, angles, num_atoms = OS.init_state(om, n, Bc, dtype)
        batch = torch.arange(Bc).repeat_interleave(n)
        N = Bc * n
        if full_sampler:
            t0 = time.perf_counter()
            OS.sample(om, n, Bc, dtype, state=(frac, types, lengths, angles, num_atoms))
            total = time.perf_counter() - t0
            per_step = total / (T - 1)
            log(f"cpu baseline {tag}: {T - 1} steps (whole sampler) in {total:.2f} s")
        else:
            times = []
            t = T - 1
            for it in range(steps + 1):
                t0 = time.perf_counter()
                scores = OS.predict_scores(om, frac, F.one_hot(types, S), torch.full((N,), t), num_atoms, lengths,
                                           angles, batch)
                noise = OS.StepNoise(torch.randn(Bc, 3, dtype=dtype), torch.randn(N, 3, dtype=dtype),
                                     torch.rand(N, S, dtype=dtype))
                frac, types, lengths, _ = OS.reverse_step(om, frac, types, lengths, angles, num_atoms, scores, t, noise)
                times.append(time.perf_counter() - t0)
                log(f"cpu baseline {tag} step {it}: {times[-1]:.2f} s")
                t -= 1
            per_step = float(np.mean(times[1:]))
        res[tag] = Bc / per_step
        if tag == "f32" and not full_sampler and per_step * (steps + 1) > 40:
            break  # keep the default run bounded on slow hosts
    what = (f"the whole {T - 1}-step sampler of batch={Bc} x {n} atoms" if full_sampler
            else f"{steps} full steps of batch={Bc} x {n} atoms after 1 warm-up")
    return {
        "value": res["f32"], "unit": "crystal-steps/s", "cores": cores, "kind": "port",
        "sample": f"{what} (oracle, torch CPU, {cores} threads){note}",
        "value_f64": res.get("f64"),
    }


if __name__ == "__main__":
    main()

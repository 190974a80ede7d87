"""Batch driver of the sampler: the role of main_diffusion_generate.py:52-94, sharded over GPUs.

Crystals are independent, so the crystal index is cut into contiguous slices, one per rank (one
process per GPU); every rank samples its slice in sub-batches and rank 0 gathers the `SampleResult`s
in crystal order.  The only communication is that final gather of host arrays
(`torch.distributed.gather_object`; RCCL/gloo are not on the data path).

    python -m torch.distributed.run --nproc-per-node 8 -m arreau_amd.generate --model_path last.ckpt \
        --num_crystals 8192 --num_atoms 20 --out out/crystals.npz
"""
import argparse
import os
from typing import Callable, Optional

import numpy as np

from .diffusion.diffusion_loss import SampleResult


def shard_range(num_items: int, world_size: int, rank: int):
    """Contiguous slice [start, stop) of `num_items` owned by `rank` (remainder to the first ranks)."""
    base, rem = divmod(num_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def concat_results(parts) -> SampleResult:
    """Crystal-order concatenation with the reference's index arrays (main_diffusion_generate.py:67-92)."""
    parts = [p for p in parts if p is not None and p.num_atoms is not None and len(p.num_atoms)]
    if not parts:
        return SampleResult(frac_x=np.empty((0, 3)), atomic_numbers=np.empty((0,)), lattice=np.empty((0, 3, 3)),
                            idx_start=np.empty((0,), dtype=np.int64), num_atoms=np.empty((0,), dtype=np.int64))
    num_atoms = np.concatenate([np.asarray(p.num_atoms) for p in parts])
    return SampleResult(
        frac_x=np.concatenate([p.frac_x for p in parts]), atomic_numbers=np.concatenate([p.atomic_numbers for p in parts]),
        lattice=np.concatenate([p.lattice for p in parts]), num_atoms=num_atoms,
        idx_start=np.cumsum(num_atoms) - num_atoms)


def generate_n_crystals(sample_fn: Callable[[int, int], SampleResult], num_crystals: int, num_atoms_per_sample: int,
                        num_crystals_per_batch: int = 256, rank: int = 0, world_size: int = 1,
                        gather: Optional[Callable] = None) -> Optional[SampleResult]:
    """sample_fn(num_atoms_per_sample, num_samples_in_batch) -> SampleResult  (e.g. PONITA_DIFFUSION.sample).
    Returns the concatenated result on rank 0 (None elsewhere when world_size > 1)."""
    start, stop = shard_range(num_crystals, world_size, rank)
    mine = []
    for s in range(start, stop, num_crystals_per_batch):
        mine.append(sample_fn(num_atoms_per_sample, min(num_crystals_per_batch, stop - s)))
    local = concat_results(mine)
    if world_size == 1:
        return local
    if gather is None:
        import torch.distributed as dist

        def gather(obj):
            out = [None] * world_size if rank == 0 else None
            dist.gather_object(obj, out, dst=0)
            return out
    parts = gather(local)
    return concat_results(parts) if rank == 0 else None


def save_sample_results(crystals: SampleResult, filename: str):
    """The reference's crystals.h5 layout (diffusion/inference/process_generated_crystals.py:8-15)."""
    from .diffusion.inference.process_generated_crystals import save_sample_results_to_hdf5
    return save_sample_results_to_hdf5(crystals, filename)


def main():
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_path", type=str, required=True)
    ap.add_argument("--num_crystals", type=int, default=10)
    ap.add_argument("--num_atoms", type=int, default=4)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", type=str, default="out/crystals.npz")
    ap.add_argument("--seed", type=int, default=None, help="seed of the host/device generators (rank is added)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ARREAU_GENERATE_BACKEND=gloo + ARREAU_GENERATE_ONE_DEVICE=1 rehearse several ranks on a one-GPU box (the only
    # communication is the final gather of host arrays, so the backend is not on the data path)
    if os.environ.get("ARREAU_GENERATE_ONE_DEVICE", "0") == "1":
        local_rank = 0
    backend = os.environ.get("ARREAU_GENERATE_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from .diffusion.inference.visualize_crystal import VisualizationSetting
    from .lightning_wrappers.diffusion import PONITA_DIFFUSION
    model = PONITA_DIFFUSION.load_from_checkpoint(args.model_path, map_location=f"cuda:{local_rank}", strict=False)
    if args.seed is not None:
        import numpy as np
        torch.manual_seed(args.seed + rank)
        np.random.seed(args.seed + rank)
    # ARREAU_GENERATE_GPU_LOCK=<file>: ranks that SHARE one device (the one-GPU rehearsal above) take turns on it -- a file
    # lock held around each sampler call.  On a node every rank owns its GPU and the variable is not set.
    lock_path = os.environ.get("ARREAU_GENERATE_GPU_LOCK")

    def fn(n, b):
        if not lock_path:
            return model.sample(n, b, VisualizationSetting.NONE, False)
        import fcntl
        with open(lock_path, "a") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                out = model.sample(n, b, VisualizationSetting.NONE, False)
                torch.cuda.synchronize()
                return out
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    res = generate_n_crystals(fn, args.num_crystals, args.num_atoms, args.batch, rank, world)
    if rank == 0:
        print("wrote", save_sample_results(res, args.out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Data-parallel training driver for the score-matching step (BASELINE config 5: global batch 512 over 8 GPUs).

The role of main_diffusion.py:293-310 + Lightning's DDP, without pytorch_lightning: one process per GPU, every rank
takes a disjoint slice of each global batch (arreau_amd.diffusion.lattice_dataset.iterate_batches), runs
PONITA_DIFFUSION.training_step (forward + backward in libarreau_hip.so), the gradients are averaged with ONE all-reduce
of a flat fp32 bucket (1.17 M parameters = 4.7 MB; RCCL when the process group is "nccl"), clipped to norm 0.5
(main_diffusion.py:297) and applied by Adam (decay / no-decay groups) with the cosine warm-up schedule stepped per epoch.

    python -m torch.distributed.run --nproc-per-node 8 -m arreau_amd.train --epochs 2 --batch_size 64
"""
import argparse
import os
from typing import Iterable, Optional

import torch

GRAD_CLIP = 0.5  # main_diffusion.py:297 gradient_clip_val


def all_reduce_gradients(parameters: Iterable[torch.nn.Parameter], world_size: int, group=None):
    """Average `.grad` over the ranks with one collective on a flat bucket (what DDP does with its single 4.7 MB bucket
    for this model).  Parameters without a gradient on this rank contribute zeros.  Returns the number of elements."""
    params = [p for p in parameters if p.requires_grad and p.numel() > 0]
    if not params:
        return 0
    if world_size <= 1:  # nothing to exchange: leave the gradients where they are
        return sum(p.numel() for p in params)
    dev, dt = params[0].device, torch.float32
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dt) for p in params])
    if world_size > 1:
        import torch.distributed as dist
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat /= world_size
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].reshape(p.shape).to(device=dev, dtype=p.dtype)
        off += n
    return off


def _trainable_parameters(model):
    cache = model.__dict__.get("_trainable_parameter_cache")
    if cache is None:
        cache = model.__dict__["_trainable_parameter_cache"] = [p for p in model.parameters()]
    return cache


def optimizer_step(model, optimizer, world_size: int = 1, clip: Optional[float] = GRAD_CLIP, always_reduce: bool = False):
    """all-reduce -> clip -> Adam step -> invalidate the engine's packed weights.  Returns the gradient norm.

    When PONITA_DIFFUSION.training_step left every gradient as a view of the engine's flat buffer (`model._grad_flat`), the
    collective, the norm and the clip run on that ONE buffer: no gather into a bucket and no scatter back, one norm instead of a
    multi-tensor norm over 70 tensors (torch.nn.utils.clip_grad_norm_'s arithmetic: total 2-norm, coefficient
    max_norm / (norm + 1e-6) clamped to 1)."""
    flat = getattr(model, "_grad_flat", None)
    params = _trainable_parameters(model)
    if flat is not None and all(p.grad is None or p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in params):
        if world_size > 1 or always_reduce:  # (always_reduce: a one-rank group still runs the collective -- bench.py's RCCL rehearsal)
            import torch.distributed as dist
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= world_size
        norm = None
        if hasattr(optimizer, "step_flat"):
            # arreau_amd.optim.ClipAdam: norm, clip coefficient, non-finite guard and the update of every tensor in two launches
            # ... and, where the module's engine keeps its own fp32 copy of a tensor, the update lands there in the same pass (then
            # only the two weights the engine reads in a derived form are rebuilt, instead of copying all 70 tensors twice)
            eng = getattr(model, "_engine", None)
            mirrors = eng.train_weight_mirrors(model) if eng is not None and hasattr(eng, "train_weight_mirrors") else None
            norm = optimizer.step_flat(flat, clip, mirrors)
            if norm is not None:
                optimizer.zero_grad(set_to_none=True)
                model._grad_flat = None
                if mirrors:
                    eng.refresh_derived_train_weights(model)
                else:
                    model.notify_parameters_changed()
                return norm
        if clip:
            norm = torch.linalg.vector_norm(flat, 2)
            flat.mul_(torch.clamp(clip / (norm + 1e-6), max=1.0))
            # A non-finite gradient (an activation beyond the fp16 range of the split-precision forward: ADVICE round 4) must not
            # reach Adam's moments before the periodic status check has switched the engine to the full-range products
            # (PONITA_DIFFUSION.training_step): such a step becomes a no-op -- on the device, no host synchronisation.
            flat.copy_(torch.where(torch.isfinite(norm), flat, torch.zeros_like(flat)))
    else:
        all_reduce_gradients(params, world_size)
        norm = torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], clip) if clip else None
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    model._grad_flat = None
    model.notify_parameters_changed()
    return norm


def train_epochs(model, dataset, epochs: int, batch_size: int, rank: int = 0, world_size: int = 1, seed: int = 0,
                 log=print):
    """`batch_size` is per rank (global batch = batch_size * world_size).  Returns the list of per-step losses."""
    from .diffusion.lattice_dataset import iterate_batches
    from .diffusion.diffusion_loss import DiffusionLossMetric
    opt = model.configure_optimizers(max_epochs=epochs)
    optimizer, scheduler = opt["optimizer"], opt["lr_scheduler"]
    losses = []
    for epoch in range(epochs):
        metric = DiffusionLossMetric()  # (diffusion_loss.py:52-65; the reference logs it per epoch with sync_dist)
        step_losses = []                # device scalars: read back once per epoch, not per step (no host sync in the loop)
        for step, batch in enumerate(iterate_batches(dataset, batch_size, shuffle=True, seed=seed + epoch, rank=rank,
                                                     world_size=world_size, drop_last=True)):
            loss = model.training_step(batch)
            optimizer_step(model, optimizer, world_size)
            step_losses.append(loss.detach())
            metric.update(loss, batch)
        scheduler.step()
        if step_losses:
            losses += [float(v) for v in torch.stack(step_losses).cpu()]
        mean = float(metric.sync().compute())  # two scalar all-reduces: loss sum and crystal count over the ranks
        if getattr(model, "_engine", None) is not None:
            model._engine.check_status()  # sticky device flags of the whole epoch
        if rank == 0:
            log(f"epoch {epoch}: {len(step_losses)} steps, loss per crystal {mean:.5f} (all ranks), last loss "
                f"{losses[-1] if losses else float('nan'):.5f}, lr {scheduler.get_last_lr()[0]:.3e}")
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", nargs="*", default=None, help="dataset files (lattice_dataset layout); synthetic if omitted")
    ap.add_argument("--num_synthetic", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch_size", type=int, default=64, help="crystals per rank")
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--seed", type=int, default=0)
    # network / diffusion flags of the reference's training CLI (main_diffusion.py:63-148; `make train`: --hidden_dim=200)
    ap.add_argument("--hidden_dim", type=int, default=128)
    ap.add_argument("--basis_dim", type=int, default=256)
    ap.add_argument("--layers", type=int, default=5)
    ap.add_argument("--widening_factor", type=int, default=4)
    ap.add_argument("--num_timesteps", type=int, default=1000)
    ap.add_argument("--radius", type=float, default=5)
    ap.add_argument("--max_neighbors", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--weight_decay", type=float, default=1e-10)
    ap.add_argument("--out", type=str, default=None, help="write a Lightning-format checkpoint here (rank 0)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("ARREAU_TRAIN_ONE_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("ARREAU_TRAIN_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from .checkpoint import default_args, save_lightning_checkpoint
    from .diffusion.lattice_dataset import CrystalDataset, synthetic_alexandria_like
    from .lightning_wrappers.diffusion import PONITA_DIFFUSION
    ds = CrystalDataset(args.data) if args.data else CrystalDataset(configs=synthetic_alexandria_like(args.num_synthetic, args.seed))
    torch.manual_seed(args.seed)  # same initial weights on every rank
    net_args = default_args(lr=args.lr, epochs=args.epochs, hidden_dim=args.hidden_dim, basis_dim=args.basis_dim,
                            layers=args.layers, widening_factor=args.widening_factor, num_timesteps=args.num_timesteps,
                            radius=args.radius, max_neighbors=args.max_neighbors, warmup=args.warmup,
                            weight_decay=args.weight_decay)
    model = PONITA_DIFFUSION(net_args, ds.z_table).to(f"cuda:{local_rank}")
    torch.manual_seed(args.seed + 1000 + rank)  # different noise per rank
    train_epochs(model, ds, args.epochs, args.batch_size, rank, world, args.seed)
    # data-parallel invariant: the replicas hold the same weights (same initial weights, same averaged gradients, ONE set
    # of calibration ratios); a drift here means a rank applied something its peers did not
    check = torch.stack([p.detach().double().sum() for p in model.parameters()]).sum().reshape(1)
    if world > 1:
        sums = [torch.zeros_like(check) for _ in range(world)]
        if dist.get_backend() == "nccl":
            dist.all_gather(sums, check)
        else:
            host = [s.cpu() for s in sums]
            dist.all_gather(host, check.cpu())
            sums = host
        vals = [float(s) for s in sums]
        if rank == 0:
            print("replica parameter checksums:", " ".join(f"{v:.12e}" for v in vals))
        if max(vals) - min(vals) > 1e-9 * max(1.0, abs(vals[0])):
            raise SystemExit(f"rank {rank}: data-parallel replicas diverged: {vals}")
    if rank == 0 and args.out:
        print("wrote", save_lightning_checkpoint(args.out, model))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

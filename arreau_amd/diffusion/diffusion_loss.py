"""DiffusionLoss: the sampling loop of the reference (diffusion/diffusion_loss.py:276-377) and the forward part of
its training loss (`__call__`, :204-274) driving the HIP engine.  The backward pass through the network kernels is not
built yet, so `__call__` returns the loss value (and the gradients with respect to the network outputs), not an
autograd graph."""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import os

import torch
import torch.nn as nn

from .d3pm import D3PM
from .diffusion_helpers import VE_pbc, VP_lattice, crystal_offsets, sample_bravais_angles
from .inference.visualize_crystal import VisualizationSetting, vis_crystal_during_sampling
from .tools.atomic_number_table import AtomicNumberTable, atomic_number_indexes_to_atomic_numbers

pos_sigma_min = 0.001
pos_sigma_max = 1.0
type_power = 2
lattice_power = 2
type_clipmax = 0.999
lattice_clipmax = 0.999


@dataclass
class SampleResult:
    """Same fields as the reference (diffusion_loss.py:39-49)."""
    frac_x: Optional[np.ndarray] = None
    atomic_numbers: Optional[np.ndarray] = None
    lattice: Optional[np.ndarray] = None
    idx_start: Optional[np.ndarray] = None
    num_atoms: Optional[np.ndarray] = None
    # extension (not in the reference): what the library did to produce this batch -- e.g. {"full_range_rerun": True} when an
    # activation left the fp16 range of the default kernels and the batch was re-run on the full-range bf16x6 kernels
    info: Optional[dict] = None


class _PinnedRing:
    """Page-locked staging buffers owned by the caller and reused round-robin: no pinned allocation per step (hipHostMalloc
    maps the block into the GPU's page tables) and an explicit rule for reuse -- a slot is taken again only after the event
    recorded behind its last copy has completed; with four slots that wait is over long before the slot comes round."""
    SLOTS = 4

    def __init__(self):
        self.slots = {}   # (dtype) -> list of [tensor or None, event or None]
        self.next = {}

    def take(self, dtype, n):
        ring = self.slots.setdefault(dtype, [[None, None] for _ in range(self.SLOTS)])
        i = self.next.get(dtype, 0)
        self.next[dtype] = (i + 1) % self.SLOTS
        slot = ring[i]
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < n:
            slot[0] = torch.empty(max(2 * n, 4096), dtype=dtype, pin_memory=True)
        return slot

_PINNED = _PinnedRing()


def _pack(stage, host, offs):
    """Copy (and cast) the host tensors into their segments of `stage` -- through numpy, on THIS thread.  torch's copy_ splits
    anything above 32768 elements over the intra-op thread pool; under a container CPU quota the pool's spinning workers used the
    quota up and the whole process was throttled for the rest of the 100 ms scheduler period, every third step or so (round 4:
    90 ms stalls of the training loop, 3.2 -> 30 ms per step)."""
    flat = stage.numpy()
    for (_, v), o in zip(host, offs):
        flat[o:o + v.numel()] = v.detach().reshape(-1).numpy()


def stage_to_device(dev, dtype, tensors):
    """`tensors` as contiguous `dtype` tensors on `dev`.  Those that live on the host travel together: they are converted
    into one pinned staging buffer (16-byte aligned segments) and cross in ONE asynchronous copy, so the host never waits
    for the stream.  Tensors already on a device are converted in place."""
    vals = [torch.as_tensor(v) for v in tensors]
    out = [None] * len(vals)
    host = [(i, v) for i, v in enumerate(vals) if v.device.type == "cpu"]
    for i, v in enumerate(vals):
        if v.device.type != "cpu":
            out[i] = v.to(device=dev, dtype=dtype).contiguous()
    if host:
        offs, total = [], 0
        for _, v in host:
            offs.append(total)
            total += -(-v.numel() // 4) * 4
        total = max(total, 4)
        if os.environ.get("ARREAU_H2D", "pinned") == "pinned":
            slot = _PINNED.take(dtype, total)
            stage = slot[0]
            _pack(stage, host, offs)
            d = torch.empty(total, dtype=dtype, device=dev)
            d.copy_(stage[:total], non_blocking=True)
            ev = slot[1] if slot[1] is not None else torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            slot[1] = ev
        else:  # ARREAU_H2D=plain: one pageable copy (the host waits for the stream to drain)
            stage = torch.empty(total, dtype=dtype)
            _pack(stage, host, offs)
            d = stage.to(dev)
        for (i, v), o in zip(host, offs):
            out[i] = d[o:o + v.numel()].view(v.shape)
    return out


class DiffusionLossMetric:
    """diffusion_loss.py:52-65 (a torchmetrics.Metric there): running sum of the step losses and of the crystals seen,
    `compute()` = their ratio.  The two states are what a data-parallel run reduces over the ranks (`dist_reduce_fx="sum"`
    in the reference): `sync()` does those two scalar all-reduces.  States stay on the device of the losses they are fed,
    so updating the metric does not synchronise the training loop."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.total_loss = None
        self.total_samples = 0

    def update(self, loss, batch=None, num_crystals=None):
        loss = torch.as_tensor(loss).detach().sum()
        self.total_loss = loss.clone() if self.total_loss is None else self.total_loss + loss
        if num_crystals is None:
            if hasattr(batch, "num_atoms"):
                num_crystals = int(torch.as_tensor(batch.num_atoms).numel())
            else:  # the reference counts torch.unique(batch.batch)
                num_crystals = int(torch.unique(torch.as_tensor(batch.batch)).numel())
        self.total_samples += int(num_crystals)

    def sync(self, group=None):
        """Sum both states over the ranks of the default (or given) process group; a no-op without one."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        backend = dist.get_backend(group)
        dev = self.total_loss.device if (self.total_loss is not None and backend == "nccl") else torch.device("cpu")
        if backend == "nccl" and dev.type != "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        tl = (self.total_loss if self.total_loss is not None else torch.zeros(())).to(dev, torch.float64).reshape(1)
        ts = torch.tensor([float(self.total_samples)], device=dev, dtype=torch.float64)
        dist.all_reduce(tl, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM, group=group)
        self.total_loss = tl.reshape(()).to(torch.float32)
        self.total_samples = int(round(float(ts)))
        return self

    def compute(self):
        if self.total_loss is None or self.total_samples == 0:
            return torch.tensor(float("nan"))
        return self.total_loss / self.total_samples


class DiffusionLoss(nn.Module):
    def __init__(self, args, num_atomic_states: int):
        super().__init__()
        self.cutoff = args.radius
        self.max_neighbors = args.max_neighbors
        self.T = args.num_timesteps
        self.pos_diffusion = VE_pbc(self.T, sigma_min=pos_sigma_min, sigma_max=pos_sigma_max)
        self.d3pm = D3PM(x0_model=None, n_T=args.num_timesteps, num_classes=num_atomic_states, forward_type="mask")
        self.lattice_diffusion = VP_lattice(num_steps=self.T, power=lattice_power, clipmax=lattice_clipmax)
        self.num_atomic_states = num_atomic_states

    def __call__(self, model, batch, t_emb_weights=None, timestep=None, noise=None, return_parts=False,
                 training=False):
        """diffusion_loss.py:204-274: sample a timestep per crystal, noise (coordinates, atom types, cell lengths),
        evaluate the score network on the noised batch and return `coord + atom-type + lattice` error (weights 1).

        `batch` carries X0 [N,3] fractional coordinates, A0 [N] class indices, L0 [B*3,3] or [B,3,3] cells and
        num_atoms [B] (the fields of the reference's PyG `Data`, lattice_dataset.py:96-104).  Random draws follow the
        reference's order: randint(1, T+1) [B,1] unless `timestep` is given (:213-221), randn_like(X0) (VE_pbc.forward),
        rand(N,S) (D3PM.get_xt), randn_like(lengths) (VP_lattice.forward), from torch's global CPU generator;
        `noise=(z_frac, u_types, z_lengths)` injects them instead (parity tests).  Everything else runs in
        libarreau_hip.so (arreau_diffusion_noise -> arreau_predict_scores -> arreau_diffusion_losses).

        `training=True` evaluates the network with arreau_train_forward (fp32, activations kept) instead of the fused
        sampling kernels, so that HipEngine.train_backward can follow.

        Returns the scalar loss (a 0-d float32 CUDA tensor); with return_parts=True also a dict of the three errors,
        the noised inputs, the network outputs and d(loss)/d(outputs)."""
        eng = model.engine(for_training=training)
        dev = eng.device
        S = self.num_atomic_states
        n_cpu = torch.as_tensor(batch.num_atoms).to("cpu", torch.int64)
        B, N = int(n_cpu.numel()), int(n_cpu.sum())
        frac0 = torch.as_tensor(batch.X0)
        if frac0.shape != (N, 3):
            raise ValueError("batch.X0 must be [sum(num_atoms), 3]")
        if timestep is None:
            t = torch.randint(1, self.T + 1, size=(B, 1)).long()
        elif torch.is_tensor(timestep) and timestep.numel() == B:
            t = timestep.reshape(B, 1).long().cpu()
        else:
            t = torch.ones((B, 1)).long() * int(timestep)
        if int(t.min()) < 1 or int(t.max()) > self.T:
            raise ValueError(f"timestep must be in 1..{self.T}")
        if noise is None:
            dt = torch.get_default_dtype()
            z_frac = torch.randn((N, 3), dtype=dt)
            u_types = torch.rand((N, S))
            z_len = torch.randn((B, 3), dtype=dt)
        else:
            z_frac, u_types, z_len = noise
        # Host -> device: ONE pinned staging buffer and one asynchronous copy per element type.  Eight pageable .to(device)
        # copies each block the host until the stream has drained, so the device idled while the host prepared and
        # enqueued the next step (round 4, rocprofv3 kernel trace of the training step: 0.25 ms of 3.2 without the profiler).
        off_host = torch.zeros(B + 1, dtype=torch.int64)
        off_host[1:] = torch.cumsum(n_cpu, 0)
        frac_d, cell_d, z_frac_d, u_types_d, z_len_d = stage_to_device(
            dev, torch.float32, [frac0, torch.as_tensor(batch.L0).reshape(-1, 3, 3), z_frac, u_types, z_len])
        t_d, types0, off = stage_to_device(dev, torch.int32, [t.reshape(B), batch.A0, off_host])
        nz = eng.diffusion_noise(frac_d, types0, cell_d, t_d, off, z_frac_d, u_types_d, z_len_d)
        evaluate = eng.train_forward if training else eng.predict_scores
        run = lambda: evaluate(nz["noisy_frac"], nz["noisy_types"], nz["noisy_lengths"], nz["angles"], t_d, off)
        # (validation: sticky device flags -> raise; non-finite outputs with fp8 operand planes in use -> repeated on fp16 planes: HipEngine.checked.
        # Reading the flags synchronises the stream, so the training step -- whose host work overlaps the device's, and whose
        # forward runs no fp8 product -- leaves that to PONITA_DIFFUSION.training_step, every STATUS_CHECK_EVERY steps.)
        eps, logits, len0 = run() if training else eng.checked(run)
        losses, grads = eng.diffusion_losses(eps, nz["target_eps"], logits, types0, nz["noisy_types"], t_d, len0,
                                             nz["lengths"], off, with_grads=True)
        if return_parts:
            return losses[0], dict(error_frac_x=losses[1], error_atomic_type=losses[2], error_lattice=losses[3],
                                   vb=losses[4], ce=losses[5], pred_eps=eps, logits=logits, pred_lengths=len0,
                                   grad_eps=grads[0], grad_logits=grads[1], grad_lengths=grads[2], timestep=t_d, **nz)
        return losses[0]

    # ------------------------------------------------------------------------------------------
    def predict_scores(self, noisy_frac_x, noisy_atom_types, t_feat, num_atoms, noisy_lengths, angles, model,
                       batch=None, t_emb_weights=None, edges=None):
        """diffusion_loss.py:112-197.  `noisy_atom_types` may be class indices [N] or one-hot [N,S]
        (the reference passes the one-hot); `t_feat` is the per-atom timestep [N] (or per-crystal [B]).
        Returns (pred_frac_eps_x [N,3], logits [N,S], pred_lengths_0 [B,3]) on the GPU."""
        eng = model.engine()
        dev = eng.device
        frac = noisy_frac_x.to(device=dev, dtype=torch.float32).contiguous()
        ty = noisy_atom_types
        if ty.dim() == 2:
            ty = ty.argmax(dim=1)
        ty = ty.to(device=dev, dtype=torch.int32).contiguous()
        lengths = noisy_lengths.to(device=dev, dtype=torch.float32).contiguous()
        ang = angles.to(device=dev, dtype=torch.float32).contiguous()
        n_cpu = torch.as_tensor(num_atoms).to("cpu", torch.int64)
        off = crystal_offsets(n_cpu, dev)
        t_feat = torch.as_tensor(t_feat)
        if t_feat.numel() == n_cpu.numel():
            t_c = t_feat.reshape(-1)
        else:  # per-atom timestep: constant inside a crystal, take each crystal's first atom
            first = (torch.cumsum(n_cpu, 0) - n_cpu).clamp(max=max(int(t_feat.numel()) - 1, 0))
            t_c = t_feat.reshape(-1).to("cpu")[first]
        t_c = t_c.to(device=dev, dtype=torch.int32).contiguous()
        return eng.predict_scores(frac, ty, lengths, ang, t_c, off, edges=edges)

    @torch.no_grad()
    def sample(self, *, model, z_table: AtomicNumberTable, t_emb_weights=None, num_atoms_per_sample,
               num_samples_in_batch: int, vis_name: str = "", visualization_setting=VisualizationSetting.NONE,
               show_bonds: bool = False, constant_atoms: Optional[torch.Tensor] = None, noise: str = "philox",
               max_steps: Optional[int] = None, use_graph: Optional[bool] = None, seed: Optional[int] = None,
               fixed_cell: bool = False, pipelined_slices: int = 1) -> SampleResult:
        """diffusion_loss.py:276-377.  The initial state is drawn on the host exactly like the reference (numpy
        uniforms for the angles, then randn lengths, randn fractional coordinates from torch's global CPU generator).
        Per-step noise:
          noise="philox" (default): the whole loop is ONE library call (arreau_sample_loop): the three draws of a step are
              generated inside the update kernels from Philox4x32-10 keyed by (seed, timestep, draw, element), the
              timestep lives on the device, nothing happens on the host between steps.  `seed` defaults to a draw from
              torch's global CPU generator (so torch.manual_seed makes runs repeatable).  `use_graph=True` replays one
              captured step as a hipGraph (same trajectory bit for bit; measured on MI355X: 1.70 vs 1.765 ms per step at
              256 x 20, nothing at 1 x 8 where the step is the sum of its kernels' durations; default: on for runs of at
              least 200 steps, which amortise the capture).
          noise="reference": randn[B,3], randn[N,3], rand[N,S] from the global CPU generator in the reference's order
              (diffusion_helpers.py:193-197, :79; d3pm.py:206), uploaded every step -- the parity mode.
          noise="device": the same loop with torch's device generator (three RNG launches per step).
        visualization_setting: LAST writes the final state, ALL every 10th timestep + final, ALL_DETAILED every timestep + final
        (the reference's schedule, diffusion_loss.py:351-370), as `<vis_name>_<timestep>.cif` / `<vis_name>_final.cif`
        structure files (inference/visualize_crystal.py; the reference renders PNGs through plotly + pymatgen).
        `fixed_cell=True` (extension, noise="philox" only): fixed-cell sampling -- the initial cell lengths are re-imposed
        after every step (arreau_sample_loop's d_fixed_lengths); coordinates and species are sampled as usual."""
        frames = visualization_setting != VisualizationSetting.NONE
        if frames and not vis_name:
            raise ValueError("visualization_setting other than NONE needs vis_name (prefix of the frame files)")
        if noise not in ("philox", "device", "reference"):
            raise ValueError("noise must be 'philox', 'device' or 'reference'")
        eng = model.engine()
        dev = eng.device
        S = len(z_table)
        B = int(num_samples_in_batch)
        # Extension over the reference (uniform n only, diffusion_loss.py:308): a sequence gives each crystal of
        # the batch its own atom count (the HIP path works on CSR offsets, so ragged batches cost nothing extra).
        if isinstance(num_atoms_per_sample, (int, np.integer)):
            num_atoms = torch.full((B,), int(num_atoms_per_sample))
        else:
            num_atoms = torch.as_tensor([int(v) for v in num_atoms_per_sample], dtype=torch.long)
            if num_atoms.numel() != B or int(num_atoms.min()) < 1:
                raise ValueError("num_atoms_per_sample must be an int or hold one positive count per crystal of the batch")
        N = int(num_atoms.sum())
        dt = torch.get_default_dtype()
        angles = torch.tensor(np.array([sample_bravais_angles("monoclinic") for _ in range(B)]))
        lengths = torch.randn([B, 3])
        frac_x = torch.randn([N, 3], dtype=dt) * pos_sigma_max
        if constant_atoms is not None:
            atom_types = torch.as_tensor(constant_atoms).reshape(-1).long()
            if atom_types.numel() != N:
                raise ValueError("constant_atoms must hold one class index per atom")
        else:
            atom_types = torch.full((N,), S - 1)

        f32 = dict(device=dev, dtype=torch.float32)
        frac_d = frac_x.to(**f32).contiguous()
        len_d = lengths.to(**f32).contiguous()
        ang_d = angles.to(**f32).contiguous()
        types_d = atom_types.to(device=dev, dtype=torch.int32).contiguous()
        const_d = types_d.clone() if constant_atoms is not None else None
        off_d = crystal_offsets(num_atoms, dev)
        lattice_d = torch.zeros((B, 3, 3), **f32)
        n_steps = self.T - 1 if max_steps is None else min(self.T - 1, int(max_steps))
        if (use_graph or fixed_cell) and noise != "philox":
            raise ValueError("graph replay and fixed-cell sampling need noise='philox' (the in-kernel generator)")

        # One stream.  Running the batch as two pipelined slices on separate streams (`pipelined_slices=2`) was 3 % faster
        # at 256 x 20 on MI355X, but its results are not reproducible -- in about one run in four one crystal differs at the
        # 1e-5 level from the one-stream loop, cause unknown -- and parity comes first: the library
        # refuses the mode unless ARREAU_ALLOW_MULTISTREAM=1 is set (an experiment, not a product mode).
        # The loop as a function of the state buffers: it runs a second time, from the saved initial state, when the default
        # fp16x3 kernels flag an overflow (see below).
        init_state = (frac_d.clone(), types_d.clone(), len_d.clone())
        rng_state = torch.random.get_rng_state()
        cuda_rng_state = torch.cuda.get_rng_state(dev) if noise == "device" else None  # (the draws of noise='device')
        if noise == "philox" and seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            rng_state = torch.random.get_rng_state()

        def run_loop(use_graph):
            eng.set_batch_layout(num_atoms, groups=max(1, int(pipelined_slices)))
            if noise == "philox":
                if use_graph is None:
                    use_graph = n_steps >= 200  # capture + instantiation (about 2 ms) against ~4 us saved per kernel boundary
                fixed = len_d.clone() if fixed_cell else None
                # Frames (diffusion_loss.py:351-370): the loop is cut at the timesteps the reference visualises -- every 10th
                # for ALL, every one for ALL_DETAILED, never the first (T - 1).  The noise is a function of (seed, timestep),
                # so a run in segments is the same trajectory as a run in one call.
                t_first, t_last = self.T - 1, self.T - n_steps
                stops = [t for t in range(t_first - 1, t_last - 1, -1)
                         if (visualization_setting == VisualizationSetting.ALL and t % 10 == 0)
                         or visualization_setting == VisualizationSetting.ALL_DETAILED]
                t_cur = t_first
                for t_stop in stops + [None]:
                    n_seg = (t_cur - t_stop + 1) if t_stop is not None else (t_cur - t_last + 1)
                    if n_seg > 0:
                        eng.sample_loop(frac_d, types_d, len_d, ang_d, off_d, t_cur, n_seg, seed, const_d, lattice_d,
                                        use_graph=bool(use_graph), fixed_lengths=fixed)
                        t_cur -= n_seg
                    if t_stop is not None:
                        vis_crystal_during_sampling(z_table, types_d.cpu().numpy(), lattice_d.cpu().numpy(),
                                                    frac_d.cpu().numpy(), vis_name + f"_{t_stop}", show_bonds, num_atoms.numpy())
            else:
                t_d = torch.empty(B, device=dev, dtype=torch.int32)
                done = 0
                for timestep in reversed(range(1, self.T)):
                    if done >= n_steps:
                        break
                    t_d.fill_(timestep)
                    eps, logits, len0 = eng.predict_scores(frac_d, types_d, len_d, ang_d, t_d, off_d)
                    if noise == "device":
                        z_l = torch.randn((B, 3), **f32)
                        z_f = torch.randn((N, 3), **f32)
                        u_t = torch.rand((N, S), **f32)
                    else:
                        z_l = torch.randn([B, 3]).to(**f32)
                        z_f = torch.randn([N, 3], dtype=dt).to(**f32)
                        u_t = torch.rand([N, S]).to(**f32)
                    eng.reverse_step(frac_d, types_d, len_d, ang_d, t_d, off_d, eps, logits, len0, z_l, z_f, u_t, lattice_d)
                    if const_d is not None:
                        types_d.copy_(const_d)
                    done += 1
                    if timestep != self.T - 1 and ((visualization_setting == VisualizationSetting.ALL and timestep % 10 == 0)
                                                   or visualization_setting == VisualizationSetting.ALL_DETAILED):
                        vis_crystal_during_sampling(z_table, types_d.cpu().numpy(), lattice_d.cpu().numpy(),
                                                    frac_d.cpu().numpy(), vis_name + f"_{timestep}", show_bonds, num_atoms.numpy())

        run_loop(use_graph)
        # Range safety without an environment variable: the fp16x3 kernels never clamp -- an activation beyond 65504 reaches the
        # outputs as NaN and sets the sticky NONFINITE flag.  When that happens on the default kernels the batch is re-run HERE,
        # from its saved initial state and with the same draws (same Philox seed / same host and device generator states), on
        # the full-range bf16x6 kernels; the engine stays on them if that run came out finite, and the result says so
        # (SampleResult.info).  Before that, when fp8 operand planes were in use (basis stash residual, cross products of the layer
        # projections): a basis value beyond e4m3's range turns into NaN there (the hardware conversion does not saturate), so the
        # batch is first re-run with two fp16 planes and three fp16 products, which the engine then keeps.
        info = None
        from .. import _hip as _h
        import warnings

        def rerun():
            eng.status(reset=True)
            frac_d.copy_(init_state[0]); types_d.copy_(init_state[1]); len_d.copy_(init_state[2])
            torch.random.set_rng_state(rng_state)
            if cuda_rng_state is not None:
                torch.cuda.set_rng_state(cuda_rng_state, dev)
            run_loop(False)

        st = eng.status(reset=False)
        if st["flags"] == _h.STATUS_NONFINITE and eng.fp8_formats_in_use(st):
            warnings.warn("arreau_amd: non-finite outputs with fp8 operand planes in use (a basis value beyond e4m3's range?); "
                          "re-running the batch with two fp16 planes and three fp16 products, which this engine keeps from now on")
            eng.set_formats(0, 0)
            rerun()
            info = {"fp16_planes_rerun": True}
            st = eng.status(reset=False)
        overflow = (st["flags"] == _h.STATUS_NONFINITE) and eng.fused_shape \
            and (st["edge_kernel"] == "fp16x3" or st["mlp_kernel"].startswith("fp16x3"))
        if overflow:
            warnings.warn("arreau_amd: an activation left the fp16 range of the split-precision kernels (weights with activation "
                          f"bounds edge {st['edge_activation_bound']:.3g} / node {st['node_activation_bound']:.3g}); re-running the "
                          "batch on the full-range bf16x6 kernels, which this engine keeps if they come out finite")
            previous = (st["edge_variant"], st["mlp_variant"])
            eng.set_variant(3, 1)
            rerun()
            if eng.status(reset=False)["flags"] & _h.STATUS_NONFINITE:
                eng.set_variant(*previous)  # not a range problem (e.g. a degenerate cell): keep the faster kernels, raise below
            info = dict(info or {}, full_range_rerun=True, kernels="bf16x6", edge_activation_bound=st["edge_activation_bound"],
                        node_activation_bound=st["node_activation_bound"])
        eng.check_status()  # sticky device flags (non-finite outputs, clamped indices): raise instead of returning them
        if frames:
            vis_crystal_during_sampling(z_table, types_d.cpu().numpy(), lattice_d.cpu().numpy(), frac_d.cpu().numpy(),
                                        vis_name + "_final", show_bonds, num_atoms.numpy())
        atomic_numbers = atomic_number_indexes_to_atomic_numbers(z_table, types_d.cpu().numpy())
        return SampleResult(num_atoms=num_atoms.numpy(), frac_x=frac_d.cpu().numpy().astype(np.float64),
                            atomic_numbers=atomic_numbers, lattice=lattice_d.cpu().numpy().astype(np.float64), info=info)

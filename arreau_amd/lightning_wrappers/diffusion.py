"""PONITA_DIFFUSION: the module surface of lightning_wrappers/diffusion.py:29-253 of the reference
(constructor arguments, attributes, state_dict keys, `load_from_checkpoint`, `forward`, `sample`)
with the per-step work running in libarreau_hip.so.  pytorch_lightning is not required; the class
is a plain nn.Module that reads and writes Lightning-format checkpoint dicts."""
import collections
import os
import pathlib
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from ..diffusion.diffusion_helpers import GaussianFourierProjection, crystal_offsets
from ..diffusion.diffusion_loss import DiffusionLoss, SampleResult
from ..diffusion.inference.visualize_crystal import VisualizationSetting
from ..diffusion.tools.atomic_number_table import AtomicNumberTable, atomic_symbols_to_indices
from ..ponita.models.ponita import PonitaFiberBundle

fourier_scale = 16
t_emb_dim = 64
OUT_DIR = f"{pathlib.Path(__file__).parent.resolve()}/../out"
DIFFUSION_DIR = f"{OUT_DIR}/diffusion"

ORI_GRID_KEY = "arreau_amd.ori_grid"  # optional extra checkpoint entry (the reference does not persist the grid)


class PONITA_DIFFUSION(nn.Module):
    def __init__(self, args, z_table: AtomicNumberTable, ori_grid=None):
        super().__init__()
        self.hparams = SimpleNamespace(args=args, z_table=z_table)
        self.register_buffer("z_table_zs", torch.tensor(list(z_table.zs), dtype=torch.int64))
        self.dataset = getattr(args, "dataset", "alexandria")
        num_atomic_states = len(z_table)
        self.num_atomic_states = num_atomic_states
        self.lr = getattr(args, "lr", 1e-3)
        self.weight_decay = getattr(args, "weight_decay", 1e-10)
        self.epochs = getattr(args, "epochs", 0)
        self.warmup = getattr(args, "warmup", 0)
        if args.layer_scale == 0.0:
            args.layer_scale = None
        self.train_augm = getattr(args, "train_augm", False)

        self.t_emb = GaussianFourierProjection(t_emb_dim // 2, fourier_scale)
        self.diffusion_loss = DiffusionLoss(args, num_atomic_states)

        in_channels_scalar = num_atomic_states + 64 + 1 + 3 + 3 + 3
        in_channels_vec = 1 + 3
        self.model = PonitaFiberBundle(
            in_channels_scalar + in_channels_vec, args.hidden_dim, num_atomic_states, 3, 0, 0, args.layers,
            output_dim_vec=1, radius=args.radius, num_ori=args.num_ori, basis_dim=args.basis_dim,
            degree=args.degree, widening_factor=args.widening_factor, layer_scale=args.layer_scale,
            multiple_readouts=args.multiple_readouts, ori_grid=ori_grid)
        self._engine = None
        self._device = torch.device("cpu")

    def __deepcopy__(self, memo):
        """copy.deepcopy(module): weights, buffers and the grid are copied; the HIP engine (a handle into the library)
        is not -- the copy packs its own on first use."""
        import copy
        eng, self._engine = self._engine, None
        saved = self._drop_transient()  # (device events cannot be copied; the caches name THIS module's parameters)
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            for k, v in self.__dict__.items():
                new.__dict__[k] = copy.deepcopy(v, memo)
        finally:
            self._engine = eng
            self.__dict__.update(saved)
        return new

    _TRANSIENT = ("_steps_in_flight", "_named_parameter_cache", "_trainable_parameter_cache", "_grad_flat")

    def _drop_transient(self):
        """per-step state of the training loop (parameter maps, the flat gradient buffer, events of steps in flight): rebuilt
        on next use"""
        return {k: self.__dict__.pop(k) for k in self._TRANSIENT if k in self.__dict__}

    # ---- device / engine management --------------------------------------------------------------
    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._engine = None  # parameters moved or changed dtype: repack on next use
        self._drop_transient()
        self._device = self.z_table_zs.device
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        state_dict = dict(state_dict)
        grid = state_dict.pop(ORI_GRID_KEY, None)
        if grid is not None:
            self.model.ori_grid = torch.as_tensor(grid).detach().cpu().clone()
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._engine = None
        self._all_callibrated = False  # re-read the `callibrated` buffers at the next training step
        return out

    def engine(self, for_training: bool = False):
        """The HIP engine for the module's current (cuda) device; packs the weights on first use.  After optimizer
        steps the engine only holds refreshed TRAINING weights (HipEngine.update_train_weights); any other use
        re-creates it from the current parameters."""
        if self._engine is not None and self._engine.stale_for_sampling and not for_training:
            self._engine = None
        if self._engine is None:
            from ..engine import HipEngine
            dev = self._device if self._device.type == "cuda" else torch.device("cuda", 0)
            self._engine = HipEngine(self, dev)
        return self._engine

    # ---- checkpoint I/O (Lightning dict format) --------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = False, **kwargs):
        from ..checkpoint import load_lightning_checkpoint
        ckpt = load_lightning_checkpoint(checkpoint_path)
        hp = ckpt["hyper_parameters"]
        model = cls(hp["args"], hp["z_table"], ori_grid=ckpt["state_dict"].get(ORI_GRID_KEY))
        model.load_state_dict(ckpt["state_dict"], strict=strict)
        if torch.cuda.is_available():
            model = model.to(map_location if map_location is not None else "cuda")
        return model

    # ---- training (BASELINE config 5) --------------------------------------------------------------
    def training_step(self, graph, timestep=None, noise=None):
        """lightning_wrappers/diffusion.py:108-118: one forward + backward of the score-matching loss on a collated batch
        (fields X0, A0, L0, num_atoms).  Leaves d(loss)/d(parameter) in `.grad` of every trainable parameter of
        `self.model` (what loss.backward() does in the reference) and returns the loss as a 0-d CUDA tensor.
        On the first training forward the conv weights are rescaled by the activation std ratios
        (FiberBundleConv.callibrate, ponita/nn/conv.py:121-123,140-146) -- after this step's gradients were taken."""
        # The step never waits for the device (inputs cross in asynchronous copies), so the host could run many steps ahead:
        # it is held to TWO steps in flight -- on the MI355X box an unbounded run-ahead stalled the launch queue for 80 ms at a
        # time (round 4, kernel trace: a gap of that length in the middle of a forward pass; 3.2 -> 20 ms per step).
        inflight = self.__dict__.setdefault("_steps_in_flight", collections.deque())
        if len(inflight) >= 2:
            inflight.popleft().synchronize()
        loss, parts = self.diffusion_loss(self, graph, self.t_emb, timestep=timestep, noise=noise, return_parts=True,
                                          training=True)
        eng = self.engine(for_training=True)
        grads = eng.train_backward(parts["grad_eps"], parts["grad_logits"], parts["grad_lengths"])
        params = self.__dict__.get("_named_parameter_cache")
        if params is None:  # (walking the module tree every step was 0.2 ms of host time; the Parameter objects are stable)
            params = self.__dict__["_named_parameter_cache"] = dict(self.named_parameters())
        covered = True
        for name, g in grads.items():
            p = params.get(name)
            if p is not None and p.requires_grad:
                if g.dtype is p.dtype and g.device == p.device:
                    p.grad = g if g.shape == p.shape else g.reshape(p.shape)  # a view of the engine's flat gradient buffer
                else:
                    p.grad = g.to(device=p.device, dtype=p.dtype).reshape(p.shape)
                    covered = False
            else:
                covered = False
        # every gradient of the step is a view of ONE buffer: the optimizer driver (arreau_amd.train.optimizer_step) may then
        # all-reduce / measure / clip that buffer instead of 70 tensors
        self._grad_flat = eng.last_grad_flat if covered else None
        self._callibrate_if_needed(eng)
        # the device status word is sticky: checking it every few steps loses nothing and keeps the step free of host
        # synchronisation (a check is a device-to-host read)
        self._train_steps = getattr(self, "_train_steps", 0) + 1
        if (self._train_steps - 1) % self.STATUS_CHECK_EVERY == 0:
            st = eng.status(reset=False)
            from .. import _hip as _h
            # (the status read above has synchronised the stream: looking at this step's loss costs nothing more.  The training
            # network does not raise NONFINITE itself -- that flag belongs to the sampling read-out --, the loss shows it)
            nonfinite = st["flags"] == _h.STATUS_NONFINITE or (st["flags"] == 0 and not bool(torch.isfinite(loss.detach()).all()))
            if nonfinite and not getattr(self, "_train_full_range", False):
                # The training forward runs fp16x3 products while the weights fit fp16; its operand bounds were taken at engine
                # creation and the weights have moved since.  An overflow surfaces here as NONFINITE: switch the
                # engine to the full-range bf16x6 products for good (the sampler's remedy), drop the flag and go on -- the
                # optimizer driver has turned the poisoned steps into no-ops (arreau_amd.train.optimizer_step).
                import warnings
                warnings.warn("arreau_amd: a training step came out non-finite on the fp16x3 forward products; this engine trains "
                              "on the full-range bf16x6 products from now on")
                eng.status(reset=True)
                eng.set_variant(3 if eng.fused_shape else -1, 1)
                self._train_full_range = True
            else:
                eng.check_status()
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(eng.device))
        inflight.append(done)
        return loss

    @torch.no_grad()
    def validation_step(self, graph, timestep=None, noise=None):
        """lightning_wrappers/diffusion.py:126-137: the same loss without a backward pass (fused sampling kernels)."""
        return self.diffusion_loss(self, graph, self.t_emb, timestep=timestep, noise=noise)

    test_step = validation_step  # :139-150

    STATUS_CHECK_EVERY = 32

    def _callibrate_if_needed(self, eng):
        layers = self.model.interaction_layers
        if getattr(self, "_all_callibrated", False):  # (the buffers live on the device: read them once, not every step)
            return
        if all(bool(layer.conv.callibrated) for layer in layers):
            self._all_callibrated = True
            return
        st = eng.conv_stats()
        # Data-parallel training: every rank saw different crystals and noise, so its activation statistics differ; the
        # replicas must apply ONE set of ratios or their weights diverge for good (only gradients are exchanged
        # afterwards).  Rank 0's statistics are used everywhere -- what a single-GPU run of the reference computes.
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                dist.broadcast(st, src=0)
            else:
                host = st.cpu()
                dist.broadcast(host, src=0)
                st = host
        st = st.cpu().double()
        with torch.no_grad():
            for l, layer in enumerate(layers):
                if bool(layer.conv.callibrated):
                    continue
                std_in, std_1, std_2 = (float(v) for v in st[l])
                layer.conv.kernel.weight.mul_(std_in / std_1)
                layer.conv.fiber_kernel.weight.mul_(std_1 / std_2)
                layer.conv.callibrated.fill_(True)
        self._all_callibrated = True
        self.notify_parameters_changed()

    @staticmethod
    def _frame_prefix(vis_name, visualization_setting):
        if visualization_setting == VisualizationSetting.NONE:
            return vis_name or ""
        import os
        if vis_name is None:
            os.makedirs(DIFFUSION_DIR, exist_ok=True)  # (:239-240)
            return f"{DIFFUSION_DIR}/step"
        if not vis_name:
            return ""  # DiffusionLoss.sample rejects it
        os.makedirs(os.path.dirname(os.path.abspath(vis_name)), exist_ok=True)
        return vis_name

    def notify_parameters_changed(self):
        """Call after an optimizer step: the HIP engine holds copies of the weights.  An engine that exists is refreshed
        for training in place (device-to-device copies, no host repack); sampling re-creates it on next use."""
        if self._engine is not None:
            self._engine.update_train_weights(self)

    def configure_optimizers(self, max_epochs=None):
        """lightning_wrappers/diffusion.py:152-218: Adam with weight decay on Linear weights only (biases, LayerNorm
        weights, layer_scale and the Fourier projection are not decayed) + the cosine warm-up schedule."""
        from .scheduler import CosineWarmupScheduler
        decay, no_decay = set(), set()
        for mn, mod in self.named_modules():
            for pn, _p in mod.named_parameters(recurse=False):
                fpn = f"{mn}.{pn}" if mn else pn
                if pn.endswith("bias"):
                    no_decay.add(fpn)
                elif pn.endswith("weight") and isinstance(mod, torch.nn.Linear):
                    decay.add(fpn)
                elif pn.endswith("weight") and isinstance(mod, (torch.nn.LayerNorm, torch.nn.Embedding)):
                    no_decay.add(fpn)
                elif pn.endswith("layer_scale") or pn.endswith("gaussian_fourier_proj_w"):
                    no_decay.add(fpn)
        param_dict = dict(self.named_parameters())
        assert not (decay & no_decay) and not (param_dict.keys() - (decay | no_decay)), "parameter grouping is not a partition"
        groups = [{"params": [param_dict[pn] for pn in sorted(decay)], "weight_decay": self.weight_decay},
                  {"params": [param_dict[pn] for pn in sorted(no_decay)], "weight_decay": 0.0}]
        # the reference's torch.optim.Adam(groups, lr); on the GPU the subclass whose step on the training step's flat gradient buffer
        # is two launches of the library (arreau_amd/optim.py: clip + update, 0.19 ms of device time per step through torch's norm,
        # scalar arithmetic and fused multi-tensor Adam; same update rule, same state_dict)
        on_gpu = all(p.is_cuda for g in groups for p in g["params"])
        if on_gpu and os.environ.get("ARREAU_TORCH_ADAM", "0") != "1":   # (ARREAU_TORCH_ADAM=1: torch's fused Adam, for A/B runs)
            from ..optim import ClipAdam
            optimizer = ClipAdam(groups, lr=self.lr)
        else:
            optimizer = torch.optim.Adam(groups, lr=self.lr, fused=True) if on_gpu else torch.optim.Adam(groups, lr=self.lr)
        scheduler = CosineWarmupScheduler(optimizer, self.warmup, max_epochs if max_epochs is not None else self.epochs)
        return {"optimizer": optimizer, "lr_scheduler": scheduler, "monitor": "val_loss"}

    # ---- operator seam ----------------------------------------------------------------------------
    def forward(self, graph):
        """model(batch) of diffusion_loss.py:183-189 / PonitaFiberBundle.forward (ponita.py:88-123).
        `graph` carries x [N,S+74], vec [N,4,3], edge_index [2,E] (sender, receiver; receiver-sorted),
        dists [E], inter_atom_direction [E,3], lattice [B,3,3], batch [N] (sorted), num_atoms [B].  Returns the
        reference's 5-tuple (logits [N,S], vec [N,1,3], global_scalar [B,3], None, [None]*L).

        The tensors are consumed as they are (arreau_ponita_forward embeds the given x and vec with the general
        x . W^T): nothing is decoded back to a sampler state, so soft type vectors, per-atom time embeddings or a
        lattice that disagrees with the length features give what the reference network gives for them."""
        eng = self.engine()
        dev = eng.device
        x = graph.x
        N = x.shape[0]
        num_atoms = getattr(graph, "num_atoms", None)
        batch = getattr(graph, "batch", None)
        if num_atoms is None:
            if batch is None:
                raise ValueError("graph needs num_atoms or batch")
            num_atoms = torch.bincount(torch.as_tensor(batch).to("cpu", torch.int64))
        n_cpu = torch.as_tensor(num_atoms).to("cpu", torch.int64)
        if int(n_cpu.sum()) != N:
            raise ValueError("num_atoms does not add up to the number of nodes")
        if batch is not None:
            expect = torch.arange(n_cpu.numel()).repeat_interleave(n_cpu)
            if not torch.equal(torch.as_tensor(batch).to("cpu", torch.int64), expect):
                raise ValueError("graph.batch must list the atoms of each crystal contiguously, crystals in order")
        f32 = lambda t: torch.as_tensor(t).to(dev, torch.float32).contiguous()
        edges = eng.edges_to_slots(graph.edge_index, graph.dists, graph.inter_atom_direction, N)
        xd, vd, ld, od = f32(x), f32(graph.vec), f32(graph.lattice), crystal_offsets(n_cpu, dev)
        logits, vec_out, gscalar = eng.checked(lambda: eng.ponita_forward(xd, vd, ld, od, edges))
        return logits, vec_out, gscalar, None, [None] * self.model.num_layers

    @torch.no_grad()
    def sample(self, num_atoms_per_sample, num_samples_in_batch: int,
               visualization_setting: VisualizationSetting = VisualizationSetting.NONE, show_bonds: bool = False,
               use_constant_atomic_symbols: Optional[list] = None, noise: str = "philox",
               max_steps: Optional[int] = None, use_graph: Optional[bool] = None,
               seed: Optional[int] = None, fixed_cell: bool = False, vis_name: Optional[str] = None,
               pipelined_slices: int = 1) -> SampleResult:
        """lightning_wrappers/diffusion.py:220-253.  `num_atoms_per_sample` may also be a sequence with one atom count
        per crystal of the batch (extension; the reference supports a single int).  Frames of a visualization_setting
        other than NONE go to `<DIFFUSION_DIR>/step_<timestep>.cif` like the reference's PNGs (`vis_name` overrides the
        prefix)."""
        z_table = AtomicNumberTable(self.z_table_zs.tolist())
        if use_constant_atomic_symbols is not None:
            if not isinstance(num_atoms_per_sample, (int, np.integer)):
                raise ValueError("use_constant_atomic_symbols needs a uniform num_atoms_per_sample")
            # one index per atom of a crystal, tiled over the batch (the reference's np.repeat at :236
            # interleaves instead of tiling and only works for single-species lists; tiling is the intent)
            idx = atomic_symbols_to_indices(z_table, use_constant_atomic_symbols)
            constant_atoms = torch.as_tensor(np.tile(idx, num_samples_in_batch))
        else:
            constant_atoms = None
        return self.diffusion_loss.sample(
            model=self, z_table=z_table, t_emb_weights=self.t_emb, num_atoms_per_sample=num_atoms_per_sample,
            num_samples_in_batch=num_samples_in_batch, vis_name=self._frame_prefix(vis_name, visualization_setting),
            visualization_setting=visualization_setting, show_bonds=show_bonds, constant_atoms=constant_atoms,
            noise=noise, max_steps=max_steps, use_graph=use_graph, seed=seed, fixed_cell=fixed_cell,
            pipelined_slices=pipelined_slices)

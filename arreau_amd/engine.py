"""HipEngine: owns the arreau_model handle (packed weights in HBM) and the step workspace.

Host-side plumbing only: tensors are allocated with torch, every arithmetic step on the sampler
state runs in libarreau_hip.so.
"""
import ctypes
import math
import os

import torch

from . import _hip


def _f32(t):
    return t.detach().to("cpu", torch.float32).contiguous()


def pack_state(module):
    """(Config, host tensors, StateDict of their pointers) of a PONITA_DIFFUSION in the reference's state_dict layout: what
    arreau_model_create and arreau_calibrate_formats take.  Host side only; the tensors must outlive every use of the struct."""
    net = module.model
    sd = module.state_dict()
    L = net.num_layers
    S = module.num_atomic_states
    cfg = _hip.Config(
        num_atomic_states=S, hidden_dim=net.hidden_dim, basis_dim=net.basis_dim, num_layers=L,
        num_ori=net.num_ori, widening_factor=net.widening_factor, degree=net.degree,
        max_neighbors=int(module.diffusion_loss.max_neighbors), num_timesteps=int(module.diffusion_loss.T),
        radius=float(module.diffusion_loss.cutoff),
        has_layer_scale=int(f"model.interaction_layers.0.layer_scale" in sd and
                            sd["model.interaction_layers.0.layer_scale"] is not None))
    il = "model.interaction_layers.{}."
    stack = lambda fmt: torch.stack([_f32(sd[fmt.format(i)]) for i in range(L)], 0).contiguous()
    host = {
        "basis_w1": _f32(sd["model.basis_fn.1.weight"]), "basis_b1": _f32(sd["model.basis_fn.1.bias"]),
        "basis_w2": _f32(sd["model.basis_fn.3.weight"]), "basis_b2": _f32(sd["model.basis_fn.3.bias"]),
        "fiber_w1": _f32(sd["model.fiber_basis_fn.1.weight"]), "fiber_b1": _f32(sd["model.fiber_basis_fn.1.bias"]),
        "fiber_w2": _f32(sd["model.fiber_basis_fn.3.weight"]), "fiber_b2": _f32(sd["model.fiber_basis_fn.3.bias"]),
        "x_embedder_w": _f32(sd["model.x_embedder.weight"]),
        "conv_kernel_w": stack(il + "conv.kernel.weight"), "conv_fiber_w": stack(il + "conv.fiber_kernel.weight"),
        "conv_bias": stack(il + "conv.bias"), "norm_w": stack(il + "norm.weight"), "norm_b": stack(il + "norm.bias"),
        "linear1_w": stack(il + "linear_1.weight"), "linear1_b": stack(il + "linear_1.bias"),
        "linear2_w": stack(il + "linear_2.weight"), "linear2_b": stack(il + "linear_2.bias"),
        "readout_w": stack("model.read_out_layers.{}.weight"), "readout_b": stack("model.read_out_layers.{}.bias"),
        "ori_grid": _f32(net.ori_grid), "t_emb_w": _f32(sd["t_emb.gaussian_fourier_proj_w"]),
        "ve_sigmas": _f32(sd["diffusion_loss.pos_diffusion.sigmas"]),
        "vp_alpha_bars": _f32(sd["diffusion_loss.lattice_diffusion.alpha_bars"]),
        "vp_betas": _f32(sd["diffusion_loss.lattice_diffusion.betas"]),
        "q_one_step_transposed": _f32(sd["diffusion_loss.d3pm.q_one_step_transposed"]),
        "q_mats": _f32(sd["diffusion_loss.d3pm.q_mats"]),
    }
    if cfg.has_layer_scale:
        host["layer_scale"] = stack(il + "layer_scale")
    expect = {"basis_w1": (net.hidden_dim, 258), "x_embedder_w": (net.hidden_dim, S + 78),
              "readout_w": (L, S + 4, net.hidden_dim), "ori_grid": (net.num_ori, 3),
              "q_mats": (cfg.num_timesteps, S, S), "ve_sigmas": (cfg.num_timesteps + 1,)}
    for k, shp in expect.items():
        if tuple(host[k].shape) != shp:
            raise ValueError(f"state_dict entry for {k} has shape {tuple(host[k].shape)}, expected {shp}")
    csd = _hip.StateDict()
    for name in _hip._SD_FIELDS:
        t = host.get(name)
        setattr(csd, name, t.data_ptr() if t is not None else None)
    return cfg, host, csd, S, L


class HipEngine:
    def __init__(self, module, device):
        """module: a PONITA_DIFFUSION (state_dict layout of the reference); device: cuda device."""
        _hip.require_gpu()
        self.device = torch.device(device)
        cfg, host, csd, S, L = pack_state(module)
        self.cfg = cfg
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().arreau_model_create(ctypes.byref(cfg), ctypes.byref(csd),
                                                      _hip.stream_ptr(self.device), ctypes.byref(self._handle)),
                       "arreau_model_create")
        self._host_keepalive = host
        self._ws = None
        self._ws_cap = (0, 0)
        self.S, self.k = S, cfg.max_neighbors
        self.stale_for_sampling = False
        # shapes without fused kernels run on the shape-general fp32 kernels (csrc/train_net.hip), which read the plain
        # weights update_train_weights refreshes: such an engine never goes stale for sampling
        self.fused_shape = (cfg.hidden_dim, cfg.basis_dim, cfg.widening_factor) == (128, 256, 4)
        self._general = not self.fused_shape or bool(os.environ.get("ARREAU_GENERAL_PATH"))

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle.value:
            _hip.lib().arreau_model_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def status(self, reset=False):
        """Sticky device-side condition flags + the kernel families the last predict_scores launched
        (arreau_model_status; synchronises the stream)."""
        st = _hip.Status()
        _hip.check(_hip.lib().arreau_model_status(self._handle, ctypes.byref(st), int(bool(reset)),
                                                   _hip.stream_ptr(self.device)), "arreau_model_status")
        return {"flags": int(st.flags), "edge_kernel": _hip.EDGE_KERNELS.get(st.edge_kernel, "none"),
                "mlp_kernel": _hip.MLP_KERNELS.get(st.mlp_kernel, "none"), "edge_variant": int(st.edge_kernel),
                "mlp_variant": int(st.mlp_kernel), "conv_variant": int(st.conv_kernel),
                "basis_row_bytes": int(st.basis_row_bytes), "conv_cross_fp8": int(st.conv_cross_fp8), "edge_activation_bound": float(st.edge_activation_bound),
                "node_activation_bound": float(st.node_activation_bound),
                "basis_fp8_share": float(st.basis_fp8_share), "cross_fp8_share": float(st.cross_fp8_share)}

    def check_status(self, reset=True):
        """Raise if a kernel flagged a condition under which its results must not be trusted."""
        st = self.status(reset=reset)
        f = st["flags"]
        if f:
            why = []
            if f & _hip.STATUS_NONFINITE:
                why.append("a network output is inf/NaN (an activation left the fp16 range of the fp16x3 kernels, "
                           "or the inputs/weights are non-finite); ARREAU_EDGE_VARIANT=3 ARREAU_MLP_VARIANT=1 selects the "
                           "full-range bf16x6 kernels")
            if f & _hip.STATUS_BAD_TIMESTEP:
                why.append("a timestep index was outside the schedule")
            if f & _hip.STATUS_BAD_TYPE:
                why.append("an atom-type index was outside [0, num_atomic_states)")
            raise _hip.ArreauHipError("arreau_hip status flags %d: %s" % (f, "; ".join(why)))
        return st

    def set_formats(self, basis_fp8=-1, cross_fp8=-1):
        """Operand formats of the message path for this model (arreau_model_set_formats; -1 keeps the current choice)."""
        _hip.check(_hip.lib().arreau_model_set_formats(self._handle, int(basis_fp8), int(cross_fp8)), "arreau_model_set_formats")

    def fp8_formats_in_use(self, st=None):
        """Did the last evaluation read an fp8 residual plane / run fp8 cross products?  (arreau_model_status)"""
        st = self.status(reset=False) if st is None else st
        return st["conv_variant"] == 2 and (st["basis_row_bytes"] == 768 or st["conv_cross_fp8"] == 1)

    def checked(self, fn):
        """Run `fn()` (one evaluation through this engine), then read the sticky flags.  A NONFINITE flag while fp8 operand
        formats were in use is first taken for what it usually is -- a basis value beyond e4m3's range (the hardware's fp8
        conversion returns NaN above 464; tools/exp/fp8_cvt_check.hip): the model is switched to two fp16 planes and three fp16
        products for good and `fn()` runs once more; its result is the one returned.  Any flag that survives raises."""
        out = fn()
        st = self.status(reset=False)
        if st["flags"] == _hip.STATUS_NONFINITE and self.fp8_formats_in_use(st):
            import warnings
            warnings.warn("arreau_amd: non-finite outputs with fp8 operand planes in use (a basis value beyond e4m3's range?); this "
                          "engine keeps two fp16 planes and three fp16 products from now on and the evaluation is repeated with them")
            self.set_formats(0, 0)
            self.status(reset=True)
            out = fn()
        self.check_status()
        return out

    def set_batch_layout(self, num_atoms, groups=0):
        """Tell the library the (host-side) atom count of every crystal of the batches that follow, so that it may run
        the score network as `groups` crystal-aligned slices (arreau_model_set_batch_layout; groups = 0: the library's
        default, ARREAU_GROUPS or off).  Slices on SEPARATE STREAMS are an experiment the library refuses unless
        ARREAU_ALLOW_MULTISTREAM=1 is set: with kernels of two streams sharing CUs results were seen to change at the
        1e-8 .. 1e-4 level in rare runs, cause unknown."""
        n = torch.as_tensor(num_atoms).to("cpu", torch.int64).reshape(-1)
        off = torch.zeros(n.numel() + 1, dtype=torch.int32)
        off[1:] = torch.cumsum(n, 0).to(torch.int32)
        _hip.check(_hip.lib().arreau_model_set_batch_layout(self._handle, ctypes.c_void_p(off.data_ptr()), int(n.numel()),
                                                            int(groups)), "arreau_model_set_batch_layout")

    def set_variant(self, edge=-1, mlp=-1):
        """Select the arithmetic of the dense kernels (edge: 0 fp32 MFMA, 3 bf16x6, 4 fp16x3; mlp: 0 fp32 MFMA,
        1 bf16x6, 2/3 fp16x3, 4 = always the (bit-identical) small-launch form of 3); -1 keeps.  edge = 5 runs the whole network on the shape-general fp32 GEMM kernels (the only
        choice for shapes other than hidden_dim 128 / basis_dim 256 / widening 4)."""
        _hip.check(_hip.lib().arreau_model_set_variant(self._handle, int(edge), int(mlp)), "arreau_model_set_variant")
        if int(edge) >= 0:
            self._general = int(edge) == _hip.VARIANT_GENERAL

    def ponita_forward(self, x, vec, lattice, offsets, edges):
        """The inner operator seam (PonitaFiberBundle.forward on the reference's batch attributes).
        x [N,S+74], vec [N,4,3], lattice [B,3,3] f32; offsets [B+1] i32; edges = slot arrays (deg, src, dir, dist).
        Returns (logits [N,S], vec_out [N,1,3], global_scalar [B,3])."""
        dev = self.device
        N, B = x.shape[0], lattice.shape[0]
        if x.shape[1] != self.S + 74 or tuple(vec.shape) != (N, 4, 3):
            raise ValueError(f"x must be [N,{self.S + 74}] and vec [N,4,3]; got {tuple(x.shape)}, {tuple(vec.shape)}")
        logits = torch.empty((N, self.S), device=dev, dtype=torch.float32)
        vec_out = torch.empty((N, 1, 3), device=dev, dtype=torch.float32)
        gscalar = torch.empty((B, 3), device=dev, dtype=torch.float32)
        ws = self.workspace(N, B)
        deg, src, direction, dist = edges
        _hip.check(_hip.lib().arreau_ponita_forward(
            self._handle, _hip.ptr(x), _hip.ptr(vec), _hip.ptr(lattice), _hip.ptr(offsets), B, N, _hip.ptr(deg),
            _hip.ptr(src), _hip.ptr(direction), _hip.ptr(dist), _hip.ptr(logits), _hip.ptr(vec_out), _hip.ptr(gscalar),
            _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)), "arreau_ponita_forward")
        return logits, vec_out, gscalar

    def sample_loop(self, frac, types, lengths, angles, offsets, t_start, n_steps, seed, const_types, lattice_out,
                    use_graph=False, fixed_lengths=None):
        """n_steps iterations of the sampling loop in one library call (arreau_sample_loop): in-place update of
        (frac, types, lengths); Philox noise keyed by (seed, timestep, draw, element)."""
        N, B = frac.shape[0], lengths.shape[0]
        ws = self.workspace(N, B)
        _hip.check(_hip.lib().arreau_sample_loop(
            self._handle, _hip.ptr(frac), _hip.ptr(types), _hip.ptr(lengths), _hip.ptr(angles), _hip.ptr(offsets), B, N,
            int(t_start), int(n_steps), int(seed) & (2 ** 64 - 1), _hip.ptr(const_types), _hip.ptr(fixed_lengths), _hip.ptr(lattice_out), _hip.ptr(ws),
            ws.numel(), int(bool(use_graph)), _hip.stream_ptr(self.device)), "arreau_sample_loop")

    def philox_fill(self, seed, timestep, kind, n, raw=False):
        """The sampler's in-kernel noise written out (arreau_philox_fill): kind 0/1 standard normal, 2 uniform [0,1)."""
        out = torch.empty(n, device=self.device, dtype=torch.float32)
        words = torch.empty((n, 4), device=self.device, dtype=torch.int32) if raw else None
        _hip.check(_hip.lib().arreau_philox_fill(int(seed) & (2 ** 64 - 1), int(timestep), int(kind), int(n), _hip.ptr(out),
                                                 _hip.ptr(words), _hip.stream_ptr(self.device)), "arreau_philox_fill")
        return (out, words) if raw else out

    def diffusion_noise(self, frac0, types0, lattice0, t_crystal, offsets, z_frac, u_types, z_lengths):
        """Forward noising of a clean batch (arreau_diffusion_noise).  Returns dict(noisy_frac, target_eps, noisy_types,
        noisy_lengths, lengths, angles)."""
        dev = self.device
        N, B = frac0.shape[0], lattice0.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        out = dict(noisy_frac=torch.empty((N, 3), **f32), target_eps=torch.empty((N, 3), **f32),
                   noisy_types=torch.empty(N, device=dev, dtype=torch.int32), noisy_lengths=torch.empty((B, 3), **f32),
                   lengths=torch.empty((B, 3), **f32), angles=torch.empty((B, 3), **f32))
        inv = torch.empty((B, 3, 3), **f32)
        _hip.check(_hip.lib().arreau_diffusion_noise(
            self._handle, _hip.ptr(frac0), _hip.ptr(types0), _hip.ptr(lattice0), _hip.ptr(t_crystal), _hip.ptr(offsets),
            B, N, _hip.ptr(z_frac), _hip.ptr(u_types), _hip.ptr(z_lengths), _hip.ptr(out["noisy_frac"]),
            _hip.ptr(out["target_eps"]), _hip.ptr(out["noisy_types"]), _hip.ptr(out["noisy_lengths"]),
            _hip.ptr(out["lengths"]), _hip.ptr(out["angles"]), _hip.ptr(inv), _hip.stream_ptr(dev)),
            "arreau_diffusion_noise")
        return out

    def diffusion_losses(self, pred_eps, target_eps, logits, types0, noisy_types, t_crystal, pred_lengths, lengths,
                         offsets, with_grads=False):
        """The three training errors (arreau_diffusion_losses).  Returns losses[6] = (loss, error_frac_x,
        error_atomic_type, error_lattice, vb, ce) [+ (grad_eps, grad_logits, grad_lengths)]."""
        dev = self.device
        N, B = pred_eps.shape[0], pred_lengths.shape[0]
        f32 = dict(device=dev, dtype=torch.float32)
        terms = torch.empty((N, 3), **f32)
        losses = torch.empty(6, **f32)
        g = (torch.empty((N, 3), **f32), torch.empty((N, self.S), **f32), torch.empty((B, 3), **f32)) if with_grads \
            else (None, None, None)
        _hip.check(_hip.lib().arreau_diffusion_losses(
            self._handle, _hip.ptr(pred_eps), _hip.ptr(target_eps), _hip.ptr(logits), _hip.ptr(types0),
            _hip.ptr(noisy_types), _hip.ptr(t_crystal), _hip.ptr(pred_lengths), _hip.ptr(lengths), _hip.ptr(offsets), B, N,
            _hip.ptr(terms), _hip.ptr(losses), _hip.ptr(g[0]), _hip.ptr(g[1]), _hip.ptr(g[2]), _hip.stream_ptr(dev)),
            "arreau_diffusion_losses")
        return (losses, g) if with_grads else losses

    # ---- training (BASELINE config 5) ---------------------------------------------------------------
    def train_forward(self, frac, types, lengths, angles, t_crystal, offsets):
        """Training-mode network evaluation (arreau_train_forward): fp32, activations kept for train_backward.
        Same inputs / outputs as predict_scores."""
        dev = self.device
        N, B = frac.shape[0], lengths.shape[0]
        eps = torch.empty((N, 3), device=dev, dtype=torch.float32)
        logits = torch.empty((N, self.S), device=dev, dtype=torch.float32)
        len0 = torch.empty((B, 3), device=dev, dtype=torch.float32)
        _hip.check(_hip.lib().arreau_train_forward(
            self._handle, _hip.ptr(frac), _hip.ptr(types), _hip.ptr(lengths), _hip.ptr(angles), _hip.ptr(t_crystal),
            _hip.ptr(offsets), B, N, _hip.ptr(eps), _hip.ptr(logits), _hip.ptr(len0), _hip.stream_ptr(dev)),
            "arreau_train_forward")
        return eps, logits, len0

    def train_backward(self, grad_eps, grad_logits, grad_len0):
        """Backward pass of the last train_forward (arreau_train_backward).  Returns {state_dict key: gradient} for every
        trainable tensor of the score network, on the device, in the reference's parameter names and shapes."""
        dev = self.device
        cfg = self.cfg
        S, C, D, L, W = self.S, cfg.hidden_dim, cfg.basis_dim, cfg.num_layers, cfg.widening_factor
        H = W * C
        shapes = {"basis_w1": (C, 258), "basis_b1": (C,), "basis_w2": (D, C), "basis_b2": (D,), "fiber_w1": (C, 3),
                  "fiber_b1": (C,), "fiber_w2": (D, C), "fiber_b2": (D,), "x_embedder_w": (C, S + 78),
                  "conv_kernel_w": (L, C, D), "conv_fiber_w": (L, C, D), "conv_bias": (L, C), "norm_w": (L, C),
                  "norm_b": (L, C), "linear1_w": (L, H, C), "linear1_b": (L, H), "linear2_w": (L, C, H),
                  "linear2_b": (L, C), "layer_scale": (L, C), "readout_w": (L, S + 4, C), "readout_b": (L, S + 4)}
        # one zero-filled buffer per step, the gradients are views of it (21 fill launches were 0.1 ms of the step); every
        # view starts on a 16-byte boundary (the kernels write some of them as float4 columns)
        sizes = {k: -(-math.prod(v) // 4) * 4 for k, v in shapes.items()}
        flat = torch.zeros(sum(sizes.values()), device=dev, dtype=torch.float32)
        g, o = {}, 0
        for k, shp in shapes.items():
            g[k] = flat[o:o + math.prod(shp)].view(shp)
            o += sizes[k]
        csd = _hip.StateDict()
        for name in _hip._SD_FIELDS:
            t = g.get(name)
            setattr(csd, name, t.data_ptr() if t is not None else None)
        _hip.check(_hip.lib().arreau_train_backward(self._handle, _hip.ptr(grad_eps), _hip.ptr(grad_logits),
                                                    _hip.ptr(grad_len0), ctypes.byref(csd), _hip.stream_ptr(dev)),
                   "arreau_train_backward")
        self.last_grad_flat = flat  # every gradient of the step is a view of this buffer (one all-reduce / norm / scale)
        return {name: (g[field] if l is None else g[field][l]) for name, field, l in self._state_names()}

    def _state_names(self):
        """(state_dict key, C-API field, layer or None) for every trainable tensor of the score network (built once)."""
        names = getattr(self, "_state_names_cache", None)
        if names is None:
            cfg = self.cfg
            names = [("model.basis_fn.1.weight", "basis_w1", None), ("model.basis_fn.1.bias", "basis_b1", None),
                     ("model.basis_fn.3.weight", "basis_w2", None), ("model.basis_fn.3.bias", "basis_b2", None),
                     ("model.fiber_basis_fn.1.weight", "fiber_w1", None), ("model.fiber_basis_fn.1.bias", "fiber_b1", None),
                     ("model.fiber_basis_fn.3.weight", "fiber_w2", None), ("model.fiber_basis_fn.3.bias", "fiber_b2", None),
                     ("model.x_embedder.weight", "x_embedder_w", None)]
            il = "model.interaction_layers.{}."
            per_layer = {"conv.kernel.weight": "conv_kernel_w", "conv.fiber_kernel.weight": "conv_fiber_w", "conv.bias": "conv_bias",
                         "norm.weight": "norm_w", "norm.bias": "norm_b", "linear_1.weight": "linear1_w",
                         "linear_1.bias": "linear1_b", "linear_2.weight": "linear2_w", "linear_2.bias": "linear2_b"}
            if cfg.has_layer_scale:
                per_layer["layer_scale"] = "layer_scale"
            for l in range(cfg.num_layers):
                for key, field in per_layer.items():
                    names.append((il.format(l) + key, field, l))
                names.append((f"model.read_out_layers.{l}.weight", "readout_w", l))
                names.append((f"model.read_out_layers.{l}.bias", "readout_b", l))
            self._state_names_cache = names
        return names

    def update_train_weights(self, module):
        """After an optimizer step: push the module's updated parameters (device tensors) into the plain fp32 weights the
        training entry points read (arreau_model_update_train_weights; device-to-device, no host repack).  The engine is
        then `stale_for_sampling` until it is rebuilt.

        The stacked [L, ...] operands of the C API live in persistent buffers, filled by ONE multi-tensor copy from the
        parameters (round 4: state_dict() + 70 conversions + 13 torch.stack per step were 0.5 ms of host time in a step whose
        host side had become the bottleneck)."""
        dev = self.device
        plan = getattr(self, "_train_weight_plan", None)
        params = getattr(module, "_named_parameter_cache", None)
        if params is None:
            params = dict(module.named_parameters())
        if plan is None or plan["module"] is not module or any(p.data_ptr() != q for p, q in zip(plan["src"], plan["ptrs"])):
            L = self.cfg.num_layers
            bufs, dst, src = {}, [], []
            for name, field, l in self._state_names():
                p = params[name]
                if field not in bufs:
                    bufs[field] = torch.empty(((L,) if l is not None else ()) + tuple(p.shape), device=dev, dtype=torch.float32)
                dst.append(bufs[field] if l is None else bufs[field][l])
                src.append(p)
            csd = _hip.StateDict()
            for name in _hip._SD_FIELDS:
                t = bufs.get(name)
                setattr(csd, name, t.data_ptr() if t is not None else None)
            plan = {"module": module, "bufs": bufs, "dst": dst, "src": src, "ptrs": [p.data_ptr() for p in src], "csd": csd}
            self._train_weight_plan = plan
        with torch.no_grad():
            torch._foreach_copy_(plan["dst"], [p.detach() for p in plan["src"]])
        _hip.check(_hip.lib().arreau_model_update_train_weights(self._handle, ctypes.byref(plan["csd"]), _hip.stream_ptr(dev)),
                   "arreau_model_update_train_weights")
        self.stale_for_sampling = not self._general

    def train_weight_mirrors(self, module):
        """{parameter: device address of the engine's own fp32 copy of it} for the optimizer's second destination
        (arreau_model_train_weight_pointers; arreau_amd/optim.py) -- every trainable tensor of the score network except the two the
        training entry points read in a derived form."""
        cached = getattr(self, "_mirror_cache", None)
        params = getattr(module, "_named_parameter_cache", None)
        if params is None:
            params = dict(module.named_parameters())
        if cached is not None and cached[0] is module and all(p.data_ptr() == q for p, q in cached[2]):
            return cached[1]
        csd = _hip.StateDict()
        _hip.check(_hip.lib().arreau_model_train_weight_pointers(self._handle, ctypes.byref(csd)), "arreau_model_train_weight_pointers")
        mirrors = {}
        for name, field, l in self._state_names():
            base = getattr(csd, field)
            p = params[name]
            if base:
                mirrors[p] = int(base) + 4 * p.numel() * (l or 0)
        self._mirror_cache = (module, mirrors, [(p, p.data_ptr()) for p in mirrors])
        return mirrors

    def refresh_derived_train_weights(self, module):
        """After an optimizer step that wrote the engine's copies itself (ClipAdam.step_flat with `mirrors`): rebuild the folded
        polynomial weight and the transposed embedder from the module's updated tensors (two small launches)."""
        params = getattr(module, "_named_parameter_cache", None)
        if params is None:
            params = dict(module.named_parameters())
        _hip.check(_hip.lib().arreau_model_refresh_derived_train_weights(
            self._handle, _hip.ptr(params["model.basis_fn.1.weight"].detach()), _hip.ptr(params["model.x_embedder.weight"].detach()),
            _hip.stream_ptr(self.device)), "arreau_model_refresh_derived_train_weights")
        self.stale_for_sampling = not self._general

    def conv_stats(self):
        """[L,3] unbiased std of (x, x_1, x_2) per layer from the last train_forward (FiberBundleConv.callibrate)."""
        st = torch.empty((self.cfg.num_layers, 3), device=self.device, dtype=torch.float32)
        _hip.check(_hip.lib().arreau_train_conv_stats(self._handle, _hip.ptr(st), _hip.stream_ptr(self.device)),
                   "arreau_train_conv_stats")
        return st

    def workspace(self, N, B):
        if self._ws is None or N > self._ws_cap[0] or B > self._ws_cap[1]:
            capN, capB = max(N, self._ws_cap[0]), max(B, self._ws_cap[1])
            nbytes = _hip.lib().arreau_workspace_bytes(ctypes.byref(self.cfg), capN, capB)
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_cap = (capN, capB)
        return self._ws

    def predict_scores(self, frac, types, lengths, angles, t_crystal, offsets, edges=None, return_edges=False):
        """frac [N,3] f32, types [N] i32, lengths/angles [B,3] f32, t_crystal [B] i32, offsets [B+1] i32 (device).
        edges = (deg [N] i32, src [N,k] i32, dir [N,k,3] f32, dist [N,k] f32) teacher-forces the graph.
        Returns (eps [N,3], logits [N,S], len0 [B,3]) [+ edges]."""
        dev = self.device
        N, B = frac.shape[0], lengths.shape[0]
        eps = torch.empty((N, 3), device=dev, dtype=torch.float32)
        logits = torch.empty((N, self.S), device=dev, dtype=torch.float32)
        len0 = torch.empty((B, 3), device=dev, dtype=torch.float32)
        ws = self.workspace(N, B)
        if edges is not None:
            deg, src, direction, dist = edges
            given = 1
        elif return_edges:
            k = self.k
            deg = torch.empty(N, device=dev, dtype=torch.int32)
            src = torch.empty((N, k), device=dev, dtype=torch.int32)
            direction = torch.empty((N, k, 3), device=dev, dtype=torch.float32)
            dist = torch.empty((N, k), device=dev, dtype=torch.float32)
            given = 0
        else:
            deg = src = direction = dist = None
            given = 0
        _hip.check(_hip.lib().arreau_predict_scores(
            self._handle, _hip.ptr(frac), _hip.ptr(types), _hip.ptr(lengths), _hip.ptr(angles), _hip.ptr(t_crystal),
            _hip.ptr(offsets), B, N, given, _hip.ptr(deg), _hip.ptr(src), _hip.ptr(direction), _hip.ptr(dist),
            _hip.ptr(eps), _hip.ptr(logits), _hip.ptr(len0), _hip.ptr(ws), ws.numel(), _hip.stream_ptr(dev)),
            "arreau_predict_scores")
        if return_edges:
            return eps, logits, len0, (deg, src, direction, dist)
        return eps, logits, len0

    def reverse_step(self, frac, types, lengths, angles, t_crystal, offsets, eps, logits, len0, z_lattice, z_frac,
                     u_types, lattice_out):
        """In-place update of (frac, types, lengths); lattice_out [B,3,3] receives the new cell."""
        B, N = lengths.shape[0], frac.shape[0]
        _hip.check(_hip.lib().arreau_reverse_step(
            self._handle, _hip.ptr(frac), _hip.ptr(types), _hip.ptr(lengths), _hip.ptr(angles), _hip.ptr(t_crystal),
            _hip.ptr(offsets), B, N, _hip.ptr(eps), _hip.ptr(logits), _hip.ptr(len0), _hip.ptr(z_lattice),
            _hip.ptr(z_frac), _hip.ptr(u_types), _hip.ptr(lattice_out), _hip.stream_ptr(self.device)),
            "arreau_reverse_step")

    def edges_to_slots(self, edge_index, dists, direction, N):
        """Receiver-sorted COO edges -> slot form (deg, src, dir, dist)."""
        dev, k = self.device, self.k
        ei = edge_index.to(device=dev, dtype=torch.int64).contiguous()
        E = ei.shape[1]
        d = dists.to(device=dev, dtype=torch.float32).contiguous()
        dr = direction.to(device=dev, dtype=torch.float32).contiguous()
        deg = torch.empty(N, device=dev, dtype=torch.int32)
        src = torch.empty((N, k), device=dev, dtype=torch.int32)
        sdir = torch.empty((N, k, 3), device=dev, dtype=torch.float32)
        sdist = torch.empty((N, k), device=dev, dtype=torch.float32)
        status = torch.zeros(1, device=dev, dtype=torch.int32)
        _hip.check(_hip.lib().arreau_edges_to_slots(_hip.ptr(ei), _hip.ptr(d), _hip.ptr(dr), E, N, k, _hip.ptr(deg),
                                                     _hip.ptr(src), _hip.ptr(sdir), _hip.ptr(sdist), _hip.ptr(status),
                                                     _hip.stream_ptr(dev)), "arreau_edges_to_slots")
        if int(status.item()) != 0:
            raise ValueError(f"edge list must be receiver-sorted with at most {k} in-edges per node")
        return deg, src, sdir, sdist

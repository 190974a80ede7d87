"""torch.optim.Adam whose step on the training step's flat gradient buffer is two launches of libarreau_hip.so.

The reference clips with `pl.Trainer(gradient_clip_val=0.5)` (main_diffusion.py:297) and steps torch.optim.Adam over two parameter
groups (lightning_wrappers/diffusion.py:152-218).  PONITA_DIFFUSION.training_step leaves every gradient as a view of ONE buffer;
`ClipAdam.step_flat` hands that buffer to arreau_optimizer_step (arreau_amd/csrc/optim.hip: norm, clip coefficient, non-finite guard
and the Adam update of all 70 tensors), where torch takes a norm, nine scalar launches, a scale, a select and seven multi-tensor
launches.  Everything else is torch.optim.Adam: parameter groups, the LR scheduler's view of them, `state_dict()` /
`load_state_dict()` (the moments are ordinary per-parameter tensors -- views of two flat buffers), and `step()` itself, which stays
the generic path for gradients that are not views of one buffer (and the only one on the CPU).
"""
import ctypes

import torch


class ClipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._table = None       # (key, handle)
        self._m_flat = self._v_flat = None

    def __del__(self):
        self._drop_table()

    def _drop_table(self):
        tab = self.__dict__.get("_table")
        if tab is not None:
            try:
                from . import _hip
                _hip.lib().arreau_optimizer_destroy(tab[1])
            except Exception:  # interpreter shutdown
                pass
            self._table = None

    def _flat_entries(self, flat):
        """[(param, offset in `flat`, group index)] when every gradient is a contiguous fp32 view of `flat`; None otherwise."""
        from . import _hip
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous() and flat.dim() == 1):
            return None
        if len(self.param_groups) > _hip.OPT_MAX_GROUPS:
            return None
        base, base_off = flat.untyped_storage().data_ptr(), flat.storage_offset()
        entries = []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
                return None
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (g.untyped_storage().data_ptr() == base and g.is_contiguous() and g.dtype == torch.float32 and
                        p.dtype == torch.float32 and p.is_contiguous() and p.device == flat.device and not g.is_sparse):
                    return None
                off = g.storage_offset() - base_off
                if off < 0 or off + p.numel() > flat.numel():
                    return None
                entries.append((p, off, gi))
        return entries or None

    def step_flat(self, flat, max_norm=None, mirrors=None):
        """Clip `flat` (the buffer all `.grad`s are views of) to `max_norm` and step.  Returns the gradient norm (0-d device tensor),
        or None when the gradients are not laid out that way -- the caller then clips and calls step().
        `mirrors` {parameter: device address}: a second destination for the updated values (HipEngine.train_weight_mirrors)."""
        from . import _hip
        entries = self._flat_entries(flat)
        if entries is None:
            return None
        betas = {tuple(g["betas"]) for g in self.param_groups}
        epss = {g["eps"] for g in self.param_groups}
        if len(betas) != 1 or len(epss) != 1:
            return None
        L = _hip.lib()
        mirrors = mirrors or {}
        key = (flat.numel(), tuple((p.data_ptr(), off, gi, mirrors.get(p, 0)) for p, off, gi in entries))
        if self._table is None or self._table[0] != key:
            self._drop_table()
            n = len(entries)
            ptrs = (ctypes.c_void_p * n)(*[p.data_ptr() for p, _, _ in entries])
            mirr = (ctypes.c_void_p * n)(*[mirrors.get(p, None) for p, _, _ in entries])
            numel = (ctypes.c_int64 * n)(*[p.numel() for p, _, _ in entries])
            offs = (ctypes.c_int64 * n)(*[off for _, off, _ in entries])
            grp = (ctypes.c_int32 * n)(*[gi for _, _, gi in entries])
            handle = ctypes.c_void_p()
            _hip.check(L.arreau_optimizer_create(n, ptrs, mirr, numel, offs, grp, len(self.param_groups), flat.numel(), ctypes.byref(handle)),
                       "arreau_optimizer_create")
            self._table = (key, handle)
        if self._m_flat is None or self._m_flat.numel() != flat.numel() or self._m_flat.device != flat.device:
            self._m_flat, self._v_flat = torch.zeros_like(flat), torch.zeros_like(flat)
            # (moments that exist already -- load_state_dict, an earlier step() -- move into the flat buffers below)
        steps = []
        for p, off, _ in entries:
            st = self.state[p]
            m_view = self._m_flat[off:off + p.numel()].view_as(p)
            if "exp_avg" not in st or st["exp_avg"].data_ptr() != m_view.data_ptr():
                v_view = self._v_flat[off:off + p.numel()].view_as(p)
                if "exp_avg" in st:
                    m_view.copy_(st["exp_avg"])
                    v_view.copy_(st["exp_avg_sq"])
                st["exp_avg"], st["exp_avg_sq"] = m_view, v_view
                if "step" not in st:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
            steps.append(st["step"])
        if any(s.is_cuda for s in steps):
            return None   # (a state_dict of torch's fused / capturable Adam: step counts on the device; torch's path handles it)
        first = float(steps[0])
        if any(float(s) != first for s in steps[1:]):
            return None   # (per-parameter step counts that differ: a state_dict stitched together by hand)
        torch._foreach_add_(steps, 1.0)
        args = _hip.AdamArgs()
        args.step = int(first) + 1
        for gi, g in enumerate(self.param_groups):
            lr = g["lr"]
            args.lr[gi] = float(lr)
            args.weight_decay[gi] = float(g["weight_decay"])
        (b1, b2), = betas
        args.beta1, args.beta2, args.eps = float(b1), float(b2), float(next(iter(epss)))
        args.max_norm = float(max_norm) if max_norm else 0.0
        norm = torch.empty((), device=flat.device, dtype=torch.float32)
        _hip.check(L.arreau_optimizer_step(self._table[1], _hip.ptr(flat), _hip.ptr(self._m_flat), _hip.ptr(self._v_flat), ctypes.byref(args),
                                           _hip.ptr(norm), _hip.stream_ptr(flat.device)), "arreau_optimizer_step")
        return norm

"""Lightning-format checkpoint I/O without pytorch_lightning, plus a synthetic-checkpoint writer.

The reference's checkpoints are Lightning 2.2.1 dicts: `state_dict`,
`hyper_parameters` = {args: argparse.Namespace, z_table: AtomicNumberTable} (pickled by
save_hyperparameters, lightning_wrappers/diffusion.py:34) and trainer bookkeeping.  The pickled
class path is `diffusion.tools.atomic_number_table.AtomicNumberTable`; it is remapped onto this
package's class when loading and restored when saving, so files round-trip with the reference.
"""
import argparse
import contextlib
import pickle
import sys
import types

import torch

from .diffusion.tools import atomic_number_table as _ant

_REF_ANT_MODULE = "diffusion.tools.atomic_number_table"


class _Stub:
    """Placeholder for classes of packages that are not installed (trainer/callback state)."""

    def __init__(self, *a, **kw):
        pass

    def __setstate__(self, state):
        self.__dict__["state"] = state


class _RemapUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == _REF_ANT_MODULE or module.endswith(".atomic_number_table"):
            return getattr(_ant, name)
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return _Stub


_pickle_module = types.ModuleType("arreau_amd._ckpt_pickle")
_pickle_module.__dict__.update({k: getattr(pickle, k) for k in dir(pickle) if not k.startswith("__")})
_pickle_module.Unpickler = _RemapUnpickler
_pickle_module.load = lambda f, **kw: _RemapUnpickler(f, **kw).load()


def load_lightning_checkpoint(path):
    """Read a Lightning checkpoint dict on the CPU (weights stay in their stored dtype)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_pickle_module)
    if "state_dict" not in ckpt or "hyper_parameters" not in ckpt:
        raise ValueError(f"{path} is not a Lightning checkpoint (needs state_dict and hyper_parameters)")
    hp = ckpt["hyper_parameters"]
    if not isinstance(hp, dict):
        hp = dict(getattr(hp, "__dict__", hp))
        ckpt["hyper_parameters"] = hp
    return ckpt


@contextlib.contextmanager
def _reference_class_paths():
    """Pickle AtomicNumberTable under the reference's module path."""
    saved_mods = {}
    chain = ["diffusion", "diffusion.tools", _REF_ANT_MODULE]
    for name in chain:
        saved_mods[name] = sys.modules.get(name)
        if name == _REF_ANT_MODULE:
            sys.modules[name] = _ant
        elif name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    old = _ant.AtomicNumberTable.__module__
    _ant.AtomicNumberTable.__module__ = _REF_ANT_MODULE
    try:
        yield
    finally:
        _ant.AtomicNumberTable.__module__ = old
        for name, mod in saved_mods.items():
            if mod is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = mod


def save_lightning_checkpoint(path, module, include_ori_grid=True):
    """Write `module` (a PONITA_DIFFUSION) as a Lightning-format checkpoint."""
    from .lightning_wrappers.diffusion import ORI_GRID_KEY
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    if include_ori_grid:
        sd[ORI_GRID_KEY] = module.model.ori_grid.detach().cpu().clone()
    ckpt = {
        "epoch": 0, "global_step": 0, "pytorch-lightning_version": "2.2.1", "state_dict": sd,
        "loops": {}, "callbacks": {}, "optimizer_states": [], "lr_schedulers": [],
        "hparams_name": "kwargs",
        "hyper_parameters": {"args": module.hparams.args, "z_table": module.hparams.z_table},
    }
    with _reference_class_paths():
        torch.save(ckpt, path)
    return path


def default_args(**overrides):
    """argparse.Namespace with the defaults of the reference's training CLI (main_diffusion.py:34-148)
    and the Makefile preset for the sampler-relevant flags (radius 5, max_neighbors 8, T 1000)."""
    d = dict(epochs=10000, warmup=10, batch_size=100, lr=1e-3, weight_decay=1e-10, log=True,
             enable_progress_bar=False, num_workers=0, seed=0, val_interval=5, train_augm=False,
             dataset="alexandria", radius=5, loop=True, num_ori=16, hidden_dim=128, basis_dim=256, degree=3,
             layers=5, widening_factor=4, layer_scale=1e-6, multiple_readouts=True, num_timesteps=1000,
             max_neighbors=8, experiment_name=None, profiler=False, gpus=1)
    d.update(overrides)
    return argparse.Namespace(**d)


def make_synthetic_model(S=90, seed=1234, trained_like=True, ori_grid=None, pooled_readout_scale=None, **arg_overrides):
    """Random-weight PONITA_DIFFUSION in the reference's layout (the real checkpoint is a download
    that is unavailable offline).  Default PyTorch initialisers under torch.manual_seed(seed), like
    the reference constructors.  `trained_like=True` additionally randomises the tensors whose
    initial values hide the network from a parity test (layer_scale = 1e-6 scales every ConvNext
    branch to nothing; zero conv bias; unit LayerNorm): layer_scale ~ U(0.1, 1), conv.bias ~ N(0, 0.1),
    norm.weight ~ U(0.5, 1.5), norm.bias ~ N(0, 0.1).  `pooled_readout_scale` multiplies the read-out rows (and biases) of
    the three per-crystal scalars (split order [S, 1, 0, 3], ponita.py:108-117): random rows make the pooled prediction
    `pred_lengths_0` a sum of ~n values of order 3 (|len0| ~ 58 at 20 atoms), whereas a trained model predicts lengths / n =
    O(1) (diffusion_loss.py:264-267); e.g. 1/32 gives a model whose three outputs are all of order one."""
    from .lightning_wrappers.diffusion import PONITA_DIFFUSION
    state = torch.random.get_rng_state()
    try:
        torch.manual_seed(seed)
        zs = list(range(1, S)) + [_ant.AtomicNumberTable.MASK_ATOMIC_NUMBER]
        model = PONITA_DIFFUSION(default_args(**arg_overrides), _ant.AtomicNumberTable(zs), ori_grid=ori_grid)
        if trained_like:
            with torch.no_grad():
                for layer in model.model.interaction_layers:
                    if layer.layer_scale is not None:
                        layer.layer_scale.uniform_(0.1, 1.0)
                    layer.conv.bias.normal_(0.0, 0.1)
                    layer.norm.weight.uniform_(0.5, 1.5)
                    layer.norm.bias.normal_(0.0, 0.1)
        if pooled_readout_scale is not None:
            with torch.no_grad():
                for ro in model.model.read_out_layers:
                    ro.weight[S + 1:S + 4] *= float(pooled_readout_scale)
                    ro.bias[S + 1:S + 4] *= float(pooled_readout_scale)
    finally:
        torch.random.set_rng_state(state)
    return model

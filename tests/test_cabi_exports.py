"""The C-ABI library builds, loads and exports every symbol include/arreau_hip.h declares.
No compute calls (runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from arreau_amd import build
    return build.build(verbose=False)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "arreau_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(arreau_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    syms = declared_symbols()
    for must in ("arreau_model_create", "arreau_predict_scores", "arreau_reverse_step", "arreau_radius_graph_pbc"):
        assert must in syms


def test_library_exports_every_declared_symbol(built_lib):
    L = ctypes.CDLL(built_lib)
    for s in declared_symbols():
        assert hasattr(L, s), f"{s} declared in include/arreau_hip.h but not exported"


def test_binding_lists_every_declared_symbol(built_lib):
    from arreau_amd import _hip
    assert sorted(_hip.EXPORTS) == declared_symbols()
    assert _hip.lib().arreau_version().decode().startswith("arreau_hip")


def test_workspace_size_is_monotone(built_lib):
    from arreau_amd import _hip
    cfg = _hip.Config(num_atomic_states=90, hidden_dim=128, basis_dim=256, num_layers=5, num_ori=16,
                      widening_factor=4, degree=3, max_neighbors=8, num_timesteps=1000, radius=5.0, has_layer_scale=1)
    a = _hip.lib().arreau_workspace_bytes(ctypes.byref(cfg), 160, 8)
    b = _hip.lib().arreau_workspace_bytes(ctypes.byref(cfg), 5120, 256)
    assert 0 < a < b
    # dominated by the per-layer edge kernels: L * N * k * O * C floats
    assert b >= 5 * 5120 * 8 * 16 * 128 * 4


def test_product_path_fails_loudly_without_gpu():
    import torch
    from arreau_amd import _hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from arreau_amd.diffusion.lattice_helpers import lattice_from_params
    with pytest.raises(_hip.ArreauHipError):
        lattice_from_params(torch.ones(1, 3), torch.ones(1, 3))

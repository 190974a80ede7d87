"""Shapes the fused kernels are not instantiated for -- the reference's `make train` preset is hidden_dim = 200
(/root/reference Makefile:7) -- run on the library's shape-general fp32 kernels (csrc/train_net.hip: exact fp32 MFMA
GEMMs + element-wise kernels): score network, inner seam, sampling loop and the training step, against the oracle.
The same path is selectable for the fused shape (edge variant 5) as one more arithmetic cross-check.
Needs an MI355X: run with `-m gpu`."""
import copy
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import geometry as OG
from oracle import ponita as OP
from oracle import sampler as OS
from oracle import training as TR
from tests.helpers import oracle_from_module, random_state, slots_from_edges
from tests.test_gpu_parity import _oracle_scores, _to_dev, assert_scores_close

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda", 0)


@pytest.fixture(scope="module", params=[dict(hidden_dim=200), dict(hidden_dim=64, basis_dim=96, widening_factor=2, layers=3)],
                ids=["make-train-preset-C200", "C64-D96-W2-L3"])
def odd_model(dev, request):
    from arreau_amd.checkpoint import make_synthetic_model
    m = make_synthetic_model(S=12, seed=77, num_timesteps=100, **request.param).to(dev)
    return m, oracle_from_module(m, torch.float32)


def test_general_shape_scores_match_oracle(dev, odd_model):
    m, om = odd_model
    eng = m.engine()
    for num_atoms, sampler_like, seed, t in (([4, 1, 6, 3], False, 1, 99), ([20] * 3, True, 4, 1), ([7, 9], False, 2, 50)):
        state = random_state(12, num_atoms, seed, sampler_like=sampler_like)
        eps_o, logits_o, len0_o, (ei, dists, direction, _c, _l) = _oracle_scores(om, *state, t)
        N, B = state[0].shape[0], len(num_atoms)
        deg, src, sdir, sdist = slots_from_edges(ei, dists, direction, N, 8)
        f, ty, le, an, off = _to_dev(dev, *state)
        t_c = torch.full((B,), t, device=dev, dtype=torch.int32)
        edges = tuple(x.to(dev).contiguous() for x in (deg, src, sdir, sdist))
        got = eng.predict_scores(f, ty, le, an, t_c, off, edges=edges)
        assert_scores_close(got, (eps_o, logits_o, len0_o), tag=("given edges", num_atoms))
        if not sampler_like:  # (tiny sampler-start cells hold exactly tied images: the selected subset may differ)
            got = eng.predict_scores(f, ty, le, an, t_c, off)  # the library's own neighbour list
            assert_scores_close(got, (eps_o, logits_o, len0_o), tag=("own edges", num_atoms))
    st = eng.status()
    assert st["edge_kernel"] == "general-fp32-gemm" and st["mlp_kernel"] == "general-fp32-gemm" and st["flags"] == 0
    with pytest.raises(RuntimeError, match="no fused kernels"):
        eng.set_variant(edge=4)


def test_general_shape_inner_seam(dev, odd_model):
    """arreau_ponita_forward on caller-assembled features (soft types, per-atom time features) for a non-fused shape."""
    m, om = odd_model
    S = 12
    state = random_state(S, [5, 3, 8], 9)
    frac, types, lengths, angles, na = state
    _, _, _, (ei, dists, direction, cart, lattice) = _oracle_scores(om, *state, 40)
    N, B = frac.shape[0], len(na)
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.softmax(torch.randn(N, S, generator=g), -1), torch.randn(N, 74, generator=g) * 0.5], 1)
    vec = torch.cat([frac[:, None, :], lattice.repeat_interleave(na, 0)], 1)
    batch = torch.arange(B).repeat_interleave(na)
    logits_o, vec_o, gs_o = OP.ponita_forward(om.sd, om.hp, x, vec, ei, dists, direction, lattice, batch, batch[ei[0]],
                                              om.ori_grid)
    deg, src, sdir, sdist = slots_from_edges(ei, dists, direction, N, 8)
    from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
    eng = m.engine()
    logits, vec_out, gs = eng.ponita_forward(x.to(dev), vec.to(dev).contiguous(), lattice.to(dev).contiguous(),
                                             crystal_offsets(na, dev),
                                             tuple(v.to(dev).contiguous() for v in (deg, src, sdir, sdist)))
    assert (logits.cpu() - logits_o).abs().max() <= TOL * max(1.0, float(logits_o.abs().max()) / 8)
    assert (vec_out.cpu().reshape(N, 3) - vec_o.reshape(N, 3)).abs().max() <= TOL * max(1.0, float(vec_o.abs().max()))
    assert (gs.cpu() - gs_o).abs().max() <= TOL * max(1.0, float(gs_o.abs().max()))


def test_general_shape_sampler_runs_and_replays(dev, odd_model):
    """The whole sampler (arreau_sample_loop) on the general kernels: eager loop and hipGraph replay give the same
    trajectory bit for bit; the first step equals the oracle's reverse step on the oracle's scores."""
    m, _ = odd_model
    from arreau_amd.diffusion.inference.visualize_crystal import VisualizationSetting
    torch.manual_seed(11); np.random.seed(11)
    a = m.sample(6, 5, VisualizationSetting.NONE, False, use_graph=False, seed=123)
    torch.manual_seed(11); np.random.seed(11)
    b = m.sample(6, 5, VisualizationSetting.NONE, False, use_graph=True, seed=123)
    assert np.isfinite(a.frac_x).all() and np.isfinite(a.lattice).all()
    assert np.array_equal(a.frac_x, b.frac_x) and np.array_equal(a.atomic_numbers, b.atomic_numbers)
    assert np.array_equal(a.lattice, b.lattice)
    assert m.engine().status()["flags"] == 0


def test_general_shape_training_step_matches_oracle_autograd(dev, odd_model):
    m, om = odd_model
    rng = np.random.RandomState(8)
    num_atoms = [3, 5, 2, 6]
    B, N, S = len(num_atoms), sum(num_atoms), 12
    lengths = torch.tensor(rng.uniform(3.5, 7.0, size=(B, 3)), dtype=torch.float32)
    angles = torch.tensor(np.deg2rad(rng.uniform(75, 105, size=(B, 3))), dtype=torch.float32)
    lattice0 = OG.lattice_from_params(lengths, angles)
    batch = SimpleNamespace(X0=torch.tensor(rng.uniform(0, 1, size=(N, 3)), dtype=torch.float32),
                            A0=torch.tensor(rng.randint(0, S - 1, size=N)), L0=lattice0.reshape(-1, 3),
                            num_atoms=torch.tensor(num_atoms))
    timestep = torch.tensor([1, 50, 100, 77])
    g = torch.Generator().manual_seed(4)
    noise = (torch.randn(N, 3, generator=g), torch.rand(N, S, generator=g), torch.randn(B, 3, generator=g))
    mm = copy.deepcopy(m)
    for layer in mm.model.interaction_layers:
        layer.conv.callibrated.fill_(True)
    loss = mm.training_step(batch, timestep=timestep, noise=noise)
    for v in om.sd.values():
        if v.is_floating_point() and v.numel() > 0:
            v.requires_grad_(True)
            v.grad = None
    loss_o = TR.diffusion_loss(om, batch.X0, batch.A0, lattice0, batch.num_atoms, timestep, *noise)
    loss_o.backward()
    want = {"model." + k: v.grad.clone() for k, v in om.sd.items() if v.requires_grad and v.grad is not None}
    for v in om.sd.values():
        v.requires_grad_(False)
    assert abs(float(loss.detach()) - float(loss_o.detach())) <= TOL * max(1.0, abs(float(loss_o)))
    got = {n: p.grad for n, p in mm.named_parameters() if p.grad is not None}
    checked = 0
    for name, w in want.items():
        if w.numel() == 0:
            continue
        err = float((got[name].cpu() - w).abs().max())
        scale = max(float(w.abs().max()), 1e-7)
        assert err <= 1e-4 * scale + 1e-7, (name, err, scale)
        checked += 1
    assert checked >= 9 + 10 * len(mm.model.interaction_layers)
    # after an optimiser step the SAME engine keeps sampling (plain weights are refreshed on the device; nothing is packed)
    opt_cfg = mm.configure_optimizers()
    opt_cfg["optimizer"].step()
    mm.notify_parameters_changed()
    eng = mm.engine()
    assert not eng.stale_for_sampling  # nothing packed to go stale: the general kernels read the refreshed plain weights
    state = random_state(12, [4, 6], 1)
    f, ty, le, an, off = _to_dev(dev, *state)
    t_c = torch.full((2,), 30, device=dev, dtype=torch.int32)
    got = eng.predict_scores(f, ty, le, an, t_c, off)
    om2 = oracle_from_module(mm, torch.float32)
    eps_o, logits_o, len0_o, _ = _oracle_scores(om2, *state, 30)
    assert_scores_close(got, (eps_o, logits_o, len0_o), tag="after optimiser step")


def test_general_path_agrees_with_fused_kernels_on_the_fused_shape(dev):
    """Edge variant 5 on the shipped shape (C=128, D=256): the fp32 GEMM network and the fused fp16x3 kernels agree to
    the parity bound, and both with the oracle."""
    from arreau_amd.checkpoint import make_synthetic_model
    from arreau_amd import _hip
    m = make_synthetic_model(S=12, seed=1234, num_timesteps=100).to(dev)
    om = oracle_from_module(m, torch.float32)
    state = random_state(12, [20] * 4, 2)
    eps_o, logits_o, len0_o, _ = _oracle_scores(om, *state, 2)
    f, ty, le, an, off = _to_dev(dev, *state)
    t_c = torch.full((4,), 2, device=dev, dtype=torch.int32)
    eng = m.engine()
    fused = eng.predict_scores(f, ty, le, an, t_c, off)
    eng.set_variant(edge=_hip.VARIANT_GENERAL)
    try:
        general = eng.predict_scores(f, ty, le, an, t_c, off)
        assert eng.status()["edge_kernel"] == "general-fp32-gemm"
    finally:
        eng.set_variant(edge=4)
    assert_scores_close(general, (eps_o, logits_o, len0_o), tag="general vs oracle")
    assert_scores_close(fused, tuple(x.cpu() for x in general), tag="fused vs general")

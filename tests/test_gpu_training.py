"""BASELINE config 5, forward part: DiffusionLoss.__call__ (forward noising + score network + three errors) on the HIP
path against the oracle (oracle/training.py, pinned on the reference's noising / D3PM functions by
tests/golden/training.npz), with every random draw injected.  Needs an MI355X: run with `-m gpu`."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import training as TR
from tests.helpers import oracle_from_module

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def setup():
    from arreau_amd.checkpoint import make_synthetic_model
    from oracle import geometry as OG
    dev = torch.device("cuda", 0)
    m = make_synthetic_model(S=12, seed=1234, num_timesteps=100).to(dev)
    om = oracle_from_module(m, torch.float32)
    rng = np.random.RandomState(8)
    num_atoms = [3, 5, 2, 1, 6]
    B, N, S = len(num_atoms), sum(num_atoms), 12
    lengths = torch.tensor(rng.uniform(3.5, 7.0, size=(B, 3)), dtype=torch.float32)
    angles = torch.tensor(np.deg2rad(rng.uniform(75, 105, size=(B, 3))), dtype=torch.float32)
    lattice0 = OG.lattice_from_params(lengths, angles)
    frac0 = torch.tensor(rng.uniform(0, 1, size=(N, 3)), dtype=torch.float32)
    types0 = torch.tensor(rng.randint(0, S - 1, size=N))
    timestep = torch.tensor([1, 50, 100, 2, 77])
    g = torch.Generator().manual_seed(4)
    noise = (torch.randn(N, 3, generator=g), torch.rand(N, S, generator=g), torch.randn(B, 3, generator=g))
    batch = SimpleNamespace(X0=frac0, A0=types0, L0=lattice0.reshape(-1, 3), num_atoms=torch.tensor(num_atoms))
    return m, om, batch, lattice0, timestep, noise


def test_forward_noising_matches_oracle(setup):
    m, om, batch, lattice0, timestep, noise = setup
    _, parts = m.diffusion_loss(m, batch, None, timestep=timestep, noise=noise, return_parts=True)
    want = TR.noise_inputs(om, batch.X0, batch.A0, lattice0, batch.num_atoms, timestep, *noise)
    np.testing.assert_allclose(parts["noisy_frac"].cpu().numpy(), want["noisy_frac"].numpy(), atol=1e-6, rtol=0)
    d = (parts["target_eps"].cpu() - want["target_eps"]).abs()
    assert torch.minimum(d, 1 - d).max() <= TOL  # both are "% 1": a value at the seam may land on either side
    assert torch.equal(parts["noisy_types"].cpu().long(), want["noisy_types"])
    np.testing.assert_allclose(parts["noisy_lengths"].cpu().numpy(), want["noisy_lengths"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(parts["lengths"].cpu().numpy(), want["lengths"].numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(parts["angles"].cpu().numpy(), want["angles"].numpy(), atol=2e-6, rtol=0)


def test_training_loss_and_output_gradients_match_oracle(setup):
    """loss, its three parts, and d(loss)/d(pred_eps, logits, pred_lengths) -- the latter against torch autograd through
    the oracle's loss functions (what the reference's backward pass hands to the network)."""
    m, om, batch, lattice0, timestep, noise = setup
    loss, parts = m.diffusion_loss(m, batch, None, timestep=timestep, noise=noise, return_parts=True)
    loss_o, po = TR.diffusion_loss(om, batch.X0, batch.A0, lattice0, batch.num_atoms, timestep, *noise,
                                   return_parts=True)
    # network outputs on the noised batch
    assert (parts["pred_eps"].cpu() - po["pred_eps"]).abs().max() <= TOL * max(1.0, float(po["pred_eps"].abs().max()))
    assert (parts["logits"].cpu() - po["logits"]).abs().max() <= TOL * max(1.0, float(po["logits"].abs().max()))
    for k in ("error_frac_x", "error_atomic_type", "error_lattice", "vb", "ce"):
        assert abs(float(parts[k]) - float(po[k])) <= TOL * max(1.0, abs(float(po[k]))), k
    assert abs(float(loss) - float(loss_o)) <= TOL * max(1.0, abs(float(loss_o)))
    # gradient seeds: autograd through the oracle's loss at the HIP path's own network outputs
    pe = parts["pred_eps"].cpu().clone().requires_grad_(True)
    lg = parts["logits"].cpu().clone().requires_grad_(True)
    pl = parts["pred_lengths"].cpu().clone().requires_grad_(True)
    t_feat = timestep.repeat_interleave(batch.num_atoms)
    l2 = (TR.compute_frac_x_error(pe, po["target_eps"]) +
          TR.d3pm_calculate_loss(om.q_one_step_transposed, om.q_mats, batch.A0, lg, po["noisy_types"], t_feat)[0] +
          F.mse_loss(pl, po["lengths"] / batch.num_atoms.unsqueeze(-1)))
    l2.backward()
    for name, got, want in (("eps", parts["grad_eps"], pe.grad), ("logits", parts["grad_logits"], lg.grad),
                            ("lengths", parts["grad_lengths"], pl.grad)):
        scale = max(float(want.abs().max()), 1e-6)
        assert (got.cpu() - want).abs().max() <= 1e-4 * scale, name


def test_training_loss_draws_its_own_noise_in_reference_order(setup):
    """Without injected noise the draws come from the global CPU generator in the reference's order
    (randint [B,1], randn [N,3], rand [N,S], randn [B,3]): reproducing them by hand gives the same loss."""
    m, om, batch, lattice0, _, _ = setup
    B, N, S = 5, 17, 12
    torch.manual_seed(77)
    loss = m.diffusion_loss(m, batch, None)
    torch.manual_seed(77)
    t = torch.randint(1, 101, size=(B, 1)).long().reshape(B)
    noise = (torch.randn(N, 3), torch.rand(N, S), torch.randn(B, 3))
    loss2 = m.diffusion_loss(m, batch, None, timestep=t, noise=noise)
    assert float(loss) == float(loss2)
    with pytest.raises(ValueError):
        m.diffusion_loss(m, batch, None, timestep=0)

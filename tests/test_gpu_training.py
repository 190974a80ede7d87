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
# every gradient tensor against oracle autograd, relative to the tensor's largest entry (round 3 allowed 2e-3; the products are
# exact-fp32 or split-precision MFMA GEMMs and the scatter is an ordered sum: rounding level is ~1e-6)
GRAD_TOL = 1e-4


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


def test_absorbing_chain_shortcut_in_the_loss_is_bitwise_the_dense_product(setup):
    """The D3PM loss terms and gradients with the mask-chain structure of Qbar exploited (diagonal + mask column: the
    default for the reference's forward_type = "mask" buffers) against the dense S x S products (ARREAU_D3PM_DENSE at model
    creation): the shortcut adds the same numbers in the same order, so every output is bit-identical."""
    import copy
    import os
    m, om, batch, lattice0, timestep, noise = setup
    _, a = m.diffusion_loss(m, batch, None, timestep=timestep, noise=noise, return_parts=True)
    os.environ["ARREAU_D3PM_DENSE"] = "1"
    try:
        m2 = copy.deepcopy(m)
        _, b = m2.diffusion_loss(m2, batch, None, timestep=timestep, noise=noise, return_parts=True)
    finally:
        del os.environ["ARREAU_D3PM_DENSE"]
    for k in ("error_frac_x", "error_atomic_type", "error_lattice", "vb", "ce", "grad_eps", "grad_logits", "grad_lengths"):
        assert torch.equal(a[k], b[k]), k
    assert float(a["grad_logits"].abs().max()) > 0


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


# ------------------------------------------------------------------------------------------- forward + backward
def _oracle_grads(om, batch, lattice0, timestep, noise):
    """Reference gradients: torch autograd through the oracle's loss (plain torch, float32)."""
    for v in om.sd.values():
        if v.is_floating_point() and v.numel() > 0:
            v.requires_grad_(True)
            v.grad = None
    loss = TR.diffusion_loss(om, batch.X0, batch.A0, lattice0, batch.num_atoms, timestep, *noise)
    loss.backward()
    grads = {"model." + k: v.grad.clone() for k, v in om.sd.items() if v.requires_grad and v.grad is not None}
    for v in om.sd.values():
        v.requires_grad_(False)
    return float(loss), grads


def test_training_forward_matches_sampling_kernels(setup):
    """arreau_train_forward (fp32 GEMM form, activations kept) and arreau_predict_scores (fused fp16x3 kernels) evaluate
    the same network: outputs agree to the parity bound."""
    from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
    m, om, batch, lattice0, timestep, noise = setup
    _, parts = m.diffusion_loss(m, batch, None, timestep=timestep, noise=noise, return_parts=True)
    eng = m.engine()
    off = crystal_offsets(batch.num_atoms, eng.device)
    a = eng.predict_scores(parts["noisy_frac"], parts["noisy_types"], parts["noisy_lengths"], parts["angles"],
                           parts["timestep"], off)
    b = eng.train_forward(parts["noisy_frac"], parts["noisy_types"], parts["noisy_lengths"], parts["angles"],
                          parts["timestep"], off)
    for name, x, y in zip(("eps", "logits", "len0"), a, b):
        assert (x - y).abs().max() <= TOL * max(1.0, float(x.abs().max())), name


def test_training_step_gradients_match_oracle_autograd(setup):
    """One training step (forward + backward in the library) against autograd through the oracle: the loss and the
    gradient of every trainable tensor of the score network."""
    import copy
    m, om, batch, lattice0, timestep, noise = setup
    mm = copy.deepcopy(m)  # training_step callibrates the conv weights afterwards; keep the fixture's model intact
    for layer in mm.model.interaction_layers:
        layer.conv.callibrated.fill_(True)
    loss = mm.training_step(batch, timestep=timestep, noise=noise)
    loss_o, want = _oracle_grads(om, batch, lattice0, timestep, noise)
    assert abs(float(loss) - loss_o) <= TOL * max(1.0, abs(loss_o))
    got = {n: p.grad for n, p in mm.named_parameters() if p.grad is not None}
    checked = 0
    for name, w in want.items():
        if w.numel() == 0:
            continue
        assert name in got, name
        g = got[name].cpu()
        scale = max(float(w.abs().max()), 1e-7)
        err = float((g - w).abs().max())
        assert err <= GRAD_TOL * scale + 1e-7, (name, err, scale)
        checked += 1
    assert checked >= 8 + 5 * 10 + 1  # basis (4) + fiber (4) + embed + per layer 10 (incl. layer_scale) + read-outs
    assert "t_emb.gaussian_fourier_proj_w" not in got  # requires_grad = False in the reference too


def test_fused_convnext_training_forward_against_the_product_form(setup, monkeypatch):
    """The training forward runs the ConvNext block through the sampling step's kernel (node_f16m.hip, TRAIN: one launch instead of a
    LayerNorm launch and two products; ARREAU_TRAIN_FUSED_MLP=0 keeps those) and that kernel also writes what the backward pass
    reads.  Both forms against each other: loss and every gradient -- the saved activations are only visible through them -- and
    again after an optimizer step, when the kernel's weight planes have been rebuilt on the device from the updated fp32 weights."""
    import copy
    from arreau_amd.train import optimizer_step
    m, om, batch, lattice0, timestep, noise = setup
    results = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("ARREAU_TRAIN_FUSED_MLP", fused)
        mm = copy.deepcopy(m)
        opt = mm.configure_optimizers(max_epochs=10)["optimizer"]
        for g in opt.param_groups:
            g["lr"] = 1e-3
        out = []
        for _ in range(2):
            loss = mm.training_step(batch, timestep=timestep, noise=noise)
            out.append((float(loss), {n: p.grad.detach().clone() for n, p in mm.named_parameters() if p.grad is not None}))
            optimizer_step(mm, opt, world_size=1)
        results[fused] = out
    for (la, ga), (lb, gb) in zip(results["1"], results["0"]):
        assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb)), (la, lb)
        assert ga.keys() == gb.keys() and len(ga) > 60
        for n in ga:
            scale = max(float(gb[n].abs().max()), 1e-30)
            assert float((ga[n] - gb[n]).abs().max()) <= 2e-5 * scale, (n, float((ga[n] - gb[n]).abs().max()) / scale)


def test_merged_launches_of_the_training_step_are_bitwise_the_one_kernel_per_operation_sequence(setup, monkeypatch):
    """Round 5: the training step merges launches -- spatial conv + spherical mix forward as one kernel, LayerNorm backward + mix
    backward as one, both gradients of the spatial conv in one launch, and the chunk sums of every column-sum pass / the k-slice
    sums of every weight-gradient product deferred to ONE launch each at the end of the backward pass.  None of that changes a sum
    or its order: against ARREAU_TRAIN_FUSE=0 ARREAU_TRAIN_SIDE_STREAM=0 (one kernel per operation, one stream) the loss and every gradient
    must be the same bits, at
    the 17-atom fixture and at the benchmark's 64 crystals, over two steps with an optimizer step between them."""
    import copy
    from arreau_amd.train import optimizer_step
    m, om, batch, lattice0, timestep, noise = setup
    big = _alexandria_like_batch(64, 12, 21)
    cases = [(batch, timestep, noise), (big[0], big[2], big[3])]
    results = {}
    for fuse in ("1", "0"):
        # "0": one kernel per operation AND everything on the caller's stream (the side stream is chosen when a model's training context
        # is created, i.e. at the first training_step of each deep copy below)
        monkeypatch.setenv("ARREAU_TRAIN_FUSE", fuse)
        monkeypatch.setenv("ARREAU_TRAIN_SIDE_STREAM", fuse)
        out = []
        for b, ts, nz in cases:
            mm = copy.deepcopy(m)
            for layer in mm.model.interaction_layers:
                layer.conv.callibrated.fill_(True)
            opt = mm.configure_optimizers(max_epochs=10)["optimizer"]
            for _ in range(2):
                loss = mm.training_step(b, timestep=ts, noise=nz)
                out.append((float(loss), {n: p.grad.detach().clone() for n, p in mm.named_parameters() if p.grad is not None}))
                optimizer_step(mm, opt, world_size=1)
        results[fuse] = out
    assert len(results["1"]) == 4
    for (la, ga), (lb, gb) in zip(results["1"], results["0"]):
        assert la == lb, (la, lb)
        assert ga.keys() == gb.keys() and len(ga) > 60
        for n in ga:
            assert torch.equal(ga[n], gb[n]), ("merged launches changed a gradient", n, float((ga[n] - gb[n]).abs().max()))


def test_first_training_forward_callibrates_conv_weights(setup):
    """FiberBundleConv.callibrate (conv.py:121-123,140-146): after the first training forward kernel.weight is scaled by
    std(x) / std(x_1) and fiber_kernel.weight by std(x_1) / std(x_2) (per layer, unbiased std), once."""
    import copy
    from oracle import ponita as OP
    from oracle import sampler as OS
    m, om, batch, lattice0, timestep, noise = setup
    mm = copy.deepcopy(m)
    w0 = [(l.conv.kernel.weight.detach().clone(), l.conv.fiber_kernel.weight.detach().clone()) for l in mm.model.interaction_layers]
    assert not any(bool(l.conv.callibrated) for l in mm.model.interaction_layers)
    mm.training_step(batch, timestep=timestep, noise=noise)
    assert all(bool(l.conv.callibrated) for l in mm.model.interaction_layers)
    # expected ratios from the oracle's internals on the same noised batch
    nz = TR.noise_inputs(om, batch.X0, batch.A0, lattice0, batch.num_atoms, timestep, *noise)
    B = len(batch.num_atoms)
    bidx = torch.arange(B).repeat_interleave(batch.num_atoms)
    x, cart, vec, lattice = OS.assemble_features(om, nz["noisy_frac"], F.one_hot(nz["noisy_types"], 12), nz["t_feat"],
                                                 batch.num_atoms, nz["noisy_lengths"], nz["angles"])
    from oracle import geometry as OG
    ei, _c, _n, dists, direction = OG.radius_graph_pbc(cart, lattice, batch.num_atoms, 5.0, 8)
    _, _, _, internals = OP.ponita_forward(om.sd, om.hp, x, vec, ei, dists, direction, lattice, bidx, bidx[ei[0]], om.ori_grid,
                                           return_internals=True)
    for l, layer in enumerate(mm.model.interaction_layers):
        s_in, s_1, s_2 = (float(v.std()) for v in internals["conv_stats"][l])
        np.testing.assert_allclose((layer.conv.kernel.weight / w0[l][0].to(layer.conv.kernel.weight.device)).mean().item(),
                                   s_in / s_1, rtol=2e-4)
        np.testing.assert_allclose((layer.conv.fiber_kernel.weight / w0[l][1].to(layer.conv.kernel.weight.device)).mean().item(),
                                   s_1 / s_2, rtol=2e-4)
    k1 = [l.conv.kernel.weight.detach().clone() for l in mm.model.interaction_layers]
    mm.training_step(batch, timestep=timestep, noise=noise)  # second step: no further rescale
    for l, layer in enumerate(mm.model.interaction_layers):
        assert torch.equal(layer.conv.kernel.weight, k1[l])


def test_optimizer_step_refreshes_training_weights_without_repacking(setup):
    """After an optimizer step the engine is refreshed in place for training (device-to-device copies,
    arreau_model_update_train_weights): the next training forward equals a freshly packed engine's, the stale engine
    refuses to sample, and asking for a sampling engine re-creates it."""
    import copy
    from arreau_amd import _hip
    from arreau_amd.diffusion.diffusion_helpers import crystal_offsets
    from arreau_amd.train import optimizer_step
    m, om, batch, lattice0, timestep, noise = setup
    mm = copy.deepcopy(m)
    opt = mm.configure_optimizers(max_epochs=10)["optimizer"]
    for g in opt.param_groups:
        g["lr"] = 1e-3
    mm.training_step(batch, timestep=timestep, noise=noise)
    w_before = mm.model.interaction_layers[0].linear_1.weight.detach().clone()
    norm = optimizer_step(mm, opt, world_size=1)
    assert float(norm) > 0 and not torch.equal(w_before, mm.model.interaction_layers[0].linear_1.weight)
    eng = mm._engine
    assert eng is not None and eng.stale_for_sampling
    _, parts = mm.diffusion_loss(mm, batch, None, timestep=timestep, noise=noise, return_parts=True, training=True)
    assert mm._engine is eng  # training keeps using the refreshed engine
    off = crystal_offsets(batch.num_atoms, eng.device)
    args = (parts["noisy_frac"], parts["noisy_types"], parts["noisy_lengths"], parts["angles"], parts["timestep"], off)
    with pytest.raises(_hip.ArreauHipError):
        eng.predict_scores(*args)
    fresh = copy.deepcopy(mm).engine()  # packs everything from the updated parameters
    a, b = eng.train_forward(*args), fresh.train_forward(*args)
    for x, y in zip(a, b):
        assert (x - y).abs().max() <= 1e-6 * max(1.0, float(x.abs().max()))
    assert mm.engine() is not eng and not mm.engine().stale_for_sampling  # a sampling engine is rebuilt on demand


def test_clip_adam_on_the_flat_gradient_buffer_is_torch_adam_after_clip_grad_norm():
    """arreau_optimizer_step (arreau_amd/optim.py: ClipAdam.step_flat) against what it replaces -- torch.nn.utils.clip_grad_norm_(0.5)
    (main_diffusion.py:297) followed by torch.optim.Adam over a decayed and an undecayed group (lightning_wrappers/diffusion.py:
    152-218) -- over six steps with a changing learning rate: parameters, both moments and the reported norm.  Tensor sizes straddle the
    kernel's 1,024-element chunks; one step has a non-finite gradient (no-op on the device: zero gradient into the update, as
    arreau_amd.train.optimizer_step does through torch); the state_dict round-trips through a plain torch.optim.Adam."""
    from arreau_amd.optim import ClipAdam
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    shapes = [(37, 29), (1024,), (5,), (3, 700), (1,)]
    decayed = [True, False, False, True, False]
    init = [torch.randn(s, generator=g) for s in shapes]
    def make(cls):
        ps = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
        groups = [{"params": [p for p, d in zip(ps, decayed) if d], "weight_decay": 1e-2},
                  {"params": [p for p, d in zip(ps, decayed) if not d], "weight_decay": 0.0}]
        return ps, cls(groups, lr=3e-3)
    ps_a, opt_a = make(ClipAdam)
    ps_b, opt_b = make(torch.optim.Adam)
    sizes = [-(-p.numel() // 4) * 4 for p in ps_a]  # 16-byte aligned views, as HipEngine.train_backward lays them out
    for step in range(6):
        flat = torch.zeros(sum(sizes) + 8, device=dev)
        off = 0
        for p, q, n in zip(ps_a, ps_b, sizes):
            grad = torch.randn(p.shape, generator=g) * (10.0 if step % 2 else 0.01)  # clipped and unclipped steps
            if step == 3:
                grad.view(-1)[0] = float("inf")
            p.grad = flat[off:off + p.numel()].view(p.shape)
            p.grad.copy_(grad)
            q.grad = grad.to(dev)
            off += n
        for o in (opt_a, opt_b):
            for grp in o.param_groups:
                grp["lr"] = 3e-3 * (1 + step)
        norm_a = opt_a.step_flat(flat, 0.5)
        assert norm_a is not None
        norm_b = torch.nn.utils.clip_grad_norm_(ps_b, 0.5)
        if not bool(torch.isfinite(norm_b)):
            for q in ps_b:
                q.grad.zero_()
        opt_b.step()
        assert (torch.isfinite(norm_b) and abs(float(norm_a) - float(norm_b)) <= 1e-6 * float(norm_b)) or (not torch.isfinite(norm_b) and not torch.isfinite(norm_a))
        for p, q in zip(ps_a, ps_b):
            assert float((p - q).abs().max()) <= 2e-6 * max(1.0, float(q.abs().max())), step
            sa, sb = opt_a.state[p], opt_b.state[q]
            assert float(sa["step"]) == float(sb["step"]) == step + 1
            for key in ("exp_avg", "exp_avg_sq"):
                assert float((sa[key] - sb[key]).abs().max()) <= 1e-6 * max(1e-12, float(sb[key].abs().max())), (step, key)
    # state_dict: same layout as torch's; loading it into a plain Adam and stepping both once more keeps them together
    ps_c, opt_c = make(torch.optim.Adam)
    with torch.no_grad():
        for c, a in zip(ps_c, ps_a):
            c.copy_(a)
    import copy
    opt_c.load_state_dict(copy.deepcopy(opt_a.state_dict()))  # (load_state_dict keeps same-device tensors as they are: without the copy the two optimizers would share moments)
    flat = torch.zeros(sum(sizes) + 8, device=dev)
    off = 0
    for p, c, n in zip(ps_a, ps_c, sizes):
        p.grad = flat[off:off + p.numel()].view(p.shape)
        p.grad.copy_(torch.randn(p.shape, generator=g) * 0.01)
        c.grad = p.grad.clone()
        off += n
    assert opt_a.step_flat(flat, None) is not None
    opt_c.step()
    for p, c in zip(ps_a, ps_c):
        assert float((p - c).abs().max()) <= 2e-6 * max(1.0, float(c.abs().max()))
    # gradients that are not views of one buffer: step_flat declines, step() is torch's
    for p in ps_a:
        p.grad = torch.zeros_like(p)
    assert opt_a.step_flat(flat, 0.5) is None
    opt_a.step()


def test_training_survives_weights_that_outgrow_the_fp16_forward(setup):
    """the training forward runs fp16x3 products chosen from the weights at engine creation, and the optimizer
    moves the weights afterwards.  Here they are moved far beyond the fp16 range of the hidden activations between two steps:
    the step comes out non-finite, the optimizer driver turns it into a no-op on the device (no NaN reaches the parameters or
    Adam's moments), the periodic status check switches the engine to the full-range bf16x6 products instead of raising, and the
    next step is finite again."""
    import copy
    import warnings
    from arreau_amd.train import optimizer_step
    m, om, batch, lattice0, timestep, noise = setup
    mm = copy.deepcopy(m)
    mm.STATUS_CHECK_EVERY = 1
    opt = mm.configure_optimizers(max_epochs=10)["optimizer"]
    loss0 = mm.training_step(batch, timestep=timestep, noise=noise)
    optimizer_step(mm, opt, world_size=1)
    assert bool(torch.isfinite(loss0))
    with torch.no_grad():  # what a diverging run might do to one layer: hidden units of order 1e7
        mm.model.interaction_layers[1].linear_1.weight *= 1.0e6
    mm.notify_parameters_changed()
    before = [p.detach().clone() for p in mm.parameters()]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        loss1 = mm.training_step(batch, timestep=timestep, noise=noise)   # overflows; its status check (every step here) switches the engine
        optimizer_step(mm, opt, world_size=1)
        loss2 = mm.training_step(batch, timestep=timestep, noise=noise)
        optimizer_step(mm, opt, world_size=1)
        loss3 = mm.training_step(batch, timestep=timestep, noise=noise)   # full-range products
    assert not bool(torch.isfinite(loss1))
    assert any("bf16x6" in str(x.message) for x in w), [str(x.message) for x in w]
    assert bool(torch.isfinite(loss3)), float(loss3)
    for p in mm.parameters():
        assert bool(torch.isfinite(p).all())
    # the poisoned step changed nothing (its gradient was zeroed on the device)
    after_first = [p.detach() for p in mm.parameters()]
    assert all(torch.isfinite(a).all() for a in after_first) and len(before) == len(after_first)
    assert mm._engine.status()["flags"] == 0


def test_two_rank_training_loop_reduces_the_loss(tmp_path):
    """arreau_amd.train end to end: two data-parallel ranks (gloo, sharing this box's GPU), synthetic Alexandria-like
    crystals, forward + backward in the library, one flat all-reduce per step, Adam with the cosine warm-up schedule;
    the ranks hold identical weights afterwards (parameter checksums of both replicas, printed by the driver) and the
    epoch-mean loss goes down."""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ARREAU_TRAIN_BACKEND="gloo", ARREAU_TRAIN_ONE_DEVICE="1", PYTHONPATH=root)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "trained.ckpt")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), "-m", "arreau_amd.train", "--num_synthetic", "99", "--epochs",
                        "6", "--warmup", "1", "--batch_size", "8", "--lr", "2e-3", "--out", out], env=env, cwd=root, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("epoch")]
    assert len(lines) == 6, p.stdout
    assert all(" 6 steps," in ln for ln in lines), lines  # 99 crystals -> 98 for two ranks -> 49 each -> 6 full batches of 8
    mean = lambda ln: float(ln.split("loss per crystal")[1].split("(")[0])  # epoch mean over all ranks
    first, last = mean(lines[0]), mean(lines[-1])
    assert np.isfinite(last) and os.path.exists(out)
    assert last < 0.9 * first, (first, last)  # (epoch 0 runs at the warm-up's 1e-6 factor: it is the untrained loss)
    sums = [ln for ln in p.stdout.splitlines() if ln.startswith("replica parameter checksums:")]
    assert len(sums) == 1, p.stdout
    a, b = (float(v) for v in sums[0].split(":")[1].split())
    assert a == b, sums  # the two replicas hold the same weights (incl. the first-step calibration ratios)
    from arreau_amd.checkpoint import load_lightning_checkpoint
    ck = load_lightning_checkpoint(out)
    assert bool(ck["state_dict"]["model.interaction_layers.0.conv.callibrated"])
    print("two-rank training: first-epoch loss", first, "last-epoch loss", last)


def _alexandria_like_batch(B, S, seed, T=100):
    """B crystals with Alexandria-PBE's size statistics (mean ~8 atoms, cap 64), every random draw of the loss injected."""
    from oracle import geometry as OG
    rng = np.random.RandomState(seed)
    num_atoms = np.clip(rng.geometric(1.0 / 8.0, size=B), 1, 64).tolist()
    N = int(sum(num_atoms))
    vol = np.array(num_atoms) / 0.055  # density of find_avg_density_of_dataset.py:40
    edge = vol ** (1.0 / 3.0)
    lengths = torch.tensor(edge[:, None] * rng.uniform(0.8, 1.25, size=(B, 3)), dtype=torch.float32)
    angles = torch.tensor(np.deg2rad(rng.uniform(75, 105, size=(B, 3))), dtype=torch.float32)
    lattice0 = OG.lattice_from_params(lengths, angles)
    batch = SimpleNamespace(X0=torch.tensor(rng.uniform(0, 1, size=(N, 3)), dtype=torch.float32),
                            A0=torch.tensor(rng.randint(0, S - 1, size=N)), L0=lattice0.reshape(-1, 3),
                            num_atoms=torch.tensor(num_atoms))
    timestep = torch.tensor(rng.randint(1, T + 1, size=B))
    g = torch.Generator().manual_seed(seed)
    noise = (torch.randn(N, 3, generator=g), torch.rand(N, S, generator=g), torch.randn(B, 3, generator=g))
    return batch, lattice0, timestep, noise, N


def test_training_step_gradients_at_the_bench_size_and_bitwise_repeatable(setup):
    """BASELINE configs[4]'s per-GPU share -- 64 crystals, ~530 atoms -- through the library's forward + backward: every
    gradient tensor within GRAD_TOL of oracle autograd, and the SAME bits when the step is repeated (round 4: the sender-side
    gradient of the spatial conv is an ordered segmented sum over a reversed adjacency; it was an fp32 atomicAdd scatter,
    reproducible only to rounding)."""
    import copy
    m, om, *_ = setup
    batch, lattice0, timestep, noise, N = _alexandria_like_batch(64, 12, 21)
    assert 350 <= N <= 800, N
    mm = copy.deepcopy(m)
    for layer in mm.model.interaction_layers:
        layer.conv.callibrated.fill_(True)
    loss = mm.training_step(batch, timestep=timestep, noise=noise)
    first = {n: p.grad.clone() for n, p in mm.named_parameters() if p.grad is not None}
    for p in mm.parameters():
        p.grad = None
    loss2 = mm.training_step(batch, timestep=timestep, noise=noise)
    assert float(loss) == float(loss2)
    for n, p in mm.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, first[n]), ("gradient not bitwise repeatable", n)
    loss_o, want = _oracle_grads(om, batch, lattice0, timestep, noise)
    # fp64 autograd through the same oracle: what both fp32 computations approximate.  A weight gradient is a sum over ~68,000
    # (edge, orientation) rows of terms of both signs, so fp32 itself -- the oracle's autograd included -- carries a
    # cancellation error that GRAD_TOL alone does not cover at this size; the library may be as far from fp64 as twice the
    # fp32 oracle is, plus GRAD_TOL of the tensor's largest entry.
    om64 = oracle_from_module(m, torch.float64)
    b64 = SimpleNamespace(X0=batch.X0.double(), A0=batch.A0, L0=batch.L0.double(), num_atoms=batch.num_atoms)
    torch.set_default_dtype(torch.float64)  # (the oracle, like the reference, builds its constant tables in the default dtype)
    try:
        _, want64 = _oracle_grads(om64, b64, lattice0.double(), timestep, tuple(x.double() for x in noise))
    finally:
        torch.set_default_dtype(torch.float32)
    # (the loss is a mean over ~530 atoms x 12 classes + 3 x 64 lattice terms of fp32 values summed in different orders by the
    # two sides: 5e-5 relative here, where the 17-atom test above holds 1e-5)
    print(f"\n[loss, 64 crystals] library {float(loss):.8f} oracle {loss_o:.8f}")
    assert abs(float(loss) - loss_o) <= 5e-5 * max(1.0, abs(loss_o))
    worst = ("", 0.0, 0.0)
    for name, w in want.items():
        if w.numel() == 0:
            continue
        w64 = want64[name]
        scale = max(float(w64.abs().max()), 1e-7)
        err = float((first[name].cpu().double() - w64).abs().max())
        err_o = float((w.double() - w64).abs().max())
        if err / scale > worst[1]:
            worst = (name, err / scale, err_o / scale)
        assert err <= GRAD_TOL * scale + 2 * err_o + 1e-7, (name, err, err_o, scale)
    print(f"\n[gradients, 64 crystals / {N} atoms] worst relative distance to fp64 autograd: library {worst[1]:.2e}, "
          f"fp32 oracle autograd {worst[2]:.2e} ({worst[0]})")


def test_make_train_preset_batch_fits_the_gradient_scratch():
    """The reference's `make train` preset (Makefile:6-7: batch_size 270, hidden_dim 200; ~2,200 atoms per batch) must run:
    round 3's batched fiber-kernel gradient put all L layers' partial sums into one scratch and failed with
    ARREAU_ECAPACITY from ~2,100 atoms at hidden_dim 200.  Checked: the step runs (the partial sums now go
    through the scratch in groups of layers that fit) and every gradient is finite, the fiber-kernel one non-zero."""
    from arreau_amd.checkpoint import make_synthetic_model
    dev = torch.device("cuda", 0)
    m = make_synthetic_model(S=12, seed=7, num_timesteps=100, hidden_dim=200).to(dev)
    for layer in m.model.interaction_layers:
        layer.conv.callibrated.fill_(True)
    batch, _lat, timestep, noise, N = _alexandria_like_batch(270, 12, 5)
    assert N > 2100, N
    loss = m.training_step(batch, timestep=timestep, noise=noise)
    assert np.isfinite(float(loss))
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    assert all(torch.isfinite(g).all() for g in grads.values())
    key = [n for n in grads if n.endswith("interaction_layers.0.conv.fiber_kernel.weight")]
    assert key and float(grads[key[0]].abs().max()) > 0


def test_reversed_adjacency_of_a_crystal_above_256_atoms(setup):
    """The ordered sender-side gradient walks a reversed adjacency that one workgroup builds per crystal, senders in passes of
    256: a 300-atom crystal (two passes, a carried base between them) next to small ones -- every gradient against oracle
    autograd, referenced to fp64 like the 64-crystal test, and bitwise repeatable."""
    import copy
    from oracle import geometry as OG
    m, om, *_ = setup
    rng = np.random.RandomState(33)
    num_atoms = [3, 300, 7]
    B, N, S = len(num_atoms), sum(num_atoms), 12
    edge = (np.array(num_atoms) / 0.055) ** (1.0 / 3.0)
    lengths = torch.tensor(edge[:, None] * rng.uniform(0.9, 1.1, size=(B, 3)), dtype=torch.float32)
    angles = torch.tensor(np.deg2rad(rng.uniform(80, 100, size=(B, 3))), dtype=torch.float32)
    lattice0 = OG.lattice_from_params(lengths, angles)
    batch = SimpleNamespace(X0=torch.tensor(rng.uniform(0, 1, size=(N, 3)), dtype=torch.float32),
                            A0=torch.tensor(rng.randint(0, S - 1, size=N)), L0=lattice0.reshape(-1, 3),
                            num_atoms=torch.tensor(num_atoms))
    timestep = torch.tensor([5, 60, 100])
    g = torch.Generator().manual_seed(2)
    noise = (torch.randn(N, 3, generator=g), torch.rand(N, S, generator=g), torch.randn(B, 3, generator=g))
    mm = copy.deepcopy(m)
    for layer in mm.model.interaction_layers:
        layer.conv.callibrated.fill_(True)
    loss = mm.training_step(batch, timestep=timestep, noise=noise)
    first = {n: p.grad.clone() for n, p in mm.named_parameters() if p.grad is not None}
    for p in mm.parameters():
        p.grad = None
    mm.training_step(batch, timestep=timestep, noise=noise)
    for n, p in mm.named_parameters():
        if p.grad is not None:
            assert torch.equal(p.grad, first[n]), ("gradient not bitwise repeatable", n)
    loss_o, want = _oracle_grads(om, batch, lattice0, timestep, noise)
    assert abs(float(loss) - loss_o) <= 5e-5 * max(1.0, abs(loss_o))
    om64 = oracle_from_module(m, torch.float64)
    b64 = SimpleNamespace(X0=batch.X0.double(), A0=batch.A0, L0=batch.L0.double(), num_atoms=batch.num_atoms)
    torch.set_default_dtype(torch.float64)
    try:
        _, want64 = _oracle_grads(om64, b64, lattice0.double(), timestep, tuple(x.double() for x in noise))
    finally:
        torch.set_default_dtype(torch.float32)
    for name, w in want.items():
        if w.numel() == 0:
            continue
        w64 = want64[name]
        scale = max(float(w64.abs().max()), 1e-7)
        err = float((first[name].cpu().double() - w64).abs().max())
        err_o = float((w.double() - w64).abs().max())
        assert err <= GRAD_TOL * scale + 2 * err_o + 1e-7, (name, err, err_o, scale)

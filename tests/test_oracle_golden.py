"""The oracle (CPU restatement) against fixtures generated from the reference's
own modules by oracle/gen_golden.py.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion as D
from oracle import geometry as G
from oracle import ponita as P
from oracle import s2grid

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def T(a, dtype=None):
    t = torch.from_numpy(np.asarray(a))
    return t.to(dtype) if dtype is not None else t


@pytest.fixture(autouse=True)
def _restore_default_dtype():
    yield
    torch.set_default_dtype(torch.float32)


# ----------------------------------------------------------------------------- radius graph
def _radius_cases():
    z = load("radius_graph.npz")
    return [(i, str(z[f"c{i}_flag"]), str(z[f"c{i}_dtype"])) for i in range(int(z["n_cases"]))]


@pytest.mark.parametrize("i,flag,tag", _radius_cases())
def test_radius_graph_matches_reference(i, flag, tag):
    z = load("radius_graph.npz")
    p = f"c{i}_"
    dt = torch.float64 if tag == "f64" else torch.float32
    torch.set_default_dtype(dt)
    cart, lattice, na = T(z[p + "cart"]), T(z[p + "lattice"]), T(z[p + "num_atoms"])
    ei, cells, cnt, dist, direction = G.radius_graph_pbc(cart, lattice, na, float(z[p + "radius"]), int(z[p + "k"]))
    assert cnt.tolist() == z[p + "count"].tolist()
    if flag.startswith("ties"):
        # choice among exactly tied images is sort-implementation dependent: compare the distance multiset
        assert ei.shape[1] == z[p + "edge_index"].shape[1]
        np.testing.assert_allclose(np.sort(dist.numpy()), np.sort(z[p + "dist"]), rtol=0, atol=1e-12)
        return
    assert torch.equal(ei, T(z[p + "edge_index"]))
    assert torch.equal(cells, T(z[p + "cells"]))
    assert torch.equal(dist, T(z[p + "dist"]))
    assert torch.equal(direction, T(z[p + "dir"]))


# ----------------------------------------------------------------------------- lattice
def test_lattice_helpers_match_reference():
    torch.set_default_dtype(torch.float64)
    z = load("lattice.npz")
    lengths = T(z["lengths"])
    assert torch.equal(G.lattice_from_params(lengths, T(z["ang_rad"])), T(z["cell_rad"]))
    assert torch.equal(G.lattice_from_params(lengths, T(z["ang_deg"])), T(z["cell_deg"]))
    kl, ka = G.matrix_to_params(T(z["known_cell"]))
    assert torch.equal(kl, T(z["known_lengths"])) and torch.equal(ka, T(z["known_angles"]))
    assert torch.equal(G.lattice_from_params(kl, ka), T(z["known_roundtrip"]))
    cart = G.frac_to_cart_coords(T(z["frac"]), T(z["cell_rad"]), T(z["num_atoms"]))
    assert torch.equal(cart, T(z["cart"]))


def test_known_cells_round_trip_lengths_angles():
    """The reference's own print-script (lattice_helpers_test.py) round-trips two cells;
    lengths and angles must be stable under cell -> params -> cell -> params."""
    torch.set_default_dtype(torch.float64)
    z = load("lattice.npz")
    kl, ka = G.matrix_to_params(T(z["known_cell"]))
    kl2, ka2 = G.matrix_to_params(G.lattice_from_params(kl, ka))
    np.testing.assert_allclose(kl2.numpy(), kl.numpy(), atol=1e-12)
    np.testing.assert_allclose(ka2.numpy(), ka.numpy(), atol=1e-12)


def test_cell_rotation_leaves_lengths_and_angles_unchanged():
    """The reference's equivariance check rotates the cell by 90 degrees about x
    (exploration/verify_model_is_equivariant.py:11-18).  The network is fed (lengths, angles) of the cell
    (diffusion_loss.py:124-127): those are rotation invariants, so the rotated crystal is the same input."""
    torch.set_default_dtype(torch.float64)
    z = load("lattice.npz")
    cell = T(z["known_cell"])
    rot = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
    l0, a0 = G.matrix_to_params(cell)
    l1, a1 = G.matrix_to_params(cell @ rot)
    np.testing.assert_allclose(l1.numpy(), l0.numpy(), atol=1e-12)
    np.testing.assert_allclose(a1.numpy(), a0.numpy(), atol=1e-12)


# ----------------------------------------------------------------------------- schedules + reverse updates
def test_schedules_match_reference():
    z = load("schedules.npz")
    for Tn in (100, 1000):
        for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            ab, be, si = D.vp_schedule(Tn, dtype=dt)
            assert ab.dtype == torch.float32 and be.dtype == dt and si.dtype == dt
            assert torch.equal(ab, T(z[f"vp{Tn}_{tag}_alpha_bars"]))
            assert torch.equal(be, T(z[f"vp{Tn}_{tag}_betas"]))
            assert torch.equal(si, T(z[f"vp{Tn}_{tag}_sigmas"]))
            assert torch.equal(D.ve_sigmas(Tn, 0.001, 1.0, dtype=dt), T(z[f"ve{Tn}_{tag}_sigmas"]))


def test_reverse_updates_match_reference():
    torch.set_default_dtype(torch.float64)
    z = load("schedules.npz")
    ab, be, _ = D.vp_schedule(1000, dtype=torch.float64)
    sig = D.ve_sigmas(1000, 0.001, 1.0, dtype=torch.float64)
    xt_l, x0_l, xt_f, eps_f = (T(z[k]) for k in ("rev_xt_l", "rev_x0_l", "rev_xt_f", "rev_eps_f"))
    for t in z["rev_ts"].tolist():
        out = D.vp_reverse_given_x0(ab, be, xt_l, x0_l, torch.tensor([t]), T(z[f"rev_l_z_{t}"]))
        assert torch.equal(out, T(z[f"rev_l_out_{t}"])), t
        tt = torch.full((xt_f.shape[0],), t)
        out = D.ve_reverse(sig, xt_f, eps_f, tt, T(z[f"rev_f_z_{t}"]))
        assert torch.equal(out, T(z[f"rev_f_out_{t}"])), t


# ----------------------------------------------------------------------------- D3PM
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_d3pm_matches_reference(tag):
    dt = torch.float64 if tag == "f64" else torch.float32
    torch.set_default_dtype(dt)
    z = load("d3pm.npz")
    S, Tn = 12, 100
    q1t, qm = D.d3pm_buffers(Tn, S, dtype=dt)
    assert torch.equal(q1t[0], T(z[f"{tag}_q_one_step_transposed_0"]))
    for ti in (0, 1, 49, 98, 99):
        assert torch.equal(qm[ti], T(z[f"{tag}_q_mats_{ti}"]))
    x_t, logits = T(z[f"{tag}_x_t"]), T(z[f"{tag}_logits"])
    for t in (Tn - 1, Tn // 2, 2, 1):
        tt = torch.full((x_t.shape[0],), t)
        post = D.d3pm_q_posterior_logits(q1t, qm, logits, x_t, tt)
        assert torch.equal(post, T(z[f"{tag}_post_{t}"])), t
        rev = D.d3pm_reverse(q1t, qm, x_t, logits, tt, T(z[f"{tag}_u_{t}"]))
        assert torch.equal(rev, T(z[f"{tag}_rev_{t}"])), t


@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_d3pm_big_chain_spot_rows(tag):
    dt = torch.float64 if tag == "f64" else torch.float32
    z = load("d3pm.npz")
    _, qm = D.d3pm_buffers(1000, 90, dtype=dt)
    for ti in (0, 499, 998, 999):
        assert torch.equal(qm[ti][0], T(z[f"big_{tag}_q_mats_{ti}_row0"]))
        assert torch.equal(qm[ti][89], T(z[f"big_{tag}_q_mats_{ti}_row89"]))


# ----------------------------------------------------------------------------- network pieces
@pytest.mark.parametrize("tag", ["f64", "f32"])
def test_network_pieces_match_reference(tag):
    dt = torch.float64 if tag == "f64" else torch.float32
    torch.set_default_dtype(dt)
    z = load("network_pieces.npz")
    g = lambda k: T(z[f"{tag}_{k}"])
    grid = g("grid")
    inv12, inv3 = P.pair_invariants(g("dir"), grid)
    assert torch.equal(inv12, g("inv12")) and torch.equal(inv3, g("inv3"))
    assert torch.equal(P.polynomial_features(g("attr"), 3), g("poly3"))
    assert torch.equal(P.polynomial_features(inv3, 3), g("poly3_fiber"))
    assert torch.equal(P.polynomial_cutoff(g("cut_d"), 5.0), g("cut"))
    assert torch.equal(P.scalar_to_sphere(g("sc"), grid), g("scalar_to_sphere"))
    assert torch.equal(P.vec_to_sphere(g("vc"), grid), g("vec_to_sphere"))
    assert torch.equal(P.sphere_to_scalar(g("sg")), g("sphere_to_scalar"))
    assert torch.equal(P.sphere_to_vec(g("sg"), grid), g("sphere_to_vec"))
    assert torch.equal(D.gaussian_fourier_projection(g("gfp_t"), g("gfp_w")), g("gfp"))
    # ConvNext block with the fixture's stand-in conv (conv(x) = 2x + 1)
    sd = {"blk." + k[len(f"{tag}_cn_sd_"):]: T(z[k]) for k in z.files if k.startswith(f"{tag}_cn_sd_")}
    x = g("cn_x")
    y = P.convnext_block(sd, "blk", x, 2 * x + 1)
    assert torch.equal(y, g("cn_y"))


@pytest.mark.parametrize("O", [8, 16])
def test_ori_grid_generator_matches_reference(O):
    torch.set_default_dtype(torch.float32)
    z = load("ori_grid.npz")
    torch.manual_seed(int(z[f"seed_{O}"]))
    grid = s2grid.uniform_grid_s2(O)
    np.testing.assert_allclose(grid.numpy(), z[f"ori_grid_{O}"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(grid.numpy(), axis=-1), 1.0, atol=1e-6)


# ------------------------------------------------------------------------------------------- training (config 5)
@pytest.mark.parametrize("tag,dtype", [("f64", torch.float64), ("f32", torch.float32)])
def test_training_noising_and_d3pm_loss_match_reference(tag, dtype):
    """oracle/training.py against tests/golden/training.npz (reference: VE_pbc.forward, min_distance_sqr_pbc,
    cart_to_frac_coords, VP_lattice.forward, matrix_to_params, D3PM.get_xt / q_posterior_logits / vb /
    calculate_loss), bit for bit in both dtypes."""
    from oracle import training as TR
    torch.set_default_dtype(dtype)
    z = load("training.npz")
    p = tag + "_"
    na = T(z["num_atoms"])
    lattice, frac0, types0 = T(z[p + "lattice"]), T(z[p + "frac0"]), T(z[p + "types0"])
    timestep = T(z[p + "timestep"])
    t_feat = timestep.repeat_interleave(na)
    lengths, angles = G.matrix_to_params(lattice)
    assert np.array_equal(lengths.numpy(), z[p + "m2p_lengths"]) and np.array_equal(angles.numpy(), z[p + "m2p_angles"])
    # VE forward noising
    fn, weps, used = TR.ve_forward(T(z[p + "ve_sigmas"]), frac0, t_feat, lattice, na, T(z[p + "ve_z"]))
    assert np.array_equal(fn.numpy(), z[p + "ve_frac_noisy"])
    assert np.array_equal(weps.numpy(), z[p + "ve_wrapped_eps"])
    assert np.array_equal(used.numpy(), z[p + "ve_used_sigmas"])
    dsq, vec = TR.min_distance_sqr_pbc(T(z[p + "md_c1"]), T(z[p + "md_c2"]), lattice, na)
    assert np.array_equal(dsq.numpy(), z[p + "md_dsq"]) and np.array_equal(vec.numpy(), z[p + "md_vec"])
    assert np.array_equal(TR.cart_to_frac_coords(T(z[p + "md_c1"]), lattice, na).numpy(), z[p + "c2f"])
    # VP forward noising of the lengths
    ht = TR.vp_forward(T(z[p + "vp_alpha_bars"]), lengths, timestep, T(z[p + "vp_eps"]))
    assert np.array_equal(ht.numpy(), z[p + "vp_ht"])
    # D3PM
    Tn, S = int(z["T"]), int(z["S"])
    q1t, qm = D.d3pm_buffers(Tn, S)
    x_t = TR.d3pm_q_sample(qm, types0, t_feat, T(z[p + "d3_u"]))
    assert np.array_equal(x_t.numpy(), z[p + "d3_xt"])
    pred_logits = T(z[p + "d3_pred_logits"])
    true_post = TR.d3pm_q_posterior_logits(q1t, qm, types0, x_t, t_feat)
    pred_post = TR.d3pm_q_posterior_logits(q1t, qm, pred_logits, x_t, t_feat)
    assert np.array_equal(true_post.numpy(), z[p + "d3_true_post"])
    assert np.array_equal(pred_post.numpy(), z[p + "d3_pred_post"])
    assert np.array_equal(TR.d3pm_vb(true_post, pred_post).numpy(), z[p + "d3_vb"])
    loss, _, _ = TR.d3pm_calculate_loss(q1t, qm, types0, pred_logits, x_t, t_feat)
    assert np.array_equal(loss.numpy(), z[p + "d3_loss"])


def test_training_frac_error_wraps_mod_one():
    """compute_frac_x_error (diffusion_loss.py:95-110; restated from source, its module needs torch_geometric):
    the distance between 0.1 and 0.9 is 0.2, not 0.8."""
    from oracle import training as TR
    pred = torch.tensor([[0.1, 0.5, 0.0], [0.25, 0.25, 0.25]])
    target = torch.tensor([[0.9, 0.5, 1.0], [0.75, 1.5, -0.25]])
    e = TR.compute_frac_x_error(pred, target)
    want = ((0.2 ** 2 + 0 + 0) + (0.5 ** 2 + 0.25 ** 2 + 0.5 ** 2)) / 2
    assert abs(float(e) - want) < 1e-6

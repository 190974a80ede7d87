"""Generate the committed golden fixtures from the reference's own importable modules.

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

It imports the pure-torch reference modules that load here and
writes small ``tests/golden/*.npz`` files holding inputs + expected outputs.
Nothing of the reference's source is written anywhere; the fixtures are data.
The GPU box never runs this script (there is no /root/reference there).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("ARREAU_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def random_cell(rng, B, lo, hi, dtype):
    """Random well-conditioned triclinic cells: lengths U(lo,hi), angles U(70,110) deg."""
    lengths = rng.uniform(lo, hi, size=(B, 3))
    ang = np.deg2rad(rng.uniform(70, 110, size=(B, 3)))
    return torch.tensor(lengths, dtype=dtype), torch.tensor(ang, dtype=dtype)


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from diffusion import d3pm as r_d3pm
    from diffusion import diffusion_helpers as r_dh
    from diffusion import lattice_helpers as r_lh
    from ponita.geometry import invariants as r_inv
    from ponita.geometry import rotation as r_rot
    from ponita.nn import convnext as r_cn
    from ponita.nn import embedding as r_emb
    from ponita.utils import to_from_sphere as r_sph
    from ponita.utils import windowing as r_win

    os.makedirs(OUT, exist_ok=True)

    # ------------------------------------------------------------------ (i) radius graph
    cases = {}
    idx = 0
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        torch.set_default_dtype(dtype)
        rng = np.random.RandomState(100)
        specs = [
            # (num_atoms list, cell range, k, flag)
            ([4, 4], (4.0, 7.0), 8, "generic"),
            ([8, 8, 8], (3.0, 5.0), 8, "dense"),
            ([1, 4, 8, 2], (4.0, 8.0), 8, "ragged"),
            ([8, 3], (9.0, 12.0), 8, "sparse"),      # below the cap: early-return path
            ([6, 5], (3.0, 6.0), 4, "k4"),
            ([1], (2.0, 2.0), 8, "ties_cubic_single"),  # tie-heavy, flagged
        ]
        for num_atoms, (lo, hi), k, flag in specs:
            B = len(num_atoms)
            if flag.startswith("ties"):
                lengths = torch.full((B, 3), lo, dtype=dtype)
                angles = torch.full((B, 3), np.pi / 2, dtype=dtype)
            else:
                lengths, angles = random_cell(rng, B, lo, hi, dtype)
            lattice = r_lh.lattice_from_params(lengths, angles)
            na = torch.tensor(num_atoms)
            frac = torch.tensor(rng.uniform(0, 1, size=(int(na.sum()), 3)), dtype=dtype)
            cart = r_dh.frac_to_cart_coords(frac, lattice, na)
            ei, cells, cnt, dist, direction = r_dh.radius_graph_pbc(
                cart, lattice, na, 5.0, k, device=cart.device, remove_self_edges=True)
            pre = f"c{idx}_"
            cases.update({
                pre + "flag": np.array(flag), pre + "dtype": np.array(tag), pre + "k": np.array(k),
                pre + "radius": np.array(5.0), pre + "num_atoms": _np(na), pre + "cart": _np(cart),
                pre + "lattice": _np(lattice), pre + "frac": _np(frac), pre + "edge_index": _np(ei),
                pre + "cells": _np(cells), pre + "count": _np(cnt), pre + "dist": _np(dist),
                pre + "dir": _np(direction),
            })
            idx += 1
    cases["n_cases"] = np.array(idx)
    np.savez_compressed(os.path.join(OUT, "radius_graph.npz"), **cases)

    # ------------------------------------------------------------------ (ii) lattice helpers
    torch.set_default_dtype(torch.float64)
    rng = np.random.RandomState(7)
    lat = {}
    lengths = torch.tensor(rng.normal(size=(6, 3)))  # may be negative like the sampler's
    ang_rad = torch.tensor(np.deg2rad(rng.uniform(60, 120, size=(6, 3))))
    ang_deg = torch.tensor(np.stack([np.full(6, 90.0), rng.uniform(90, 180, size=6), np.full(6, 90.0)], 1))
    lat["lengths"], lat["ang_rad"], lat["ang_deg"] = _np(lengths), _np(ang_rad), _np(ang_deg)
    lat["cell_rad"] = _np(r_lh.lattice_from_params(lengths, ang_rad))
    lat["cell_deg"] = _np(r_lh.lattice_from_params(lengths, ang_deg))  # degrees consumed as radians
    known = torch.tensor([  # the two cells of diffusion/lattice_helpers_test.py:9-20 (data)
        [[5.28526086, 0.0, 0.0], [2.64263043, 4.57717017, 0.0], [2.64263043, 1.52572339, 4.31539742]],
        [[5.52431857, 0.0, 0.0], [2.76215929, 4.78420022, 0.0], [2.76215929, 1.59473341, 4.51058723]],
    ])
    kl, ka = r_lh.matrix_to_params(known)
    lat["known_cell"], lat["known_lengths"], lat["known_angles"] = _np(known), _np(kl), _np(ka)
    lat["known_roundtrip"] = _np(r_lh.lattice_from_params(kl, ka))
    na = torch.tensor([2, 1, 3, 1, 2, 2])
    frac = torch.tensor(rng.uniform(size=(int(na.sum()), 3)))
    lat["num_atoms"], lat["frac"] = _np(na), _np(frac)
    lat["cart"] = _np(r_dh.frac_to_cart_coords(frac, torch.tensor(lat["cell_rad"]), na))
    np.savez_compressed(os.path.join(OUT, "lattice.npz"), **lat)

    # ------------------------------------------------------------------ (iii)+(iv) schedules and reverse updates
    sch = {}
    for T in (100, 1000):
        for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
            torch.set_default_dtype(dtype)  # the reference's buffers depend on the default dtype
            vp = r_dh.VP_lattice(num_steps=T, power=2, clipmax=0.999)
            sch[f"vp{T}_{tag}_alpha_bars"], sch[f"vp{T}_{tag}_betas"], sch[f"vp{T}_{tag}_sigmas"] = (
                _np(vp.alpha_bars), _np(vp.betas), _np(vp.sigmas))
            ve = r_dh.VE_pbc(T, sigma_min=0.001, sigma_max=1.0)
            sch[f"ve{T}_{tag}_sigmas"] = _np(ve.sigmas)
    torch.set_default_dtype(torch.float64)
    T = 1000
    vp = r_dh.VP_lattice(num_steps=T, power=2, clipmax=0.999)
    ve = r_dh.VE_pbc(T, sigma_min=0.001, sigma_max=1.0)
    rng = np.random.RandomState(11)
    B, N = 3, 7
    xt_l = torch.tensor(rng.normal(size=(B, 3)))
    x0_l = torch.tensor(rng.normal(size=(B, 3)))
    xt_f = torch.tensor(rng.uniform(size=(N, 3)))
    eps_f = torch.tensor(rng.normal(size=(N, 3)))
    sch["rev_xt_l"], sch["rev_x0_l"], sch["rev_xt_f"], sch["rev_eps_f"] = map(_np, (xt_l, x0_l, xt_f, eps_f))
    ts = [T - 1, T // 2, 2, 1]
    sch["rev_ts"] = np.array(ts)
    for t in ts:
        torch.manual_seed(1000 + t)
        z = torch.randn_like(xt_l)
        torch.manual_seed(1000 + t)
        out = vp.reverse_given_x0(xt_l, x0_l, torch.tensor([t]))
        sch[f"rev_l_z_{t}"], sch[f"rev_l_out_{t}"] = _np(z), _np(out)
        tt = torch.full((N,), t)
        torch.manual_seed(2000 + t)
        z = torch.randn_like(xt_f)
        torch.manual_seed(2000 + t)
        out = ve.reverse(xt_f, eps_f, tt, None, None)
        sch[f"rev_f_z_{t}"], sch[f"rev_f_out_{t}"] = _np(z), _np(out)
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **sch)

    # ------------------------------------------------------------------ (iv)+(viii) D3PM
    dd = {}
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        torch.set_default_dtype(dtype)
        S, T = 12, 100
        d = r_d3pm.D3PM(x0_model=None, n_T=T, num_classes=S, forward_type="mask")
        rng = np.random.RandomState(5)
        N = 9
        x_t = torch.tensor(rng.randint(0, S, size=N))
        x_t[::3] = S - 1
        logits = torch.tensor(rng.normal(size=(N, S)) * 2, dtype=dtype)
        dd[f"{tag}_x_t"], dd[f"{tag}_logits"] = _np(x_t), _np(logits)
        dd[f"{tag}_q_one_step_transposed_0"] = _np(d.q_one_step_transposed[0])
        for ti in (0, 1, 49, 98, 99):
            dd[f"{tag}_q_mats_{ti}"] = _np(d.q_mats[ti])
        for t in (T - 1, T // 2, 2, 1):
            tt = torch.full((N,), t)
            dd[f"{tag}_post_{t}"] = _np(d.q_posterior_logits(logits, x_t, tt))
            torch.manual_seed(300 + t)
            u = torch.rand((N, S))
            torch.manual_seed(300 + t)
            dd[f"{tag}_u_{t}"], dd[f"{tag}_rev_{t}"] = _np(u), _np(d.reverse(x_t, logits, tt))
    # spot rows of the T=1000, S=90 product chain (SURVEY 8c viii), float64 and float32
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        torch.set_default_dtype(dtype)
        d = r_d3pm.D3PM(x0_model=None, n_T=1000, num_classes=90, forward_type="mask")
        for ti in (0, 499, 998, 999):
            dd[f"big_{tag}_q_mats_{ti}_row0"] = _np(d.q_mats[ti][0])
            dd[f"big_{tag}_q_mats_{ti}_row89"] = _np(d.q_mats[ti][89])
    np.savez_compressed(os.path.join(OUT, "d3pm.npz"), **dd)

    # ------------------------------------------------------------------ (v)+(vi) network pieces
    nn = {}
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        torch.set_default_dtype(dtype)
        rng = np.random.RandomState(3)
        O, E, Nn = 8, 10, 5
        grid = torch.tensor(rng.normal(size=(O, 3)), dtype=dtype)
        grid = grid / grid.norm(dim=-1, keepdim=True)
        direction = torch.tensor(rng.normal(size=(E, 3)) * 2, dtype=dtype)
        a2, a3 = r_inv.invariant_attr_r3s2_fiber_bundle(None, grid, None, direction, separable=True)
        nn[f"{tag}_grid"], nn[f"{tag}_dir"], nn[f"{tag}_inv12"], nn[f"{tag}_inv3"] = map(_np, (grid, direction, a2, a3))
        attr = torch.tensor(rng.normal(size=(E, O, 6)), dtype=dtype)
        nn[f"{tag}_attr"], nn[f"{tag}_poly3"] = _np(attr), _np(r_emb.PolynomialFeatures(3)(attr))
        nn[f"{tag}_poly3_fiber"] = _np(r_emb.PolynomialFeatures(3)(a3))
        d = torch.tensor(np.concatenate([rng.uniform(0, 5.5, size=14), [0.0, 5.0, 4.999999]]), dtype=dtype)
        nn[f"{tag}_cut_d"], nn[f"{tag}_cut"] = _np(d), _np(r_win.PolynomialCutoff(5.0)(d))
        sc = torch.tensor(rng.normal(size=(Nn, 3)), dtype=dtype)
        vc = torch.tensor(rng.normal(size=(Nn, 4, 3)), dtype=dtype)
        sg = torch.tensor(rng.normal(size=(Nn, O, 2)), dtype=dtype)
        nn[f"{tag}_sc"], nn[f"{tag}_vc"], nn[f"{tag}_sg"] = map(_np, (sc, vc, sg))
        nn[f"{tag}_scalar_to_sphere"] = _np(r_sph.scalar_to_sphere(sc, grid))
        nn[f"{tag}_vec_to_sphere"] = _np(r_sph.vec_to_sphere(vc, grid))
        nn[f"{tag}_sphere_to_scalar"] = _np(r_sph.sphere_to_scalar(sg))
        nn[f"{tag}_sphere_to_vec"] = _np(r_sph.sphere_to_vec(sg, grid))
        w = torch.tensor(rng.normal(size=32) * 16, dtype=dtype)
        tv = torch.tensor(rng.uniform(0, 1, size=(4, 1)), dtype=torch.float32)
        proj = r_dh.GaussianFourierProjection(32, 16.0)
        proj.gaussian_fourier_proj_w.data = w
        nn[f"{tag}_gfp_w"], nn[f"{tag}_gfp_t"], nn[f"{tag}_gfp"] = _np(w), _np(tv), _np(proj(tv))
        # ConvNext with a stand-in conv (the real conv needs torch_geometric): conv(x) = (2x+1, None)
        C = 16

        class StandIn(torch.nn.Module):
            def forward(self, x, edge_index, edge_attr, **kw):
                return 2 * x + 1, None

        torch.manual_seed(17)
        blk = r_cn.ConvNext(C, StandIn(), act=torch.nn.GELU(), layer_scale=1e-6, widening_factor=4)
        blk.layer_scale.data = torch.tensor(rng.uniform(0.1, 1.0, size=C), dtype=dtype)
        blk.norm.weight.data = torch.tensor(rng.uniform(0.5, 1.5, size=C), dtype=dtype)
        blk.norm.bias.data = torch.tensor(rng.normal(size=C) * 0.1, dtype=dtype)
        xin = torch.tensor(rng.normal(size=(Nn, O, C)), dtype=dtype)
        yout, _ = blk(xin, None, None)
        nn[f"{tag}_cn_x"], nn[f"{tag}_cn_y"] = _np(xin), _np(yout)
        for k, v in blk.state_dict().items():
            nn[f"{tag}_cn_sd_{k}"] = _np(v)
    np.savez_compressed(os.path.join(OUT, "network_pieces.npz"), **nn)

    # ------------------------------------------------------------------ (vii) orientation grids
    torch.set_default_dtype(torch.float32)
    gg = {}
    for O in (8, 16):
        torch.manual_seed(4242 + O)
        gg[f"ori_grid_{O}"] = _np(r_rot.uniform_grid_s2(O, show_pbar=False))
        gg[f"seed_{O}"] = np.array(4242 + O)
    np.savez_compressed(os.path.join(OUT, "ori_grid.npz"), **gg)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} bytes")


if __name__ == "__main__":
    main()

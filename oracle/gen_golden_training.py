"""Generate tests/golden/training.npz from the reference's own importable modules (forward noising + D3PM loss pieces).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden_training

Separate from oracle/gen_golden.py so that the round-1 fixtures stay byte-identical.  The reference functions draw
their noise internally (randn_like / rand); the fixture stores that noise by reseeding the global generator and
repeating the same first draw, so the oracle (which takes noise as an input) can be compared value for value.
Nothing of the reference's source is written anywhere; the fixture is data.
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("ARREAU_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from diffusion import d3pm as r_d3pm
    from diffusion import diffusion_helpers as r_dh
    from diffusion import lattice_helpers as r_lh

    T, S = 100, 12
    num_atoms = [3, 5, 2, 1]
    out = {"T": np.array(T), "S": np.array(S), "num_atoms": np.array(num_atoms)}
    for dtype, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        torch.set_default_dtype(dtype)
        rng = np.random.RandomState(11)
        B, N = len(num_atoms), sum(num_atoms)
        na = torch.tensor(num_atoms)
        lengths = torch.tensor(rng.uniform(3.0, 7.0, size=(B, 3)), dtype=dtype)
        angles = torch.tensor(np.deg2rad(rng.uniform(70, 110, size=(B, 3))), dtype=dtype)
        lattice = r_lh.lattice_from_params(lengths, angles)
        frac0 = torch.tensor(rng.uniform(0, 1, size=(N, 3)), dtype=dtype)
        types0 = torch.tensor(rng.randint(0, S - 1, size=N))
        timestep = torch.tensor([[1], [50], [T], [2]])  # [B,1] like diffusion_loss.py:213-216
        t_feat = timestep.repeat_interleave(na, dim=0)  # [N,1]
        p = tag + "_"
        out.update({p + "lattice": _np(lattice), p + "frac0": _np(frac0), p + "types0": _np(types0),
                    p + "timestep": _np(timestep.squeeze(1))})
        # matrix_to_params (lattice_helpers.py:16-35)
        len_m, ang_m = r_lh.matrix_to_params(lattice)
        out[p + "m2p_lengths"], out[p + "m2p_angles"] = _np(len_m), _np(ang_m)
        # VE_pbc.forward (diffusion_helpers.py:43-63)
        ve = r_dh.VE_pbc(T, sigma_min=0.001, sigma_max=1.0)
        torch.manual_seed(5)
        z = torch.randn_like(frac0)
        torch.manual_seed(5)
        frac_noisy, wrapped_eps, used = ve(frac0, t_feat, lattice, na)
        out.update({p + "ve_sigmas": _np(ve.sigmas), p + "ve_z": _np(z), p + "ve_frac_noisy": _np(frac_noisy),
                    p + "ve_wrapped_eps": _np(wrapped_eps), p + "ve_used_sigmas": _np(used)})
        # min_distance_sqr_pbc / cart_to_frac_coords on their own (diffusion_helpers.py:233-325)
        c1 = torch.tensor(rng.uniform(-2, 9, size=(N, 3)), dtype=dtype)
        c2 = torch.tensor(rng.uniform(-2, 9, size=(N, 3)), dtype=dtype)
        dsq, vec = r_dh.min_distance_sqr_pbc(c1, c2, lattice, na, c1.device, return_vector=True)
        out.update({p + "md_c1": _np(c1), p + "md_c2": _np(c2), p + "md_dsq": _np(dsq), p + "md_vec": _np(vec),
                    p + "c2f": _np(r_dh.cart_to_frac_coords(c1, lattice, na))})
        # VP_lattice.forward (diffusion_helpers.py:156-163)
        vp = r_dh.VP_lattice(num_steps=T, power=2, clipmax=0.999)
        torch.manual_seed(6)
        eps_l = torch.randn_like(len_m)
        torch.manual_seed(6)
        ht, eps_ret = vp(len_m, timestep)
        assert torch.equal(eps_l, eps_ret)
        out.update({p + "vp_alpha_bars": _np(vp.alpha_bars), p + "vp_eps": _np(eps_l), p + "vp_ht": _np(ht)})
        # D3PM (d3pm.py:67-163)
        d = r_d3pm.D3PM(x0_model=None, n_T=T, num_classes=S, forward_type="mask")
        tf = t_feat.squeeze()
        torch.manual_seed(7)
        u = torch.rand((N, S))
        torch.manual_seed(7)
        x_t = d.get_xt(types0, tf)
        assert torch.equal(x_t, d.q_sample(types0, tf, u))
        pred_logits = torch.tensor(rng.normal(size=(N, S)) * 2, dtype=dtype)
        true_post = d.q_posterior_logits(types0, x_t, tf)
        pred_post = d.q_posterior_logits(pred_logits, x_t, tf)
        out.update({p + "d3_u": _np(u), p + "d3_xt": _np(x_t), p + "d3_pred_logits": _np(pred_logits),
                    p + "d3_true_post": _np(true_post), p + "d3_pred_post": _np(pred_post),
                    p + "d3_vb": _np(d.vb(true_post, pred_post)),
                    p + "d3_loss": _np(d.calculate_loss(types0, pred_logits, x_t, tf))})
    torch.set_default_dtype(torch.float32)
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, "training.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

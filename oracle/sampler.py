"""Oracle: per-step feature assembly, one denoising step and the sampling loop (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import diffusion as D
from . import geometry as G
from . import ponita as P

POS_SIGMA_MIN, POS_SIGMA_MAX = 0.001, 1.0  # diffusion/diffusion_loss.py:30-31


@dataclass
class OracleModel:
    """Everything PONITA_DIFFUSION owns that the sampler reads
    (lightning_wrappers/diffusion.py:32-102), as plain tensors."""
    sd: dict  # state_dict of the network, "model." prefix stripped
    hp: dict  # num_layers, degree, radius, S, max_neighbors, T
    ori_grid: torch.Tensor  # [O,3]
    t_emb_w: torch.Tensor  # [32]  t_emb.gaussian_fourier_proj_w
    ve_sigmas: torch.Tensor  # [T+1]
    vp_alpha_bars: torch.Tensor  # [T+1] float32
    vp_betas: torch.Tensor  # [T+1] default dtype holding float32-computed values
    q_one_step_transposed: torch.Tensor  # [T,S,S]
    q_mats: torch.Tensor  # [T,S,S]

    @staticmethod
    def from_state_dict(full_sd: dict, hp: dict, ori_grid: torch.Tensor, dtype=torch.float32):
        """Split a PONITA_DIFFUSION state_dict (keys listed in SURVEY.md section 5).

        Float tensors are cast to ``dtype`` except VP alpha_bars, which stays
        float32 like the reference buffer (diffusion_helpers.py:141)."""
        cast = lambda v: v.to(dtype) if v.is_floating_point() else v
        net = {k[len("model."):]: cast(v) for k, v in full_sd.items() if k.startswith("model.")}
        return OracleModel(
            sd=net, hp=hp, ori_grid=ori_grid.to(dtype),
            t_emb_w=cast(full_sd["t_emb.gaussian_fourier_proj_w"]),
            ve_sigmas=cast(full_sd["diffusion_loss.pos_diffusion.sigmas"]),
            vp_alpha_bars=full_sd["diffusion_loss.lattice_diffusion.alpha_bars"].float(),
            vp_betas=cast(full_sd["diffusion_loss.lattice_diffusion.betas"]),
            q_one_step_transposed=cast(full_sd["diffusion_loss.d3pm.q_one_step_transposed"]),
            q_mats=cast(full_sd["diffusion_loss.d3pm.q_mats"]),
        )


def assemble_features(m: OracleModel, frac, types_onehot, t_feat, num_atoms, lengths, angles):
    """Node features of diffusion/diffusion_loss.py:124-158.

    Returns (x [N,S+74], cart [N,3], vec [N,4,3], lattice [B,3,3]).  The time
    embedding is of betas[t] (float32 buffer), not of t (:126-127).
    """
    lattice = G.lattice_from_params(lengths, angles)
    t = m.vp_betas[t_feat].view(-1, 1)
    t_emb = D.gaussian_fourier_projection(t, m.t_emb_w)
    rep = lambda v: torch.repeat_interleave(v, num_atoms, dim=0)
    n_feat = rep(num_atoms).unsqueeze(-1)
    scaled = (lengths / num_atoms.unsqueeze(-1)).abs()
    x = torch.cat([types_onehot, t_emb, n_feat, rep(lengths), rep(angles), rep(scaled)], dim=1)
    cart = G.frac_to_cart_coords(frac, lattice, num_atoms)
    vec = torch.cat([frac.unsqueeze(1), rep(lattice)], dim=1)
    return x, cart, vec, lattice


def predict_scores(m: OracleModel, frac, types_onehot, t_feat, num_atoms, lengths, angles,
                   batch, edges=None, return_graph=False):
    """diffusion/diffusion_loss.py:112-197.

    ``edges`` = (edge_index, dists, direction) teacher-forces a neighbour list
    (used for parity of the network independent of tie-breaking); otherwise the
    oracle's own radius_graph_pbc is used like the reference does at :164-174.
    Returns (pred_frac_eps [N,3], logits [N,S], pred_lengths_0 [B,3]).
    """
    x, cart, vec, lattice = assemble_features(m, frac, types_onehot, t_feat, num_atoms, lengths, angles)
    if edges is None:
        edge_index, _cells, _cnt, dists, direction = G.radius_graph_pbc(
            cart, lattice, num_atoms, m.hp["radius"], m.hp["max_neighbors"], remove_self_edges=True)
    else:
        edge_index, dists, direction = edges
    batch_of_edge = batch[edge_index[0]]
    dt = frac.dtype
    logits, vec_out, gscalar = P.ponita_forward(
        m.sd, m.hp, x.to(dt), vec.to(dt), edge_index, dists, direction, lattice.to(dt), batch,
        batch_of_edge, m.ori_grid)
    out = (vec_out.squeeze(1), logits, gscalar)
    if return_graph:
        return out + ((edge_index, dists, direction, cart, lattice),)
    return out


@dataclass
class StepNoise:
    """The three draws of one loop iteration, in the reference's order:
    randn[B,3] (diffusion_helpers.py:193-197), randn[N,3] (:79), rand[N,S] (d3pm.py:206)."""
    z_lattice: torch.Tensor
    z_frac: torch.Tensor
    u_types: torch.Tensor


def reverse_step(m: OracleModel, frac, atom_types, lengths, angles, num_atoms, scores, timestep: int,
                 noise: StepNoise):
    """The four updates of diffusion/diffusion_loss.py:338-347 for one timestep."""
    eps_x, logits, len0 = scores
    N = frac.shape[0]
    t = torch.full((N,), timestep, dtype=torch.long)
    tvec = torch.tensor([timestep])
    len_scaled = len0 * num_atoms.unsqueeze(-1)
    lengths = D.vp_reverse_given_x0(m.vp_alpha_bars, m.vp_betas, lengths, len_scaled, tvec, noise.z_lattice)
    lattice = G.lattice_from_params(lengths, angles)
    frac = D.ve_reverse(m.ve_sigmas, frac, eps_x, t, noise.z_frac)
    atom_types = D.d3pm_reverse(m.q_one_step_transposed, m.q_mats, atom_types, logits, t, noise.u_types)
    return frac, atom_types, lengths, lattice


@dataclass
class SampleTrace:
    """Optional per-step record used by parity tests."""
    steps: list = field(default_factory=list)


def init_state(m: OracleModel, n_per: int, B: int, dtype, np_rng=None):
    """Sampler initial state, diffusion/diffusion_loss.py:294-316.

    Draw order: B numpy uniforms (angles, DEGREES), randn[B,3] (lengths,
    default dtype), randn[N,3] (frac, default dtype) * pos_sigma_max; all atom
    types start in the mask state S-1."""
    np_rng = np.random if np_rng is None else np_rng
    angles = torch.tensor(np.array([(90, np_rng.uniform(90, 180), 90) for _ in range(B)])).to(dtype)
    lengths = torch.randn([B, 3], dtype=dtype)
    frac = torch.randn([B * n_per, 3], dtype=dtype) * POS_SIGMA_MAX
    num_atoms = torch.full((B,), n_per)
    atom_types = torch.full((B * n_per,), m.hp["S"] - 1)
    return frac, atom_types, lengths, angles, num_atoms


def sample(m: OracleModel, n_per: int, B: int, dtype=torch.float32, trace: Optional[SampleTrace] = None,
           max_steps: Optional[int] = None, state=None):
    """DiffusionLoss.sample, diffusion/diffusion_loss.py:276-377 (no visualisation).

    Runs timesteps T-1 .. 1 (T-1 iterations, :318) drawing noise from torch's
    global CPU generator in the reference's order.  Returns (frac, atom_types,
    lengths, lattice) as tensors.
    """
    S, T = m.hp["S"], m.hp["T"]
    if state is None:
        state = init_state(m, n_per, B, dtype)
    frac, atom_types, lengths, angles, num_atoms = state
    batch = torch.arange(0, B).repeat_interleave(n_per)
    N = B * n_per
    lattice = None
    done = 0
    for timestep in reversed(range(1, T)):
        t = torch.full((N,), timestep)
        scores = predict_scores(m, frac, F.one_hot(atom_types, S), t, num_atoms, lengths, angles, batch)
        noise = StepNoise(torch.randn([B, 3], dtype=dtype), torch.randn([N, 3], dtype=dtype),
                          torch.rand([N, S], dtype=dtype))
        if trace is not None:
            trace.steps.append(dict(t=timestep, frac=frac.clone(), types=atom_types.clone(),
                                    lengths=lengths.clone(), scores=scores, noise=noise))
        frac, atom_types, lengths, lattice = reverse_step(
            m, frac, atom_types, lengths, angles, num_atoms, scores, timestep, noise)
        done += 1
        if max_steps is not None and done >= max_steps:
            break
    return frac, atom_types, lengths, lattice

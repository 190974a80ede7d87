"""Shared helpers for the parity tests (oracle side + state builders)."""
import numpy as np
import torch

from oracle.sampler import OracleModel


def oracle_from_module(module, dtype=torch.float32):
    net = module.model
    hp = dict(num_layers=net.num_layers, degree=net.degree, radius=float(module.diffusion_loss.cutoff),
              S=module.num_atomic_states, max_neighbors=int(module.diffusion_loss.max_neighbors),
              T=int(module.diffusion_loss.T))
    sd = {k: v.detach().cpu() for k, v in module.state_dict().items()}
    return OracleModel.from_state_dict(sd, hp, net.ori_grid.detach().cpu(), dtype)


def random_state(S, num_atoms, seed, sampler_like=False, cell=(4.0, 8.0)):
    """A sampler state (frac, types, lengths, angles, num_atoms) on the CPU in float32.

    sampler_like=True draws it like the sampler's start (lengths ~ N(0,1), frac ~ N(0,1), monoclinic
    angles in degrees); otherwise physically reasonable cells with angles in radians."""
    rng = np.random.RandomState(seed)
    na = torch.tensor(num_atoms)
    B, N = len(num_atoms), int(sum(num_atoms))
    if sampler_like:
        lengths = torch.tensor(rng.normal(size=(B, 3)), dtype=torch.float32)
        angles = torch.tensor(np.stack([np.full(B, 90.0), rng.uniform(90, 180, B), np.full(B, 90.0)], 1),
                              dtype=torch.float32)
        frac = torch.tensor(rng.normal(size=(N, 3)), dtype=torch.float32)
    else:
        lengths = torch.tensor(rng.uniform(*cell, size=(B, 3)), dtype=torch.float32)
        angles = torch.tensor(np.deg2rad(rng.uniform(70, 110, size=(B, 3))), dtype=torch.float32)
        frac = torch.tensor(rng.uniform(0, 1, size=(N, 3)), dtype=torch.float32)
    types = torch.tensor(rng.randint(0, S, size=N))
    types[::3] = S - 1
    return frac, types, lengths, angles, na


def slots_from_edges(edge_index, dists, direction, N, k):
    """Receiver-sorted COO edges -> (deg, src, dir, dist) slot arrays on the CPU (test-side reference
    for arreau_edges_to_slots)."""
    deg = torch.zeros(N, dtype=torch.int32)
    src = torch.full((N, k), -1, dtype=torch.int32)
    sdir = torch.zeros((N, k, 3), dtype=torch.float32)
    sdist = torch.zeros((N, k), dtype=torch.float32)
    for e in range(edge_index.shape[1]):
        r = int(edge_index[1, e])
        s = int(deg[r])
        src[r, s] = int(edge_index[0, e])
        sdir[r, s] = direction[e].float()
        sdist[r, s] = dists[e].float()
        deg[r] += 1
    return deg, src, sdir, sdist

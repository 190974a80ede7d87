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


def make_heavy_tailed(module, seed=5, df=3.0, boost=50.0, boosted_rows=3):
    """Replace the weights the low-precision operands of the message path are built from -- every layer's
    `conv.kernel.weight` and the two Linears of `basis_fn` -- by heavy-tailed draws (Student-t with `df` degrees of freedom,
    scaled to the standard deviation of the tensor they replace) and multiply a few output rows of every kernel by `boost`:
    the regime in which a 4-significand-bit operand (the e4m3 cross products, conv_proj.hip) or a shared block exponent
    (the Q16 stash) hurts first, which Gaussian / uniform initialisers never visit.  In place; returns the module.
    (|kernel weight| stays below 7, so the fp8 cross products remain selected: arreau_model.x8_ok.)"""
    g = torch.Generator().manual_seed(seed)
    t_dist = torch.distributions.StudentT(df)

    def redraw(w):
        std = float(w.std())
        torch.manual_seed(int(torch.randint(0, 2 ** 31, (1,), generator=g)))
        z = t_dist.sample(w.shape)
        return (z * (std / float(z.std()))).to(w.dtype)

    state = torch.random.get_rng_state()
    try:
        with torch.no_grad():
            net = module.model
            for lin in (net.basis_fn[1], net.basis_fn[3]):
                lin.weight.copy_(redraw(lin.weight))
            for layer in net.interaction_layers:
                w = redraw(layer.conv.kernel.weight)
                rows = torch.randperm(w.shape[0], generator=g)[:boosted_rows]
                w[rows] *= boost
                w.clamp_(-6.5, 6.5)
                layer.conv.kernel.weight.copy_(w)
    finally:
        torch.random.set_rng_state(state)
    return module

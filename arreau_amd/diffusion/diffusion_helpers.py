"""Host-side mirror of the reference's diffusion/diffusion_helpers.py for the sampling path:
schedule buffers with the same names/dtypes, and HIP-backed `frac_to_cart_coords` and
`radius_graph_pbc` with the reference's signatures and return conventions."""
import numpy as np
import torch
import torch.nn as nn

from .. import _hip


class GaussianFourierProjection(nn.Module):
    """Parameter holder (diffusion_helpers.py:14-21); evaluated inside the prep kernel."""

    def __init__(self, embedding_size=256, scale=1.0):
        super().__init__()
        self.gaussian_fourier_proj_w = nn.Parameter(torch.randn(embedding_size) * scale, requires_grad=False)


class VE_pbc(nn.Module):
    """Geometric sigma ladder, T+1 points (diffusion_helpers.py:33-41)."""

    def __init__(self, num_steps, sigma_min, sigma_max):
        super().__init__()
        self.T, self.sigma_min, self.sigma_max = num_steps, sigma_min, sigma_max
        self.register_buffer("sigmas", torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), self.T + 1)))


class VP_lattice(nn.Module):
    """Cosine schedule (diffusion_helpers.py:139-154).  Like the reference, `t` is float32 so
    alpha_bars is float32, while betas/sigmas take the default dtype through the zero they are
    concatenated with."""

    def __init__(self, num_steps=1000, s=0.0001, power=2, clipmax=0.999):
        super().__init__()
        t = torch.arange(0, num_steps + 1, dtype=torch.float)
        f_t = torch.cos((np.pi / 2) * ((t / num_steps) + s) / (1 + s)) ** power
        alpha_bars = f_t / f_t[0]
        betas = torch.cat([torch.zeros([1]), 1 - (alpha_bars[1:] / alpha_bars[:-1])], dim=0).clamp_max(clipmax)
        sigmas = torch.sqrt(betas[1:] * ((1 - alpha_bars[:-1]) / (1 - alpha_bars[1:])))
        self.register_buffer("alpha_bars", alpha_bars)
        self.register_buffer("betas", betas)
        self.register_buffer("sigmas", torch.cat([torch.zeros([1]), sigmas], dim=0))


def crystal_offsets(num_atoms: torch.Tensor, device) -> torch.Tensor:
    """num_atoms [B] -> CSR offsets [B+1] int32 on `device` (host-side index plumbing)."""
    n = torch.as_tensor(num_atoms).to("cpu", torch.int64)
    off = torch.zeros(n.numel() + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(n, 0)
    return off.to(device=device, dtype=torch.int32)


def frac_to_cart_coords(frac_coords: torch.Tensor, lattice: torch.Tensor, num_atoms: torch.Tensor) -> torch.Tensor:
    """diffusion_helpers.py:223-230 on the GPU."""
    _hip.require_gpu()
    frac = frac_coords.to(torch.float32).contiguous()
    lat = lattice.to(device=frac.device, dtype=torch.float32).contiguous()
    off = crystal_offsets(num_atoms, frac.device)
    out = torch.empty_like(frac)
    _hip.check(_hip.lib().arreau_frac_to_cart(_hip.ptr(frac), _hip.ptr(lat), _hip.ptr(off), lat.shape[0],
                                               frac.shape[0], _hip.ptr(out), _hip.stream_ptr(frac.device)),
               "arreau_frac_to_cart")
    return out


def radius_graph_pbc_slots(cart_coords, lattice, num_atoms, radius, max_num_neighbors_threshold):
    """Neighbour list in the library's receiver-major slot form.
    Returns (deg [N], src [N,k], cell [N,k], dir [N,k,3], dist [N,k], offsets [B+1])."""
    _hip.require_gpu()
    cart = cart_coords.to(torch.float32).contiguous()
    dev = cart.device
    lat = lattice.to(device=dev, dtype=torch.float32).contiguous()
    off = crystal_offsets(num_atoms, dev)
    N, k = cart.shape[0], int(max_num_neighbors_threshold)
    deg = torch.empty(N, device=dev, dtype=torch.int32)
    src = torch.empty((N, k), device=dev, dtype=torch.int32)
    cell = torch.empty((N, k), device=dev, dtype=torch.int32)
    direction = torch.empty((N, k, 3), device=dev, dtype=torch.float32)
    dist = torch.empty((N, k), device=dev, dtype=torch.float32)
    _hip.check(_hip.lib().arreau_radius_graph_pbc(
        _hip.ptr(cart), _hip.ptr(lat), _hip.ptr(off), lat.shape[0], N, float(radius), k, _hip.ptr(deg),
        _hip.ptr(src), _hip.ptr(cell), _hip.ptr(direction), _hip.ptr(dist), _hip.stream_ptr(dev)),
        "arreau_radius_graph_pbc")
    return deg, src, cell, direction, dist, off


def radius_graph_pbc(cart_coords, lattice, num_atoms, radius, max_num_neighbors_threshold, device=None,
                     topk_per_pair=None, remove_self_edges=True):
    """Same signature and return tuple as diffusion_helpers.py:328-337,548-555:
    (edge_index [2,E] int64 = (sender, receiver), -unit_cell [E,3], num_neighbors_image [B],
    atomic_distance [E], neighbor_direction [E,3]).  Exactly tied distances are resolved by
    (d^2, enumeration index) instead of the reference's unstable sort."""
    if topk_per_pair is not None or not remove_self_edges:
        raise NotImplementedError("the sampling path calls radius_graph_pbc with topk_per_pair=None, "
                                  "remove_self_edges=True (diffusion_loss.py:164-174)")
    deg, src, cell, direction, dist, off = radius_graph_pbc_slots(cart_coords, lattice, num_atoms, radius,
                                                                  max_num_neighbors_threshold)
    dev = deg.device
    N, k = src.shape
    eoff = torch.empty(N + 1, device=dev, dtype=torch.int32)
    edge_index = torch.empty((2, N * k), device=dev, dtype=torch.int64)
    cell_off = torch.empty((N * k, 3), device=dev, dtype=torch.float32)
    odist = torch.empty(N * k, device=dev, dtype=torch.float32)
    odir = torch.empty((N * k, 3), device=dev, dtype=torch.float32)
    _hip.check(_hip.lib().arreau_compact_edges(
        _hip.ptr(deg), _hip.ptr(src), _hip.ptr(cell), _hip.ptr(direction), _hip.ptr(dist), N, k, _hip.ptr(eoff),
        _hip.ptr(edge_index), _hip.ptr(cell_off), _hip.ptr(odist), _hip.ptr(odir), _hip.stream_ptr(dev)),
        "arreau_compact_edges")
    eoff_h = eoff.cpu()
    E = int(eoff_h[-1])
    off_h = off.cpu().long()
    per_crystal = (eoff_h[off_h[1:]] - eoff_h[off_h[:-1]]).to(device=dev, dtype=torch.int64)
    return edge_index[:, :E], cell_off[:E], per_crystal, odist[:E], odir[:E]


def sample_bravais_angles(lattice_type: str):
    """diffusion_helpers.py:739-774: angles in DEGREES (the sampler consumes them as radians)."""
    if lattice_type in ("cubic", "tetragonal", "orthorhombic"):
        return np.array([90, 90, 90])
    if lattice_type == "monoclinic":
        return np.array([90, np.random.uniform(90, 180), 90])
    if lattice_type == "triclinic":
        return np.array([np.random.uniform(60, 120) for _ in range(3)])
    if lattice_type == "hexagonal":
        return np.array([90, 90, 120])
    if lattice_type == "rhombohedral":
        angle = np.random.uniform(60, 120)
        return np.array([angle, angle, angle])
    raise ValueError(f"Invalid lattice type: {lattice_type}")

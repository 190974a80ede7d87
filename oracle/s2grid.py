"""Oracle: the S2 orientation grid generator (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math

import torch


def _to_euclid(g):
    """(beta, gamma) -> unit vector.  ponita/geometry/rotation.py:877-896."""
    b, c = g[..., 0], g[..., 1]
    return torch.stack([torch.sin(b) * torch.cos(c), torch.sin(b) * torch.sin(c), torch.cos(b)], dim=-1)


def uniform_grid_s2(n: int, steps: int = 100, step_size: float = 0.1, alpha: float = 0.001) -> torch.Tensor:
    """n roughly uniform unit vectors on S2, [n,3].

    ponita/geometry/rotation.py:947-1009 + repulsion.py:31-90: draw n normal
    vectors, normalise, convert to spherical angles (rotation.py:917-930); then
    ``steps`` plain-SGD steps (lr = step_size) on the mean Coulomb energy
    d^-2 of geodesic distances (acos of the clamped dot, eps 1e-7, :933-934;
    distances divided by pi), dropping each point's zero self-distance by
    sorting and skipping column 0; annealed Gaussian noise
    (steps-epoch)/steps*alpha is added to the gradient before the step.  Draw
    order from torch's global generator: randn(n,3), then randn(n,2) per step.
    """
    x = torch.randn((n, 3))
    x = x / torch.linalg.norm(x, dim=-1, keepdim=True)
    grid = torch.stack([torch.acos(x[..., 2]), torch.atan2(x[..., 1], x[..., 0])], dim=-1)
    grid.requires_grad_(True)
    for epoch in range(steps):
        grid.grad = None
        p = _to_euclid(grid)
        dots = (p[:, None] * p).sum(-1)
        dist = torch.acos(torch.clamp(dots, -1 + 1e-7, 1 - 1e-7)).sort(dim=-1)[0][:, 1:]
        energy = ((dist / math.pi) ** (-2)).mean()
        energy.backward()
        with torch.no_grad():
            grid.grad += (steps - epoch) / steps * alpha * torch.randn(grid.grad.shape)
            grid -= step_size * grid.grad
    return _to_euclid(grid.detach())

"""Orientation grid on S2 (host logic, runs once per model construction like the reference's
ponita/geometry/rotation.py:947-1009 + repulsion.py:31-90).  The reference never stores the grid
in the checkpoint (it is a plain attribute of a transform), so every load there gets a different
random grid; this build persists it as an optional extra checkpoint key (see checkpoint.py)."""
import math

import torch


def spherical_to_euclid(g: torch.Tensor) -> torch.Tensor:
    beta, gamma = g[..., 0], g[..., 1]
    return torch.stack([torch.sin(beta) * torch.cos(gamma), torch.sin(beta) * torch.sin(gamma), torch.cos(beta)], -1)


def uniform_grid_s2(n: int, steps: int = 100, step_size: float = 0.1, alpha: float = 0.001,
                    generator: torch.Generator = None) -> torch.Tensor:
    """n unit vectors [n,3] spread by Coulomb repulsion of geodesic distances: random start,
    `steps` SGD steps with annealed gradient noise.  Same procedure and draw order as the reference
    (randn(n,3) then randn(n,2) per step) when `generator` is None (global CPU generator)."""
    x = torch.randn((n, 3), generator=generator)
    x = x / torch.linalg.norm(x, dim=-1, keepdim=True)
    grid = torch.stack([torch.acos(x[..., 2]), torch.atan2(x[..., 1], x[..., 0])], dim=-1).requires_grad_(True)
    for epoch in range(steps):
        grid.grad = None
        p = spherical_to_euclid(grid)
        geo = torch.acos(torch.clamp((p[:, None] * p).sum(-1), -1 + 1e-7, 1 - 1e-7))
        geo = geo.sort(dim=-1)[0][:, 1:]  # drop each point's own (zero) distance
        ((geo / math.pi) ** (-2)).mean().backward()
        with torch.no_grad():
            noise = torch.randn(grid.grad.shape, generator=generator)
            grid -= step_size * (grid.grad + (steps - epoch) / steps * alpha * noise)
    return spherical_to_euclid(grid.detach())

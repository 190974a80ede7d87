"""Oracle: lattice algebra and the periodic-boundary radius graph (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import itertools

import torch

# 27 periodic images in the order the reference enumerates them
# (diffusion/diffusion_helpers.py:10): itertools.product((-1,0,1), repeat=3).
IMAGE_CELLS = list(itertools.product((-1, 0, 1), repeat=3))


def lattice_from_params(lengths: torch.Tensor, angles: torch.Tensor) -> torch.Tensor:
    """lengths [B,3], angles [B,3] (consumed as radians) -> cell matrix [B,3,3].

    Follows diffusion/lattice_helpers.py:55-105 (pymatgen convention, rows are
    the lattice vectors a, b, c; the arccos argument is clamped to [-1, 1] as
    abs_cap does at :38-51).
    """
    a, b, c = lengths[:, 0], lengths[:, 1], lengths[:, 2]
    ca, cb, cg = (torch.cos(angles[:, i]) for i in range(3))
    sa, sb = torch.sin(angles[:, 0]), torch.sin(angles[:, 1])
    gamma_star = torch.arccos(torch.clamp((ca * cb - cg) / (sa * sb), -1.0, 1.0))
    zero = torch.zeros_like(a)
    rows = [
        a * sb, zero, a * cb,
        -b * sa * torch.cos(gamma_star), b * sa * torch.sin(gamma_star), b * ca,
        zero, zero, c,
    ]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def matrix_to_params(matrix: torch.Tensor):
    """Cell matrix [B,3,3] -> (lengths [B,3], angles [B,3] in radians).

    Follows diffusion/lattice_helpers.py:16-35.  The reference allocates the
    angle tensor with torch.zeros (default dtype); we do the same.
    """
    lengths = torch.sqrt((matrix ** 2).sum(-1))
    angles = torch.zeros((matrix.shape[0], 3), device=matrix.device)
    for i in range(3):
        j, k = (i + 1) % 3, (i + 2) % 3
        cosv = (matrix[:, j, :] * matrix[:, k, :]).sum(-1) / (lengths[:, j] * lengths[:, k])
        angles[:, i] = torch.acos(torch.clamp(cosv, -1.0, 1.0))
    return lengths, angles


def frac_to_cart_coords(frac: torch.Tensor, lattice: torch.Tensor, num_atoms: torch.Tensor):
    """x_cart[n] = frac[n] @ lattice[crystal(n)].  diffusion/diffusion_helpers.py:223-230."""
    per_node = torch.repeat_interleave(lattice, num_atoms, dim=0)
    return torch.einsum("bi,bij->bj", frac, per_node)


def radius_graph_pbc(cart, lattice, num_atoms, radius, max_neighbors, remove_self_edges=True):
    """Periodic radius graph with a per-receiver neighbour cap.

    Restates diffusion/diffusion_helpers.py:328-564 (topk_per_pair=None path):

    * candidates are all ordered pairs (receiver i, sender j) inside one crystal
      times the 27 images, enumerated receiver-major, sender, image (:351-402);
    * direction = pos_j + image_offset - pos_i, d2 = |direction|^2 (:404-409),
      image_offset = lattice^T @ cell (:391-396);
    * keep 1e-4 < d2 <= radius^2 (:432-436);
    * if some receiver has more than ``max_neighbors`` candidates, keep per
      receiver the ``max_neighbors`` smallest d2 using a sort over a padded
      [N, max_count] matrix filled with radius^2+1 (:494-528).  The sort is
      torch's default (unstable) sort exactly as in the reference, so the
      choice among exactly tied d2 values is torch-build dependent;
    * surviving edges keep the enumeration order (masking preserves order).

    Returns (edge_index [2,E] = (sender, receiver), cell_offsets [E,3] (the
    reference returns the negated image cell, :551), per-crystal edge counts
    [B], dist [E], direction [E,3]).
    """
    dev, dt = cart.device, cart.dtype
    n_per = num_atoms.long()
    B = n_per.numel()
    N = cart.shape[0]
    n_sq = n_per * n_per
    first_atom = torch.cumsum(n_per, 0) - n_per

    # receiver / sender index of every ordered pair, receiver-major
    pair_crystal = torch.repeat_interleave(torch.arange(B, device=dev), n_sq)
    first_pair = torch.cumsum(n_sq, 0) - n_sq
    local = torch.arange(int(n_sq.sum()), device=dev) - first_pair[pair_crystal]
    n_of_pair = n_per[pair_crystal]
    recv = torch.div(local, n_of_pair, rounding_mode="trunc") + first_atom[pair_crystal]
    send = local % n_of_pair + first_atom[pair_crystal]

    cells = torch.tensor(IMAGE_CELLS, device=dev, dtype=dt)  # [27,3]
    # offsets[b, :, c] = lattice[b]^T @ cells[c]   (bmm as in :391-393)
    offsets = torch.bmm(lattice.transpose(1, 2), cells.t().unsqueeze(0).expand(B, -1, -1))
    off_pair = offsets[pair_crystal]  # [P,3,27]

    p_recv = cart[recv].unsqueeze(-1).expand(-1, -1, 27)
    p_send = cart[send].unsqueeze(-1).expand(-1, -1, 27) + off_pair
    direction = p_send - p_recv  # [P,3,27]
    d2 = (direction ** 2).sum(1).reshape(-1)  # [P*27]

    recv27 = recv.unsqueeze(1).expand(-1, 27).reshape(-1)
    send27 = send.unsqueeze(1).expand(-1, 27).reshape(-1)
    cell27 = cells.unsqueeze(0).expand(recv.numel(), -1, -1).reshape(-1, 3)

    keep = d2 <= radius * radius
    if remove_self_edges:
        keep = keep & (d2 > 0.0001)

    recv_k, send_k, cell_k = recv27[keep], send27[keep], cell27[keep]
    dir_k = direction.transpose(1, 2).reshape(-1, 3)[keep]
    d2_k = d2[keep]

    count = torch.zeros(N, device=dev, dtype=torch.long)
    count.index_add_(0, recv_k, torch.ones_like(recv_k))
    max_count = int(count.max()) if count.numel() else 0

    capped = count.clamp(max=max_neighbors) if max_neighbors > 0 else count
    cum_capped = torch.cat([capped.new_zeros(1), torch.cumsum(capped, 0)])
    cum_atoms = torch.cat([n_per.new_zeros(1), torch.cumsum(n_per, 0)])
    edges_per_crystal = cum_capped[cum_atoms[1:]] - cum_capped[cum_atoms[:-1]]

    if max_count > max_neighbors and max_neighbors > 0:
        pad = torch.full((N * max_count,), radius * radius + 1.0, device=dev, dtype=dt)
        first_edge = torch.cumsum(count, 0) - count
        slot = torch.arange(recv_k.numel(), device=dev) - first_edge[recv_k]
        pad[recv_k * max_count + slot] = d2_k
        sorted_d2, order = torch.sort(pad.view(N, max_count), dim=1)
        sorted_d2 = sorted_d2[:, :max_neighbors]
        order = order[:, :max_neighbors] + first_edge.view(-1, 1)
        chosen = order[sorted_d2 <= radius * radius]
        sel = torch.zeros(recv_k.numel(), device=dev, dtype=torch.bool)
        sel[chosen] = True
        recv_k, send_k, cell_k = recv_k[sel], send_k[sel], cell_k[sel]
        dir_k, d2_k = dir_k[sel], d2_k[sel]

    edge_index = torch.stack((send_k, recv_k))
    return edge_index, -cell_k, edges_per_crystal, torch.sqrt(d2_k), dir_k

"""Oracle: the Ponita fiber-bundle score network, functional form (CPU, torch).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

All functions take the reference's ``state_dict`` tensors (keys as written by
``PONITA_DIFFUSION``, prefix ``model.``) and materialise what the reference
materialises, in the same op order, so it doubles as the timed CPU baseline.
"""
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------
def polynomial_features(x: torch.Tensor, degree: int) -> torch.Tensor:
    """[..., d] -> [..., d + d^2 + ... + d^degree] by repeated outer products.

    ponita/nn/embedding.py:10-14: level k is outer(level k-1, x) flattened, so
    the 6-attribute, degree-3 case yields 6 + 36 + 216 = 258 columns with
    repeated monomials kept.
    """
    levels = [x]
    for _ in range(1, degree):
        levels.append((levels[-1].unsqueeze(-1) * x.unsqueeze(-2)).flatten(-2, -1))
    return torch.cat(levels, -1)


def polynomial_cutoff(d: torch.Tensor, r_max: float, p: float = 6.0) -> torch.Tensor:
    """Smooth envelope of ponita/utils/windowing.py:21-29, times (d < r_max)."""
    u = d / r_max
    env = (
        1.0
        - ((p + 1.0) * (p + 2.0) / 2.0) * torch.pow(u, p)
        + p * (p + 2.0) * torch.pow(u, p + 1)
        - (p * (p + 1.0) / 2) * torch.pow(u, p + 2)
    )
    return env * (d < r_max)


def scalar_to_sphere(scalar, ori_grid):
    """[N,c] -> [N,O,c] (replication).  ponita/utils/to_from_sphere.py:7-8."""
    return scalar.unsqueeze(-2).repeat_interleave(ori_grid.shape[-2], dim=-2)


def vec_to_sphere(vec, ori_grid):
    """[N,c,3] x [O,3] -> [N,O,c].  ponita/utils/to_from_sphere.py:4-5."""
    return torch.einsum("bcd,nd->bnc", vec, ori_grid)


def sphere_to_scalar(sig):
    """mean over the orientation axis.  ponita/utils/to_from_sphere.py:13-14."""
    return sig.mean(dim=-2)


def sphere_to_vec(sig, ori_grid):
    """[N,O,c] x [O,3] -> [N,c,3] / O.  ponita/utils/to_from_sphere.py:10-11."""
    return torch.einsum("bnc,nd->bcd", sig, ori_grid) / ori_grid.shape[-2]


def pair_invariants(direction, ori_grid):
    """Separable R3xS2 invariants, ponita/geometry/invariants.py:10-31.

    Returns ([E,O,2] = (dir.o, |dir - (dir.o) o|), [O,O,1] = o_a . o_b).
    """
    rel = direction[:, None, :]
    ga, gb = ori_grid[None, :, :], ori_grid[:, None, :]
    inv1 = (rel * ga).sum(-1, keepdim=True)
    inv2 = (rel - inv1 * ga).norm(dim=-1, keepdim=True)
    inv3 = (ga * gb).sum(-1, keepdim=True)
    return torch.cat([inv1, inv2], -1), inv3


def edge_attributes(direction, dists, lattice, batch_of_edge, ori_grid):
    """[E,O,6] attribute tensor and the [O,O,1] fiber attribute.

    ponita/transforms/invariants.py:69-88: the two pair invariants followed by
    (dist, cos(dir,a), cos(dir,b), cos(dir,c)) replicated over orientations;
    cosines use torch.nn.CosineSimilarity(dim=-1) (eps 1e-8, :26).
    """
    r3s2, fiber_attr = pair_invariants(direction, ori_grid)
    lat = torch.index_select(lattice, 0, batch_of_edge)
    cos = [F.cosine_similarity(direction, lat[:, i, :], dim=-1) for i in range(3)]
    scal = scalar_to_sphere(torch.stack([dists, cos[0], cos[1], cos[2]], dim=-1), ori_grid)
    return torch.cat([r3s2, scal], -1), fiber_attr, scal


def _basis_mlp(sd, prefix, attr, degree):
    """PolynomialFeatures -> Linear -> GELU -> Linear -> GELU (ponita.py:65-66)."""
    h = polynomial_features(attr, degree)
    h = F.gelu(F.linear(h, sd[prefix + ".1.weight"], sd[prefix + ".1.bias"]))
    return F.gelu(F.linear(h, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"]))


def fiber_bundle_conv(sd, prefix, x, edge_index, kernel_basis, fiber_kernel_basis, stats=None):
    """Separable depth-wise conv: spatial message passing then spherical conv.

    ponita/nn/conv.py:105-129 with message :131-133 and the PyG propagate
    semantics of :234-262: x_j = x[edge_index[0]] (sender), sum-aggregate onto
    edge_index[1] (receiver) with dim_size = N.
    """
    kernel = F.linear(kernel_basis, sd[prefix + ".kernel.weight"])  # [E,O,C]
    messages = kernel * x[edge_index[0]]
    x_1 = torch.zeros_like(x).index_add_(0, edge_index[1], messages)
    fiber_kernel = F.linear(fiber_kernel_basis, sd[prefix + ".fiber_kernel.weight"])  # [O,O,C]
    x_2 = torch.einsum("boc,opc->bpc", x_1, fiber_kernel) / fiber_kernel.shape[-2]
    if stats is not None:  # what FiberBundleConv.callibrate reads (conv.py:121-123): x.std(), x_1.std(), x_2.std()
        stats.append((x, x_1, x_2))
    return x_2 + sd[prefix + ".bias"], messages


def convnext_block(sd, prefix, x, conv_out):
    """LayerNorm -> Linear C->WC -> GELU -> Linear WC->C -> layer_scale -> +input.

    ponita/nn/convnext.py:20-33 (LayerNorm eps 1e-5 = torch default).
    """
    C = x.shape[-1]
    h = F.layer_norm(conv_out, (C,), sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"])
    h = F.gelu(F.linear(h, sd[prefix + ".linear_1.weight"], sd[prefix + ".linear_1.bias"]))
    h = F.linear(h, sd[prefix + ".linear_2.weight"], sd[prefix + ".linear_2.bias"])
    ls = sd.get(prefix + ".layer_scale")
    if ls is not None:
        h = ls * h
    return h + x


def ponita_forward(sd, hp, x, vec, edge_index, dists, direction, lattice, batch, batch_of_edge,
                   ori_grid, return_internals=False):
    """PonitaFiberBundle.forward, ponita/models/ponita.py:88-123.

    sd : state_dict slice with the ``model.`` prefix stripped.
    hp : dict(num_layers, degree, radius, S) -- S = scalar output width.
    x [N,S+74], vec [N,4,3]; returns (logits [N,S], vec_out [N,1,3],
    global_scalar [B,3]).
    """
    L, degree, S = hp["num_layers"], hp["degree"], hp["S"]
    # lift (ponita/transforms/position_orientation_graph.py:82-86)
    xs = torch.cat([scalar_to_sphere(x, ori_grid), vec_to_sphere(vec, ori_grid)], dim=-1)
    attr, fiber_attr, _ = edge_attributes(direction, dists, lattice, batch_of_edge, ori_grid)

    kernel_basis = _basis_mlp(sd, "basis_fn", attr, degree) * polynomial_cutoff(
        dists, hp["radius"]).unsqueeze(-1).unsqueeze(-1)
    fiber_kernel_basis = _basis_mlp(sd, "fiber_basis_fn", fiber_attr, degree)

    h = F.linear(xs, sd["x_embedder.weight"])
    internals = {"attr": attr, "kernel_basis": kernel_basis, "x0": h, "x": [], "conv_stats": []}
    readouts = []
    for i in range(L):
        pre = f"interaction_layers.{i}"
        conv_out, _messages = fiber_bundle_conv(sd, pre + ".conv", h, edge_index, kernel_basis,
                                                fiber_kernel_basis, stats=internals["conv_stats"])
        h = convnext_block(sd, pre, h, conv_out)
        internals["x"].append(h)
        readouts.append(F.linear(h, sd[f"read_out_layers.{i}.weight"], sd[f"read_out_layers.{i}.bias"]))
        # edge_readout_layers have zero output width on this path (ponita.py:83,106): nothing to compute
    readout = sum(readouts) / len(readouts)
    r_scalar, r_vec, _r_gvec, r_gscalar = torch.split(readout, [S, 1, 0, 3], dim=-1)
    logits = sphere_to_scalar(r_scalar)
    vec_out = sphere_to_vec(r_vec, ori_grid)
    gs_nodes = sphere_to_scalar(r_gscalar)
    B = int(batch.max()) + 1
    global_scalar = torch.zeros(B, 3, dtype=gs_nodes.dtype).index_add_(0, batch, gs_nodes)
    if return_internals:
        return logits, vec_out, global_scalar, internals
    return logits, vec_out, global_scalar

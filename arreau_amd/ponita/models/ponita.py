"""PonitaFiberBundle: parameter container with the reference's state_dict layout
(ponita/models/ponita.py:31-86); the forward pass runs in libarreau_hip.so.

Keys produced (prefix `model.` inside PONITA_DIFFUSION): basis_fn.{1,3}.{weight,bias},
fiber_basis_fn.{1,3}.{weight,bias}, windowing_fn.{p,r_max}, x_embedder.weight,
interaction_layers.{i}.conv.{kernel.weight,fiber_kernel.weight,bias,callibrated},
interaction_layers.{i}.{linear_1,linear_2,norm}.{weight,bias}, interaction_layers.{i}.layer_scale,
read_out_layers.{i}.{weight,bias}, edge_readout_layers.{i}.{weight,bias} (zero-width).
"""
import torch
import torch.nn as nn

from ..geometry.rotation import uniform_grid_s2

NUM_EDGE_ATTR = 6  # (inv1, inv2, dist, cos a, cos b, cos c), ponita/transforms/invariants.py:85-88


def _poly_width(d: int, degree: int) -> int:
    return sum(d ** k for k in range(1, degree + 1))  # PolynomialFeatures, ponita/nn/embedding.py:10-14


class _Placeholder(nn.Module):
    """Parameter-free stage of the reference's nn.Sequential (PolynomialFeatures / GELU): keeps
    the Sequential indices 1 and 3 for the two Linear layers."""


class PolynomialCutoff(nn.Module):
    """Buffers only (ponita/utils/windowing.py:14-17); the envelope is evaluated in the edge kernel."""

    def __init__(self, r_max, p=6):
        super().__init__()
        self.register_buffer("p", torch.tensor(p, dtype=torch.get_default_dtype()))
        self.register_buffer("r_max", torch.tensor(r_max, dtype=torch.get_default_dtype()))


class FiberBundleConv(nn.Module):
    """ponita/nn/conv.py:71-103 (separable, depth-wise): parameters only."""

    def __init__(self, channels, attr_dim):
        super().__init__()
        self.kernel = nn.Linear(attr_dim, channels, bias=False)
        self.fiber_kernel = nn.Linear(attr_dim, channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("callibrated", torch.tensor(False))


class ConvNext(nn.Module):
    """ponita/nn/convnext.py:4-18: parameters only."""

    def __init__(self, channels, conv, layer_scale=1e-6, widening_factor=4):
        super().__init__()
        self.conv = conv
        self.linear_1 = nn.Linear(channels, widening_factor * channels)
        self.linear_2 = nn.Linear(widening_factor * channels, channels)
        if layer_scale is not None:
            self.layer_scale = nn.Parameter(torch.ones(channels) * layer_scale)
        else:
            self.register_buffer("layer_scale", None)
        self.norm = nn.LayerNorm(channels)


class PonitaFiberBundle(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, output_dim_global_scalar, output_dim_global_vec,
                 output_dim_edge_scalar, num_layers, output_dim_vec=0, radius=None, num_ori=20, basis_dim=None,
                 degree=3, widening_factor=4, layer_scale=None, multiple_readouts=True, ori_grid=None, **kwargs):
        super().__init__()
        if not multiple_readouts:
            raise NotImplementedError("arreau_amd implements multiple_readouts=True (the diffusion default)")
        if output_dim_global_vec != 0 or output_dim_edge_scalar != 0:
            raise NotImplementedError("global-vector / edge read-outs are zero-width on the diffusion path")
        self.output_dim, self.output_dim_vec = output_dim, output_dim_vec
        self.output_dim_global_scalar = output_dim_global_scalar
        self.hidden_dim, self.num_layers, self.num_ori = hidden_dim, num_layers, num_ori
        self.degree, self.widening_factor, self.radius = degree, widening_factor, radius
        basis_dim = hidden_dim if basis_dim is None else basis_dim
        self.basis_dim = basis_dim
        # the S2 grid the reference builds inside PositionOrientationGraph (position_orientation_graph.py:29-32)
        self.ori_grid = uniform_grid_s2(num_ori) if ori_grid is None else torch.as_tensor(ori_grid).clone()

        def basis(in_width):
            return nn.Sequential(_Placeholder(), nn.Linear(in_width, hidden_dim), _Placeholder(),
                                 nn.Linear(hidden_dim, basis_dim), _Placeholder())

        self.basis_fn = basis(_poly_width(NUM_EDGE_ATTR, degree))
        self.fiber_basis_fn = basis(_poly_width(1, degree))
        self.windowing_fn = PolynomialCutoff(radius)
        self.x_embedder = nn.Linear(input_dim, hidden_dim, False)
        self.interaction_layers = nn.ModuleList()
        self.read_out_layers = nn.ModuleList()
        self.edge_readout_layers = nn.ModuleList()
        n_out = output_dim + output_dim_vec + output_dim_global_scalar + output_dim_global_vec
        for _ in range(num_layers):
            conv = FiberBundleConv(hidden_dim, basis_dim)
            self.interaction_layers.append(ConvNext(hidden_dim, conv, layer_scale=layer_scale,
                                                    widening_factor=widening_factor))
            self.read_out_layers.append(nn.Linear(hidden_dim, n_out))
            self.edge_readout_layers.append(nn.Linear(hidden_dim + 4, output_dim_edge_scalar))

    def forward(self, graph):
        raise RuntimeError("PonitaFiberBundle.forward is evaluated by PONITA_DIFFUSION.forward (HIP engine)")

"""lattice_from_params on the GPU (reference: diffusion/lattice_helpers.py:55-105)."""
import torch

from .. import _hip


def lattice_from_params(lengths: torch.Tensor, angles: torch.Tensor) -> torch.Tensor:
    """lengths [B,3], angles [B,3] (consumed as radians) on a cuda device -> [B,3,3] fp32."""
    _hip.require_gpu()
    lengths = lengths.to(torch.float32).contiguous()
    angles = angles.to(device=lengths.device, dtype=torch.float32).contiguous()
    B = lengths.shape[0]
    out = torch.empty((B, 3, 3), device=lengths.device, dtype=torch.float32)
    _hip.check(_hip.lib().arreau_lattice_from_params(_hip.ptr(lengths), _hip.ptr(angles), B, _hip.ptr(out),
                                                      _hip.stream_ptr(lengths.device)), "arreau_lattice_from_params")
    return out

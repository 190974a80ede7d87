"""ctypes binding of libarreau_hip.so (the C ABI declared in include/arreau_hip.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.
torch is used only for device memory and streams.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_void_p, POINTER, Structure

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARREAU_HIP_LIB: load another build of the same library (A/B timing of two kernel versions on one box)
LIB_PATH = os.environ.get("ARREAU_HIP_LIB") or os.path.join(_HERE, "csrc", "libarreau_hip.so")

EXPORTS = [
    "arreau_last_error", "arreau_version", "arreau_model_create", "arreau_model_destroy",
    "arreau_model_config", "arreau_workspace_bytes", "arreau_lattice_from_params", "arreau_frac_to_cart",
    "arreau_radius_graph_pbc", "arreau_compact_edges", "arreau_edges_to_slots", "arreau_predict_scores",
    "arreau_reverse_step", "arreau_profile_edge_kernel", "arreau_edge_kernel_time_ms", "arreau_conv_kernel_time_ms",
    "arreau_model_status", "arreau_model_set_variant", "arreau_ponita_forward",
    "arreau_diffusion_noise", "arreau_diffusion_losses", "arreau_sample_loop", "arreau_philox_fill",
    "arreau_train_forward", "arreau_train_backward", "arreau_train_conv_stats", "arreau_model_update_train_weights",
    "arreau_debug_sgemm", "arreau_optimizer_create", "arreau_optimizer_step", "arreau_optimizer_destroy",
    "arreau_model_train_weight_pointers", "arreau_model_refresh_derived_train_weights",
    "arreau_model_set_batch_layout", "arreau_model_set_formats", "arreau_debug_set_pollution", "arreau_debug_leftover_fraction",
]

STATUS_NONFINITE, STATUS_BAD_TIMESTEP, STATUS_BAD_TYPE = 1, 2, 4
EDGE_KERNELS = {0: "fp32-mfma", 1: "fp32-mfma", 2: "fp32-mfma", 3: "bf16x6", 4: "fp16x3", 5: "general-fp32-gemm"}
MLP_KERNELS = {0: "fp32-mfma", 1: "bf16x6", 2: "fp16x3-32x32x16", 3: "fp16x3-16x16x32",
               5: "general-fp32-gemm"}
VARIANT_GENERAL = 5  # edge variant selecting the shape-general fp32 network (any hidden_dim / basis_dim / widening)


class ArreauHipError(RuntimeError):
    pass


OPT_MAX_GROUPS = 4


class AdamArgs(Structure):
    """arreau_adam_args (include/arreau_hip.h)"""
    _fields_ = [("step", c_int64), ("lr", c_double * OPT_MAX_GROUPS), ("weight_decay", c_double * OPT_MAX_GROUPS),
                ("beta1", c_double), ("beta2", c_double), ("eps", c_double), ("max_norm", c_double)]


class Config(Structure):
    _fields_ = [
        ("num_atomic_states", c_int32), ("hidden_dim", c_int32), ("basis_dim", c_int32),
        ("num_layers", c_int32), ("num_ori", c_int32), ("widening_factor", c_int32), ("degree", c_int32),
        ("max_neighbors", c_int32), ("num_timesteps", c_int32), ("radius", c_float),
        ("has_layer_scale", c_int32),
    ]


class Status(Structure):
    _fields_ = [("flags", c_int32), ("edge_kernel", c_int32), ("mlp_kernel", c_int32), ("conv_kernel", c_int32),
                ("basis_row_bytes", c_int32), ("conv_cross_fp8", c_int32), ("edge_activation_bound", c_float), ("node_activation_bound", c_float),
                ("basis_fp8_share", c_float), ("cross_fp8_share", c_float)]


_SD_FIELDS = [
    "basis_w1", "basis_b1", "basis_w2", "basis_b2", "fiber_w1", "fiber_b1", "fiber_w2", "fiber_b2",
    "x_embedder_w", "conv_kernel_w", "conv_fiber_w", "conv_bias", "norm_w", "norm_b", "linear1_w",
    "linear1_b", "linear2_w", "linear2_b", "layer_scale", "readout_w", "readout_b", "ori_grid", "t_emb_w",
    "ve_sigmas", "vp_alpha_bars", "vp_betas", "q_one_step_transposed", "q_mats",
]


class StateDict(Structure):
    _fields_ = [(name, c_void_p) for name in _SD_FIELDS]


_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArreauHipError(
            f"{LIB_PATH} not found: build it with `python -m arreau_amd.build` (hipcc, gfx950). "
            "arreau_amd has no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    L.arreau_last_error.restype = c_char_p
    L.arreau_version.restype = c_char_p
    L.arreau_model_create.argtypes = [POINTER(Config), POINTER(StateDict), c_void_p, POINTER(c_void_p)]
    L.arreau_model_destroy.argtypes = [c_void_p]
    L.arreau_model_destroy.restype = None
    L.arreau_model_config.argtypes = [c_void_p, POINTER(Config)]
    L.arreau_workspace_bytes.argtypes = [POINTER(Config), c_int64, c_int64]
    L.arreau_workspace_bytes.restype = c_size_t
    L.arreau_lattice_from_params.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]
    L.arreau_frac_to_cart.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]
    L.arreau_radius_graph_pbc.argtypes = [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_int32,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    L.arreau_compact_edges.argtypes = [c_void_p] * 5 + [c_int32, c_int32] + [c_void_p] * 6
    L.arreau_edges_to_slots.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32] + [c_void_p] * 6
    L.arreau_predict_scores.argtypes = ([c_void_p] * 7 + [c_int32, c_int32, c_int32] + [c_void_p] * 7 +
                                        [c_void_p, c_size_t, c_void_p])
    L.arreau_reverse_step.argtypes = [c_void_p] * 7 + [c_int32, c_int32] + [c_void_p] * 8
    L.arreau_model_status.argtypes = [c_void_p, POINTER(Status), c_int32, c_void_p]
    L.arreau_model_set_variant.argtypes = [c_void_p, c_int32, c_int32]
    L.arreau_ponita_forward.argtypes = ([c_void_p] * 5 + [c_int32, c_int32] + [c_void_p] * 7 +
                                        [c_void_p, c_size_t, c_void_p])
    L.arreau_diffusion_noise.argtypes = [c_void_p] * 6 + [c_int32, c_int32] + [c_void_p] * 11
    L.arreau_diffusion_losses.argtypes = [c_void_p] * 10 + [c_int32, c_int32] + [c_void_p] * 6
    L.arreau_sample_loop.argtypes = ([c_void_p] * 6 + [c_int32, c_int32, c_int32, c_int32, ctypes.c_uint64] +
                                     [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p])
    L.arreau_philox_fill.argtypes = [ctypes.c_uint64, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p]
    L.arreau_train_forward.argtypes = [c_void_p] * 7 + [c_int32, c_int32] + [c_void_p] * 4
    L.arreau_train_backward.argtypes = [c_void_p] * 4 + [POINTER(StateDict), c_void_p]
    L.arreau_train_conv_stats.argtypes = [c_void_p, c_void_p, c_void_p]
    L.arreau_debug_sgemm.argtypes = [c_int32, c_int32, c_int32, c_int32, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int32,
                                     c_float, c_float, c_void_p]
    L.arreau_model_update_train_weights.argtypes = [c_void_p, POINTER(StateDict), c_void_p]
    L.arreau_model_set_batch_layout.argtypes = [c_void_p, c_void_p, c_int32, c_int32]
    if hasattr(L, "arreau_model_set_formats") or not os.environ.get("ARREAU_HIP_LIB"):  # (an older build under test: tools/ab.sh)
        L.arreau_model_set_formats.argtypes = [c_void_p, c_int32, c_int32]
    if hasattr(L, "arreau_optimizer_create") or not os.environ.get("ARREAU_HIP_LIB"):
        L.arreau_optimizer_create.argtypes = [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64), POINTER(c_int32),
                                              c_int32, c_int64, POINTER(c_void_p)]
        L.arreau_model_train_weight_pointers.argtypes = [c_void_p, POINTER(StateDict)]
        L.arreau_model_refresh_derived_train_weights.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
        L.arreau_optimizer_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(AdamArgs), c_void_p, c_void_p]
        L.arreau_optimizer_destroy.argtypes = [c_void_p]
        L.arreau_optimizer_destroy.restype = None
    L.arreau_debug_set_pollution.argtypes = [ctypes.c_uint32]
    L.arreau_debug_leftover_fraction.argtypes = [ctypes.c_uint32, POINTER(c_double), POINTER(c_double), c_void_p]
    L.arreau_profile_edge_kernel.argtypes = [c_int32]
    L.arreau_edge_kernel_time_ms.argtypes = [POINTER(c_double), POINTER(c_int64)]
    L.arreau_conv_kernel_time_ms.argtypes = [POINTER(c_double), POINTER(c_int64)]
    for name in EXPORTS:
        if os.environ.get("ARREAU_HIP_LIB") and not hasattr(L, name):
            continue
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int:
            fn.restype = c_int32
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().arreau_last_error().decode(errors="replace")
        raise ArreauHipError(f"{what} failed (code {rc}): {msg}")


def ptr(t):
    """Device (or host) address of a contiguous tensor, or NULL for None."""
    if t is None:
        return None
    assert t.is_contiguous(), "arreau_amd passes contiguous tensors only"
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu():
    if not torch.cuda.is_available():
        raise ArreauHipError("arreau_amd needs an AMD GPU (gfx950); there is no CPU fallback.")

"""The dense product of the training step (csrc/sgemm.h) called through the C ABI (arreau_debug_sgemm): every operand layout the
step uses -- X W^T (both operands contiguous along k), dY W (the weight read across its rows), dY^T X (both operands of a weight
gradient) and X^T W^T -- in its three arithmetics (exact fp32 MFMA, fp16x3, bf16x6), at ragged sizes (rows, columns and K that are
no multiples of the 64 / 128 tiles or of the 32-wide k-step), with split-K (a long reduction over few tiles) and with alpha / beta,
against torch's fp64 product.  Reference for the role: torch.nn.functional.linear + autograd's matmuls behind ponita.py:65-66,
conv.py:110-116, convnext.py:24-30."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# relative to the largest entry of the exact product; measured (tools/exp/sgemm_bench.hip, profiles/r04_sgemm_bench.txt): exact fp32
# 4e-7 .. 2e-6, fp16x3 1.4e-7 .. 9e-7, bf16x6 3e-7 .. 1.7e-6 from K = 96 to K = 68,096
TOL = 2e-6


def _operand(rows, cols, contiguous_along_cols, gen, dev):
    """a [rows, cols] matrix whose memory is row-major (contiguous along cols) or column-major; returns (tensor view, stride0, stride1)"""
    if contiguous_along_cols:
        t = torch.rand((rows, cols), generator=gen, dtype=torch.float32).sub_(0.5).to(dev)
        return t, cols, 1
    t = torch.rand((cols, rows), generator=gen, dtype=torch.float32).sub_(0.5).to(dev)
    return t.t(), 1, rows


def _run(mode, M, N, K, a_k, b_k, alpha=1.0, beta=0.0, seed=0, a_scale=1.0):
    from arreau_amd import _hip, build
    build.build(verbose=False)
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(seed)
    A, as0, as1 = _operand(M, K, a_k, gen, dev)          # A(m, k)
    A.mul_(a_scale)
    B, bs0, bs1 = _operand(K, N, not b_k, gen, dev)      # B(k, n): contiguous along k = column-major storage
    C0 = torch.rand((M, N), generator=gen, dtype=torch.float32).to(dev)
    C = C0.clone()
    base_a = A if a_k else A.t()
    base_b = B if not b_k else B.t()
    assert base_a.is_contiguous() and base_b.is_contiguous()
    _hip.check(_hip.lib().arreau_debug_sgemm(mode, M, N, K, _hip.ptr(base_a), as0, as1, _hip.ptr(base_b), bs0, bs1, _hip.ptr(C), N, alpha, beta,
                                              _hip.stream_ptr(dev)), "arreau_debug_sgemm")
    ref = alpha * (A.double() @ B.double()) + beta * C0.double()
    err = float((C.double() - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
    return err


LAYOUTS = {"X W^T (k, k)": (True, True), "dY W (k, n)": (True, False), "dY^T X (m, n)": (False, False), "X^T W^T (m, k)": (False, True)}


@pytest.mark.parametrize("layout", list(LAYOUTS))
@pytest.mark.parametrize("mode", [0, 1, 2], ids=["exact fp32", "fp16x3", "bf16x6"])
def test_product_in_every_layout_at_ragged_sizes(layout, mode):
    a_k, b_k = LAYOUTS[layout]
    worst = 0.0
    # (M, N, K): 64 x 64 tiles with ragged edges; 128 x 128 tiles (>= 512 of them); K below the split kernel's threshold; K % 32 != 0
    for M, N, K in ((333, 260, 196), (8501, 132, 260), (9000, 900, 132), (500, 132, 40), (257, 129, 1028)):
        worst = max(worst, _run(mode, M, N, K, a_k, b_k, seed=M + K))
    print(f"\n[sgemm {layout}, mode {mode}] worst error relative to the largest entry: {worst:.2e}")
    assert worst <= TOL


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["exact fp32", "fp16x3", "bf16x6"])
def test_weight_gradient_shapes_take_split_k_and_stay_deterministic(mode):
    """dW = dY^T X with K = all rows: a handful of output tiles under a long reduction (k-slices + the ordered reduce kernel)."""
    errs = [_run(mode, 640, 256, 20000, False, False, seed=3), _run(mode, 96, 128, 8501, False, False, seed=4)]
    assert max(errs) <= TOL
    assert _run(mode, 640, 256, 20000, False, False, seed=3) == errs[0]   # same sums in the same order


def test_alpha_and_beta():
    for mode in (0, 1, 2):
        assert _run(mode, 300, 200, 256, True, True, alpha=0.2, beta=1.0) <= TOL
        assert _run(mode, 300, 200, 256, True, False, alpha=-1.5, beta=0.5) <= TOL


def test_bf16x6_keeps_its_precision_on_gradient_sized_operands():
    """Products with a gradient operand run as bf16x6 because gradients come in any magnitude (1e-9 behind layer_scale = 1e-6): three
    8-bit pieces of an fp32 value keep its full exponent range, where an fp16 plane would go subnormal."""
    for a_k, b_k in ((True, False), (False, False)):
        assert _run(2, 2000, 256, 640, a_k, b_k, a_scale=1e-9) <= TOL
        assert _run(0, 2000, 256, 640, a_k, b_k, a_scale=1e-9) <= TOL

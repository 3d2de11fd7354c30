"""The gated project GEMMs of the squeeze-excite blocks (kernels_conv.hip launch_pw_gemm16_gated) on their own, through
bh_debug_gated_gemm: every kernel behind the dispatch -- streaming (N <= 48; the two-workgroup variant for K <= 32), row-streaming
(N = 64 .. 240, D in NHWC and blocked), 128 x 128 staged tiles -- on shapes no model of the suite has (K % 32 != 0, N % 16 != 0, rows
that are not whole 16-row tiles, tiles that straddle segments), against float64, and against each other: a row's bits do not depend
on the kernel its launch happened to take."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lib, A, gate, W, bias, R, P, terms, blocked):
    M, K = A.shape
    N = W.shape[1]
    out = np.empty((M, N), np.float32)
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.bh_debug_gated_gemm(0, ptr(A), ptr(gate), ptr(W), ptr(bias), ptr(R), ptr(out), M, K, N, P, terms, blocked)
    assert rc == 0, rc
    return out


def _operands(seed, n_seg, P, K, N, residual):
    rng = np.random.default_rng(seed)
    M = n_seg * P
    A = (rng.standard_normal((M, K)) * rng.uniform(0.2, 3.0, (1, K))).astype(np.float32)     # channels of different scale
    gate = rng.uniform(0.0, 1.0, (n_seg, K)).astype(np.float32)                                 # sigmoid outputs
    W = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32) if residual else None
    return A, gate, W, bias, R


def _reference(A, gate, W, bias, R, P):
    Ag = A.astype(np.float64) * np.repeat(gate.astype(np.float64), P, axis=0)
    ref = Ag @ W.astype(np.float64) + bias.astype(np.float64)
    bound = np.abs(Ag) @ np.abs(W.astype(np.float64)) + np.abs(bias)       # what the rounding errors scale with
    if R is not None:
        ref += R; bound += np.abs(R)
    return ref, bound


# (n_seg, rows per segment, K, N, residual, blocked) -- M = n_seg * P >= 4 096 reaches the streaming kernels
SHAPES = [
    (5, 1008, 24, 24, True, 0),      # streaming kernel, one k step: two workgroups a CU (the 24 -> 24 block's shape)
    (5, 1000, 40, 24, False, 0),     # ... K % 32 != 0, rows not whole tiles (M % 16 = 8), tiles straddling segments
    (14, 300, 100, 44, True, 0),     # ... three column tiles, N % 16 != 0
    (33, 128, 72, 64, True, 0),      # row-streaming kernel, N = 64
    (65, 64, 304, 100, True, 1),     # ... seven column tiles over blocked rows, K % 32 != 0
    (17, 256, 816, 136, True, 1),    # ... the 8x32 stage's shape, blocked
    (17, 256, 816, 136, False, 0),   # ... and in NHWC
    (640, 64, 336, 232, True, 1),    # ... fifteen column tiles (the 4x16 stage's width; N > 144 streams from 40 960 rows up), blocked
    (641, 64, 176, 232, True, 0),
    (64, 64, 336, 232, True, 1),     # the same width below 40 960 rows: staged tiles
    (83, 50, 60, 84, True, 0),       # ... partial last tile, tiles straddling segments, gate rows of several segments a pass
    (64, 64, 128, 384, True, 0),     # 128 x 128 staged tiles (N = 384)
    (3, 64, 304, 100, True, 1),      # fewer than 4 096 rows: staged tiles over blocked rows
    (256, 16, 1392, 232, False, 1),  # 16-row segments: the gate rows of a pass no longer fit LDS beside W -> staged tiles
]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "n%d_P%d_K%d_N%d_r%d_b%d" % s)
def test_gated_gemm_matches_float64(shape):
    from birda_amd import _lib
    lib = _lib.load()
    n_seg, P, K, N, residual, blocked = shape
    A, gate, W, bias, R = _operands(K * 131 + N, n_seg, P, K, N, residual)
    ref, bound = _reference(A, gate, W, bias, R, P)
    for terms, tol in ((3, 4e-7), (1, 1.5e-3)):     # split f16 (three products a MAC): ~2^-22 of sum |a||w|; plain f16: ~2^-10
        got = _run(lib, A, gate, W, bias, R, P, terms, blocked)
        err = np.abs(got - ref) / bound
        assert np.isfinite(got).all() and err.max() <= tol, (shape, terms, float(err.max()))


@pytest.mark.parametrize("shape", [(650, 64, 336, 232, True), (20, 256, 816, 136, True), (6, 1008, 24, 24, True), (6, 1008, 192, 32, True),
                                   (90, 64, 128, 64, False)], ids=lambda s: "n%d_P%d_K%d_N%d_r%d" % s)
def test_a_rows_bits_do_not_depend_on_the_kernel(shape):
    """The first segments of a large launch (streaming / row-streaming kernels) against the same segments as a launch of their own
    (fewer than 4 096 rows: staged tiles), D in NHWC and -- where the block would use it -- blocked."""
    from birda_amd import _lib
    lib = _lib.load()
    n_seg, P, K, N, residual = shape
    A, gate, W, bias, R = _operands(7 * K + N, n_seg, P, K, N, residual)
    few = max(1, 2048 // P)
    assert few * P < 4096 <= n_seg * P
    for blocked in ((0, 1) if (K % 16 == 0 and P % 16 == 0 and 6 <= -(-N // 16) <= 15) else (0,)):
        for terms in (3, 1):
            big = _run(lib, A, gate, W, bias, R, P, terms, blocked)
            small = _run(lib, A[:few * P].copy(), gate[:few].copy(), W, bias, None if R is None else R[:few * P].copy(), P, terms, blocked)
            assert np.array_equal(big[:few * P], small), (shape, blocked, terms)
    if K % 16 == 0 and P % 16 == 0:      # and the two layouts of D give the same bits
        assert np.array_equal(_run(lib, A, gate, W, bias, R, P, 3, 0), _run(lib, A, gate, W, bias, R, P, 3, 1))


def test_gated_gemm_refuses_what_it_cannot_take():
    from birda_amd import _lib
    lib = _lib.load()
    A, gate, W, bias, R = _operands(1, 4, 16, 18, 8, False)      # K % 4 != 0
    out = np.empty((64, 8), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.bh_debug_gated_gemm(0, p(A), p(gate), p(W), p(bias), None, p(out), 64, 18, 8, 16, 3, 0) != 0
    A, gate, W, bias, R = _operands(1, 4, 16, 24, 8, False)      # blocked rows need K % 16 == 0
    assert lib.bh_debug_gated_gemm(0, p(A), p(gate), p(W), p(bias), None, p(out), 64, 24, 8, 16, 3, 1) != 0

"""GPU unit tests of the dense device kernels through the C-ABI test hooks (fp64, compared
with NumPy on the same inputs; tolerance = a few ulps of the accumulated magnitude)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gemm(lib, pkg, M, N, K, ta, tb, tri=0, lower=0, alpha=1.0, beta=0.0, seed=0):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((K, M) if ta else (M, K))
    B = rng.standard_normal((N, K) if tb else (K, N))
    Cm = rng.standard_normal((M, N))
    opA = A.T if ta else A
    opB = B.T if tb else B
    if tri & 1: opA = np.tril(opA); A = opA.T.copy() if ta else opA.copy()
    if tri & 2: opA = np.triu(opA); A = opA.T.copy() if ta else opA.copy()
    if tri & 4: opB = np.tril(opB); B = opB.T.copy() if tb else opB.copy()
    if tri & 8: opB = np.triu(opB); B = opB.T.copy() if tb else opB.copy()
    ref = alpha * (opA @ opB) + beta * Cm
    out = Cm.copy()
    A = np.ascontiguousarray(A); B = np.ascontiguousarray(B)
    st = lib.gmrf_test_gemm(0, M, N, K, int(ta), int(tb), tri, lower, alpha, pkg._cabi.ptr(A), A.shape[1],
                            pkg._cabi.ptr(B), B.shape[1], beta, pkg._cabi.ptr(out), N)
    pkg._cabi.check(st)
    return out, ref, Cm


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_all_layouts(lib, pkg, ta, tb):
    out, ref, _ = _gemm(lib, pkg, 128, 192, 80, ta, tb, alpha=-0.5, beta=1.0, seed=ta * 2 + tb)
    assert np.max(np.abs(out - ref)) < 1e-12 * 80


def test_gemm_asymmetric_identity(lib, pkg):
    # A = I with an asymmetric B catches a swapped C/D register map (guide section 3)
    M = N = K = 64
    A = np.eye(64)
    B = np.arange(64 * 64, dtype=np.float64).reshape(64, 64)
    out = np.zeros((64, 64))
    pkg._cabi.check(lib.gmrf_test_gemm(0, M, N, K, 0, 0, 0, 0, 1.0, pkg._cabi.ptr(A), 64, pkg._cabi.ptr(B), 64,
                                       0.0, pkg._cabi.ptr(out), 64))
    assert np.array_equal(out, B)


@pytest.mark.parametrize("tri,ta,tb", [(1, 0, 0), (2, 1, 0), (4, 0, 0), (8, 0, 1), (2 | 4, 1, 0), (1, 0, 0)])
def test_gemm_triangular_k_ranges(lib, pkg, tri, ta, tb):
    out, ref, _ = _gemm(lib, pkg, 256, 256, 256, ta, tb, tri=tri, seed=tri)
    assert np.max(np.abs(out - ref)) < 1e-12 * 256


def test_gemm_lower_only_leaves_upper_tiles(lib, pkg):
    out, ref, c0 = _gemm(lib, pkg, 256, 256, 64, 0, 1, lower=1, alpha=-1.0, beta=1.0, seed=5)
    for bm in range(4):
        for bn in range(4):
            blk = (slice(bm * 64, bm * 64 + 64), slice(bn * 64, bn * 64 + 64))
            if bn <= bm:
                assert np.max(np.abs(out[blk] - ref[blk])) < 1e-11
            else:
                assert np.array_equal(out[blk], c0[blk])


BIG = 2048   # gmrf_test_gemm: take the 128 x 128 kernel


@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("tri", [0, 1, 4, 8])
def test_gemm_big_tile_kernel(lib, pkg, tb, tri):
    M, N, K = (256, 256, 256) if tri else (256, 384, 208)
    out, ref, _ = _gemm(lib, pkg, M, N, K, 0, tb, tri=tri | BIG, alpha=-0.75, beta=1.0, seed=40 + tri + tb)
    assert np.max(np.abs(out - ref)) < 1e-12 * 256


def test_gemm_big_matches_small_bitwise(lib, pkg):
    # same k order per output element in both kernels (k ascending inside each MFMA slot chain)
    a, ref, _ = _gemm(lib, pkg, 256, 256, 128, 0, 1, tri=BIG, seed=77)
    b, _, _ = _gemm(lib, pkg, 256, 256, 128, 0, 1, tri=0, seed=77)
    assert np.array_equal(a, b)
    c, _, _ = _gemm(lib, pkg, 256, 256, 128, 0, 0, tri=BIG, seed=78)
    d, _, _ = _gemm(lib, pkg, 256, 256, 128, 0, 0, tri=0, seed=78)
    assert np.array_equal(c, d)


LL = 4096    # gmrf_test_gemm: the 32 x 32 low-latency kernel


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("tri", [0, 1, 4])
def test_gemm_low_latency_kernel_matches_bitwise(lib, pkg, ta, tb, tri):
    if tri and ta:
        tri = 2                      # A upper triangular for the transposed operand
    a, ref, _ = _gemm(lib, pkg, 256, 192 if not tri else 256, 256, ta, tb, tri=tri | LL, alpha=-0.5, beta=1.0, seed=90 + tri)
    b, _, _ = _gemm(lib, pkg, 256, 192 if not tri else 256, 256, ta, tb, tri=tri, alpha=-0.5, beta=1.0, seed=90 + tri)
    assert np.max(np.abs(a - ref)) < 1e-12 * 256
    assert np.array_equal(a, b)      # same summation order per element as the 64 x 64 kernel


def test_gemm_big_lower_only(lib, pkg):
    out, ref, c0 = _gemm(lib, pkg, 384, 384, 64, 0, 0, tri=BIG, lower=1, alpha=-1.0, beta=1.0, seed=6)
    for bm in range(3):
        for bn in range(3):
            blk = (slice(bm * 128, bm * 128 + 128), slice(bn * 128, bn * 128 + 128))
            if bn <= bm:
                assert np.max(np.abs(out[blk] - ref[blk])) < 1e-11
            else:
                assert np.array_equal(out[blk], c0[blk])


def _spd(n, seed, cond_boost=0.0):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    return G @ G.T + (n + cond_boost) * np.eye(n)


@pytest.mark.parametrize("seed", [1, 7])
def test_potrf_tile_and_inverse(lib, pkg, seed):
    A = _spd(64, seed)
    t = A.copy(); inv = np.zeros((64, 64)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(info)))
    L = np.linalg.cholesky(A)
    assert info.value == 0
    assert np.max(np.abs(t - L)) / np.max(np.abs(L)) < 1e-13
    assert np.allclose(np.triu(t, 1), 0.0) and np.allclose(np.triu(inv, 1), 0.0)
    assert np.max(np.abs(inv @ L - np.eye(64))) < 1e-12


@pytest.mark.parametrize("where", [3, 17, 40, 63])          # one pivot in each wave's panel
def test_potrf_tile_reports_non_spd(lib, pkg, where):
    A = _spd(64, 2)
    A[where, where] = -1.0
    t = A.copy(); inv = np.zeros((64, 64)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_tile(0, pkg._cabi.ptr(t), pkg._cabi.ptr(inv), C.byref(info)))
    assert info.value == 1


@pytest.mark.parametrize("bs", [64, 128, 256, 1024, 2048])
def test_potrf_block_with_inverse(lib, pkg, bs):
    A = _spd(bs, bs)
    S = np.tril(A).copy()           # only the lower triangle is read
    Linv = np.zeros((bs, bs)); info = C.c_int32(0)
    pkg._cabi.check(lib.gmrf_test_potrf_block(0, bs, pkg._cabi.ptr(S), pkg._cabi.ptr(Linv), C.byref(info)))
    L = np.linalg.cholesky(A)
    assert info.value == 0
    assert np.max(np.abs(np.tril(S) - L)) / np.max(np.abs(L)) < 1e-13
    assert np.allclose(np.triu(S, 1), 0.0)
    assert np.max(np.abs(np.tril(Linv) @ L - np.eye(bs))) < 1e-11
    assert np.allclose(np.triu(Linv, 1), 0.0)


def test_microbench_rates(lib, pkg):
    tf = C.c_double(0.0); gb = C.c_double(0.0)
    pkg._cabi.check(lib.gmrf_test_mfma_f64_rate(0, C.byref(tf)))
    pkg._cabi.check(lib.gmrf_test_hbm_rate(0, 1 << 30, C.byref(gb)))
    print(f"\nmeasured fp64 MFMA rate {tf.value:.1f} TFLOP/s, HBM read {gb.value:.0f} GB/s")
    assert tf.value > 20.0 and gb.value > 1000.0


def test_multi_rank_rehearsal_on_one_gpu(rehearsal):
    """The N > 1 path on real device buffers (tests/conftest.py ran tools/rehearse_driver.py before this
    process touched the GPU): two ranks on cuda:0 over gloo -- rank 0 factors block ranges, the Linv / C
    ranges of a BATCH of two problems are broadcast into rank 1's caller-owned storage, both ranks take
    the mean and draw their own sample ids, one all-reduce -- and the library's RCCL communicator
    (gmrf_comm_*, the path a Julia host takes) with a world of one rank."""
    assert rehearsal, "the rehearsal did not run (no GPU visible at session start?)"
    res = rehearsal.get("result")
    assert res is not None, rehearsal.get("tail")
    assert res.get("two_ranks_rc") == 0, res.get("two_ranks.log")
    assert res.get("single_rc") == 0, res.get("single.log")
    assert res["ok"], res["checks"]


@pytest.mark.gpu
def test_streams_on_distinct_hardware_queues(pkg):
    """gmrf_streams_create: n usable streams, of which n_distinct were measured to overlap pairwise; a handle on
    such a stream gives the results of a handle on the default stream."""
    ss = pkg.StreamSet(4)
    assert len(ss.pointers) == 4 and all(p != 0 for p in ss.pointers) and len(set(ss.pointers)) == 4
    assert 1 <= ss.n_distinct <= 4
    w = pkg.workloads.random_block_tridiagonal(4, 128, seed=3)
    x0 = pkg.ldiv(pkg.tridiagonal_cholesky(w.Q, 4), w.rhs)
    for p in ss.pointers:
        F = pkg.TridiagonalCholeskyFactor(stream=p).factor(w.Q, 4)
        assert np.array_equal(pkg.ldiv(F, w.rhs), x0)
        F.close()
    ss.close()
    assert ss.pointers == []


# gmrf_test_gemm: the LDS-DMA kernels (global_load_lds staging, gemm_f64_dma.hpp) with 64 x 64 / 128 x 64 / 64 x 128 tiles
DMA = {"64x64": 8192, "128x64": 16384, "64x128": 32768}


@pytest.mark.parametrize("shape", list(DMA))
@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("tri", [0, 1, 4, 8])
def test_gemm_dma_kernels_match_bitwise(lib, pkg, shape, tb, tri):
    """Every tile shape of the LDS-DMA GEMM against NumPy, and bitwise against the register-staged 64 x 64 kernel
    (same k order per output element), on full and triangular K ranges, both B layouts, with an addend."""
    M, N, K = (256, 256, 256) if tri else (256, 384, 208)
    a, ref, _ = _gemm(lib, pkg, M, N, K, 0, tb, tri=tri | DMA[shape], alpha=-0.75, beta=1.0, seed=140 + tri + tb)
    b, _, _ = _gemm(lib, pkg, M, N, K, 0, tb, tri=tri, alpha=-0.75, beta=1.0, seed=140 + tri + tb)
    assert np.max(np.abs(a - ref)) < 1e-12 * 256
    assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", ["64x64", "128x64"])
@pytest.mark.parametrize("tb", [0, 1])
def test_gemm_dma_lower_only(lib, pkg, shape, tb):
    out, ref, c0 = _gemm(lib, pkg, 384, 384, 96, 0, tb, tri=DMA[shape], lower=1, alpha=-1.0, beta=1.0, seed=16)
    old, _, _ = _gemm(lib, pkg, 384, 384, 96, 0, tb, tri=0, lower=1, alpha=-1.0, beta=1.0, seed=16)
    for bm in range(6):
        for bn in range(6):
            blk = (slice(bm * 64, bm * 64 + 64), slice(bn * 64, bn * 64 + 64))
            if bn <= bm:
                assert np.max(np.abs(out[blk] - ref[blk])) < 1e-11 and np.array_equal(out[blk], old[blk])
            else:
                assert np.array_equal(out[blk], c0[blk])


@pytest.mark.parametrize("tb", [0, 1])
@pytest.mark.parametrize("tri", [0, 2, 2 | 4, 8])
def test_gemm_dma_a_stored_k_by_m(lib, pkg, tb, tri):
    """Round 4: the LDS-DMA GEMM with A stored [k][m] (the products A^T B of selected inversion, both operands lying [k][.]):
    against NumPy, and bitwise against the register-staged kernel's A-transposed form (same k order per output element), on
    full and triangular K ranges, both B layouts, with an addend; a lower-only launch leaves the upper tiles alone."""
    M, N, K = (256, 256, 256) if tri else (192, 320, 208)
    a, ref, _ = _gemm(lib, pkg, M, N, K, 1, tb, tri=tri | DMA["64x64"], alpha=-0.75, beta=1.0, seed=400 + tri + tb)
    b, _, _ = _gemm(lib, pkg, M, N, K, 1, tb, tri=tri, alpha=-0.75, beta=1.0, seed=400 + tri + tb)
    assert np.max(np.abs(a - ref)) < 1e-12 * 256
    assert np.array_equal(a, b)
    out, ref, c0 = _gemm(lib, pkg, 384, 384, 96, 1, tb, tri=DMA["64x64"], lower=1, alpha=-1.0, beta=1.0, seed=416 + tb)
    old, _, _ = _gemm(lib, pkg, 384, 384, 96, 1, tb, tri=0, lower=1, alpha=-1.0, beta=1.0, seed=416 + tb)
    for bm in range(6):
        for bn in range(6):
            blk = (slice(bm * 64, bm * 64 + 64), slice(bn * 64, bn * 64 + 64))
            if bn <= bm:
                assert np.max(np.abs(out[blk] - ref[blk])) < 1e-11 and np.array_equal(out[blk], old[blk])
            else:
                assert np.array_equal(out[blk], c0[blk])
    if tri == 0:          # the staging pipeline under load (64 K steps per workgroup)
        a, ref, _ = _gemm(lib, pkg, 1024, 1024, 1024, 1, tb, tri=DMA["64x64"], seed=430 + tb)
        b, _, _ = _gemm(lib, pkg, 1024, 1024, 1024, 1, tb, tri=0, seed=430 + tb)
        assert np.max(np.abs(a - ref)) < 1e-12 * 1024 and np.array_equal(a, b)


@pytest.mark.parametrize("shape", list(DMA))
def test_gemm_dma_long_k_every_cu_busy(lib, pkg, shape):
    """1024^3: 128 - 256 workgroups, 64 K steps each -- the staging pipeline (counted vmcnt waits, one barrier per step,
    buffer re-use) under load; any stale or early fragment read shows as a difference from the register-staged kernel."""
    for tb in (0, 1):
        a, ref, _ = _gemm(lib, pkg, 1024, 1024, 1024, 0, tb, tri=DMA[shape], seed=300 + tb)
        b, _, _ = _gemm(lib, pkg, 1024, 1024, 1024, 0, tb, tri=0, seed=300 + tb)
        assert np.max(np.abs(a - ref)) < 1e-12 * 1024
        assert np.array_equal(a, b)

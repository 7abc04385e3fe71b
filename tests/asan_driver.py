"""Child process of tests/test_host_logic.py::test_host_side_under_address_sanitizer: loads the sanitizer build of the
library (argv[1]) -- the ASan runtime is preloaded by the parent -- and drives every entry point that needs no GPU with
valid, ragged and invalid inputs.  Any AddressSanitizer / UBSan report aborts the process (non-zero exit code)."""
import ctypes as C
import sys

import numpy as np
import scipy.sparse as sp

lib = C.CDLL(sys.argv[1])
vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
P = C.POINTER


def ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)


lib.gmrf_test_symbolic_csc.argtypes = [i64, i64, vp, vp, i32, vp]


def symbolic(A, N, base=0):
    A = sp.csc_matrix(A); A.sort_indices()
    cp, rv = A.indptr.astype(np.int64) + base, A.indices.astype(np.int64) + base
    out = np.zeros(8, dtype=np.int64)
    st = lib.gmrf_test_symbolic_csc(A.shape[0], N, ptr(cp), ptr(rv), base, ptr(out))
    return st, out


rng = np.random.default_rng(0)
# 1. block-tridiagonal matrices of several block sizes (padded and unpadded), 0- and 1-based
for n, N in ((512, 8), (640, 5), (4096, 16), (96, 3), (64, 1), (30, 30)):
    bs = n // N
    R = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=1, format="coo")
    keep = np.abs(R.row // bs - R.col // bs) <= 1
    A = sp.coo_matrix((R.data[keep], (R.row[keep], R.col[keep])), shape=(n, n))
    A = (A + A.T + sp.identity(n) * 10).tocsc()
    for base in (0, 1):
        st, out = symbolic(A, N, base)
        assert st == 0, (n, N, base, st)
        assert out[2] > 0 and 0 <= out[0] < 64 * 2 ** int(np.ceil(np.log2(max(1, (bs + 63) // 64))))
# 2. the structured FEM stencil of the BASELINE Darcy configs (7-point pattern squared: the tile plan route)
nx = 48
idx = np.arange(nx * nx).reshape(nx, nx)
rows, cols = [], []
for dy, dx in ((0, 0), (0, 1), (1, 0), (1, 1), (0, -1), (-1, 0), (-1, -1)):
    a = idx[max(0, -dy):nx - max(0, dy), max(0, -dx):nx - max(0, dx)]
    b = idx[max(0, dy):nx - max(0, -dy), max(0, dx):nx - max(0, -dx)]
    rows.append(a.ravel()); cols.append(b.ravel())
S = sp.coo_matrix((np.ones(sum(r.size for r in rows)), (np.concatenate(rows), np.concatenate(cols))), shape=(nx * nx, nx * nx)).tocsr()
Q = (S @ S @ S).tocsc()
st, out = symbolic(Q, nx // 4)
assert st == 0 and out[4] == 1 and out[5] == 1, out
# 3. entries outside the block tri-band, out-of-range rows, shapes that do not divide: status codes, no memory errors
A = sp.identity(256, format="lil"); A[200, 3] = 1.0; A[3, 200] = 1.0
assert symbolic(A, 8)[0] == -3
assert symbolic(sp.identity(100), 7)[0] == -2
cp = np.arange(11, dtype=np.int64); rv = np.full(10, 99, dtype=np.int64); out = np.zeros(8, dtype=np.int64)
assert lib.gmrf_test_symbolic_csc(10, 2, ptr(cp), ptr(rv), 0, ptr(out)) == -2
cp = np.array([0, 2, 1, 3, 4, 5, 6, 7, 8, 9, 10], dtype=np.int64); rv = np.arange(10, dtype=np.int64)
assert lib.gmrf_test_symbolic_csc(10, 2, ptr(cp), ptr(rv), 0, ptr(out)) == -2
# 4. posterior assembler, symbolic phase (device = -1)
lib.gmrf_assemble_create.argtypes = [i32, vp, i64, vp, vp, i64, vp, vp, i32, P(vp)]
lib.gmrf_assemble_pattern.argtypes = [vp, P(i64), P(i64), vp, vp, i32]
lib.gmrf_assemble_destroy.argtypes = [vp]
for n, m in ((40, 30), (200, 260), (5, 1)):
    Qm = (sp.random(n, n, density=0.1, random_state=2) + sp.identity(n)).tocsc(); Qm = (Qm + Qm.T).tocsc(); Qm.sort_indices()
    J = sp.random(m, n, density=0.08, random_state=3, format="csr"); J.sort_indices()
    h = vp()
    qp, qi, jp, ji = (x.astype(np.int64) for x in (Qm.indptr, Qm.indices, J.indptr, J.indices))
    assert lib.gmrf_assemble_create(-1, None, n, ptr(qp), ptr(qi), m, ptr(jp), ptr(ji), 0, C.byref(h)) == 0
    nnz, npr = i64(0), i64(0)
    assert lib.gmrf_assemble_pattern(h, C.byref(nnz), C.byref(npr), None, None, 0) == 0
    cpo, rvo = np.zeros(n + 1, dtype=np.int64), np.zeros(nnz.value, dtype=np.int64)
    assert lib.gmrf_assemble_pattern(h, None, None, ptr(cpo), ptr(rvo), 1) == 0
    ref = (abs(Qm) + abs(J.T @ J)).tocsc(); ref.sort_indices()
    assert nnz.value >= ref.nnz and cpo[-1] == nnz.value + 1
    lib.gmrf_assemble_destroy(h)
ji_bad = np.array([0, 7], dtype=np.int64); jp1 = np.array([0, 2], dtype=np.int64)
Q5 = sp.identity(5, format="csc")
h = vp()
assert lib.gmrf_assemble_create(-1, None, 5, ptr(Q5.indptr.astype(np.int64)), ptr(Q5.indices.astype(np.int64)), 1, ptr(jp1), ptr(ji_bad), 0, C.byref(h)) == -2
# 5. FEM patterns (device = -1): Darcy 7-point, Burgers P1 / P2 lines, shallow-water K / S and quadrature points
lib.gmrf_darcy_p1_create.argtypes = [i32, vp, i64, i64, P(vp)]
lib.gmrf_darcy_p1_pattern.argtypes = [vp, P(i64), vp, vp, i32]
lib.gmrf_darcy_p1_destroy.argtypes = [vp]
for nx_, ny_ in ((2, 2), (9, 5), (64, 64)):
    h = vp(); assert lib.gmrf_darcy_p1_create(-1, None, nx_, ny_, C.byref(h)) == 0
    nnz = i64(0); lib.gmrf_darcy_p1_pattern(h, C.byref(nnz), None, None, 0)
    rp, ci = np.zeros(nx_ * ny_ + 1, dtype=np.int64), np.zeros(nnz.value, dtype=np.int64)
    assert lib.gmrf_darcy_p1_pattern(h, None, ptr(rp), ptr(ci), 0) == 0 and rp[-1] == nnz.value and ci.max() == nx_ * ny_ - 1
    lib.gmrf_darcy_p1_destroy(h)
lib.gmrf_burgers_p1_create.argtypes = [i32, vp, i64, i64, dbl, dbl, P(vp)]
lib.gmrf_burgers_p2_create.argtypes = [i32, vp, i64, i64, dbl, dbl, P(vp)]
lib.gmrf_burgers_p1_pattern.argtypes = [vp, P(i64), vp, vp, i32]
lib.gmrf_burgers_p1_destroy.argtypes = [vp]
for create, ns, nt in ((lib.gmrf_burgers_p1_create, 3, 2), (lib.gmrf_burgers_p1_create, 17, 5), (lib.gmrf_burgers_p2_create, 6, 2), (lib.gmrf_burgers_p2_create, 64, 7)):
    h = vp(); assert create(-1, None, ns, nt, 0.1, 0.01, C.byref(h)) == 0
    nnz = i64(0); lib.gmrf_burgers_p1_pattern(h, C.byref(nnz), None, None, 0)
    rows_ = (nt - 1) * ns
    rp, ci = np.zeros(rows_ + 1, dtype=np.int64), np.zeros(nnz.value, dtype=np.int64)
    assert lib.gmrf_burgers_p1_pattern(h, None, ptr(rp), ptr(ci), 0) == 0 and rp[-1] == nnz.value and 0 <= ci.min() and ci.max() < ns * nt
    lib.gmrf_burgers_p1_destroy(h)
h = vp(); assert lib.gmrf_burgers_p2_create(-1, None, 7, 3, 0.1, 0.01, C.byref(h)) == -2
lib.gmrf_shallow_water_p1_create.argtypes = [i32, vp, i64, i64, P(vp)]
lib.gmrf_shallow_water_p1_pattern.argtypes = [vp, i32, P(i64), vp, vp, i32]
lib.gmrf_shallow_water_p1_qpoints.argtypes = [vp, vp]
lib.gmrf_shallow_water_p1_destroy.argtypes = [vp]
for nx_, ny_ in ((2, 2), (7, 4), (33, 21)):
    h = vp(); assert lib.gmrf_shallow_water_p1_create(-1, None, nx_, ny_, C.byref(h)) == 0
    for which in (0, 1):
        nnz = i64(0); lib.gmrf_shallow_water_p1_pattern(h, which, C.byref(nnz), None, None, 0)
        rp, ci = np.zeros(3 * nx_ * ny_ + 1, dtype=np.int64), np.zeros(nnz.value, dtype=np.int64)
        assert lib.gmrf_shallow_water_p1_pattern(h, which, None, ptr(rp), ptr(ci), 0) == 0 and rp[-1] == nnz.value
    xy = np.zeros((2 * (nx_ - 1) * (ny_ - 1), 3, 2))
    assert lib.gmrf_shallow_water_p1_qpoints(h, ptr(xy)) == 0 and 0 < xy.min() and xy.max() < 1
    assert lib.gmrf_shallow_water_p1_pattern(h, 5, None, None, None, 0) == -2
    lib.gmrf_shallow_water_p1_destroy(h)
# 6. entry points that must refuse without a device, sizes queries
lib.gmrf_bt_create.argtypes = [i32, vp, P(vp)]
h = vp(); assert lib.gmrf_bt_create(0, None, C.byref(h)) == -6
lib.gmrf_bt_storage_bytes.argtypes = [i64, i64, i64, P(i64), P(i64), P(i64)]
a, b, c = i64(0), i64(0), i64(0)
assert lib.gmrf_bt_storage_bytes(65536, 64, 32, C.byref(a), C.byref(b), C.byref(c)) == 0 and a.value == 32 * 64 * 1024 * 1024 * 8
assert lib.gmrf_bt_storage_bytes(10, 3, 1, C.byref(a), C.byref(b), C.byref(c)) == -2
print("asan driver ok")

/* ORACLE (test infrastructure, never shipped, never on the measured GPU path).
 *
 * Dependency-free C restatement of /root/reference/src/tridiagonal_cholesky.jl for boxes
 * without SciPy: dense blocks, column-by-column Cholesky (what LAPACK dpotf2 computes),
 * forward substitution for C = B L^-T (:74), D - C C^T (:77), and the two sweeps (:43-52,
 * :24-33; the three defects of the reference's solve half corrected).
 * PARITY UNPINNED (no Julia here, no golden vectors in the reference); pinned against the
 * NumPy oracle and the golden fixtures in tests/test_oracle.py.
 *
 * Layout: row-major dense blocks.  Ld[N][bs][bs] lower factors, Cs[N-1][bs][bs].
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

/* in-place lower Cholesky of a bs x bs row-major matrix; returns 0 or the failing column + 1 */
static int potrf_lower(double* a, int64_t bs) {
    for (int64_t j = 0; j < bs; ++j) {
        double d = a[j * bs + j];
        for (int64_t k = 0; k < j; ++k) d -= a[j * bs + k] * a[j * bs + k];
        if (!(d > 0.0)) return (int)(j + 1);
        d = sqrt(d);
        a[j * bs + j] = d;
        for (int64_t i = j + 1; i < bs; ++i) {
            double s = a[i * bs + j];
            for (int64_t k = 0; k < j; ++k) s -= a[i * bs + k] * a[j * bs + k];
            a[i * bs + j] = s / d;
        }
        for (int64_t c = j + 1; c < bs; ++c) a[j * bs + c] = 0.0;
    }
    return 0;
}

/* dense blocks D[N][bs][bs] (lower triangle read), B[N-1][bs][bs] (block (i+1,i)) ->
 * Ld, Cs.  Returns 0 or the 1-based index of the block that is not positive definite. */
int bt_factor_dense(int64_t N, int64_t bs, const double* D, const double* B, double* Ld, double* Cs) {
    const int64_t bb = bs * bs;
    for (int64_t i = 0; i < N; ++i) {
        double* L = Ld + i * bb;
        for (int64_t e = 0; e < bb; ++e) L[e] = D[i * bb + e];
        if (i > 0) {
            const double* Lp = Ld + (i - 1) * bb;
            double* C = Cs + (i - 1) * bb;
            /* C = B Lp^-T : row r of C solves Lp c = b_r  (forward_solve(chos[end], B')', :74) */
            for (int64_t r = 0; r < bs; ++r)
                for (int64_t j = 0; j < bs; ++j) {
                    double s = B[(i - 1) * bb + r * bs + j];
                    for (int64_t k = 0; k < j; ++k) s -= Lp[j * bs + k] * C[r * bs + k];
                    C[r * bs + j] = s / Lp[j * bs + j];
                }
            /* L = D - C C^T, lower triangle (:77) */
            for (int64_t r = 0; r < bs; ++r)
                for (int64_t c = 0; c <= r; ++c) {
                    double s = 0.0;
                    for (int64_t k = 0; k < bs; ++k) s += C[r * bs + k] * C[c * bs + k];
                    L[r * bs + c] -= s;
                }
        }
        if (potrf_lower(L, bs) != 0) return (int)(i + 1);
    }
    return 0;
}

/* y = L^-1 b (mode 1), y = L^-T b (mode 2), y = A^-1 b (mode 0); b, y length N*bs; may alias */
void bt_solve_dense(int64_t N, int64_t bs, const double* Ld, const double* Cs, const double* b, double* y,
                    int mode) {
    const int64_t bb = bs * bs, n = N * bs;
    if (y != b) for (int64_t i = 0; i < n; ++i) y[i] = b[i];
    if (mode == 0 || mode == 1) {
        for (int64_t i = 0; i < N; ++i) {
            const double* L = Ld + i * bb;
            double* yi = y + i * bs;
            if (i > 0) {
                const double* C = Cs + (i - 1) * bb;
                const double* yp = y + (i - 1) * bs;
                for (int64_t r = 0; r < bs; ++r) {
                    double s = 0.0;
                    for (int64_t k = 0; k < bs; ++k) s += C[r * bs + k] * yp[k];
                    yi[r] -= s;                                    /* b_i - Cs[i-1] x[i-1], :49 */
                }
            }
            for (int64_t r = 0; r < bs; ++r) {
                double s = yi[r];
                for (int64_t k = 0; k < r; ++k) s -= L[r * bs + k] * yi[k];
                yi[r] = s / L[r * bs + r];
            }
        }
    }
    if (mode == 0 || mode == 2) {
        for (int64_t i = N - 1; i >= 0; --i) {
            const double* L = Ld + i * bb;
            double* yi = y + i * bs;
            if (i < N - 1) {
                const double* C = Cs + i * bb;
                const double* yn = y + (i + 1) * bs;
                for (int64_t c = 0; c < bs; ++c) {
                    double s = 0.0;
                    for (int64_t k = 0; k < bs; ++k) s += C[k * bs + c] * yn[k];
                    yi[c] -= s;                                    /* b_i - Cs[i]' x[i+1], :30 */
                }
            }
            for (int64_t r = bs - 1; r >= 0; --r) {
                double s = yi[r];
                for (int64_t k = r + 1; k < bs; ++k) s -= L[k * bs + r] * yi[k];
                yi[r] = s / L[r * bs + r];
            }
        }
    }
}

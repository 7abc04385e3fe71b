"""Generates tests/golden/*.npz: small seeded input/output vectors of the block-tridiagonal
path.  The reference itself cannot run here (no Julia), so the expected outputs
come from an independent dense route (NumPy Cholesky / inverse of the full matrix), NOT from
the oracle and NOT from the HIP path; both are tested against these files.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
HERE = os.path.dirname(os.path.abspath(__file__))


def dense_reference(w, k=6, seed=11):
    A = w.Q.toarray()
    n, N = w.n, w.n_blocks
    bs = n // N
    Lfull = np.linalg.cholesky(A)           # block Cholesky == the dense factor's tri-band blocks
    chos = np.stack([Lfull[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] for i in range(N)])
    Cs = np.stack([Lfull[(i + 1) * bs:(i + 2) * bs, i * bs:(i + 1) * bs] for i in range(N - 1)]) if N > 1 else np.zeros((0, bs, bs))
    mean = np.linalg.solve(A, w.rhs)
    Z = np.random.default_rng(seed).standard_normal((n, k))
    fwd = np.linalg.solve(Lfull, Z)
    bwd = np.linalg.solve(Lfull.T, Z)
    var = np.diag(np.linalg.inv(A))
    logdet = 2.0 * np.log(np.diag(Lfull)).sum()
    csc = w.Q.tocsc()
    return dict(n=n, n_blocks=N, colptr=csc.indptr.astype(np.int64), rowval=csc.indices.astype(np.int64),
                nzval=csc.data, rhs=w.rhs, chos=chos, Cs=Cs, mean=mean, Z=Z, forward=fwd, backward=bwd,
                var=var, logdet=logdet)


CASES = {
    "darcy16": lambda: pkg.workloads.darcy(16),                   # n = 256, 4 blocks of 64
    "burgers32x6": lambda: pkg.workloads.burgers(32, 6),          # n = 192, 6 blocks of 32 (padded to 64 on GPU)
    "elliptic16": lambda: pkg.workloads.elliptic(16),             # n = 256, 8 blocks of 32
    "rand5x24": lambda: pkg.workloads.random_block_tridiagonal(5, 24, seed=9),
}

if __name__ == "__main__":
    for name, mk in CASES.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **dense_reference(mk()))
        print("wrote", name)

"""Oracle-backed engine for the CPU tests of the multi-rank driver (posterior.ShardedPosterior):
same protocol as HipEngine, NumPy numerics, torch CPU tensors as the broadcast unit."""
import numpy as np
import torch

from oracle import bt_oracle as O
from tests.philox_ref import philox_normals_np


class OracleEngine:
    def __init__(self, workload):
        self.w = workload
        N, bs = workload.n_blocks, workload.block_size
        self.L = torch.zeros(N, bs, bs, dtype=torch.float64)
        self.C = torch.zeros(max(N - 1, 1), bs, bs, dtype=torch.float64)
        self.Li = torch.zeros(N, bs, bs, dtype=torch.float64)     # unused numerically, broadcast like the HIP path
        self.F = None
        self.calls = []

    def prepare(self, is_root, shared_storage, dist=None):
        self.calls.append(("prepare", is_root, shared_storage))
        self._pending = []

    def _factor_all(self):
        F = O.tridiagonal_cholesky(self.w.Q, self.w.n_blocks)
        self._full = F

    def factor(self):
        self._factor_all()
        self.F = self._full

    def factor_range_async(self, i0, i1, first):
        if first:
            self._factor_all()
        for i in range(i0, i1):
            self.L[i] = torch.from_numpy(self._full.chos[i])
            self.Li[i] = torch.from_numpy(np.linalg.inv(self._full.chos[i]))
            if i > 0:
                self.C[i - 1] = torch.from_numpy(self._full.Cs[i - 1])

    def factor_end(self):
        self.adopt_commit()

    def adopt_commit(self):
        # like the HIP path, only Linv and C travel: every rank (the root too, so that all ranks hold
        # bitwise the same factor) rebuilds the triangular blocks from the inverses it holds
        N = self.w.n_blocks
        self.F = O.TridiagonalCholeskyFactor(self.w.n, [np.linalg.inv(self.Li[i].numpy()) for i in range(N)],
                                             [self.C[i].numpy() for i in range(N - 1)])

    def share_range(self, dist, i0, i1, is_root=True):
        ts = [self.Li[i0:i1]]
        c0, c1 = max(i0 - 1, 0), max(i1 - 1, 0)
        if c1 > c0:
            ts.append(self.C[c0:c1])
        self._pending.extend(dist.broadcast(t, src=0, async_op=True) for t in ts)

    def share_finish(self, is_root=True):
        for hnd in self._pending:
            hnd.wait()
        self._pending = []

    batch = 1

    def mean(self):
        return O.ldiv(self.F, self.w.rhs)

    def sample(self, k, mean, seed, first_id, keep=True):
        Z = philox_normals_np(seed, self.w.n, first_id, k)
        return O.sample(self.F, mean, Z)

    def synchronize(self):
        pass

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


class OracleGatherEngine:
    """The all-gather form of sharing a batch of factors (posterior.HipGatherEngine's protocol, NumPy numerics): the
    `batch_total` problems are `scales[p] * Q`; rank r factors the problems [r * b, (r + 1) * b), the inverses and
    coupling blocks of each block range are all-gathered, every rank then holds all of them."""

    def __init__(self, workload, batch_total, world, rank, scales):
        self.w, self.batch, self.world, self.rank = workload, batch_total, world, rank
        self.b = batch_total // world
        self.scales = list(scales)
        N, bs = workload.n_blocks, workload.block_size
        self.Li_own = torch.zeros(self.b, N, bs, bs, dtype=torch.float64)
        self.C_own = torch.zeros(self.b, max(N - 1, 1), bs, bs, dtype=torch.float64)
        self.Li = torch.zeros(batch_total, N, bs, bs, dtype=torch.float64)
        self.C = torch.zeros(batch_total, max(N - 1, 1), bs, bs, dtype=torch.float64)
        self.F = None

    def prepare(self, is_root, shared_storage, dist=None):
        pass

    def factor_range_async(self, i0, i1, first):
        if first:
            self._full = [O.tridiagonal_cholesky(self.scales[self.rank * self.b + p] * self.w.Q, self.w.n_blocks) for p in range(self.b)]
        for p in range(self.b):
            for i in range(i0, i1):
                self.Li_own[p, i] = torch.from_numpy(np.linalg.inv(self._full[p].chos[i]))
                if i > 0:
                    self.C_own[p, i - 1] = torch.from_numpy(self._full[p].Cs[i - 1])

    def _gather(self, dist, own, out):
        if dist is None or self.world == 1:
            out.copy_(own)
            return
        parts = [torch.empty_like(own) for _ in range(self.world)]
        dist.all_gather(parts, own.contiguous())
        out.copy_(torch.cat(parts, dim=0))

    def share_range(self, dist, i0, i1, is_root=True):
        self._gather(dist, self.Li_own[:, i0:i1], self.Li[:, i0:i1])
        c0, c1 = max(i0 - 1, 0), max(i1 - 1, 0)
        if c1 > c0:
            self._gather(dist, self.C_own[:, c0:c1], self.C[:, c0:c1])

    def share_finish(self, is_root=True):
        pass

    def factor_end(self):
        pass

    def adopt_commit(self):
        N = self.w.n_blocks
        self.F = [O.TridiagonalCholeskyFactor(self.w.n, [np.linalg.inv(self.Li[p, i].numpy()) for i in range(N)],
                                              [self.C[p, i].numpy() for i in range(N - 1)]) for p in range(self.batch)]

    def mean(self):
        return np.stack([O.ldiv(F, self.w.rhs) for F in self.F])

    def sample(self, k, mean, seed, first_id, keep=True):
        return np.stack([O.sample(self.F[p], mean[p], philox_normals_np(seed, self.w.n, first_id + p * k, k)) for p in range(self.batch)])

    def synchronize(self):
        pass

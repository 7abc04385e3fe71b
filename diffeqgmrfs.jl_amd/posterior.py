"""Posterior job = factor the posterior precision once, posterior mean (1 forward + 1 backward
sweep), k_s samples (1 backward sweep), optionally marginal variances -- the per-problem loop of
/root/reference/scripts/darcy/solve_darcy_gmrf-fem.jl:176-198 (`condition_on_observations`,
`mean`, `rand`, `std`) on the block-tridiagonal path.

Sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" on CPU for the tests):
  * samples are independent: rank r draws the sample ids [r*k_s, (r+1)*k_s) -- Philox keyed
    by (seed, sample id, dof), so a sample does not depend on the number of ranks;
  * the factor is shared: rank 0 factors block ranges and broadcasts each finished range of
    L / C / Linv blocks while the next range is being factored (the only collective on the
    data path); the variance accumulators are summed with one all-reduce.

The engine object does the numerics.  `HipEngine` drives libgmrf_hip.so; the CPU tests plug an
oracle-backed engine into the same driver to cover the N > 1 control flow without a GPU.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of `total` items: (first, count) of `rank`; remainders go to the low ranks."""
    base, rem = divmod(total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def block_groups(n_blocks: int, group: int) -> List[Tuple[int, int]]:
    return [(i, min(n_blocks, i + group)) for i in range(0, n_blocks, group)]


class HipEngine:
    """libgmrf_hip.so behind the driver protocol; factor storage lives in torch tensors so that
    torch.distributed can broadcast it in place."""

    def __init__(self, pkg, workload, device_index: int = 0, batch: int = 1, values=None, rhs=None):
        """`values` (batch, nnz) / `rhs` (batch, n): one row per independent problem on the
        workload's sparsity pattern (default: the workload itself, repeated)."""
        import torch
        self.torch = torch
        self.pkg = pkg
        self.w = workload
        self.batch = batch
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.stream = torch.cuda.current_stream(self.dev)
        self.F = pkg.TridiagonalCholeskyFactor(device=device_index, stream=self.stream.cuda_stream, batch=batch)
        if values is None:
            values = np.tile(np.ascontiguousarray(workload.Q.data), (batch, 1))
        if rhs is None:
            rhs = np.tile(np.ascontiguousarray(workload.rhs), (batch, 1))
        self.values_host = np.ascontiguousarray(values, dtype=np.float64).reshape(batch, -1)
        self.nz = torch.from_numpy(self.values_host).to(self.dev)
        self.rhs = torch.from_numpy(np.ascontiguousarray(rhs, dtype=np.float64).reshape(batch, 1, -1)).to(self.dev)
        self.buffers = None
        self._analysed = False

    # --- storage shared with torch
    def _attach_storage(self):
        import ctypes as C
        lib = self.pkg._cabi.load()
        bl, bc, bi = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self.pkg._cabi.check(lib.gmrf_bt_storage_bytes(self.w.n, self.w.n_blocks, C.byref(bl), C.byref(bc), C.byref(bi)))
        t = self.torch
        self.buffers = [t.zeros(b.value // 8, dtype=t.float64, device=self.dev) for b in (bl, bc, bi)]
        self.pkg._cabi.check(lib.gmrf_bt_set_storage(self.F._h, self.w.n, self.w.n_blocks,
                                                     *[self.pkg._cabi.ptr(b) for b in self.buffers]))
        self.block_elems = [self.buffers[0].numel() // self.w.n_blocks] * 3
        self.F._set_shape(self.w.n, self.w.n_blocks)

    def prepare(self, is_root: bool, shared_storage: bool):
        """Untimed set-up: symbolic analysis (root), storage, graph capture."""
        if shared_storage:
            self._attach_storage()
        if is_root or not shared_storage:
            self.F.factor(self.w.Q, self.w.n_blocks, values=self.values_host)   # analyse + first numeric factor
            self._analysed = True
        elif not shared_storage:
            self.F.adopt_shape(self.w.n, self.w.n_blocks)

    # --- numeric phases
    def factor(self):
        self.F.refactor(self.nz)

    def factor_range_async(self, i0: int, i1: int, first: bool):
        if first:
            self.F.factor_begin_values(self.nz)
        self.F.factor_step_async(i0, i1)

    def factor_end(self):
        self.F.factor_end()

    def adopt_commit(self):
        self.F.adopt_commit()

    def slices(self, i0: int, i1: int):
        """The tensors holding blocks [i0, i1) of L, C (blocks i0-1 .. i1-2) and Linv."""
        e = self.block_elems[0]
        out = [self.buffers[0][i0 * e:i1 * e], self.buffers[2][i0 * e:i1 * e]]
        c0, c1 = max(i0 - 1, 0), max(i1 - 1, 0)
        if c1 > c0:
            out.append(self.buffers[1][c0 * e:c1 * e])
        return out

    def mean(self):
        """(batch, n) posterior means."""
        return self.F.solve_batch(self.rhs)[:, 0, :]

    def sample(self, k: int, mean, seed: int, first_id: int):
        """(batch, k, n) samples; problem p draws the ids first_id + p*k + s."""
        return self.F.sample_batch(k, mean=mean, seed=seed, first_id=first_id, like=self.rhs)

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)


class ShardedPosterior:
    """One posterior job across `world` ranks (see module docstring)."""

    def __init__(self, engine, dist=None, rank: int = 0, world: int = 1, k_samples: int = 64,
                 seed: int = 0x5EED, group: int = 8, replicate_factor: bool = False):
        self.e, self.dist, self.rank, self.world = engine, dist, rank, world
        self.k, self.seed, self.group = k_samples, seed, group
        self.replicate = replicate_factor or world == 1
        self.groups = block_groups(engine.w.n_blocks, group)

    def prepare(self):
        self.e.prepare(is_root=(self.rank == 0), shared_storage=not self.replicate)
        if self.dist is not None and self.world > 1:
            self.dist.barrier()

    def _factor_and_share(self):
        if self.replicate:
            self.e.factor()
            return
        handles = []
        for gi, (i0, i1) in enumerate(self.groups):
            if self.rank == 0:
                self.e.factor_range_async(i0, i1, first=(gi == 0))
            for t in self.e.slices(i0, i1):
                handles.append(self.dist.broadcast(t, src=0, async_op=True))
        for hnd in handles:
            hnd.wait()
        if self.rank == 0:
            self.e.factor_end()
        else:
            self.e.adopt_commit()

    def step(self, step_index: int = 0):
        """factor (+ broadcast) -> mean -> this rank's k samples.  Returns (mean, samples)."""
        self._factor_and_share()
        mu = self.e.mean()
        nb = getattr(self.e, "batch", 1)
        first = (step_index * self.world + self.rank) * self.k * nb
        X = self.e.sample(self.k, mu, self.seed, first)
        return mu, X

    def solves_per_step(self) -> int:
        """Posterior solves of one step over all ranks.  Shared factor (broadcast): one mean and
        k samples per rank.  Replicated / independent problems: every rank handles its own batch
        of problems, each with its own mean and k samples."""
        nb = getattr(self.e, "batch", 1)
        if self.replicate:
            return self.world * nb * (1 + self.k)
        return 1 + self.k * self.world

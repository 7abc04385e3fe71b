"""Posterior job = factor the posterior precision once, posterior mean (1 forward + 1 backward
sweep), k_s samples (1 backward sweep), optionally marginal variances -- the per-problem loop of
/root/reference/scripts/darcy/solve_darcy_gmrf-fem.jl:176-198 (`condition_on_observations`,
`mean`, `rand`, `std`) on the block-tridiagonal path.

Sharding across the GPUs of one node (one process per GPU):
  * samples are independent: rank r draws the sample ids [r*k_s, (r+1)*k_s) -- Philox keyed
    by (seed, sample id, dof), so a sample does not depend on the number of ranks;
  * the factor is shared: rank 0 factors block ranges and each finished range of Linv / C blocks
    (what the sweeps read; the L blocks stay on rank 0) is broadcast while the next range is being
    factored -- the only collective on the data path; variance accumulators are summed with one
    all-reduce.  A range travels as ONE packed image (gmrf_bt_pack_blocks_async: the lower-triangular
    64 x 64 tiles of the block inverses, the stored windows of the coupling blocks, the blocks'
    log-determinant parts): darcy256 0.56 GB per posterior instead of 0.83 GB of raw buffers.

Two transports move the factor, both RCCL on a GPU box:
  * "cabi"  : the library's own communicator (gmrf_comm_*, include/gmrf_hip.h) -- what a Julia host
              uses; broadcasts run on the communicator's stream beside the factorisation;
  * "torch" : torch.distributed broadcasts of caller-owned storage tensors (gmrf_bt_set_storage);
              backend "nccl" is RCCL, "gloo" serves the CPU tests.

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
    """libgmrf_hip.so behind the driver protocol."""

    def __init__(self, pkg, workload, device_index: int = 0, batch: int = 1, values=None, rhs=None,
                 keep_l: bool = True, transport: str = "torch", comm=None):
        """`values` (batch, nnz) / `rhs` (batch, n): one row per independent problem on the
        workload's sparsity pattern (default: the workload itself, repeated).  `keep_l` False: the L
        blocks are not retained (gmrf_bt_set_keep_l).  `transport` / `comm`: see the module docstring
        (`comm` = an api.Comm for "cabi")."""
        import torch
        self.torch = torch
        self.pkg = pkg
        self.w = workload
        self.batch = batch
        self.dev = torch.device("cuda", device_index)
        torch.cuda.set_device(self.dev)
        self.stream = torch.cuda.current_stream(self.dev)
        if self.stream.cuda_stream == 0:
            # On the legacy default stream the library would give the handle a stream of its OWN (gmrf_bt_create: stream 0 =
            # none given), and the torch work of this engine (copies, collectives on packed images) would not be ordered with
            # the handle's launches: the engine takes a stream of its own and makes it the thread's current stream.
            self.stream = torch.cuda.Stream(self.dev)
            torch.cuda.set_stream(self.stream)
        self.F = pkg.TridiagonalCholeskyFactor(device=device_index, stream=self.stream.cuda_stream, batch=batch)
        if not keep_l:
            self.F.set_keep_l(False)
        self.keep_l = keep_l
        self.transport, self.comm = transport, comm
        if transport == "cabi" and comm is None:
            raise ValueError('transport "cabi" needs an api.Comm')
        if values is None:
            values = np.tile(np.ascontiguousarray(workload.Q.data), (batch, 1))
        if rhs is None:
            rhs = np.tile(np.ascontiguousarray(workload.rhs), (batch, 1))
        self.values_host = np.ascontiguousarray(values, dtype=np.float64).reshape(batch, -1)
        self.nz = torch.from_numpy(self.values_host).to(self.dev)
        self.rhs = torch.from_numpy(np.ascontiguousarray(rhs, dtype=np.float64).reshape(batch, 1, -1)).to(self.dev)
        self._stage = {}
        self.bytes_moved = 0
        self._pending = []

    def prepare(self, is_root: bool, shared_storage: bool, dist=None):
        """Untimed set-up: symbolic analysis (root), the layout of the stored coupling blocks to every
        rank, storage, graph capture."""
        if not shared_storage:
            self.F.factor(self.w.Q, self.w.n_blocks, values=self.values_host)   # analyse + first numeric factor
            return
        if is_root:
            self.F.factor(self.w.Q, self.w.n_blocks, values=self.values_host)
            layout = self.F.get_layout()
        else:
            layout = None
        # the layout record (a few int64) travels once, before the timed region
        if self.transport == "cabi":
            cnt = np.array([0 if layout is None else layout.size], dtype=np.int64)
            self.comm.bcast_host(cnt, 0)
            if layout is None:
                layout = np.zeros(int(cnt[0]), dtype=np.int64)
            self.comm.bcast_host(layout, 0)
        else:
            box = [layout]
            dist.broadcast_object_list(box, src=0)
            layout = np.asarray(box[0], dtype=np.int64)
        if not is_root:
            self.F.adopt_layout(self.w.n, self.w.n_blocks, layout)

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
        self.F.adopt_commit(False)

    def _staging(self, i0: int, i1: int):
        """(batch, packed_size) device tensor that holds the transport image of blocks [i0, i1) (transport "torch")."""
        key = (i0, i1)
        if key not in self._stage:
            t = self.torch
            self._stage[key] = t.empty((self.batch, self.F.packed_size(i0, i1)), dtype=t.float64, device=self.dev)
        return self._stage[key]

    def share_range(self, dist, i0: int, i1: int, is_root: bool = True):
        """Enqueue the broadcast of the finished blocks [i0, i1) from rank 0 (every rank calls this)."""
        if self.transport == "cabi":
            self.comm.bcast_blocks_async(self.F, i0, i1, root=0, with_l=False)
            return
        buf = self._staging(i0, i1)
        if is_root:
            self.F.pack_blocks_async(i0, i1, buf)          # on the handle's stream = torch's current stream
        self._pending.append((dist.broadcast(buf, src=0, async_op=True), i0, i1, buf))
        self.bytes_moved += buf.numel() * 8

    def share_finish(self, is_root: bool = True):
        if self.transport == "cabi":
            self.comm.wait(self.F)
            return
        for hnd, i0, i1, buf in self._pending:
            hnd.wait()                                     # the current stream waits for the transfer
            if not is_root:
                self.F.unpack_blocks_async(i0, i1, buf)
        self._pending = []

    def transport_bytes(self, reset: bool = False) -> float:
        """Factor bytes this rank's transport has moved so far."""
        if self.transport == "cabi":
            return self.comm.bytes_moved(reset)
        v = self.bytes_moved
        if reset:
            self.bytes_moved = 0
        return float(v)

    def mean(self):
        """(batch, n) posterior means."""
        return self.F.solve_batch(self.rhs)[:, 0, :]

    def sample(self, k: int, mean, seed: int, first_id: int, keep: bool = True):
        """(batch, k, n) samples; problem p draws the ids first_id + p*k + s.  A batched handle draws at most 128 per
        call: larger k goes in chunks of 128 (chunk c of problem p: ids first_id + c*128*batch + p*kc + s); with
        `keep` False only the last chunk is returned (throughput runs that do not hold k x n x batch doubles)."""
        if self.batch == 1 or k <= 128:
            return self.F.sample_batch(k, mean=mean, seed=seed, first_id=first_id, like=self.rhs)
        outs = []
        for c0 in range(0, k, 128):
            kc = min(128, k - c0)
            x = self.F.sample_batch(kc, mean=mean, seed=seed, first_id=first_id + c0 * self.batch, like=self.rhs)
            outs = outs + [x] if keep else [x]
        return self.torch.cat(outs, dim=1) if keep else outs[0]

    def synchronize(self):
        self.torch.cuda.synchronize(self.dev)


class HipGatherEngine(HipEngine):
    """The all-gather form of sharing a BATCH of factors (round 4): the `batch_total` posteriors of a step are dealt to
    the ranks in contiguous shares of b = batch_total / world; every rank factors ITS share (handle `F`, batch b), the
    packed images of each finished block range are all-gathered into `F_all` (batch batch_total: problem r * b + p = problem
    p of rank r), and every rank then takes the means of ALL posteriors and draws its own sample ids on `F_all`.  Against
    the root broadcast the factorisation is spread over the ranks and every link of the fabric carries 1 / world of the
    images (gmrf_bt_allgather_blocks_async, or torch.distributed.all_gather_into_tensor of the packed image)."""

    def __init__(self, pkg, workload, device_index: int, batch_total: int, world: int, rank: int, values_all, rhs_all,
                 keep_l: bool = False, transport: str = "torch", comm=None):
        if batch_total % world:
            raise ValueError("the shared batch must be a multiple of the number of ranks")
        b = batch_total // world
        values_all = np.ascontiguousarray(values_all, dtype=np.float64).reshape(batch_total, -1)
        rhs_all = np.ascontiguousarray(rhs_all, dtype=np.float64).reshape(batch_total, -1)
        super().__init__(pkg, workload, device_index=device_index, batch=b, values=values_all[rank * b:(rank + 1) * b],
                         rhs=rhs_all[rank * b:(rank + 1) * b], keep_l=keep_l, transport=transport, comm=comm)
        t = self.torch
        self.share_batch, self.batch, self.world, self.rank = b, batch_total, world, rank      # `batch`: what mean / sample address
        self.F_all = pkg.TridiagonalCholeskyFactor(device=device_index, stream=self.stream.cuda_stream, batch=batch_total)
        self.F_all.set_keep_l(False)
        self.rhs = t.from_numpy(rhs_all.reshape(batch_total, 1, -1)).to(self.dev)
        self._own, self._all = {}, {}

    def prepare(self, is_root: bool, shared_storage: bool, dist=None):
        self.F.factor(self.w.Q, self.w.n_blocks, values=self.values_host)      # every rank analyses the pattern and factors once
        self.F_all.adopt_layout(self.w.n, self.w.n_blocks, self.F.get_layout())

    def adopt_commit(self):
        self.F_all.adopt_commit(False)

    def share_range(self, dist, i0: int, i1: int, is_root: bool = True):
        if self.transport == "cabi":
            self.comm.allgather_blocks_async(self.F, self.F_all, i0, i1)
            return
        t = self.torch
        key = (i0, i1)
        if key not in self._own:
            seg = self.F.packed_size(i0, i1)
            self._own[key] = t.empty((self.share_batch, seg), dtype=t.float64, device=self.dev)
            self._all[key] = t.empty((self.batch, seg), dtype=t.float64, device=self.dev)
        own, allb = self._own[key], self._all[key]
        self.F.pack_blocks_async(i0, i1, own)
        if dist.get_backend() == "nccl":
            self._pending.append((dist.all_gather_into_tensor(allb, own, async_op=True), i0, i1, allb))
        else:
            # gloo has no all-gather of device tensors (rehearsals on one GPU): every rank broadcasts its share into its slot
            b = self.share_batch
            allb[self.rank * b:(self.rank + 1) * b].copy_(own)
            hs = [dist.broadcast(allb[r * b:(r + 1) * b], src=r, async_op=True) for r in range(self.world)]
            for h in hs[:-1]:
                self._pending.append((h, None, None, None))
            self._pending.append((hs[-1], i0, i1, allb))
        self.bytes_moved += (allb.numel() - own.numel()) * 8

    def share_finish(self, is_root: bool = True):
        if self.transport == "cabi":
            self.comm.wait(self.F_all)
            return
        for hnd, i0, i1, allb in self._pending:
            hnd.wait()
            if allb is not None:
                self.F_all.unpack_blocks_async(i0, i1, allb)
        self._pending = []

    def mean(self):
        return self.F_all.solve_batch(self.rhs)[:, 0, :]

    def sample(self, k: int, mean, seed: int, first_id: int, keep: bool = True):
        if k <= 128:
            return self.F_all.sample_batch(k, mean=mean, seed=seed, first_id=first_id, like=self.rhs)
        outs = []
        for c0 in range(0, k, 128):
            kc = min(128, k - c0)
            x = self.F_all.sample_batch(kc, mean=mean, seed=seed, first_id=first_id + c0 * self.batch, like=self.rhs)
            outs = outs + [x] if keep else [x]
        return self.torch.cat(outs, dim=1) if keep else outs[0]


class ShardedPosterior:
    """One posterior job across `world` ranks (see module docstring)."""

    def __init__(self, engine, dist=None, rank: int = 0, world: int = 1, k_samples: int = 64,
                 seed: int = 0x5EED, group: int = 8, replicate_factor: bool = False, force_shared: bool = False,
                 keep_samples: bool = True, timing: bool = False, share: str = "broadcast"):
        """`force_shared`: run the shared-factor protocol (ranged factorisation, broadcasts, commit) even
        with a world of one rank -- rehearsals of the multi-GPU path on a one-GPU box.  `timing`: device events
        around the phases of a step (`phase_ms()` after a synchronisation); HipEngine only."""
        self.e, self.dist, self.rank, self.world = engine, dist, rank, world
        self.k, self.seed, self.group = k_samples, seed, group
        self.replicate = replicate_factor or (world == 1 and not force_shared)
        self.groups = block_groups(engine.w.n_blocks, group)
        self.keep_samples = keep_samples
        # "broadcast": rank 0 factors, every range is broadcast; "allgather": every rank factors its share of the engine's
        # batch and the ranges are all-gathered (HipGatherEngine, or an engine with the same protocol)
        self.share = share
        if share not in ("broadcast", "allgather"):
            raise ValueError(share)
        if share == "allgather":
            self.replicate = False
        self.timing = timing and hasattr(engine, "torch")
        self._ev = None

    def prepare(self):
        self.e.prepare(is_root=(self.rank == 0), shared_storage=not self.replicate, dist=self.dist)
        if self.dist is not None and self.world > 1:
            self.dist.barrier()

    def _mark(self, i: int):
        if self.timing:
            self._ev[i].record(self.e.torch.cuda.current_stream(self.e.dev))

    def _factor_and_share(self):
        if self.timing and self._ev is None:
            self._ev = [self.e.torch.cuda.Event(enable_timing=True) for _ in range(4)]
        self._mark(0)
        if self.replicate:
            self.e.factor()
            self._mark(1); self._mark(2)
            return
        if self.share == "allgather":
            # every rank factors its share of the batch range by range; each finished range is all-gathered beside the next
            for gi, (i0, i1) in enumerate(self.groups):
                self.e.factor_range_async(i0, i1, first=(gi == 0))
                self.e.share_range(self.dist, i0, i1, True)
            self._mark(1)
            self.e.factor_end()
            self.e.share_finish(True)
            self._mark(2)
            self.e.adopt_commit()
            return
        root = self.rank == 0
        for gi, (i0, i1) in enumerate(self.groups):
            if root:
                self.e.factor_range_async(i0, i1, first=(gi == 0))
            self.e.share_range(self.dist, i0, i1, root)
        self._mark(1)                      # root: the last range has been enqueued behind the factorisation
        self.e.share_finish(root)
        self._mark(2)                      # the stream has waited for the last transfer (receivers: and unpacked)
        if root:
            self.e.factor_end()
        else:
            self.e.adopt_commit()

    def step(self, step_index: int = 0):
        """factor (+ broadcast) -> mean -> this rank's k samples.  Returns (mean, samples)."""
        self._factor_and_share()
        mu = self.e.mean()
        nb = getattr(self.e, "batch", 1)
        first = (step_index * self.world + self.rank) * self.k * nb
        X = self.e.sample(self.k, mu, self.seed, first) if self.keep_samples else self.e.sample(self.k, mu, self.seed, first, keep=False)
        self._mark(3)
        return mu, X

    def phase_ms(self):
        """Device times of the last step on this rank (after a synchronisation): factorisation (root; receivers: 0),
        what the stream then still waited for the transfers, mean + samples."""
        if not self.timing or self._ev is None:
            return None
        e = self._ev
        return {"factor_ms": e[0].elapsed_time(e[1]), "transfer_wait_ms": e[1].elapsed_time(e[2]),
                "mean_and_samples_ms": e[2].elapsed_time(e[3])}

    def solves_per_step(self) -> int:
        """Posterior solves of one step over all ranks.  Shared factor (broadcast): per problem one
        mean and k samples per rank.  Replicated / independent problems: every rank handles its own
        batch of problems, each with its own mean and k samples."""
        nb = getattr(self.e, "batch", 1)
        if self.replicate:
            return self.world * nb * (1 + self.k)
        return nb * (1 + self.k * self.world)

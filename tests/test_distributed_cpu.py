"""world_size-2 gloo test of the N > 1 path: factor on rank 0, block-range broadcast, sample
sharding by Philox sample id (results independent of the number of ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    from importlib import import_module
    pkg = g.load_package()
    post = import_module(g.PKG_NAME + ".posterior")
    from tests.oracle_engine import OracleEngine
    w = pkg.workloads.make("darcy16")
    eng = OracleEngine(w)
    job = post.ShardedPosterior(eng, dist=dist, rank=rank, world=world, k_samples=5, seed=42, group=3)
    job.prepare()
    mu, X = job.step(0)
    # variance accumulators: each rank adds its samples, one all-reduce finishes the estimator
    acc = torch.from_numpy(((X - mu[:, None]) ** 2).sum(axis=1))
    dist.all_reduce(acc)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), mu=mu, X=X, Li=eng.Li.numpy(), L=eng.L.numpy(), C=eng.C.numpy(), acc=acc.numpy(),
             solves=job.solves_per_step())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sample_sharding(tmp_path, pkg):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, start_method="spawn")
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    # the factor rank 1 received is the one rank 0 computed: the inverses and the coupling blocks travel,
    # the triangular blocks themselves stay on the root
    assert np.array_equal(r0["Li"], r1["Li"]) and np.array_equal(r0["C"], r1["C"])
    assert np.abs(r0["L"]).max() > 0 and np.abs(r1["L"]).max() == 0
    assert np.array_equal(r0["mu"], r1["mu"])
    assert int(r0["solves"]) == 1 + 5 * 2
    # single-process run drawing the same 10 sample ids gives the same samples
    from importlib import import_module
    import __graft_entry__ as g
    post = import_module(g.PKG_NAME + ".posterior")
    from tests.oracle_engine import OracleEngine
    w = pkg.workloads.make("darcy16")
    eng = OracleEngine(w)
    job = post.ShardedPosterior(eng, k_samples=10, seed=42)
    job.prepare()
    mu, X = job.step(0)
    assert np.allclose(np.concatenate([r0["X"], r1["X"]], axis=1), X, rtol=1e-10, atol=1e-12)
    assert np.allclose(r0["acc"], ((X - mu[:, None]) ** 2).sum(axis=1), rtol=1e-12)
    assert np.array_equal(r0["acc"], r1["acc"])


def _gather_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as g
    from importlib import import_module
    pkg = g.load_package()
    post = import_module(g.PKG_NAME + ".posterior")
    from tests.oracle_engine import OracleGatherEngine
    w = pkg.workloads.make("darcy16")
    eng = OracleGatherEngine(w, batch_total=2, world=world, rank=rank, scales=[1.0, 1.5])
    job = post.ShardedPosterior(eng, dist=dist, rank=rank, world=world, k_samples=5, seed=42, group=3, share="allgather")
    job.prepare()
    mu, X = job.step(0)
    np.savez(os.path.join(outdir, f"g{rank}.npz"), mu=mu, X=X, Li=eng.Li.numpy(), C=eng.C.numpy(), Li_own=eng.Li_own.numpy(),
             solves=job.solves_per_step())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_of_a_shared_batch(tmp_path, pkg):
    """Round 4: the all-gather form of the shared-factor job -- every rank factors its share of the batch, block ranges are
    all-gathered, every rank takes the means of ALL posteriors and draws its own sample ids (world 2 over gloo, oracle
    numerics).  Same factors on both ranks, each rank factored only its own problem, samples independent of the rank count."""
    world = 2
    mp.start_processes(_gather_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, start_method="spawn")
    g0 = np.load(tmp_path / "g0.npz")
    g1 = np.load(tmp_path / "g1.npz")
    assert np.array_equal(g0["Li"], g1["Li"]) and np.array_equal(g0["C"], g1["C"]) and np.array_equal(g0["mu"], g1["mu"])
    # problem r of the gathered batch is what rank r factored (and nobody else did)
    assert np.array_equal(g0["Li"][0], g0["Li_own"][0]) and np.array_equal(g0["Li"][1], g1["Li_own"][0])
    assert not np.array_equal(g0["Li_own"], g1["Li_own"])
    assert int(g0["solves"]) == 2 * (1 + 5 * 2)
    # one process, the same protocol with a world of one: steps 0 and 1 draw the sample ids of ranks 0 and 1
    from importlib import import_module
    import __graft_entry__ as g
    post = import_module(g.PKG_NAME + ".posterior")
    from tests.oracle_engine import OracleGatherEngine
    w = pkg.workloads.make("darcy16")
    eng = OracleGatherEngine(w, batch_total=2, world=1, rank=0, scales=[1.0, 1.5])
    job = post.ShardedPosterior(eng, k_samples=5, seed=42, group=3, share="allgather")
    job.prepare()
    mu, X0 = job.step(0)
    _, X1 = job.step(1)
    assert np.allclose(mu, g0["mu"], rtol=1e-12, atol=1e-14)
    assert np.allclose(X0, g0["X"], rtol=1e-10, atol=1e-12) and np.allclose(X1, g1["X"], rtol=1e-10, atol=1e-12)
    # the two problems are different posteriors
    assert not np.allclose(mu[0], mu[1])

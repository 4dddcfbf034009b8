"""The N>1 path on CPU: two gloo processes, each owning a shard of the global
chain ids, build their summary blocks (here from the oracle's draws of exactly
those chains), all-gather them and aggregate; the result must equal the
single-process summaries over all chains."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

CHAINS_PER_RANK, NSWEEPS, SEED, P = 3, 25, 99, 12


def _case():
    from cases import regression_data, spike_slab_prior, suf_from_xy
    X, y, _ = regression_data(300, P, 3, seed=12)
    suf = suf_from_xy(X, y)
    prior = spike_slab_prior(suf, 3)
    g0 = np.zeros(P, np.uint8)
    g0[0] = 1
    return suf, prior, g0


def _block_for(chain_ids):
    """summary block of a set of global chains, laid out like the engine's"""
    from boom_amd import dist as bd
    from oracle_lib import Oracle, ssvs_options
    suf, prior, g0 = _case()
    O = Oracle()
    blk = np.zeros(bd.summary_block_size(P))
    blk[3 * P + bd.ACC_MIN_MARGIN] = np.inf
    for c in chain_ids:
        o = O.ssvs_run(suf, prior, ssvs_options(), ("philox", SEED, c), g0, NSWEEPS,
                       want_margin=True)
        blk[:P] += o["gamma"].sum(axis=0)
        blk[P:2 * P] += o["beta"].sum(axis=0)
        blk[2 * P:3 * P] += (o["beta"] ** 2).sum(axis=0)
        blk[3 * P + bd.ACC_SWEEPS] += NSWEEPS
        blk[3 * P + bd.ACC_SIGSQ] += o["sigsq"].sum()
        blk[3 * P + bd.ACC_K] += o["gamma"].sum()
        blk[3 * P + bd.ACC_MIN_MARGIN] = min(blk[3 * P + bd.ACC_MIN_MARGIN], o["min_margin"])
    return blk


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boom_amd import dist as bd
    off, n = bd.shard(CHAINS_PER_RANK, rank)
    blk = torch.from_numpy(_block_for(range(off, off + n)))
    blocks = bd.gather_blocks(blk, world)
    tmax = bd.max_over_ranks(1.0 + rank, world, "cpu")
    agg = bd.aggregate(blocks, P)
    if rank == 0:
        q.put((blocks.shape, tmax, agg))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    shape, tmax, agg = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    from boom_amd import dist as bd
    assert shape == (2, bd.summary_block_size(P))
    assert tmax == 2.0
    whole = bd.aggregate(_block_for(range(2 * CHAINS_PER_RANK))[None, :], P)
    assert agg["sweeps"] == whole["sweeps"] == 2 * CHAINS_PER_RANK * NSWEEPS
    for k in ("inclusion_prob", "beta_mean", "beta_second_moment"):
        assert np.allclose(agg[k], whole[k], rtol=1e-13, atol=1e-15)
    assert abs(agg["sigsq_mean"] - whole["sigsq_mean"]) < 1e-13
    assert agg["min_margin"] == whole["min_margin"]


def test_shard_ids_are_disjoint_and_cover():
    from boom_amd import dist as bd
    ids = []
    for r in range(8):
        off, n = bd.shard(1024, r)
        ids.extend(range(off, off + n))
    assert ids == list(range(8 * 1024))


# ---- config-4 data path: design matrix sharded by rows, ONE all-reduce of the
# sufficient-statistics block (here the partial blocks come from the oracle's
# NeRegSuf restatement; on GPUs from ba_suf_partial_device -- same layout)
N_ROWS, P_SUF = 257, 9


def _suf_partial_block(lo, hi):
    from boom_amd import dist as bd
    from cases import regression_data
    from oracle_lib import Oracle
    X, y, _ = regression_data(N_ROWS, P_SUF, 3, seed=77)
    s = Oracle().neregsuf(X[lo:hi], y[lo:hi])
    blk = np.zeros(bd.suf_block_size(P_SUF))
    pp = P_SUF * P_SUF
    blk[:pp] = s["xtx"].T.ravel()          # column-major
    blk[pp:pp + P_SUF] = s["xty"]
    blk[pp + P_SUF] = s["yty"]
    blk[pp + P_SUF + 1] = s["sumy"]
    blk[pp + P_SUF + 2:] = s["xsum"]
    return blk


def _suf_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from boom_amd import dist as bd
    lo, hi = bd.row_shard(N_ROWS, rank, world)
    blk = torch.from_numpy(_suf_partial_block(lo, hi))
    bd.reduce_suf_block(blk, world)
    q.put((rank, lo, hi, blk.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_sufficient_statistics_all_reduce():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_suf_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    from boom_amd import dist as bd
    from cases import regression_data
    # the shards tile the rows, every rank holds the same total, bit for bit
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == N_ROWS
    assert np.array_equal(got[0][3], got[1][3])
    X, y, _ = regression_data(N_ROWS, P_SUF, 3, seed=77)
    suf = bd.unpack_suf_block(got[0][3], P_SUF, N_ROWS)
    assert np.allclose(suf["xtx"], X.T @ X, rtol=1e-13, atol=1e-11)
    assert np.allclose(suf["xty"], X.T @ y, rtol=1e-13, atol=1e-11)
    assert abs(suf["yty"] - y @ y) < 1e-11 * (y @ y)
    assert abs(suf["sumy"] - y.sum()) < 1e-11
    assert np.allclose(suf["xsum"], X.sum(axis=0), rtol=1e-13, atol=1e-11)
    assert suf["n"] == N_ROWS

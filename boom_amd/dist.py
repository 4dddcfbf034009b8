"""Multi-GPU bookkeeping for the chain-sharded engine (one process per GPU).

Chains are independent, so the only distributed pieces are (i) which global
chain ids a rank owns and (ii) ONE all-gather of the posterior-summary block
at the end (RCCL over xGMI on GPUs; any torch.distributed backend works, the
CPU tests use gloo).  Nothing here touches the data path.
"""
import numpy as np

import time

SUMMARY_SCALARS = 16
# wall milliseconds of the last collectives of this process (device-synchronised on both
# sides when the tensors are on a GPU): bench.py prints them in the --gpus N line
timings = {}


def _timed(name, tensor, fn):
    import torch
    if tensor.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    if tensor.is_cuda:
        torch.cuda.synchronize()
    timings[name] = (time.perf_counter() - t0) * 1e3
    return out

ACC_SWEEPS, ACC_SIGSQ, ACC_SIGSQ2, ACC_K, ACC_ACCEPTS, ACC_PROPOSALS, ACC_MIN_MARGIN = range(7)


def shard(chains_per_rank, rank):
    """global chain ids of `rank` under weak scaling: [offset, offset+chains)"""
    return rank * chains_per_rank, chains_per_rank


def summary_block_size(p):
    return 3 * p + SUMMARY_SCALARS


def _staged(t):
    """the tensor a collective runs on: the tensor itself with RCCL, a host copy
    when the job's backend is gloo but the data sit on a GPU (single-GPU dry runs
    of the multi-rank path)"""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() == "gloo":
        return t.cpu(), True
    return t, False


def gather_blocks(block, world):
    """all-gather one summary block per rank -> (world, 3p+16) numpy array.
    `block` is a 1-d float64 torch tensor on the backend's device."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return block.detach().cpu().numpy()[None, :]
    blk, _ = _staged(block)
    out = [torch.empty_like(blk) for _ in range(world)]
    _timed("all_gather_ms", blk, lambda: dist.all_gather(out, blk))
    timings["all_gather_bytes"] = blk.numel() * blk.element_size() * world
    return torch.stack(out).cpu().numpy()


def max_over_ranks(value, world, device):
    import torch
    import torch.distributed as dist
    if world == 1:
        return float(value)
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def suf_block_size(p):
    """doubles in a sufficient-statistics block: X'X | X'y | y'y, sum y | col sums"""
    return p * p + 2 * p + 2


def row_shard(n, rank, world):
    """rows [lo, hi) of an n-row design matrix owned by `rank`"""
    return (n * rank) // world, (n * (rank + 1)) // world


def reduce_suf_block(block, world):
    """sum the ranks' partial sufficient-statistics blocks in place: ONE
    all-reduce (RCCL on GPUs; gloo in the CPU tests).  `block` is a 1-d float64
    torch tensor on the backend's device."""
    import torch.distributed as dist
    if world > 1:
        blk, staged = _staged(block)
        _timed("all_reduce_ms", blk, lambda: dist.all_reduce(blk, op=dist.ReduceOp.SUM))
        timings["all_reduce_bytes"] = blk.numel() * blk.element_size()
        if staged:
            block.copy_(blk)
    return block


def unpack_suf_block(block, p, n_total):
    """the block as the dict the tests / oracle use"""
    b = np.asarray(block, dtype=np.float64)
    pp = p * p
    return dict(xtx=b[:pp].reshape(p, p).T.copy(), xty=b[pp:pp + p].copy(),
                yty=float(b[pp + p]), sumy=float(b[pp + p + 1]),
                xsum=b[pp + p + 2:].copy(), n=float(n_total))


def build_suf_row_sharded(engine, X_local, y_local, n_total, world):
    """Config-4 data path: this rank's rows -> local MFMA syrk -> one all-reduce
    of (X'X | X'y | y'y | sum y | sum x) -> installed on the engine.  X_local is
    a column-major (p, n_local)-shaped CUDA tensor (n_local x p matrix), y_local
    a CUDA vector."""
    import torch
    p, n_local = X_local.shape
    block = torch.empty(suf_block_size(p), dtype=torch.float64, device=X_local.device)
    engine.suf_partial_device(n_local, p, X_local.data_ptr(), y_local.data_ptr(),
                              block.data_ptr())
    reduce_suf_block(block, world)
    if world > 1:
        torch.cuda.synchronize()
    engine.set_suf_from_block_device(n_total, p, block.data_ptr())
    return block


def aggregate(blocks, p):
    """whole-job posterior summaries from the gathered per-rank blocks"""
    blocks = np.asarray(blocks, dtype=np.float64)
    sc = blocks[:, 3 * p:]
    sweeps = float(sc[:, ACC_SWEEPS].sum())
    inc = blocks[:, :p].sum(axis=0)
    bsum = blocks[:, p:2 * p].sum(axis=0)
    bsq = blocks[:, 2 * p:3 * p].sum(axis=0)
    out = dict(
        sweeps=sweeps,
        inclusion_prob=inc / max(sweeps, 1.0),
        beta_mean=bsum / max(sweeps, 1.0),
        beta_second_moment=bsq / max(sweeps, 1.0),
        sigsq_mean=float(sc[:, ACC_SIGSQ].sum() / max(sweeps, 1.0)),
        mean_model_size=float(sc[:, ACC_K].sum() / max(sweeps, 1.0)),
        accepts=float(sc[:, ACC_ACCEPTS].sum()),
        proposals=float(sc[:, ACC_PROPOSALS].sum()),
        min_margin=float(sc[:, ACC_MIN_MARGIN].min()),
    )
    return out

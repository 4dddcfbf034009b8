"""Multi-GPU bookkeeping for the chain-sharded engine (one process per GPU).

Chains are independent, so the only distributed pieces are (i) which global
chain ids a rank owns and (ii) ONE all-gather of the posterior-summary block
at the end (RCCL over xGMI on GPUs; any torch.distributed backend works, the
CPU tests use gloo).  Nothing here touches the data path.
"""
import numpy as np

SUMMARY_SCALARS = 16
ACC_SWEEPS, ACC_SIGSQ, ACC_SIGSQ2, ACC_K, ACC_ACCEPTS, ACC_PROPOSALS, ACC_MIN_MARGIN = range(7)


def shard(chains_per_rank, rank):
    """global chain ids of `rank` under weak scaling: [offset, offset+chains)"""
    return rank * chains_per_rank, chains_per_rank


def summary_block_size(p):
    return 3 * p + SUMMARY_SCALARS


def gather_blocks(block, world):
    """all-gather one summary block per rank -> (world, 3p+16) numpy array.
    `block` is a 1-d float64 torch tensor on the backend's device."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return block.detach().cpu().numpy()[None, :]
    out = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(out, block)
    return torch.stack(out).cpu().numpy()


def max_over_ranks(value, world, device):
    import torch
    import torch.distributed as dist
    if world == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


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

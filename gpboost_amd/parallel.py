"""Multi-GPU composition of the Vecchia likelihood (one process per GPU, torch.distributed over RCCL).

The per-point work is independent: every rank keeps the coordinates, y and the neighbour
table replicated, evaluates a contiguous block of the Vecchia ordering, and ONE all-reduce of <= 7 doubles
per evaluation combines the partial sums -- no other data-path collective.  torch is used for the process
group only; the numbers come from lib_gpboost_amd.so.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous block of the ordering for `rank`; block edges are multiples of 16 (one workgroup = 16 points)."""
    per = -(-n // world)
    per = -(-per // 16) * 16
    i0 = min(rank * per, n)
    i1 = min(i0 + per, n)
    if rank == world - 1:
        i1 = n
    return i0, i1


def allreduce_terms(t):
    """Sum the partial terms over ranks (in place on `t`, a torch tensor on this rank's device or CPU)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def nll_from_terms(n, yPy, logdet, sigma2):
    """include/GPBoost/re_model_template.h:3132"""
    return yPy / 2. / sigma2 + logdet / 2. + n / 2. * (np.log(sigma2) + np.log(2 * np.pi))


def grad_from_terms(n, t7, sigma2):
    """include/GPBoost/re_model_template.h:1994,2004"""
    t7 = np.asarray(t7, dtype=np.float64)
    return np.array([-t7[0] / sigma2 / 2. + n / 2., t7[3] / sigma2 + t7[4], t7[5] / sigma2 + t7[6]])

"""Data-parallel communicator: the RCCL (torch.distributed backend "nccl" on ROCm) stand-in for the
mpi4py communicator the reference threads through `learn(..., comm=)` (ppo2/model.py:97-100,
129-131; common/mpi_adam_optimizer.py:18-51; common/mpi_util.py:15-26).

One process per GPU.  The only data-path collective is ONE in-place all-reduce(sum) of the flat
fp32 gradient per minibatch step over xGMI; weights/Adam slots are broadcast from rank 0 once after
initialisation.  The class is backend-agnostic so the N>1 logic is testable with gloo on CPU."""
import torch
import torch.distributed as dist


class Comm(object):
    def __init__(self, group=None):
        assert dist.is_initialized(), 'torch.distributed is not initialised'
        self.group = group

    def Get_size(self):
        return dist.get_world_size(self.group)

    def Get_rank(self):
        return dist.get_rank(self.group)

    def allreduce_sum_(self, t):
        """mpi_adam_optimizer.py:39: Allreduce(flat_grad, SUM), in place."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def total_weight(self, w):
        """mpi_adam_optimizer.py:25-27: sum of the per-rank weights (f32)."""
        dev = 'cuda' if dist.get_backend(self.group) == 'nccl' else 'cpu'
        t = torch.tensor([float(w)], dtype=torch.float32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def bcast_(self, t, root=0):
        """mpi_util.py:15-26 sync_from_root."""
        dist.broadcast(t, src=root, group=self.group)
        return t

    def check_synced(self, t):
        """mpi_adam_optimizer.py:53-68: every rank must hold the same value (here: min == max of a
        checksum instead of a gather to rank 0)."""
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        assert torch.equal(lo, hi), 'ranks have different weights: {} vs {}'.format(lo, hi)


def default_comm():
    """`if MPI is not None and comm is None: comm = MPI.COMM_WORLD` (ppo2/model.py:31-32)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return Comm()
    return None


def shard_envs(num_envs, rank, world):
    """Contiguous environment shard [lo, hi) owned by `rank` (GPU g owns envs
    [g*N/G, (g+1)*N/G)); remainders go to the lowest ranks so sizes differ by at most one."""
    base, extra = divmod(int(num_envs), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)

"""Data-parallel communicator: the RCCL (torch.distributed backend "nccl" on ROCm) stand-in for the
mpi4py communicator the reference threads through `learn(..., comm=)` (ppo2/model.py:97-100,
129-131; common/mpi_adam_optimizer.py:18-51; common/mpi_util.py:15-26).

One process per GPU.  The only data-path collective is ONE in-place all-reduce(sum) of the flat
fp32 gradient per minibatch step over xGMI; weights/Adam slots are broadcast from rank 0 once after
initialisation.  The class is backend-agnostic so the N>1 logic is testable with gloo on CPU."""
import ctypes
import os

import torch
import torch.distributed as dist


class Comm(object):
    """`native`: handle of the in-library RCCL communicator (`mrl_comm`, include/mrl.h).  When present the
    gradient all-reduce is issued by libmrl.so itself from inside the backward pass (Model attaches it to its
    mrl_model) and the broadcasts below go through it too; torch.distributed is then only the bootstrap channel
    for the RCCL unique id and the control plane (total weight, check_synced)."""

    def __init__(self, group=None):
        assert dist.is_initialized(), 'torch.distributed is not initialised'
        self.group = group
        self.native = None
        self.native_error = None

    def enable_native(self):
        """Create the mrl_comm of this rank: rank 0 draws the RCCL unique id, torch.distributed ships it.

        Collective and fail-safe: every rank takes part in the id broadcast (rank 0 sends None when it could not
        draw one); an all-reduce(MIN) of a success flag decides BEFORE the (collective) ncclCommInitRank whether all
        ranks enter it, a second one agrees on its outcome, a third on a probe all-reduce through the new communicator
        -- when ANY rank failed anywhere (library not loadable, no id, ncclCommInitRank error, wrong probe result) every
        rank drops its handle and the collectives stay in torch.distributed; nobody blocks alone.
        Returns True when the in-library communicator is in use on all ranks."""
        from .. import _lib
        size, rank = self.Get_size(), self.Get_rank()
        box, ok, why = [None], 1, ''
        lib = None
        try:
            lib = _lib.load()
            if rank == 0:
                buf = ctypes.create_string_buffer(128)
                _lib.check(lib.mrl_comm_unique_id(buf), 'mrl_comm_unique_id')
                box[0] = bytes(buf.raw)
        except Exception as exc:                     # the broadcast below must still run on every rank
            ok, why = 0, repr(exc)
        dist.broadcast_object_list(box, src=0, group=self.group)
        dev = 'cuda' if dist.get_backend(self.group) == 'nccl' else 'cpu'

        def agree(ok):
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item())

        # phase 0: ncclCommInitRank is itself collective (it waits for `size` participants), so NO rank may enter it
        # unless EVERY rank loaded the library and received an id -- a rank that skipped it would leave the others blocked
        if box[0] is None:
            ok, why = 0, why or 'rank 0 could not draw an RCCL unique id'
        if not agree(ok):
            self.native = None
            self.native_error = why or 'another rank could not load the library'
            return False
        # phase 1: communicator creation, agreed on
        h = None
        try:
            h = ctypes.c_void_p()
            _lib.check(lib.mrl_comm_create(box[0], size, rank, ctypes.byref(h)), 'mrl_comm_create')
        except Exception as exc:
            ok, why, h = 0, repr(exc), None
        flag = torch.tensor([agree(ok)], dtype=torch.int32)
        if int(flag.item()) == 1 and dev == 'cuda':
            # phase 2: the first collective through the new communicator, checked, and agreed on again
            try:
                probe = torch.ones(1024, dtype=torch.float32, device='cuda')
                _lib.check(lib.mrl_allreduce_grads(h, _lib.ptr(probe), probe.numel(), _lib.stream_ptr()),
                           'mrl_allreduce_grads (probe)')
                torch.cuda.synchronize()
                if not bool((probe == float(size)).all().item()):
                    raise RuntimeError('probe all-reduce returned %r, expected %d' % (float(probe[0].item()), size))
            except Exception as exc:
                ok, why = 0, repr(exc)
            flag = torch.tensor([agree(ok)], dtype=torch.int32)
        if int(flag.item()) == 1:
            self.native = h
            self.native_error = None
            return True
        if h is not None:
            try:
                lib.mrl_comm_destroy(h)
            except Exception:
                pass
        self.native = None
        self.native_error = why or 'another rank could not create its communicator'
        return False

    def close(self):
        """destroy the in-library communicator (idempotent)"""
        h, self.native = self.native, None
        if h is not None:
            try:
                from .. import _lib
                _lib.load().mrl_comm_destroy(h)
            except Exception:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Get_size(self):
        return dist.get_world_size(self.group)

    def Get_rank(self):
        return dist.get_rank(self.group)

    def allreduce_sum_(self, t):
        """mpi_adam_optimizer.py:39: Allreduce(flat_grad, SUM), in place."""
        if self.native is not None and t.is_cuda and t.dtype == torch.float32:
            from .. import _lib
            _lib.check(_lib.load().mrl_allreduce_grads(self.native, _lib.ptr(t), t.numel(), _lib.stream_ptr()),
                       'mrl_allreduce_grads')
            return t
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
        if self.native is not None and t.is_cuda:
            from .. import _lib
            _lib.check(_lib.load().mrl_broadcast_state(self.native, _lib.ptr(t), t.numel() * t.element_size(), int(root),
                                                       _lib.stream_ptr()), 'mrl_broadcast_state')
            return t
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
    """`if MPI is not None and comm is None: comm = MPI.COMM_WORLD` (ppo2/model.py:31-32).  Over the RCCL backend
    the communicator is the in-library one (MRL_NATIVE_COMM=0 keeps the collectives in torch.distributed)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        comm = Comm()
        if dist.get_backend() == 'nccl' and os.environ.get('MRL_NATIVE_COMM', '1') != '0':
            comm.enable_native()
        return comm
    return None


def shard_envs(num_envs, rank, world):
    """Contiguous environment shard [lo, hi) owned by `rank` (GPU g owns envs
    [g*N/G, (g+1)*N/G)); remainders go to the lowest ranks so sizes differ by at most one."""
    base, extra = divmod(int(num_envs), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)

"""MicrobatchedModel with the reference's plug-point signature (ppo2/microbatched_model.py:5-75):

    learn(..., model_fn=functools.partial(MicrobatchedModel, microbatch_size=2))

A minibatch step becomes nmicrobatches gradient passes of `microbatch_size` samples -- the workspace (activations and
their gradients, ~173 KB/sample for NatureCNN) is sized for ONE microbatch, which is the reason to use it on a
device: a minibatch larger than HBM allows.  Semantics follow the reference exactly:
  * advantages are normalised once over the whole minibatch (:40-43); every slice loss is a mean over its slice;
  * each slice gradient is [rank-averaged and] clipped by its own global norm before it is summed -- the
    reference adds up `self.grads`, which model.py:105-112 defines post-clip (:57-64);
  * the sum is divided by nmicrobatches and applied by Adam with no further clip (:66-70);
  * returned stats are the mean of the per-slice stats (:72).
Kernels: mrl_model_grad_micro, mrl_clip_accumulate, mrl_adam_clip_step (include/mrl.h).
"""
import numpy as np
import torch

from .. import ops
from .model import Model


class MicrobatchedModel(Model):
    def __init__(self, *, policy, ob_space, ac_space, nbatch_act, nbatch_train, nsteps, ent_coef, vf_coef,
                 max_grad_norm, mpi_rank_weight=1, comm=None, microbatch_size, **kw):
        if nbatch_train % microbatch_size != 0:
            raise AssertionError('microbatch_size ({}) should divide nbatch_train ({}) evenly'.format(
                microbatch_size, nbatch_train))
        self.nmicrobatches = nbatch_train // microbatch_size
        self.minibatch_size = nbatch_train
        # like the reference, the parent is built for microbatch-sized training batches
        Model.__init__(self, policy=policy, ob_space=ob_space, ac_space=ac_space, nbatch_act=nbatch_act,
                       nbatch_train=microbatch_size, nsteps=nsteps, ent_coef=ent_coef, vf_coef=vf_coef,
                       max_grad_norm=max_grad_norm, mpi_rank_weight=mpi_rank_weight, comm=comm,
                       microbatch_size=microbatch_size, **kw)
        self.grad_sum = torch.zeros_like(self.grads)

    def _micro_steps(self, lr, B, grad_call):
        """grad_call(mb0, mbn, stats_row) fills self.grads for one slice; returns the averaged stats (host list)."""
        assert B % self.microbatch_size == 0, 'minibatch of {} samples is not a multiple of microbatch_size {}'.format(
            B, self.microbatch_size)
        nmicro = B // self.microbatch_size
        stats = torch.empty((nmicro, 5), dtype=torch.float32, device=self.device)
        for k in range(nmicro):
            grad_call(k * self.microbatch_size, self.microbatch_size, stats[k])
            if self.multi and not self.native_dp:     # native: mrl_model_grad_micro already summed over ranks
                if self.mpi_rank_weight != 1:
                    self.grads.mul_(float(self.mpi_rank_weight))
                self.comm.allreduce_sum_(self.grads)
            ops.clip_accumulate(self.grads, self.grad_sum, self.max_grad_norm, self.total_weight, k == 0, self._scratch)
        one = np.float32(1)
        alpha = np.float32(lr) * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)
        # sum / nmicrobatches, then the optimizer's apply op alone (no clip): microbatched_model.py:66-70
        ops.adam_clip_step(self.params, self.grad_sum, self.adam_m, self.adam_v, alpha, self.beta1, self.beta2,
                           self.epsilon, None, float(nmicro), self._scratch, None)
        self.beta1_power = np.float32(self.beta1_power * self.beta1)
        self.beta2_power = np.float32(self.beta2_power * self.beta2)
        self._train_calls += 1
        return stats

    def train_indexed(self, lr, cliprange, rollout, idx_dev, stats_out=None, states=None):
        assert states is None, 'microbatches with recurrent models are not supported yet'     # microbatched_model.py:38
        B = idx_dev.numel()

        def call(mb0, mbn, row):
            self.dm.grad_micro(self.params, rollout.obs, rollout.actions, rollout.returns, rollout.values,
                               rollout.neglogpacs, idx_dev, B, mb0, mbn, rollout.T, rollout.N, cliprange, self.ent_coef,
                               self.vf_coef, self.grads, row)
        stats = self._micro_steps(lr, B, call).mean(dim=0)
        if stats_out is not None:
            stats_out.copy_(stats)
            return stats_out
        return stats

    def train_epoch(self, lr, cliprange, rollout, inds_dev):
        B = self.minibatch_size
        return torch.stack([self.train_indexed(lr, cliprange, rollout, inds_dev[k * B:(k + 1) * B])
                            for k in range(inds_dev.numel() // B)])

    def train(self, lr, cliprange, obs, returns, masks, actions, values, neglogpacs, states=None):
        assert states is None, 'microbatches with recurrent models are not supported yet'
        obs = self._to_dev_obs(obs)
        B = obs.shape[0]
        act = self._field(actions, self.dm.action_dtype)
        ret, val, nlp = (self._field(x, torch.float32) for x in (returns, values, neglogpacs))

        def call(mb0, mbn, row):
            self.dm.grad_micro(self.params, obs, act, ret, val, nlp, None, B, mb0, mbn, 1, 1, cliprange, self.ent_coef,
                               self.vf_coef, self.grads, row)
        stats = self._micro_steps(lr, B, call).cpu().numpy()
        return np.mean(stats, axis=0).tolist()

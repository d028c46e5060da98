"""PPO2 driver with the reference's signature and plug points (ppo2/ppo2.py:21-218):

    learn(*, network, env, total_timesteps, eval_env=None, seed=None, nsteps=2048, ent_coef=0.0,
          lr=3e-4, vf_coef=0.5, max_grad_norm=0.5, gamma=0.99, lam=0.95, log_interval=10,
          nminibatches=4, noptepochs=4, cliprange=0.2, save_interval=0, load_path=None,
          model_fn=None, update_fn=None, init_fn=None, mpi_rank_weight=1, comm=None,
          **network_kwargs) -> model

One update = Runner.run() (rollout into HBM + GAE kernel) followed by noptepochs x nminibatches
minibatch steps.  Permutations come from np.random.shuffle on the host exactly like ppo2.py:157-160
(same global NumPy stream as the reference -> identical minibatches for a given seed); the
indices are uploaded once per epoch and the gather happens inside the first-layer loaders.
Loss statistics stay on the device until the end of the update (one host sync per update).
"""
import os
import os.path as osp
import time
from collections import deque

import numpy as np
import torch

from .. import logger
from ..common import explained_variance, set_global_seeds
from ..common.math_util import safemean
from ..common.policies import build_policy
from ..common.schedules import constfn
from .runner import Runner, RolloutField


def _is_root(comm):
    if comm is not None:
        return comm.Get_rank() == 0
    import torch.distributed as dist
    return (not (dist.is_available() and dist.is_initialized())) or dist.get_rank() == 0


def _as_schedule(x, what):
    """lr / cliprange may be a float (constant) or a function of the remaining-progress fraction."""
    if isinstance(x, float):
        return constfn(x)
    if not callable(x):
        raise AssertionError('%s must be a float or a callable of frac' % what)
    return x


class _Bookkeeping(object):
    """Everything learn() does besides stepping and training: episode-info windows, the log record
    of one update (same keys as the reference, ppo2.py:191-209) and checkpointing."""

    def __init__(self, nsteps, nbatch, loss_names, root, with_eval):
        self.nsteps, self.nbatch, self.loss_names, self.root = nsteps, nbatch, loss_names, root
        self.train_eps = deque(maxlen=100)
        self.eval_eps = deque(maxlen=100) if with_eval else None
        self.t0 = time.perf_counter()

    def record(self, update, fps, values, returns, lossvals, now):
        host = lambda f: f.to_numpy() if isinstance(f, RolloutField) else np.asarray(f)
        kv = [('misc/serial_timesteps', update * self.nsteps), ('misc/nupdates', update),
              ('misc/total_timesteps', update * self.nbatch), ('fps', fps),
              ('misc/explained_variance', float(explained_variance(host(values), host(returns)))),
              ('eprewmean', safemean([e['r'] for e in self.train_eps])),
              ('eplenmean', safemean([e['l'] for e in self.train_eps]))]
        if self.eval_eps is not None:
            kv += [('eval_eprewmean', safemean([e['r'] for e in self.eval_eps])),
                   ('eval_eplenmean', safemean([e['l'] for e in self.eval_eps]))]
        kv.append(('misc/time_elapsed', now - self.t0))
        kv += [('loss/' + name, val) for val, name in zip(lossvals, self.loss_names)]
        for k, v in kv:
            logger.logkv(k, v)
        if self.root:
            logger.dumpkvs()
        else:
            logger.getkvs().clear()

    def checkpoint(self, model, update):
        folder = osp.join(logger.get_dir(), 'checkpoints')
        os.makedirs(folder, exist_ok=True)
        target = osp.join(folder, '%.5i' % update)
        print('Saving to', target)
        model.save(target)


def learn(*, network, env, total_timesteps, eval_env=None, seed=None, nsteps=2048, ent_coef=0.0, lr=3e-4,
          vf_coef=0.5, max_grad_norm=0.5, gamma=0.99, lam=0.95,
          log_interval=10, nminibatches=4, noptepochs=4, cliprange=0.2,
          save_interval=0, load_path=None, model_fn=None, update_fn=None, init_fn=None, mpi_rank_weight=1,
          comm=None, **network_kwargs):
    set_global_seeds(seed)
    lr, cliprange = _as_schedule(lr, 'lr'), _as_schedule(cliprange, 'cliprange')
    total_timesteps = int(total_timesteps)

    policy = build_policy(env, network, **network_kwargs)
    nenvs = env.num_envs
    nbatch = nenvs * nsteps
    nbatch_train = nbatch // nminibatches
    root = _is_root(comm)

    if model_fn is None:
        from .model import Model as model_fn
    model = model_fn(policy=policy, ob_space=env.observation_space, ac_space=env.action_space, nbatch_act=nenvs,
                     nbatch_train=nbatch_train, nsteps=nsteps, ent_coef=ent_coef, vf_coef=vf_coef,
                     max_grad_norm=max_grad_norm, comm=comm, mpi_rank_weight=mpi_rank_weight)
    if load_path is not None:
        model.load(load_path)

    # a model with train_indexed() reads the HBM rollout in place (our Model); any other model_fn product
    # (the reference's protocol only) gets host arrays and the reference's gather loop
    fast = hasattr(model, 'train_indexed')
    runner = Runner(env=env, model=model, nsteps=nsteps, gamma=gamma, lam=lam, return_host=(False if fast else None))
    eval_runner = None
    if eval_env is not None:
        eval_runner = Runner(env=eval_env, model=model, nsteps=nsteps, gamma=gamma, lam=lam, return_host=False)
    book = _Bookkeeping(nsteps, nbatch, model.loss_names, root, eval_env is not None)

    if init_fn is not None:
        init_fn()

    nupdates = total_timesteps // nbatch          # 0 -> build/load only (used by the reference's tests)
    for update in range(1, nupdates + 1):
        assert nbatch % nminibatches == 0
        tstart = time.perf_counter()
        frac = 1.0 - (update - 1.0) / nupdates
        lrnow, cliprangenow = lr(frac), cliprange(frac)
        chatty = update % log_interval == 0

        if chatty and root:
            logger.info('Stepping environment...')
        obs, returns, masks, actions, values, neglogpacs, states, epinfos = runner.run()
        book.train_eps.extend(epinfos)
        if eval_runner is not None:
            book.eval_eps.extend(eval_runner.run()[-1])
        if chatty and root:
            logger.info('Done.')
        step_stats = []
        if states is not None:
            # recurrent version (ppo2.py:167-180): minibatches are made of whole env trajectories -- the ENV indices are
            # shuffled (global NumPy stream), every minibatch takes nenvs // nminibatches envs with all their steps in order
            assert nenvs % nminibatches == 0
            envsperbatch = nenvs // nminibatches
            envinds = np.arange(nenvs)
            flatinds = np.arange(nenvs * nsteps).reshape(nenvs, nsteps)
            for _ in range(noptepochs):
                np.random.shuffle(envinds)
                for start in range(0, nenvs, envsperbatch):
                    mbenvinds = envinds[start:start + envsperbatch]
                    mbflatinds = flatinds[mbenvinds].ravel()
                    if fast:
                        mbstates = states[torch.from_numpy(mbenvinds).to(states.device)]
                        step_stats.append(model.train_indexed(lrnow, cliprangenow, runner.rollout,
                                                              torch.from_numpy(mbflatinds).to(model.device), states=mbstates))
                    else:
                        slices = (arr[mbflatinds] for arr in (obs, returns, masks, actions, values, neglogpacs))
                        step_stats.append(model.train(lrnow, cliprangenow, *slices, states[mbenvinds]))

        # noptepochs x nminibatches steps; permutations from the global NumPy stream (ppo2.py:157-160)
        inds = np.arange(nbatch)
        for _ in range(noptepochs if states is None else 0):
            np.random.shuffle(inds)
            if fast:
                # (asynchronous upload where the model offers one: the host goes on to draw the next permutation while the
                # device is still working through this epoch)
                inds_dev = (model.indices_to_device(inds) if hasattr(model, 'indices_to_device')
                            else torch.from_numpy(inds).to(model.device))
                if hasattr(model, 'train_epoch'):           # one replayable launch graph per epoch where that pays
                    step_stats.extend(model.train_epoch(lrnow, cliprangenow, runner.rollout, inds_dev).unbind(0))
                else:
                    for lo in range(0, nbatch, nbatch_train):
                        step_stats.append(model.train_indexed(lrnow, cliprangenow, runner.rollout,
                                                              inds_dev[lo:lo + nbatch_train]))
            else:
                for lo in range(0, nbatch, nbatch_train):
                    pick = inds[lo:lo + nbatch_train]
                    step_stats.append(model.train(lrnow, cliprangenow, obs[pick], returns[pick], masks[pick],
                                                  actions[pick], values[pick], neglogpacs[pick]))
        if fast:
            lossvals = torch.stack(step_stats).mean(dim=0).cpu().numpy()     # the update's one sync point
        else:
            lossvals = np.mean(step_stats, axis=0)
        tnow = time.perf_counter()
        fps = int(nbatch / (tnow - tstart))

        if update_fn is not None:
            update_fn(update)
        if chatty or update == 1:
            book.record(update, fps, values, returns, lossvals, tnow)
        if save_interval and (update % save_interval == 0 or update == 1) and logger.get_dir() and root:
            book.checkpoint(model, update)

    return model

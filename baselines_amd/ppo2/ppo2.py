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


def learn(*, network, env, total_timesteps, eval_env=None, seed=None, nsteps=2048, ent_coef=0.0, lr=3e-4,
          vf_coef=0.5, max_grad_norm=0.5, gamma=0.99, lam=0.95,
          log_interval=10, nminibatches=4, noptepochs=4, cliprange=0.2,
          save_interval=0, load_path=None, model_fn=None, update_fn=None, init_fn=None, mpi_rank_weight=1,
          comm=None, **network_kwargs):
    set_global_seeds(seed)

    if isinstance(lr, float):
        lr = constfn(lr)
    else:
        assert callable(lr)
    if isinstance(cliprange, float):
        cliprange = constfn(cliprange)
    else:
        assert callable(cliprange)
    total_timesteps = int(total_timesteps)

    policy = build_policy(env, network, **network_kwargs)

    nenvs = env.num_envs
    ob_space = env.observation_space
    ac_space = env.action_space
    nbatch = nenvs * nsteps
    nbatch_train = nbatch // nminibatches
    is_mpi_root = _is_root(comm)

    if model_fn is None:
        from .model import Model
        model_fn = Model

    model = model_fn(policy=policy, ob_space=ob_space, ac_space=ac_space, nbatch_act=nenvs,
                     nbatch_train=nbatch_train, nsteps=nsteps, ent_coef=ent_coef, vf_coef=vf_coef,
                     max_grad_norm=max_grad_norm, comm=comm, mpi_rank_weight=mpi_rank_weight)

    if load_path is not None:
        model.load(load_path)
    fast = hasattr(model, 'train_indexed')
    runner = Runner(env=env, model=model, nsteps=nsteps, gamma=gamma, lam=lam,
                    return_host=(False if fast else None))
    if eval_env is not None:
        eval_runner = Runner(env=eval_env, model=model, nsteps=nsteps, gamma=gamma, lam=lam, return_host=False)

    epinfobuf = deque(maxlen=100)
    if eval_env is not None:
        eval_epinfobuf = deque(maxlen=100)

    if init_fn is not None:
        init_fn()

    tfirststart = time.perf_counter()

    nupdates = total_timesteps // nbatch
    for update in range(1, nupdates + 1):
        assert nbatch % nminibatches == 0
        tstart = time.perf_counter()
        frac = 1.0 - (update - 1.0) / nupdates
        lrnow = lr(frac)
        cliprangenow = cliprange(frac)

        if update % log_interval == 0 and is_mpi_root:
            logger.info('Stepping environment...')

        obs, returns, masks, actions, values, neglogpacs, states, epinfos = runner.run()
        if eval_env is not None:
            eval_epinfos = eval_runner.run()[-1]

        if update % log_interval == 0 and is_mpi_root:
            logger.info('Done.')

        epinfobuf.extend(epinfos)
        if eval_env is not None:
            eval_epinfobuf.extend(eval_epinfos)

        mblossvals = []
        assert states is None, 'recurrent policies are outside the supported hot path (SURVEY.md 8 f4)'
        inds = np.arange(nbatch)
        for _ in range(noptepochs):
            np.random.shuffle(inds)
            if fast:
                inds_dev = torch.from_numpy(inds).to(model.device)
            for start in range(0, nbatch, nbatch_train):
                end = start + nbatch_train
                if fast:
                    mblossvals.append(model.train_indexed(lrnow, cliprangenow, runner.rollout, inds_dev[start:end]))
                else:
                    mbinds = inds[start:end]
                    slices = (arr[mbinds] for arr in (obs, returns, masks, actions, values, neglogpacs))
                    mblossvals.append(model.train(lrnow, cliprangenow, *slices))

        if fast:
            lossvals = torch.stack(mblossvals).mean(dim=0).cpu().numpy()     # the update's one sync point
        else:
            lossvals = np.mean(mblossvals, axis=0)
        tnow = time.perf_counter()
        fps = int(nbatch / (tnow - tstart))

        if update_fn is not None:
            update_fn(update)

        if update % log_interval == 0 or update == 1:
            v_host = values.to_numpy() if isinstance(values, RolloutField) else np.asarray(values)
            r_host = returns.to_numpy() if isinstance(returns, RolloutField) else np.asarray(returns)
            ev = explained_variance(v_host, r_host)
            logger.logkv('misc/serial_timesteps', update * nsteps)
            logger.logkv('misc/nupdates', update)
            logger.logkv('misc/total_timesteps', update * nbatch)
            logger.logkv('fps', fps)
            logger.logkv('misc/explained_variance', float(ev))
            logger.logkv('eprewmean', safemean([epinfo['r'] for epinfo in epinfobuf]))
            logger.logkv('eplenmean', safemean([epinfo['l'] for epinfo in epinfobuf]))
            if eval_env is not None:
                logger.logkv('eval_eprewmean', safemean([epinfo['r'] for epinfo in eval_epinfobuf]))
                logger.logkv('eval_eplenmean', safemean([epinfo['l'] for epinfo in eval_epinfobuf]))
            logger.logkv('misc/time_elapsed', tnow - tfirststart)
            for (lossval, lossname) in zip(lossvals, model.loss_names):
                logger.logkv('loss/' + lossname, lossval)
            if is_mpi_root:
                logger.dumpkvs()
            else:
                logger.getkvs().clear()
        if save_interval and (update % save_interval == 0 or update == 1) and logger.get_dir() and is_mpi_root:
            checkdir = osp.join(logger.get_dir(), 'checkpoints')
            os.makedirs(checkdir, exist_ok=True)
            savepath = osp.join(checkdir, '%.5i' % update)
            print('Saving to', savepath)
            model.save(savepath)

    return model

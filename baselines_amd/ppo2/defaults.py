"""Default PPO2 hyper-parameters per environment family.

`baselines.run` looks up `<alg>.defaults.<env_type>()` (run.py:169-178), so the module keeps the
reference's function names (ppo2/defaults.py:1-25); the values are the published ones and are what
bench.py uses for the Atari- and MuJoCo-shaped workloads.
"""

_COMMON = {'lam': 0.95, 'gamma': 0.99, 'log_interval': 1}

_FAMILY = {
    # family: (nsteps, nminibatches, noptepochs, ent_coef, base_lr, cliprange, extra)
    'mujoco': (2048, 32, 10, 0.0, 3e-4, 0.2, {'value_network': 'copy'}),
    'atari': (128, 4, 4, 0.01, 2.5e-4, 0.1, {}),
}


def _linear_decay(base):
    def schedule(frac):
        return base * frac
    return schedule


def _build(family):
    nsteps, nminibatches, noptepochs, ent_coef, base_lr, cliprange, extra = _FAMILY[family]
    cfg = dict(_COMMON, nsteps=nsteps, nminibatches=nminibatches, noptepochs=noptepochs, ent_coef=ent_coef,
               lr=_linear_decay(base_lr), cliprange=cliprange)
    cfg.update(extra)
    return cfg


def mujoco():
    return _build('mujoco')


def atari():
    return _build('atari')


def retro():
    return _build('atari')

"""Scalar schedules used by the drivers (reference: common/schedules.py:76-99 and the
`constfn` / `lr(frac)` convention of ppo2/ppo2.py:16-19,82-85)."""


def constfn(val):
    def f(_):
        return val
    return f


class ConstantSchedule(object):
    def __init__(self, value):
        self._v = value

    def value(self, t):
        return self._v


class LinearSchedule(object):
    """initial_p + min(t / schedule_timesteps, 1) * (final_p - initial_p)"""

    def __init__(self, schedule_timesteps, final_p, initial_p=1.0):
        self.schedule_timesteps = schedule_timesteps
        self.final_p = final_p
        self.initial_p = initial_p

    def value(self, t):
        fraction = min(float(t) / self.schedule_timesteps, 1.0)
        return self.initial_p + fraction * (self.final_p - self.initial_p)

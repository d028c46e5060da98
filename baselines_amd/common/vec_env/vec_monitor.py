"""VecMonitor: per-env episode return / length bookkeeping for host VecEnvs (reference:
common/vec_env/vec_monitor.py:7-55, writer format bench/monitor.py:100-121).

When an env reports done its info dict gains `info['episode'] = {'r': return, 'l': length, 't': seconds since start}` --
the record `Runner.run` collects (ppo2/runner.py:39-41) and `learn` averages into eprewmean / eplenmean
(ppo2.py:201-202).  With `filename` every finished episode is also appended to a `*.monitor.csv` in the reference's
wire format (a `# {json header}` line, the csv header `r,l,t[,extra keys]`, one row per episode), so the reference's
`load_results` / plotting tools read our runs.  Return accumulators are float32 and lengths int32 like the
reference's ('f' / 'i' arrays).  Device-resident envs (SyntheticVecEnv) keep these counters on the device themselves.
"""
import csv
import json
import os
import time
from collections import deque

import numpy as np

from .vec_env import VecEnvWrapper

MONITOR_EXT = 'monitor.csv'


class EpisodeLog(object):
    """append-only `monitor.csv` writer"""

    def __init__(self, filename, header, extra_keys=()):
        if not filename.endswith(MONITOR_EXT):
            filename = os.path.join(filename, MONITOR_EXT) if os.path.isdir(filename) else filename + '.' + MONITOR_EXT
        self.path = filename
        self._fh = open(filename, 'wt')
        self._fh.write('# {} \n'.format(json.dumps(header)))
        self._rows = csv.DictWriter(self._fh, fieldnames=('r', 'l', 't') + tuple(extra_keys))
        self._rows.writeheader()
        self._fh.flush()

    def write_row(self, record):
        self._rows.writerow(record)
        self._fh.flush()

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


class VecMonitor(VecEnvWrapper):
    def __init__(self, venv, filename=None, keep_buf=0, info_keywords=()):
        VecEnvWrapper.__init__(self, venv)
        if getattr(venv, 'device_resident', False):
            raise ValueError('device-resident envs report finished episodes themselves (fin_r / fin_l)')
        self.eprets = self.eplens = None
        self.epcount = 0
        self.tstart = time.time()
        self.info_keywords = tuple(info_keywords)
        self.results_writer = EpisodeLog(filename, {'t_start': self.tstart}, self.info_keywords) if filename else None
        self.keep_buf = keep_buf
        if keep_buf:
            self.epret_buf, self.eplen_buf = deque([], maxlen=keep_buf), deque([], maxlen=keep_buf)

    def reset(self):
        obs = self.venv.reset()
        self.eprets = np.zeros(self.num_envs, np.float32)
        self.eplens = np.zeros(self.num_envs, np.int32)
        return obs

    def step_wait(self):
        obs, rews, dones, infos = self.venv.step_wait()
        self.eprets += rews
        self.eplens += 1
        out = list(infos)
        for e in np.flatnonzero(np.asarray(dones)):
            record = {'r': self.eprets[e], 'l': self.eplens[e], 't': round(time.time() - self.tstart, 6)}
            merged = dict(infos[e])
            for key in self.info_keywords:
                record[key] = merged[key]
            merged['episode'] = record
            out[e] = merged
            if self.keep_buf:
                self.epret_buf.append(record['r'])
                self.eplen_buf.append(record['l'])
            self.epcount += 1
            self.eprets[e] = 0
            self.eplens[e] = 0
            if self.results_writer is not None:
                self.results_writer.write_row(record)
        return obs, rews, dones, out

    def close(self):
        if self.results_writer is not None:
            self.results_writer.close()
        return VecEnvWrapper.close(self)

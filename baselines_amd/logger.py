"""Key-value logger with the call surface the PPO2 driver uses (reference: logger.py:193-218,
372-399: logkv / logkv_mean / dumpkvs / info / get_dir / configure; key names kept:
fps, loss/*, misc/*, eprewmean, eplenmean).  Writers: a stdout table and, when a directory is
configured (argument or $OPENAI_LOGDIR), progress.csv + progress.json.  Observability only."""
import json
import os
import sys
from collections import OrderedDict, defaultdict


class _Logger(object):
    def __init__(self, dir=None, stdout=True, files=True):
        self.name2val = OrderedDict()
        self.name2cnt = defaultdict(int)
        self.dir = dir
        self.stdout = stdout
        self.files = files
        self._csv_keys = None
        if dir:
            os.makedirs(dir, exist_ok=True)

    def logkv(self, key, val):
        self.name2val[key] = val

    def logkv_mean(self, key, val):
        old, cnt = self.name2val.get(key, 0.0), self.name2cnt[key]
        self.name2val[key] = old * cnt / (cnt + 1) + val / (cnt + 1)
        self.name2cnt[key] = cnt + 1

    def dumpkvs(self):
        d = OrderedDict(self.name2val)
        if self.stdout and d:
            kw = max(len(k) for k in d)
            vals = {k: ('%-8.3g' % v if hasattr(v, '__float__') else str(v)) for k, v in d.items()}
            vw = max(len(v) for v in vals.values())
            dash = '-' * (kw + vw + 7)
            lines = [dash] + ['| %s | %s |' % (k.ljust(kw), vals[k].ljust(vw)) for k in sorted(d)] + [dash]
            sys.stdout.write('\n'.join(lines) + '\n')
            sys.stdout.flush()
        if self.dir and self.files and d:
            with open(os.path.join(self.dir, 'progress.json'), 'at') as f:
                f.write(json.dumps({k: (float(v) if hasattr(v, '__float__') else v) for k, v in d.items()}) + '\n')
            path = os.path.join(self.dir, 'progress.csv')
            new = [k for k in d if k not in (self._csv_keys or [])]
            if self._csv_keys is None:
                self._csv_keys = list(d.keys())
                with open(path, 'wt') as f:
                    f.write(','.join(self._csv_keys) + '\n')
            elif new:
                # keys that appear later extend the header: the file is rewritten with the wider header and the old rows
                # padded (the reference's CSVOutputFormat does the same, logger.py:104-121)
                with open(path, 'rt') as f:
                    rows = f.read().splitlines()[1:]
                self._csv_keys += new
                with open(path, 'wt') as f:
                    f.write(','.join(self._csv_keys) + '\n')
                    for r in rows:
                        f.write(r + ',' * len(new) + '\n')
            with open(path, 'at') as f:
                f.write(','.join(str(d.get(k, '')) for k in self._csv_keys) + '\n')
        self.name2val.clear()
        self.name2cnt.clear()
        return d


_current = _Logger(dir=os.environ.get('OPENAI_LOGDIR'), stdout=True)


def configure(dir=None, format_strs=None, comm=None, log_suffix=''):
    """format_strs == [] silences every writer, the files included (what run.py does for non-root ranks,
    run.py:209-214); None: stdout + files; a list: 'stdout' and/or 'csv'/'json' select the writers."""
    global _current
    _current = _Logger(dir=dir or os.environ.get('OPENAI_LOGDIR'),
                       stdout=(format_strs is None or 'stdout' in format_strs),
                       files=(format_strs is None or any(f in format_strs for f in ('csv', 'json', 'log'))))


def logkv(key, val):
    _current.logkv(key, val)


def logkv_mean(key, val):
    _current.logkv_mean(key, val)


def logkvs(d):
    for k, v in d.items():
        logkv(k, v)


def dumpkvs():
    return _current.dumpkvs()


def getkvs():
    return _current.name2val


def info(*args):
    if _current.stdout:
        print(*args)
        sys.stdout.flush()


def get_dir():
    return _current.dir


record_tabular = logkv
dump_tabular = dumpkvs

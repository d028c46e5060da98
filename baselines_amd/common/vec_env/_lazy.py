"""`torch` resolved on first attribute access, so that importing the vec_env package (env worker subprocesses do)
does not import torch."""


class _Lazy(object):
    def __init__(self, name):
        self._name, self._mod = name, None

    def __getattr__(self, attr):
        if self._mod is None:
            import importlib
            self._mod = importlib.import_module(self._name)
        return getattr(self._mod, attr)


torch = _Lazy('torch')

"""Functional stand-in for the little of `sacred` the reference's experiment scripts use at import time (TEST INFRASTRUCTURE
ONLY): `Experiment(...)` whose decorators hand the decorated function back unchanged -- so that `make_world`,
`trajectory_loss` and `run_world_fixed_dt` of experiments/trajectory_fitting/optim_sphere.py can be imported and called with
explicit arguments -- and whose `automain` does NOT run the experiment."""
import types


class _Observers(list):
    pass


class Experiment:
    def __init__(self, name=None, **_k):
        self.name = name
        self.observers = _Observers()
        self.captured_out_filter = None

    def _same(self, f=None, **_k):
        return f if f is not None else (lambda g: g)

    config = capture = command = named_config = main = _same

    def automain(self, f):       # the reference's scripts end in @ex.automain: importing them must not start a run
        return f

    def log_scalar(self, *a, **k):
        pass


observers = types.SimpleNamespace(FileStorageObserver=lambda *a, **k: None)
utils = types.ModuleType("sacred.utils")
utils.apply_backspaces_and_linefeeds = lambda s: s

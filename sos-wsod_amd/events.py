"""Minimal metric sink with the reference's EventStorage interface (uwsod/detectron2/utils/events.py:232-432).
The hot path calls get_event_storage().put_scalar(...) (roi_heads.py:370-373, fast_rcnn_oicr.py:245-256);
here scalars may be device tensors and are only materialised when a writer asks (no per-iteration host sync)."""
from collections import defaultdict
from contextlib import contextmanager

_CURRENT_STORAGE_STACK = []


def get_event_storage():
    assert len(_CURRENT_STORAGE_STACK), "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT_STORAGE_STACK[-1]


def has_event_storage():
    return len(_CURRENT_STORAGE_STACK) > 0


class EventStorage:
    def __init__(self, start_iter=0):
        self._history = defaultdict(list)
        self._latest = {}
        self._iter = start_iter
        self._prefix = ""

    def put_scalar(self, name, value, smoothing_hint=True):
        name = self._prefix + name
        self._latest[name] = value
        self._history[name].append((self._iter, value))

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint)

    def latest(self):
        return {k: (float(v) if hasattr(v, "item") else v) for k, v in self._latest.items()}

    def history(self, name):
        return self._history[name]

    def step(self):
        self._iter += 1

    @property
    def iter(self):
        return self._iter

    def __enter__(self):
        _CURRENT_STORAGE_STACK.append(self)
        return self

    def __exit__(self, *a):
        assert _CURRENT_STORAGE_STACK[-1] is self
        _CURRENT_STORAGE_STACK.pop()

    @contextmanager
    def name_scope(self, name):
        old = self._prefix
        self._prefix = name.rstrip("/") + "/"
        yield
        self._prefix = old

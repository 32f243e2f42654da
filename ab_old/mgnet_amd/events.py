"""Minimal event storage with detectron2's `get_event_storage().put_scalar(name, value)` surface (mg_net.py:362-371).
Scalars may be device tensors: they are kept as-is (no host sync inside forward) and resolved by `latest()`."""
import contextlib

_CURRENT = []


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._latest = {}

    def put_scalar(self, name, value, smoothing_hint=True):
        self._latest[name] = value

    def put_scalars(self, **kw):
        for k, v in kw.items():
            self.put_scalar(k, v)

    def latest(self):
        return {k: (float(v) if hasattr(v, "item") else v) for k, v in self._latest.items()}

    def step(self):
        self.iter += 1

    def __enter__(self):
        _CURRENT.append(self)
        return self

    def __exit__(self, *a):
        _CURRENT.pop()


_DEFAULT = EventStorage()


def get_event_storage():
    return _CURRENT[-1] if _CURRENT else _DEFAULT

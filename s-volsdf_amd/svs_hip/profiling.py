"""Per-launch timing of C-ABI entry points with HIP events on the stream each launch goes to (bench.py's roofline
block).  `torch.cuda.Event` only sees torch's current stream -- which is exactly the stream every wrapper in ops.py /
train.py passes as `hip_stream` -- so an event pair recorded around the ctypes call brackets the kernels of that call
on their own stream, also when ray groups run on side streams.  Eager launches only: events recorded while a hipGraph
is being captured cannot be read back.
"""
import torch

from . import lib as _lib


class LaunchTimer:
    """with LaunchTimer(["svs_sdf_outputs", ...]) as t: ...steps...;  t.summary() -> {name: dict(n, ms_mean, ms_total)}"""

    def __init__(self, names):
        self.names = list(names)
        self.events = {n: [] for n in self.names}
        self._orig = {}
        self.meta = {n: [] for n in self.names}

    def __enter__(self):
        L = _lib.load()
        for n in self.names:
            fn = getattr(L, n)
            self._orig[n] = fn
            setattr(L, n, self._wrap(n, fn))
        return self

    def _wrap(self, name, fn):
        def call(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            self.events[name].append((e0, e1))
            self.meta[name].append(args)
            return rc
        return call

    def __exit__(self, *exc):
        L = _lib.load()
        for n, fn in self._orig.items():
            setattr(L, n, fn)
        return False

    def times_ms(self, name):
        return [a.elapsed_time(b) for a, b in self.events[name]]

"""Feeds fixture random draws through torch's CPU generator call sites (test infrastructure)."""
import contextlib

import numpy as np
import torch


@contextlib.contextmanager
def inject_rng(draws):
    """Replaces the CPU-generator call sites of the model (reference order: jitter, u, randperm, randint, uniform_)
    so that they return the given draws; honours the out= form used by the pinned staging of the draws."""
    o = (torch.rand, torch.randperm, torch.randint, torch.Tensor.uniform_)
    q = [draws["jitter"], draws["u"]] + ([draws["jitter_bg"]] if "jitter_bg" in draws else [])

    def give(v, k):
        t = torch.from_numpy(np.ascontiguousarray(v))
        if k.get("out") is not None:
            k["out"].reshape(-1)[:t.numel()].copy_(t.reshape(-1))
            return k["out"]
        return t

    torch.rand = lambda *s, **k: give(q.pop(0), k)
    torch.randperm = lambda n, **k: give(draws["perm"], k)
    torch.randint = lambda h, s, **k: give(draws["eik_idx"], k)
    torch.Tensor.uniform_ = lambda self, a, b: self.copy_(torch.from_numpy(draws["eik_points"]))
    try:
        yield
    finally:
        torch.rand, torch.randperm, torch.randint, torch.Tensor.uniform_ = o

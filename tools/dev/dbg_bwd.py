import sys, os
sys.path[:0] = ["/root/repo/tests", "/root/repo/tests/golden", "/root/repo/oracle", "/root/repo/s-volsdf_amd"]
import numpy as np, torch
import test_gpu_backward as tb
src = open("/root/repo/tests/test_gpu_backward.py").read()
# run the test body with per-tensor printing: monkeypatch the asserts by wrapping
import pytest
class Ops: pass
from svs_hip import ops
dev = torch.device("cuda:0")
import re
body = tb.test_mlp_backward_vs_autograd
import inspect
code = inspect.getsource(body).replace("assert e < bound,", "print('%-28s %.2e' % (f'{l}.{name}', e)); assert e < 1.0,")
ns = dict(tb.__dict__)
exec(code, ns)
ns["test_mlp_backward_vs_autograd"](dev, ops, "f16x2")

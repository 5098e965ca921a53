"""Which torch copies does bench.py's planned step make per step?  Runs bench.py in-process with Tensor.copy_ / .to / .cuda
wrapped; prints the Python origin of every device copy made during TrainStep call number 400 (dev aid; GPU box)."""
import os, runpy, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "s-volsdf_amd"))
import torch
from svs_hip import trainer
state = {"calls": 0, "on": False}
orig_call = trainer.TrainStep.__call__
def call(self, *a, **k):
    state["calls"] += 1
    state["on"] = state["calls"] == 400
    try:
        return orig_call(self, *a, **k)
    finally:
        state["on"] = False
trainer.TrainStep.__call__ = call
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def spy(self, *a, **k):
        if state["on"]:
            print(name, tuple(self.shape), self.dtype, self.device, [type(x).__name__ if not torch.is_tensor(x) else (tuple(x.shape), str(x.device)) for x in a][:2],
                  "|", " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[-6:-1]), flush=True)
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, spy)
for n in ("copy_", "to", "cuda", "contiguous", "clone", "zero_", "fill_"):
    wrap(n)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-exact-f32", "--no-gpu-torch", "--no-volopt-loop", "--no-extras", "--no-kernel-timing",
            "--steps", "300", "--warmup", "10", "--settle", "0.2"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

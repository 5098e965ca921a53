"""How many steps of a VolOpt run on the analytic scene (tools/chamfer_parity.py, with the synthetic MVS prior) end with a
non-finite gradient that the fused clip + guard + Adam launch zeroes (svs_optim.hip: info[1]) -- dev aid.
    python tools/dev/count_drops.py [steps] [seed]"""
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import chamfer_parity as cp      # noqa: E402  (sets sys.path)
import numpy as np               # noqa: E402
import torch                     # noqa: E402
import test_gpu_volopt as tv     # noqa: E402

steps, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1500, int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.chdir(tempfile.mkdtemp(prefix="svs_drops_"))
torch.manual_seed(seed); random.seed(seed); np.random.seed(seed)
v = tv.build(cp.make_args(512, True))
v._preview = lambda *x, **k: None
v.save_checkpoints = lambda *x, **k: None
v.get_mvs_input(cp.build_prior(v.train_dataset))
drops, norms = [], []
orig = v.train_step


def spy(batch, use_mvs=False, **kw):
    out = orig(batch, use_mvs, **kw)
    info = v.step_fn.opt.info.cpu()
    norms.append(float(info[0]))
    if float(info[1]) != 0.0:
        drops.append(v.iter_step)
    return out


v.train_step = spy
v.run(opt_stepN=steps)
n = np.asarray(norms)
print(f"precision {os.environ.get('SVS_MLP_PRECISION', 'f16x2')}: {len(norms)} steps, {len(drops)} dropped (non-finite gradient) at {drops[:20]}; "
      f"gradient norm median {np.median(n):.3g}, max {np.nanmax(n):.3g}, non-finite norms {int((~np.isfinite(n)).sum())}")

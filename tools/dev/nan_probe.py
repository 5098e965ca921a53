"""Find the first non-finite model output of the bmvs long run (tools/long_run.py bmvs full) and what produced it."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tools")]
import long_run as lr
lr.MODEL, lr.MODE = "bmvs", "full"
from svs_hip import trainer
orig = trainer.TrainStep.__call__
state = dict(step=0, found=False)

def call(self, model_input, ground_truth, mvs=None, fast=1):
    out = orig(self, model_input, ground_truth, mvs=mvs, fast=fast)
    lo, mo = out
    if not state["found"]:
        bad = {k: int((~torch.isfinite(v)).sum()) for k, v in mo.items() if torch.is_tensor(v) and v.dtype.is_floating_point and not torch.isfinite(v).all()}
        lbad = {k: float(v) for k, v in lo.items() if not np.isfinite(float(v))}
        if bad or lbad:
            state["found"] = True
            print("step", state["step"], "non-finite outputs:", bad, "losses:", lbad, flush=True)
            for k in ("rgb_values",):
                v = mo[k]
                rows = torch.nonzero(~torch.isfinite(v).all(-1)).flatten()[:8]
                print(" rays", rows.tolist(), "uv", model_input["uv"][0][rows].tolist())
                for kk in ("weights", "depth_values", "depth_values_all"):
                    if kk in mo:
                        print(" ", kk, mo[kk][rows].flatten()[:12].tolist())
            res = self._results
            for gi, (l, o) in enumerate(res):
                for kk, vv in o.items():
                    if torch.is_tensor(vv) and vv.dtype.is_floating_point and not torch.isfinite(vv).all():
                        print("  group", gi, kk, "non-finite", int((~torch.isfinite(vv)).sum()))
            keeps = [h[0] for h in self._hold]
            for gi, k in enumerate(keeps):
                for kk, vv in k.items():
                    if torch.is_tensor(vv) and vv.dtype.is_floating_point and not torch.isfinite(vv).all():
                        print("  keep", gi, kk, "non-finite", int((~torch.isfinite(vv)).sum()), tuple(vv.shape))
            print(" beta", float(self.model.density.get_beta()))
    state["step"] += 1
    return out
trainer.TrainStep.__call__ = call
lr.run(None, int(sys.argv[1]) if len(sys.argv) > 1 else 1000)

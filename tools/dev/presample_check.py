"""tools/dev/presample_check.py: one default-mode (two ray groups) 1024-ray step; prints hashes of the step's forward results
(colours, depths, weights, loss terms -- the forward has no atomics: bits must not depend on SVS_PRESAMPLE).  Dev aid, GPU only."""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
import synth                                                             # noqa: E402
from svs_hip.trainer import TrainStep                                    # noqa: E402
from volsdf.model.loss import VolSDFLoss                                 # noqa: E402
from volsdf.utils.conf import dtu_model_conf                             # noqa: E402
from volsdf.model.network import VolSDFNetwork                           # noqa: E402

dev = torch.device("cuda:0")
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sha = lambda t: hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]
R = 1024
K, pose = synth.make_camera()
inp = {"intrinsics": G(K)[None], "uv": G(synth.make_uv(R, seed=4))[None], "pose": G(pose)[None]}
rs = np.random.default_rng(6)
gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32))}
views = synth.make_mvs_views(2)
mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"]), z_mvs=G(v["z_mvs"])) for v in views], same_view=0,
           img_res=(576, 768), inverse_depth=False)
m = VolSDFNetwork(dtu_model_conf())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}, strict=True)
m = m.to(dev).train()
loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                  anneal_rgb=200, gce=0.5, confi=1e-3)
loss.iter_step = 250
ts = TrainStep(m, loss, groups=[(0, 976), (976, 1024)])
torch.manual_seed(13)
print("SVS_PRESAMPLE =", os.environ.get("SVS_PRESAMPLE", "(default)"))
for i in range(2):
    lo, out = ts(inp, gt, mvs=mvs)
    torch.cuda.synchronize()
    print(f"step {i}: rgb {sha(out['rgb_values'])} depth {sha(out['depth_values'])} weights {sha(out['weights'])} "
          f"loss {float(lo['loss']).hex()} eik {float(lo['eikonal_loss']).hex()}")
    break   # (the second step starts from parameters that carry the first step's atomic-order noise)

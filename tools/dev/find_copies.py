"""Which tensor.copy_ / .to() calls does a planned train step still make?  (dev aid; run on the GPU box)"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from test_gpu_graph import _fresh, G
from svs_hip.trainer import TrainStep
dev = torch.device("cuda:0")
R = 256
K, pose = synth.make_camera()
rs = np.random.default_rng(1)
inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=3), dev)[None], "pose": G(pose, dev)[None]}
gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
m, loss = _fresh(dev, sys.argv[1] if len(sys.argv) > 1 else "dtu")
from volsdf.model.loss import VolSDFLoss
loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                  anneal_rgb=200, gce=0.5, confi=1e-3)
views = synth.make_mvs_views(3)
mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views], same_view=0,
           img_res=(576, 768), inverse_depth=False)
ts = TrainStep(m, loss, graph="plan")
for _ in range(260):
    ts(inp, gt, mvs=mvs)
orig = torch.Tensor.copy_
def spy(self, src, *a, **k):
    if self.is_cuda or src.is_cuda:
        print("copy_", tuple(self.shape), self.device, "<-", tuple(src.shape), src.device, "|", " <- ".join(f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]))
    return orig(self, src, *a, **k)
torch.Tensor.copy_ = spy
ts(inp, gt, mvs=mvs)
torch.Tensor.copy_ = orig
torch.cuda.synchronize()

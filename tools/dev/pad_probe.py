"""Padded batches (ray counts that are not a multiple of the kernels' tile) through eager launches and launch plans (dev aid;
run on the GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from test_gpu_graph import _fresh, G
from svs_hip.trainer import TrainStep
dev = torch.device("cuda:0")
for kind in ("dtu", "bmvs"):
    for R in (100, 250):
        K, pose = synth.make_camera()
        rs = np.random.default_rng(1)
        inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=3), dev)[None], "pose": G(pose, dev)[None]}
        gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
        res = {}
        for graph in (False, "plan"):
            m, loss = _fresh(dev, kind)
            ts = TrainStep(m, loss, graph=graph)
            torch.manual_seed(1)
            try:
                for step in range(4):
                    lo, out = ts(inp, gt)
                res[graph] = (float(lo["loss"]), out["rgb_values"].shape)
            except Exception as e:
                res[graph] = repr(e)[:300]
        print(kind, R, res)

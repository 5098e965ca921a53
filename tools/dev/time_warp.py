"""dev aid: time of the fused warp + variance (split-volume output) at the three stage shapes of config 3"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
import synth
from svs_hip import costvol
dev = torch.device("cuda:0")
feats, proj, depth_values = synth.make_mvs_sample(3, img_hw=(512, 640))
G = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for st, (D, s) in enumerate(((192, 4), (32, 2), (8, 1))):
    key = f"stage{st + 1}"
    fs = [G(f[key])[None] for f in feats]
    h, w = fs[0].shape[-2:]
    dv = torch.linspace(425.0, 935.0, D, device=dev).view(1, D, 1, 1).expand(1, D, h, w).contiguous()
    pm = G(proj[key])[None]
    for split in (True, False):
        ts = []
        for _ in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); costvol.warp_variance(fs, pm, dv, split=split); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"{os.environ.get('SVS_LIB_PATH', 'default')[-12:]:12s} {key} C={fs[0].shape[1]} split={split}: {min(ts):.3f} ms")

"""tools/dev/det_hash.py [steps]: SHA-256 of the gradients and parameters after a few DETERMINISTIC optimisation steps (both
models) and of a 1024-ray forward's outputs -- two builds of the library that claim bit-identical arithmetic (SVS_LIB_PATH
selects one) must print the same lines.  Dev aid, GPU only."""
import hashlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
import synth                                                             # noqa: E402
from svs_hip import lib as _lib                                          # noqa: E402
from svs_hip.trainer import TrainStep                                    # noqa: E402
from volsdf.model.loss import VolSDFLoss                                 # noqa: E402

dev = torch.device("cuda:0")
G = lambda a, d: torch.from_numpy(np.ascontiguousarray(a)).to(d)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16]


def run(kind, R):
    K, pose = synth.make_camera()
    inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=4), dev)[None], "pose": G(pose, dev)[None]}
    rs = np.random.default_rng(6)
    gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
    views = synth.make_mvs_views(2)
    mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views], same_view=0,
               img_res=(576, 768), inverse_depth=False)
    from volsdf.utils.conf import dtu_model_conf, bmvs_model_conf
    torch.manual_seed(3)
    if kind == "dtu":
        from volsdf.model.network import VolSDFNetwork
        m = VolSDFNetwork(dtu_model_conf())
        m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}, strict=True)
    else:
        from volsdf.model.network_bg import VolSDFNetworkBG
        m = VolSDFNetworkBG(bmvs_model_conf())
    m = m.to(dev).train()
    loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0,
                      anneal_rgb=200, gce=0.5, confi=1e-3)
    loss.iter_step = 250
    ts = TrainStep(m, loss)
    assert ts.deterministic
    torch.manual_seed(13)
    for i in range(steps):
        lo, out = ts(inp, gt, mvs=mvs)
        print(f"{kind} R={R} step {i}: grad {sha(ts.fp.grad)} rgb {sha(out['rgb_values'])} loss {float(lo['loss']).hex()}")
    torch.cuda.synchronize()
    print(f"{kind} R={R} params {sha(ts.fp.flat)}")


L = _lib.load()
L.svs_set_deterministic(1)
print("library:", os.environ.get("SVS_LIB_PATH", "(default)"))
run("dtu", 1024)
run("bmvs", 256)

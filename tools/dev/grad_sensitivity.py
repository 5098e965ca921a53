"""How far the float64 gradient of a bmvs step moves when every parameter is perturbed by eps * N(0,1) relative noise:
the conditioning of the quantity tests/test_gpu_bg.py::test_bg_step_gradient_at_bench_geometry compares (dev aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests/golden", "oracle", "s-volsdf_amd", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import synth, torch_ref as tref
import test_gpu_bg as tb

dev = torch.device("cuda:0")
R, it = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 250
from svs_hip.trainer import TrainStep
from volsdf.model.loss import VolSDFLoss
G = tb.G
m = tb._model(dev, 0.1)
loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3)
loss.iter_step = it
K, pose = synth.make_camera()
inp = {"intrinsics": G(K, dev)[None], "uv": G(synth.make_uv(R, seed=17), dev)[None], "pose": G(pose, dev)[None]}
rs = np.random.default_rng(5)
gt = {"rgb": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev), "rgb_smooth": G(rs.uniform(0, 1, (1, R, 3)).astype(np.float32), dev)}
views = synth.make_mvs_views(2)
mvs = dict(views=[dict(K=v["K"], c2w=v["c2w"], cost=G(v["cost"], dev), z_mvs=G(v["z_mvs"], dev)) for v in views], same_view=0, img_res=(576, 768), inverse_depth=False)
ts = TrainStep(m, loss, lr=5e-4, groups="auto", graph=False)
p0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
torch.manual_seed(3)
ts(inp, gt, mvs=mvs); torch.cuda.synchronize()
norm = float(ts.opt.info[0]); coef = min(1.0, 1.0 / (norm + 1e-6))
got = {n: (p.grad / coef).double().cpu() for n, p in m.named_parameters()}
keeps = [h[0] for h in ts._hold]; outs = [r[1] for r in ts._results]
cat = lambda xs: torch.cat(xs, 0).double()
z, dirs, ds = (cat([k[n] for k in keeps]) for n in ("z_vals", "ray_dirs", "depth_scale"))
z_max, z_bg, bg_depth = (cat([k[n] for k in keeps]) for n in ("z_max", "z_bg", "bg_depth"))
Nb = z_bg.shape[1]
bg_pts = cat([k["bg_pts"].reshape(-1, Nb, 4) for k in keeps]); eik = cat([k["src"].points for k in keeps])
pj, pi = cat([o["pj"] for o in outs]), cat([o["pi"] for o in outs])

def autograd(pp):
    p = {k: pp[k].detach().double().clone().requires_grad_(True) for k in pp}
    out = tref.forward_differentiable_bg(p, keeps[0]["cam_loc"].double(), dirs, z, z_max, eik, ds, z_bg, bg_pts, bg_depth=bg_depth, device=dev)
    out["pj"], out["pi"] = pj, pi
    out["depth_values"] = out["depth_values_all"]
    tref.loss_fn(out, gt["rgb"].reshape(-1, 3).double(), gt["rgb_smooth"].reshape(-1, 3).double(), it).backward()
    return {k: (v.grad.cpu() if v.grad is not None else torch.zeros(v.shape, dtype=torch.float64)) for k, v in p.items()}

ref = autograd(p0)
g = torch.Generator().manual_seed(1)
for eps in (1e-7, 1e-6, 1e-5):
    worst = {}
    for t in range(3):
        pp = {k: v.double() * (1 + eps * torch.randn(v.shape, generator=g, dtype=torch.float64).to(dev)) for k, v in p0.items()}
        gg = autograd(pp)
        for n in ref:
            den = float(ref[n].abs().max()) + 1e-30
            worst[n] = max(worst.get(n, 0), float((gg[n] - ref[n]).abs().max()) / den)
    print("eps", eps, {n.replace("rendering_network", "rn"): f"{e:.1e}" for n, e in worst.items() if n.startswith("rendering") or n == "density.beta"})
print("ours", {n.replace("rendering_network", "rn"): f"{float((got[n] - ref[n]).abs().max()) / (float(ref[n].abs().max()) + 1e-30):.1e}" for n in ref if n.startswith("rendering") or n == "density.beta"})

"""dev aid: GPU time of each of the first steps after a synchronise (events on the origin stream between steps), and when the
host returned from enqueueing each -- where a short timed region loses its time (profiles/r06_bench_steps_sweep.txt)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")]
import numpy as np, torch, synth
from volsdf.utils.conf import dtu_model_conf
from svs_hip.trainer import TrainStep
from volsdf.model.loss import VolSDFLoss
from volsdf.model.network import VolSDFNetwork
dev = torch.device("cuda:0")
params = synth.make_params(0)
model = VolSDFNetwork(dtu_model_conf()); model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}); model.to(dev).train()
K, pose = synth.make_camera(); R = 1024
inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(np.ascontiguousarray(synth.make_uv(R, seed=100)))[None].to(dev),
       "pose": torch.from_numpy(pose)[None].to(dev)}
rs = np.random.default_rng(11)
gt = {"rgb": torch.from_numpy(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)).to(dev), "rgb_smooth": torch.from_numpy(rs.uniform(0, 1, (1, R, 3)).astype(np.float32)).to(dev)}
gen = torch.Generator(device=dev); gen.manual_seed(7)
views = []
for dx in (0.0, 0.3, -0.3):
    Kj, Pj = synth.make_camera(center=(dx, 0.0, -2.5), tilt=-0.12 * dx / 0.3)
    prob = torch.softmax(torch.randn(192, 288, 384, device=dev, generator=gen), 0)
    zm = torch.linspace(1.5, 3.5, 192, device=dev).view(-1, 1, 1) * torch.ones(1, 288, 384, device=dev)
    views.append(dict(K=Kj, c2w=Pj, cost=prob, z_near=zm[0].contiguous(), z_far=zm[-1].contiguous()))
mvs = dict(views=views, same_view=0, img_res=(576, 768), inverse_depth=False)
loss = VolSDFLoss(rgb_loss="torch.nn.L1Loss", eikonal_weight=0.1, rgb_weight=1.0, mvs_weight=1.0, sparse_weight=1.0, anneal_rgb=200, gce=0.5, confi=1e-3)
ts = TrainStep(model, loss, lr=5e-4, groups="auto", graph="auto")
for _ in range(300): ts(inp, gt, mvs=mvs)
torch.cuda.synchronize()
for rep in range(3):
    n = 12
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        ts(inp, gt, mvs=mvs); ev[i + 1].record(); host.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize(); wall = 1e3 * (time.perf_counter() - t0)
    print("gpu ms per step:", " ".join(f"{ev[i].elapsed_time(ev[i + 1]):.2f}" for i in range(n)), f"| wall {wall:.2f} = {wall / n:.3f} per step")
    print("host returned at:", " ".join(f"{h:.2f}" for h in host))

"""ms of the sampler's sdf-only evaluation (131 072 points) with the library named by SVS_LIB_PATH (dev aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests/golden", "s-volsdf_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch, synth
from svs_hip import ops
from volsdf.utils.conf import dtu_model_conf
from volsdf.model.network import VolSDFNetwork
dev = torch.device("cuda:0")
m = VolSDFNetwork(dtu_model_conf())
m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}, strict=True)
m.to(dev).train()
pk = m.packed_mlp()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
g = torch.Generator(device="cpu").manual_seed(0)
cam = torch.tensor([0.0, 0.0, -2.5], device=dev)
dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
z = torch.sort(torch.rand(R, 128, generator=g) * 4 + 0.5, -1)[0].to(dev)
src = ops.PointSource(cam=cam, dirs=dirs, z=z)
for _ in range(20):
    ops.sdf_vals(pk, src, 3.0, 20.0)
torch.cuda.synchronize()
best = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.sdf_vals(pk, src, 3.0, 20.0)
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / 50)
print(os.environ.get("SVS_LIB_PATH", "default").split("_")[-1], "SVS_SDF_KERNEL=" + os.environ.get("SVS_SDF_KERNEL", "32"), R, "rays: ms", " ".join(f"{t:.4f}" for t in best))

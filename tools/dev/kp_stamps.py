"""Per-wave cycle stamps of the K-split-pair sdf-only kernel built with -DKP_STAMP (dev aid)."""
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
R = 1024
g = torch.Generator(device="cpu").manual_seed(0)
cam = torch.tensor([0.0, 0.0, -2.5], device=dev)
dirs = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=-1).to(dev)
z = torch.sort(torch.rand(R, 128, generator=g) * 4 + 0.5, -1)[0].to(dev)
src = ops.PointSource(cam=cam, dirs=dirs, z=z)
for _ in range(30):
    out = ops.sdf_vals(pk, src, 3.0, 20.0)
torch.cuda.synchronize()
o = out.cpu().numpy().reshape(-1, 128)[:, :16].reshape(-1, 8, 2)
print("per wave (mean over workgroups): loop cycles", o[:, :, 0].mean(0).round(0), " sync cycles", o[:, :, 1].mean(0).round(0))
print("tiles timed: 55; per tile: loop %.0f sync %.0f" % (o[:, :, 0].mean() / 55, o[:, :, 1].mean() / 55))

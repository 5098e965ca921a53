"""Timeline of the two waves of pair 0 (waves 0 and 4) of the last workgroup, K-split-pair sdf-only kernel built with
-DKP_TRACE: s_memtime in front of every MFMA block (M), behind it (V = start of the vector block), before / after the tile
barrier (dev aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("tests/golden", "s-volsdf_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch, synth
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
for _ in range(10):
    out = ops.sdf_vals(pk, src, 3.0, 20.0)
torch.cuda.synchronize()
o = out.cpu().numpy().reshape(-1)
G = int(os.environ.get("KP_GROUP", "1"))
per_tile = 2 * (8 // G) + 2
base = o[640:648]
b0 = base.min()
for w in (0, 4, 1, 5):
    t = o[w * 80:(w + 1) * 80] + (base[w] - b0)
    print(f"wave {w} (start +{base[w] - b0:.0f}):")
    for tile in range(3):
        row = t[tile * per_tile:(tile + 1) * per_tile]
        blocks = " ".join(f"M{row[2 * i]:.0f}-V{row[2 * i + 1]:.0f}" for i in range(8 // G))
        print(f"   tile {tile}: {blocks} | send {row[-2]:.0f} barrier-> {row[-1]:.0f}")

import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd")]
from svs_hip import costvol
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
cin, shape = 8, (3, 5, 40)
x = rng.normal(0, 1, (cin,) + shape).astype(np.float32)
w = (rng.normal(0, 1, (cin, 27, 1)) / np.sqrt(27 * cin)).astype(np.float32)
wt = torch.from_numpy(w).double().permute(2, 0, 1).reshape(1, cin, 3, 3, 3)
ref = torch.nn.functional.conv3d(torch.from_numpy(x).double()[None], wt, padding=1)[0, 0].numpy()
got = costvol.conv3d(torch.from_numpy(x).to(dev), torch.from_numpy(w).to(dev), None, relu=False)[0].cpu().numpy()
err = np.abs(got - ref)
print("max err", err.max())
bad = np.argwhere(err > 1e-4)
print(len(bad), "bad of", err.size)
print("bad x histogram:", np.bincount(bad[:, 2], minlength=shape[2]))
print("bad y histogram:", np.bincount(bad[:, 1], minlength=shape[1]))
print("bad z histogram:", np.bincount(bad[:, 0], minlength=shape[0]))

"""FeatureNet (row f1) at config-3 image size: the HIP convolutions against the same modules run through torch/MIOpen."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import synth  # noqa: E402
from models.CasMVSNet import FeatureNet  # noqa: E402

dev = torch.device("cuda:0")
net = FeatureNet(8, 3, 4, "fpn")
net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.make_featurenet_params(1).items()})
net.to(dev).eval()
x = torch.rand(1, 3, 512, 640, device=dev)


def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def torch_path():
    blk = lambda b, c: F.relu(b.bn(b.conv(c)))
    c0 = blk(net.conv0[1], blk(net.conv0[0], x))
    c1 = c0
    for b in net.conv1:
        c1 = blk(b, c1)
    c2 = c1
    for b in net.conv2:
        c2 = blk(b, c2)
    f = F.interpolate(c2, scale_factor=2, mode="nearest") + net.inner1(c1)
    f2 = F.interpolate(f, scale_factor=2, mode="nearest") + net.inner2(c0)
    return net.out1(c2), net.out2(f), net.out3(f2)


with torch.no_grad():
    print(f"FeatureNet 512x640: HIP {timed(lambda: net(x)):.3f} ms, torch/MIOpen {timed(torch_path):.3f} ms")

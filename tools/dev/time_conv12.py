"""dev aid: conv1 -> conv2 of the U-Net at the three stage sizes, fused (split volume between them) vs separate"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd")]
from svs_hip import costvol
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for shp in ((192, 128, 160), (32, 256, 320), (8, 512, 640)):
    x = torch.randn(8, *shp, device=dev)
    w1 = torch.randn(8, 27, 16, device=dev) * 0.1; b1 = torch.randn(16, device=dev)
    w2 = torch.randn(16, 27, 16, device=dev) * 0.1; b2 = torch.randn(16, device=dev)
    c1 = costvol.conv3d(x, w1, b1, stride=2)
    sv = costvol.conv3d(x, w1, b1, stride=2, split_out=True)
    print(shp, f"conv1 float {t(lambda: costvol.conv3d(x, w1, b1, stride=2)):.3f}  conv1 split {t(lambda: costvol.conv3d(x, w1, b1, stride=2, split_out=True)):.3f}  "
          f"conv2 old {t(lambda: costvol.conv3d(c1, w2, b2)):.3f}  conv2 rows {t(lambda: costvol.conv3d(sv, w2, b2)):.3f} ms")

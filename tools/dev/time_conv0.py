"""dev aid: time of conv0 (32 -> 8, 192x128x160) and of the prob layer with the library SVS_LIB_PATH names"""
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
for cin, cout, shp in ((32, 8, (192, 128, 160)), (16, 8, (32, 256, 320)), (8, 8, (8, 512, 640)), (8, 1, (192, 128, 160)), (8, 1, (32, 256, 320)), (8, 1, (8, 512, 640)), (16, 16, (96, 64, 80))):
    x = torch.randn(cin, *shp, device=dev)
    wt = torch.randn(cin, 27, cout, device=dev) * 0.1
    b = torch.randn(cout, device=dev)
    ms = t(lambda: costvol.conv3d(x, wt, b))
    if cout == 8:
        sv = costvol.SplitVolume.pack(x)
        print(f"  pair: {t(lambda: costvol.conv3d(sv, wt, b)):.3f} ms", end="")
    print(f"{os.environ.get('SVS_LIB_PATH', 'default')[-24:]:24s} {cin}->{cout} {shp}: {ms:.3f} ms")

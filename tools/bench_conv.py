"""per-layer timing of the cost-regularisation U-Net at the three stage sizes of config 3
(C=32, 192x128x160; C=16, 32x256x320; C=8, 8x512x640).  usage: python tools/bench_conv.py [stage]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
from svs_hip import costvol
dev = torch.device("cuda:0")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
STAGE = int(sys.argv[1]) if len(sys.argv) > 1 else 1
C0, D, H, W = {1: (32, 192, 128, 160), 2: (16, 32, 256, 320), 3: (8, 8, 512, 640)}[STAGE]
tot = 0.0
for name, cin, cout, (d, h, w), stride, tr in (("conv0", C0, 8, (D, H, W), 1, False), ("conv1", 8, 16, (D, H, W), 2, False),
                                                ("conv2", 16, 16, (D // 2, H // 2, W // 2), 1, False), ("conv3", 16, 32, (D // 2, H // 2, W // 2), 2, False),
                                                ("conv4", 32, 32, (D // 4, H // 4, W // 4), 1, False), ("conv5", 32, 64, (D // 4, H // 4, W // 4), 2, False),
                                                ("conv6", 64, 64, (D // 8, H // 8, W // 8), 1, False), ("conv7", 64, 32, (D // 8, H // 8, W // 8), 2, True),
                                                ("conv9", 32, 16, (D // 4, H // 4, W // 4), 2, True), ("conv11", 16, 8, (D // 2, H // 2, W // 2), 2, True),
                                                ("prob", 8, 1, (D, H, W), 1, False)):
    x = torch.randn(cin, d, h, w, device=dev)
    wt = torch.randn(cin, 27, cout, device=dev) * 0.1
    b = torch.randn(cout, device=dev)
    od = (d * 2, h * 2, w * 2) if tr else ((d - 1) // stride + 1, (h - 1) // stride + 1, (w - 1) // stride + 1)
    xin = costvol.SplitVolume.pack(x) if name == "conv0" else x              # conv0 reads the producer's split volume
    skip = torch.randn(cout, *od, device=dev) if tr else None                # the transposed layers add a skip
    b = None if name == "prob" else b
    ms = t(lambda: costvol.conv3d(xin, wt, b, skip=skip, stride=stride, transposed=tr, relu=name != "prob"))
    macs = od[0] * od[1] * od[2] * cout * cin * (27 / 8 if tr else 27)
    tot += ms
    print(f"{name:7s} {cin:3d}->{cout:3d} {ms:7.3f} ms  {2 * macs / ms / 1e9:7.1f} TFLOP/s")
print(f"sum {tot:.3f} ms")

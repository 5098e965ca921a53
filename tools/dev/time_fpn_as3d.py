"""dev aid: FeatureNet's wide 3x3 layers run as 3-D convolutions of a one-plane volume on the cost-volume MFMA kernels
(the 3x3 taps embedded in the middle z slice of a 3x3x3 kernel): time and error against float64 conv2d, beside the
float32 VALU kernel the pyramid uses today (svs_hip.ops / csrc/svs_conv2d.hip)."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "s-volsdf_amd"), os.path.join(ROOT, "tests", "golden")]
from svs_hip import costvol
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for name, cin, cout, H, W in (("conv1.1 16->16", 16, 16, 256, 320), ("conv2.1 32->32", 32, 32, 128, 160), ("out2 32->16", 32, 16, 256, 320),
                             ("out3 32->8", 32, 8, 512, 640)):
    x = torch.randn(cin, H, W, generator=g).to(dev)
    w2 = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    w3 = torch.zeros(cin, 27, cout, device=dev)
    w3[:, 9:18, :] = w2.permute(1, 2, 3, 0).reshape(cin, 9, cout)          # [Cin][kz ky kx][Cout], the middle z slice
    ref = torch.nn.functional.conv2d(x.double()[None], w2.double(), b.double(), padding=1)[0].clamp(min=0)
    out = costvol.conv3d(x[:, None].contiguous(), w3, b, relu=True)[:, 0]
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); costvol.conv3d(x[:, None], w3, b, relu=True); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t2 = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.nn.functional.conv2d(x[None], w2, b, padding=1); e1.record(); torch.cuda.synchronize()
        t2.append(e0.elapsed_time(e1) * 1e3)
    print(f"{name:16s} {H}x{W}: as 3-D conv {min(ts):6.1f} us (median {sorted(ts)[10]:6.1f}), rel err {err:.1e}; torch conv2d {min(t2):6.1f} us")

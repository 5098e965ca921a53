"""dev aid: the two multi-layer weight-gradient launches of a step, replayed alone at several ray counts
    python tools/dev/time_wgrad.py [rays ...]      (SVS_LIB_PATH selects a variant build, tools/dev/ab_defs.sh)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import torch
import synth
from volsdf.utils.conf import dtu_model_conf
from svs_hip import lib
from svs_hip.train import MlpBackward
from volsdf.model.network import VolSDFNetwork

L = lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = VolSDFNetwork(dtu_model_conf()).to(dev).train()
for R in [int(x) for x in sys.argv[1:]] or [256, 512, 1024]:
    K, pose = synth.make_camera()
    inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(synth.make_uv(R, seed=1))[None].to(dev),
           "pose": torch.from_numpy(pose)[None].to(dev)}
    keep = {}
    m._forward_impl(inp, 1, keep)
    n_total, n_main = keep["src"].n, keep["rgb"].shape[0]
    bw = MlpBackward(dev)
    sdf_p, rgb_p = m.mlp_params()
    d_rgb = torch.randn(n_main, 3, device=dev) * 1e-3
    d_sdf = torch.randn(n_main, 1, device=dev) * 1e-3
    d_gt = torch.randn(n_total - n_main, 3, device=dev) * 1e-3
    rec = []
    orig = L.svs_wgrad_multi
    L.svs_wgrad_multi = lambda *a: (rec.append(a), orig(*a))[1]
    bw.run(sdf_p, rgb_p, keep, d_rgb, d_sdf, d_gt)
    L.svs_wgrad_multi = orig
    torch.cuda.synchronize()
    for a in rec:
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        fn = lambda: lib.check(orig(a[0], a[1], a[2], st))
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        print("rays %5d  wgrad_multi %d jobs  %.1f us" % (R, a[1], e0.elapsed_time(e1) / 20 * 1e3), flush=True)

"""Per-kernel timings at bench sizes (1024 rays): isolates each hot kernel with events on the launch stream.
Development aid for the roofline work; prints one JSON object.  BK_RAYS=<n>: another batch size (default 1024); BK_STAMPS=1 with a
-DSVS_ABL=65536 build of svs_mlp_bwd_h2.hip (tools/dev/ab_defs.sh + SVS_LIB_PATH): pass B's per-tile cycle stamps and clock."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402
from volsdf.utils.conf import dtu_model_conf  # noqa: E402
from svs_hip import lib, ops  # noqa: E402
from svs_hip.train import KBLOCK, MlpBackward, _off  # noqa: E402
from volsdf.model.network import VolSDFNetwork  # noqa: E402

F_SDF, F_RGB = 1_049_088, 533_504


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = torch.device("cuda:0")
    L = lib.load()
    R = int(os.environ.get("BK_RAYS", "1024"))
    params = synth.make_params(0)
    m = VolSDFNetwork(dtu_model_conf())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m.to(dev).train()
    K, pose = synth.make_camera()
    inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(synth.make_uv(R, seed=1))[None].to(dev),
           "pose": torch.from_numpy(pose)[None].to(dev)}
    keep = {}
    out = m._forward_impl(inp, 1, keep)
    pk = m.packed_mlp()
    src = keep["src"]
    n_total, n_main = src.n, keep["rgb"].shape[0]
    res = {}
    zs = torch.sort(torch.rand(R, 128, device=dev) * 5 + 0.5, -1)[0]
    src128 = ops.PointSource(cam=keep["cam_loc"], dirs=keep["ray_dirs"], z=zs)
    t = timeit(lambda: ops.sdf_vals(pk, src128, 3.0, 20.0))
    res["sdf_only"] = dict(ms=t, tflops=131072 * 918016 / t / 1e9)
    t = timeit(lambda: ops.sdf_outputs(pk, src, 3.0, 20.0, clamp_n=n_main, keep={}))
    res["sdf_full"] = dict(ms=t, tflops=n_total * 2 * F_SDF / t / 1e9)
    src_main = ops.PointSource(cam=keep["cam_loc"], dirs=keep["ray_dirs"], z=keep["z_vals"])
    g = keep["rgb"].new_zeros(n_main, 3).normal_()
    t = timeit(lambda: ops.rgb_eval(pk, src_main, g, keep["ray_dirs"], keep["feat_tiles"], keep={}))
    res["rgb"] = dict(ms=t, tflops=n_main * F_RGB / t / 1e9)
    # backward pieces: run one full backward to populate the scratch, then time the kernels alone
    bw = MlpBackward(dev)
    sdf_p, rgb_p = m.mlp_params()
    d_rgb = torch.randn(n_main, 3, device=dev) * 1e-3
    d_sdf = torch.randn(n_main, 1, device=dev) * 1e-3
    d_gt = torch.randn(n_total - n_main, 3, device=dev) * 1e-3
    res["backward_total"] = dict(ms=timeit(lambda: bw.run(sdf_p, rgb_p, keep, d_rgb, d_sdf, d_gt), n=5))
    # the two multi-layer weight-gradient launches of a step, replayed alone (arguments recorded from one backward)
    rec = []
    orig = L.svs_wgrad_multi
    L.svs_wgrad_multi = lambda *a: (rec.append(a), orig(*a))[1]
    bw.run(sdf_p, rgb_p, keep, d_rgb, d_sdf, d_gt)
    L.svs_wgrad_multi = orig
    torch.cuda.synchronize()
    for a in rec:
        name = "wgrad_multi_%d_layers" % a[1]
        t = timeit(lambda: lib.check(orig(a[0], a[1], a[2], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))))
        res[name] = dict(ms=t)
    P = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dn = torch.empty(n_main, 3, device=dev)
    prec = bw.streams.precision
    h2 = prec in (1, 2)
    am = bw.accum.absmax
    N = lambda x: P(x) if h2 else None
    t = timeit(lambda: lib.check(L.svs_rgb_bwd(n_main, P(d_rgb), P(keep["rgb"]), P(keep["rbuf"]), P(bw.streams.rgb), prec,
                                               P(bw.zbuf), P(bw.feat_bar), P(dn), N(am), st())))
    res["rgb_bwd"] = dict(ms=t, tflops=n_main * F_RGB / t / 1e9)
    d_grad = torch.cat([dn, d_gt], 0)
    hbuf, gbuf, mask = keep["hbuf"], keep["gbuf"], keep["clamp_mask"]
    dsf = torch.zeros(n_total, device=dev)
    row0 = torch.zeros(257, device=dev)
    t = timeit(lambda: lib.check(L.svs_sdf_bwd_a(*src.args(), P(d_grad), P(mask), P(hbuf), P(gbuf), P(bw.streams.sdf), prec,
                                                 P(bw.ubuf), P(bw.a2buf), P(bw.pebuf), N(am), st())))
    res["sdf_bwd_a"] = dict(ms=t, tflops=n_total * 0.9 * F_SDF / t / 1e9)
    t = timeit(lambda: lib.check(L.svs_sdf_bwd_b(n_total, P(dsf), P(mask), P(bw.feat_bar), n_main, P(hbuf), P(gbuf), P(bw.a2buf), P(bw.ubuf),
                                                 P(bw.streams.sdf), prec, P(bw.abuf), P(bw.sbar), N(am), st())))
    res["sdf_bwd_b"] = dict(ms=t, tflops=n_total * F_SDF / t / 1e9)
    if os.environ.get("BK_STAMPS"):      # a -DSVS_ABL=65536 build: cycle stamps of wave 0 of every workgroup in sbar_out
        torch.cuda.synchronize()
        o = bw.sbar[: (n_total // 128) * 128].reshape(-1, 128)[:, :19].double().cpu().numpy()
        np.set_printoptions(linewidth=200, suppress=True)
        print("workgroups", o.shape[0], "shader MHz during the kernel %.0f" % (o[:, 17] / o[:, 18] * 100).mean(), file=sys.stderr)
        print("per tile index, cycles summed over 8 stages: MFMA part", o[:, :8].mean(0).round(0), file=sys.stderr)
        print("                                    wait + barrier part", o[:, 8:16].mean(0).round(0), file=sys.stderr)
        print("last epilogues %.0f  total %.0f  (sum of parts %.0f)" % (o[:, 16].mean(), o[:, 17].mean(), o[:, :17].sum(1).mean()), file=sys.stderr)
    from svs_hip.train import block_stride, record_off
    LS = block_stride(n_total)
    R = lambda buf, n, nb, l: _off(buf, record_off(n, nb, l)) if h2 else None
    dW = torch.zeros(256, 288, device=dev); db = torch.zeros(256, device=dev)
    l = 2
    t = timeit(lambda: lib.check(L.svs_wgrad(_off(bw.abuf, l * LS), _off(hbuf, (l - 1) * LS), KBLOCK, KBLOCK,
                                             _off(gbuf, l * LS), _off(bw.ubuf, l * LS), KBLOCK, KBLOCK,
                                             None, 0, n_total, prec, N(am), R(bw.abuf, n_total, 8, l), R(bw.ubuf, n_total, 9, l),
                                             P(dW), 288, P(db), st())))
    res["wgrad_2pair"] = dict(ms=t, tflops=2 * 2 * 256 * 256 * n_total / t / 1e9)
    t = timeit(lambda: lib.check(L.svs_wgrad(P(bw.feat_bar), _off(hbuf, 7 * LS), KBLOCK, KBLOCK, None, None, 0, 0,
                                             None, 0, n_main, prec, _off(am, 2) if h2 else None, R(bw.feat_bar, n_main, 1, 0), None,
                                             P(dW), 288, P(db), st())))
    res["wgrad_1pair"] = dict(ms=t, tflops=2 * 256 * 256 * n_main / t / 1e9)
    if True:
        res["lin8_row0"] = dict(ms=timeit(lambda: lib.check(L.svs_lin8_row0_grad(P(hbuf), P(bw.ubuf), P(bw.sbar), n_total, prec, P(row0), st()))))
    print(json.dumps({k: {a: round(b, 3) for a, b in v.items()} for k, v in res.items()}))


if __name__ == "__main__":
    main()

"""dev aid: do the HBM-bound backward sweeps keep their speed on a subset of the CUs, and do a matrix-core-bound and an
HBM-bound kernel overlap when each gets its own CUs (hipExtStreamCreateWithCUMask)?  1024-ray sizes."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, synth
from volsdf.utils.conf import dtu_model_conf
from svs_hip import lib, ops
from svs_hip.train import MlpBackward
from volsdf.model.network import VolSDFNetwork

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[32 * w + b]) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def mask(kind, n):
    """n CUs of 256: 'first' = bits 0..n-1; 'xcd' = the first n/8 of every group of 32; 'stride' = spread evenly"""
    b = [0] * 256
    if kind == "first":
        for i in range(n): b[i] = 1
    elif kind == "xcd":
        for x in range(8):
            for i in range(n // 8): b[32 * x + i] = 1
    else:
        for i in range(n): b[(i * 256) // n] = 1
    return b


def main():
    dev = torch.device("cuda:0")
    L = lib.load()
    R = 1024
    params = synth.make_params(0)
    m = VolSDFNetwork(dtu_model_conf())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    m.to(dev).train()
    K, pose = synth.make_camera()
    inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(synth.make_uv(R, seed=1))[None].to(dev),
           "pose": torch.from_numpy(pose)[None].to(dev)}
    keep = {}
    m._forward_impl(inp, 1, keep)
    pk = m.packed_mlp()
    src = keep["src"]
    n_total, n_main = src.n, keep["rgb"].shape[0]
    bw = MlpBackward(dev)
    sdf_p, rgb_p = m.mlp_params()
    d_rgb = torch.randn(n_main, 3, device=dev) * 1e-3
    d_sdf = torch.randn(n_main, 1, device=dev) * 1e-3
    d_gt = torch.randn(n_total - n_main, 3, device=dev) * 1e-3
    bw.run(sdf_p, rgb_p, keep, d_rgb, d_sdf, d_gt)
    torch.cuda.synchronize()
    P = lambda x: ctypes.c_void_p(x.data_ptr())
    prec = bw.streams.precision
    am = bw.accum.absmax
    dn = torch.zeros(n_main, 3, device=dev)
    d_grad = torch.cat([dn, d_gt], 0)
    hbuf, gbuf, cmask = keep["hbuf"], keep["gbuf"], keep["clamp_mask"]
    dsf = torch.zeros(n_total, device=dev)
    zs = torch.sort(torch.rand(R, 128, device=dev) * 5 + 0.5, -1)[0]
    src128 = ops.PointSource(cam=keep["cam_loc"], dirs=keep["ray_dirs"], z=zs)

    def st():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    kernels = {
        "sdf_only": lambda: ops.sdf_vals(pk, src128, 3.0, 20.0),
        "sdf_full": lambda: ops.sdf_outputs(pk, src, 3.0, 20.0, clamp_n=n_main, keep={}),
        "bwd_a": lambda: lib.check(L.svs_sdf_bwd_a(*src.args(), P(d_grad), P(cmask), P(hbuf), P(gbuf), P(bw.streams.sdf), prec,
                                                  P(bw.ubuf), P(bw.a2buf), P(bw.pebuf), P(am), st())),
        "bwd_b": lambda: lib.check(L.svs_sdf_bwd_b(n_total, P(dsf), P(cmask), P(bw.feat_bar), n_main, P(hbuf), P(gbuf), P(bw.a2buf), P(bw.ubuf),
                                                  P(bw.streams.sdf), prec, P(bw.abuf), P(bw.sbar), P(am), st())),
    }

    def timeit(fn, stream, n=6):
        with torch.cuda.stream(stream):
            fn(); stream.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(n): fn()
            e1.record(stream); stream.synchronize()
        return e0.elapsed_time(e1) / n

    res = {}
    full = torch.cuda.Stream()
    for k, fn in kernels.items():
        res[f"{k}/all"] = round(timeit(fn, full), 3)
    for kind in ("first", "xcd", "stride"):
        for n in (64, 96, 128, 160, 192):
            s = masked_stream(mask(kind, n))
            for k, fn in kernels.items():
                res[f"{k}/{kind}{n}"] = round(timeit(fn, s), 3)
    print(json.dumps(res))
    # overlap: sdf_full on nA CUs concurrently with bwd_a on the other 256 - nA
    for kind in ("xcd",):
        for nA in (128, 160, 192):
            ba = mask(kind, nA)
            bb = [1 - x for x in ba]
            sa, sb = masked_stream(ba), masked_stream(bb)
            for ka, kb in (("sdf_full", "bwd_a"), ("sdf_full", "bwd_b"), ("sdf_only", "bwd_a")):
                torch.cuda.synchronize()
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                sa.wait_event(e0); sb.wait_event(e0)
                with torch.cuda.stream(sa):
                    for _ in range(4): kernels[ka]()
                    e1.record(sa)
                with torch.cuda.stream(sb):
                    for _ in range(4): kernels[kb]()
                    e2.record(sb)
                torch.cuda.synchronize()
                print(f"{kind} {nA}/{256 - nA}: {ka} {e0.elapsed_time(e1) / 4:.3f} ms || {kb} {e0.elapsed_time(e2) / 4:.3f} ms  "
                      f"(alone on all CUs: {res[ka + '/all']} + {res[kb + '/all']} = {res[ka + '/all'] + res[kb + '/all']:.3f})")


if __name__ == "__main__":
    main()

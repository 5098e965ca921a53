"""Diagnostic: builds svs_mlp_h2.hip with -DSVS_ABL=<mask> (svs_mlp_h2_dev.h) into variant libraries and times the forward
kernels with each (results of the ablated kernels are wrong by construction; only the time means anything).

    python tools/ablate_fwd.py build 0 1 3 11 4 12      # here (cross-compile), writes s-volsdf_amd/lib/abl/
    python tools/ablate_fwd.py run 0 1 3 11 4 12        # on the GPU box: one child process per variant
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "s-volsdf_amd")
sys.path.insert(0, PKG)
import build as B  # noqa: E402

ABL = os.path.join(B.LIBDIR, "abl")


def build(masks):
    B.build(verbose=False)
    os.makedirs(ABL, exist_ok=True)
    objs = [os.path.join(B.LIBDIR, u.rsplit(".", 1)[0] + ".o") for u in B.UNITS if u != "svs_mlp_h2.hip"]
    for m in masks:
        tag = os.environ.get("ABL_TAG", "")
        obj = os.path.join(ABL, f"h2_{m}{tag}.o")
        extra = [f"-D{d}" for d in os.environ.get("ABL_DEFS", "").split()]   # further -D switches under test
        subprocess.check_call([B._hipcc()] + B.BASE_FLAGS + extra + [f"-DSVS_ABL={m}", "-I", B.CSRC, "-c",
                                                           os.path.join(B.CSRC, "svs_mlp_h2.hip"), "-o", obj])
        subprocess.check_call([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o",
                               os.path.join(ABL, f"lib_{m}{tag}.so"), obj] + objs)
        print("built", m, flush=True)


def child(mask):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from svs_hip import lib
    lib.LIB_PATH = os.path.join(ABL, f"lib_{mask}{os.environ.get('ABL_TAG', '')}.so")
    import torch
    import synth
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
    z128 = torch.sort(torch.rand(R, 128, generator=g) * 4 + 0.5, -1)[0].to(dev)
    z100 = torch.sort(torch.rand(R, 100, generator=g) * 4 + 0.5, -1)[0].to(dev)
    s128 = ops.PointSource(cam=cam, dirs=dirs, z=z128)
    s100 = ops.PointSource(cam=cam, dirs=dirs, z=z100)

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    res = {"mask": mask, "tag": os.environ.get("ABL_TAG", "")}
    res["sdf_only_ms"] = round(timeit(lambda: ops.sdf_vals(pk, s128, 3.0, 20.0)), 4)
    if mask & 16:   # in-kernel stamps of wave 0 of every workgroup (see sdf_only_h2_kernel)
        o = ops.sdf_vals(pk, s128, 3.0, 20.0).view(-1, 128)[:, :6].double().cpu()
        med = o.median(0).values
        res["cycles_pe_trunk_head_total"] = [int(v) for v in med[:4]]
        res["clock_ghz"] = round(float(med[3] / med[4]) * 0.1, 3)
        start = o[:, 5]
        res["wg_start_spread_us"] = round(float((start.max() - start.min()) % (1 << 24)) / 100.0, 1)
    res["sdf_full_ms"] = round(timeit(lambda: ops.sdf_outputs(pk, s100, 3.0, 20.0, clamp_n=R * 98, keep={})), 4)
    if not (mask & 16):
        # the radiance forward (training launch: r blocks stored) on the same points
        _, grads, feat, _, _ = ops.sdf_outputs(pk, s100, 3.0, 20.0, clamp_n=R * 98, keep={})
        res["rgb_ms"] = round(timeit(lambda: ops.rgb_eval(pk, s100, grads, dirs, feat, keep={})), 4)
        res["rgb_render_ms"] = round(timeit(lambda: ops.rgb_eval(pk, s100, grads, dirs, feat)), 4)
    if mask & 16:   # sdf_full_h2_kernel's stamps: pe, trunk, head, features, reverse 7..1, reverse 0 + Jacobian, total
        o = ops.sdf_outputs(pk, s100, 3.0, 20.0, clamp_n=R * 98, keep={})
        o = o[1].reshape(-1, 384)[:, :8].double().cpu()
        med = o.median(0).values
        res["full_cycles_pe_trunk_head_feat_rev_rev0_total"] = [int(v) for v in med[:7]]
        res["full_clock_ghz"] = round(float(med[6] / med[7]) * 0.1, 3)
        o = ops.sdf_outputs(pk, s100, 3.0, 20.0, clamp_n=R * 98)      # render mode: no gbuf stores
        o = o[1].reshape(-1, 384)[:, :8].double().cpu()
        res["render_cycles"] = [int(v) for v in o.median(0).values[:7]]
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    cmd, masks = sys.argv[1], [int(a) for a in sys.argv[2:]]
    if cmd == "build":
        build(masks)
    elif cmd == "child":
        child(masks[0])
    else:
        for rep in range(2):
            for m in masks:
                subprocess.call([sys.executable, os.path.abspath(__file__), "child", str(m)])

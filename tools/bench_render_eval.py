"""Secondary measurement (not the driver's bench line): whole-image EVAL rendering, VolOpt.render_step's path
(volsdf/vsdf.py:237-287): a 768x576 view, model.eval(), fast=-1 (the sampler's full up-sampling loop, up to 5 rounds, with
the convergence decision per 500-ray chunk), through svs_hip/renderer.py::render_image (8000-ray launches) and through the
reference's own schedule -- one forward call per 500-ray chunk -- over the same model.  Prints one JSON object."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402


def measure(chunk_loop_too=True):
    from volsdf.utils.conf import dtu_model_conf
    from volsdf.model.network import VolSDFNetwork
    from svs_hip.renderer import depth_image, render_image
    dev = torch.device("cuda:0")
    H, W = 576, 768
    m = VolSDFNetwork(dtu_model_conf())
    sd = {k: torch.from_numpy(v) for k, v in synth.make_params(0).items()}
    m.load_state_dict(sd, strict=True)
    m.to(dev).eval()
    K, pose = synth.make_camera()
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    uv = np.stack([xs.reshape(-1), ys.reshape(-1)], -1)
    N = H * W
    inp = {"intrinsics": torch.from_numpy(K)[None].to(dev), "uv": torch.from_numpy(uv)[None].to(dev),
           "pose": torch.from_numpy(pose)[None].to(dev)}
    res = {"image": [H, W], "rays": N, "fast": -1}

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps, out

    t, out = timed(lambda: render_image(m, inp, N, split_n_pixels=500, rays_per_launch=8000), 3)
    res["render_image_s"] = t
    res["render_image_rays_per_s"] = N / t
    d = depth_image(out, (H, W))
    res["depth_finite"] = bool(torch.isfinite(d).all())

    def chunk_loop():
        with torch.no_grad():
            for lo in range(0, N, 500):
                o = m(dict(inp, uv=inp["uv"][:, lo:lo + 500]), fast=-1)
        return o
    if chunk_loop_too:
        t, _ = timed(chunk_loop, 1)
        res["chunk_loop_500_s"] = t
        res["chunk_loop_500_rays_per_s"] = N / t
    return res


def main():
    print(json.dumps(measure()))


if __name__ == "__main__":
    main()

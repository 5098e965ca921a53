"""Secondary measurements for the rows after the renderer, one JSON object:
depth fusion of one 1200x1600 reference view against 2 source views, and the Chamfer protocol on a DTU-size cloud,
with the CPU oracle (numpy / sklearn kd-tree, all host cores) timed beside each on a bounded sample."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "s-volsdf_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import synth  # noqa: E402


def gpu_time(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    from evals import eval_dtu
    from svs_hip import fusion
    import chamfer_oracle, fusion_oracle
    res = {}
    views = synth.make_fusion_views(1, hw=(1200, 1600), n_views=3)
    dv = {v: {k: (torch.from_numpy(a).cuda() if k in ("depth", "confidence", "img") else a) for k, a in views[v].items()} for v in views}
    t = gpu_time(lambda: fusion.fuse_view(dv[0], [dv[1], dv[2]], conf=0.3))
    res["fuse_view_1200x1600_2src_ms"] = t * 1e3
    small = synth.make_fusion_views(1, hw=(300, 400), n_views=3)
    t0 = time.perf_counter(); fusion_oracle.fuse_view(small[0], [small[1], small[2]], conf=0.3); t1 = time.perf_counter()
    res["fuse_view_cpu_oracle_ms_scaled_to_1200x1600"] = (t1 - t0) * 16 * 1e3

    n = 4_000_000
    rng = np.random.default_rng(0)
    d = rng.normal(0, 1, (n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pred = d * 150.0 + rng.normal(0, 0.2, (n, 3))
    s = rng.normal(0, 1, (n // 2, 3)); s /= np.linalg.norm(s, axis=1, keepdims=True); stl = s * 150.0
    P, Sd = torch.from_numpy(pred).cuda(), torch.from_numpy(stl).cuda()
    res["points_pred"], res["points_stl"] = n, n // 2
    res["downsample_0.2_ms"] = gpu_time(lambda: eval_dtu.radius_downsample(P, 0.2), 1) * 1e3
    res["nn_pred_to_stl_ms"] = gpu_time(lambda: eval_dtu.nearest_neighbor(Sd, P, 20.0), 2) * 1e3
    res["nn_stl_to_pred_ms"] = gpu_time(lambda: eval_dtu.nearest_neighbor(P, Sd, 20.0), 2) * 1e3
    m = 400_000
    t0 = time.perf_counter(); chamfer_oracle.nn_distance(stl, pred[:m]); t1 = time.perf_counter()
    res["nn_cpu_sklearn_ms_scaled"] = (t1 - t0) * (n / m) * 1e3
    t0 = time.perf_counter(); chamfer_oracle.radius_downsample(pred[:m], 0.2); t1 = time.perf_counter()
    res["downsample_cpu_sklearn_ms_scaled"] = (t1 - t0) * (n / m) * 1e3
    res["cpu_cores"] = os.cpu_count()
    print(json.dumps(res))


if __name__ == "__main__":
    main()

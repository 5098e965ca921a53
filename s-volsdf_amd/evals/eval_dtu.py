"""DTU Chamfer evaluation on the HIP path (reference: evals/eval_dtu.py:52-196, modes 'pcd' and 'mesh'; SURVEY.md
section 8 row f4).

The reference is a command-line script around sklearn's kd-tree; the protocol is exposed here as functions on arrays
(`evaluate_scan`) and on the DTU file layout (`evaluate_scan_files`, `main`).  Every per-point step runs on the GPU
(csrc/svs_cloud.hip): greedy radius down-sampling, observation-mask filter, both nearest-neighbour passes, the
truncated means.  Only the file readers (PLY, .mat) and the random shuffle of the input order are host code.
"""
import argparse
import ctypes
import os

import numpy as np
import torch

from svs_hip import lib as _lib
from svs_hip.ops import _ptr, _stream


def trun_n_d(n, d):
    return int(n * 10 ** d) / 10 ** d


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _cloud(a):
    if torch.is_tensor(a):
        return a.detach().to(device=_dev(), dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float64))).to(_dev())


def _origin(clouds, pad):
    """Padded bounding box of the clouds (one library call per cloud, one read-back for all)."""
    L = _lib.load()
    clouds = [c for c in clouds if c.shape[0]]
    dev = clouds[0].device
    ws = torch.empty(L.svs_cloud_bounds_workspace_bytes() // 8, dtype=torch.float64, device=dev)
    box = torch.empty(len(clouds), 6, dtype=torch.float64, device=dev)
    for k, c in enumerate(clouds):
        _lib.check(L.svs_cloud_bounds(_ptr(c), c.shape[0], _ptr(ws), _ptr(box[k]), _stream()), "svs_cloud_bounds")
    box = box.cpu().numpy()
    return box[:, :3].min(0) - pad, box[:, 3:].max(0) + pad


def nearest_neighbor(ref, query, max_radius, cell=None, return_index=False):
    """dist (nq,) float64 [, idx (nq,) int32]: NearestNeighbors(n_neighbors=1).fit(ref).kneighbors(query) for
    neighbours closer than max_radius (larger results are upper bounds, inf if nothing is near)."""
    L = _lib.load()
    ref, query = _cloud(ref), _cloud(query)
    nr, nq = ref.shape[0], query.shape[0]
    dist = torch.empty(nq, dtype=torch.float64, device=ref.device)
    idx = torch.empty(nq, dtype=torch.int32, device=ref.device) if return_index else None
    if nr == 0:
        raise ValueError("Found array with 0 sample(s) while a minimum of 1 is required by NearestNeighbors.")   # sklearn's fit
    if nq == 0:
        return (dist, idx) if return_index else dist
    lo, hi = _origin([ref, query], 1e-6)
    if cell is None:
        # about two points per cell of the surface the cloud samples; never more than 2^20 cells per axis
        ext = float((hi - lo).max())
        cell = max(ext / (1 << 20), min(max_radius, 2.0 * ext / max(nr, 1) ** 0.5))
    ws = torch.empty(L.svs_cloud_grid_bytes(nr), dtype=torch.uint8, device=ref.device)
    org = (ctypes.c_double * 3)(*[float(v) for v in lo])
    _lib.check(L.svs_cloud_nn(_ptr(ref), nr, _ptr(query), nq, org, float(cell), float(max_radius), _ptr(ws), _ptr(dist), _ptr(idx),
                              _stream()), "svs_cloud_nn")
    return (dist, idx) if return_index else dist


def sample_mesh(vertices, triangles, thresh):
    """The cloud the reference's mesh mode evaluates (evals/eval_dtu.py:65-90): the vertices followed by the points of a
    regular grid on every triangle of non-zero area, spaced so that the samples are about `thresh` apart.  The
    per-triangle quantities are the script's numpy expressions; the points come from the library.  Returns (n,3) float64
    on the device."""
    L = _lib.load()
    vertices = np.asarray(vertices, np.float64)
    tri_vert = vertices[np.asarray(triangles)]
    v1 = tri_vert[:, 1] - tri_vert[:, 0]
    v2 = tri_vert[:, 2] - tri_vert[:, 0]
    l1 = np.linalg.norm(v1, axis=-1, keepdims=True)
    l2 = np.linalg.norm(v2, axis=-1, keepdims=True)
    area2 = np.linalg.norm(np.cross(v1, v2), axis=-1, keepdims=True)
    non_zero_area = (area2 > 0)[:, 0]
    l1, l2, area2, v1, v2, tri_vert = [arr[non_zero_area] for arr in [l1, l2, area2, v1, v2, tri_vert]]
    thr = thresh * np.sqrt(l1 * l2 / area2)
    n1 = np.floor(l1 / thr)
    n2 = np.floor(l2 / thr)
    verts_d = _cloud(vertices)
    n_tri = len(n1)
    if n_tri == 0:
        return verts_d
    if not (np.isfinite(n1).all() and np.isfinite(n2).all()):
        raise ValueError("sample_mesh: a triangle's sample counts are not finite")
    rec = torch.from_numpy(np.ascontiguousarray(np.concatenate([n1, n2, v1, v2, tri_vert[:, 0]], 1))).to(_dev())
    counts = torch.empty(n_tri, dtype=torch.int64, device=rec.device)
    _lib.check(L.svs_mesh_sample_count(_ptr(rec), n_tri, _ptr(counts), _stream()), "svs_mesh_sample_count")
    ends = torch.cumsum(counts, 0)
    offsets = (ends - counts).contiguous()
    total = int(ends[-1].item())
    out = torch.empty(verts_d.shape[0] + total, 3, dtype=torch.float64, device=rec.device)
    out[:verts_d.shape[0]] = verts_d
    new_pts = out[verts_d.shape[0]:]
    if total:
        _lib.check(L.svs_mesh_sample_points(_ptr(rec), n_tri, _ptr(offsets), _ptr(new_pts), _stream()), "svs_mesh_sample_points")
    return out


def radius_downsample(pts, radius):
    """Boolean keep-mask (n,) of the greedy radius down-sampling in index order (eval_dtu.py:104-118)."""
    L = _lib.load()
    pts = _cloud(pts)
    n = pts.shape[0]
    state = torch.empty(n, dtype=torch.uint8, device=pts.device)
    if n == 0:
        return state.bool()
    lo, hi = _origin([pts], 1e-6)
    if float((hi - lo).max()) / radius >= (1 << 21):
        raise ValueError("cloud extent / radius exceeds the 2^21 cells per axis of the grid")
    ws = torch.empty(L.svs_cloud_grid_bytes(n), dtype=torch.uint8, device=pts.device)
    org = (ctypes.c_double * 3)(*[float(v) for v in lo])
    open_ = torch.zeros(1, dtype=torch.int32, device=pts.device)
    _lib.check(L.svs_cloud_downsample_begin(_ptr(pts), n, org, float(radius), _ptr(ws), _ptr(state), _stream()),
               "svs_cloud_downsample_begin")
    for _ in range(n + 1):
        _lib.check(L.svs_cloud_downsample_round(org, n, float(radius), _ptr(ws), _ptr(state), _ptr(open_), _stream()),
                   "svs_cloud_downsample_round")
        if int(open_.item()) == 0:
            break
    return state == 1


def compact(pts, mask):
    L = _lib.load()
    pts = _cloud(pts)
    n = pts.shape[0]
    mask = mask.to(torch.uint8).contiguous()
    out = torch.empty_like(pts)
    ws = torch.empty(max(n, 1), dtype=torch.int32, device=pts.device)
    count = torch.zeros(1, dtype=torch.int32, device=pts.device)
    _lib.check(L.svs_cloud_compact(_ptr(pts), _ptr(mask), n, _ptr(ws), _ptr(out), _ptr(count), _stream()), "svs_cloud_compact")
    return out[:int(count.item())]


def mean_below(dist, max_dist):
    L = _lib.load()
    ws = torch.empty(L.svs_cloud_mean_workspace_bytes() // 8, dtype=torch.float64, device=dist.device)
    out = torch.empty(2, dtype=torch.float64, device=dist.device)
    _lib.check(L.svs_cloud_mean_below(_ptr(dist), dist.shape[0], float(max_dist), _ptr(ws), _ptr(out), _stream()), "svs_cloud_mean_below")
    return float(out[0].item())


def evaluate_scan(data_pcd, stl, ObsMask, BB, Res, ground_plane, downsample_density=0.2, patch_size=60, max_dist=20,
                  shuffle_rng=None, details=False):
    """evals/eval_dtu.py:99-192 for one scan -> (mean_d2s accuracy, mean_s2d completeness, overall) in mm.
    data_pcd (n,3): predicted cloud; stl (m,3): ground-truth scan; ObsMask (d0,d1,d2), BB (2,3), Res: ObsMask*.mat;
    ground_plane (4,): Plane*.mat 'P'.  shuffle_rng: numpy Generator for the reference's random index shuffle (:100-101;
    None = a fresh default_rng() like the reference, False = keep the given order)."""
    L = _lib.load()
    data_pcd = np.array(data_pcd, np.float64)
    if shuffle_rng is not False:
        (np.random.default_rng() if shuffle_rng is None else shuffle_rng).shuffle(data_pcd, axis=0)
    pts = _cloud(data_pcd)
    keep = radius_downsample(pts, downsample_density)
    data_down = compact(pts, keep)

    BB = np.ascontiguousarray(np.asarray(BB, np.float32))
    obs = torch.from_numpy(np.ascontiguousarray(np.asarray(ObsMask) != 0).astype(np.uint8)).to(pts.device)
    res = float(np.asarray(Res, np.float64).reshape(-1)[0])
    n = data_down.shape[0]
    inbound = torch.empty(n, dtype=torch.uint8, device=pts.device)
    in_obs = torch.empty(n, dtype=torch.uint8, device=pts.device)
    bb = (ctypes.c_float * 6)(*[float(v) for v in BB.reshape(-1)])
    _lib.check(L.svs_cloud_obs_filter(_ptr(data_down), n, bb, res, float(patch_size), _ptr(obs), *obs.shape, _ptr(inbound),
                                      _ptr(in_obs), _stream()), "svs_cloud_obs_filter")
    data_in = compact(data_down, inbound)
    data_in_obs = compact(data_down, in_obs)

    stl_d = _cloud(stl)
    dist_d2s = nearest_neighbor(stl_d, data_in_obs, max_dist)
    mean_d2s = mean_below(dist_d2s, max_dist)

    above = torch.empty(stl_d.shape[0], dtype=torch.uint8, device=pts.device)
    plane = (ctypes.c_double * 4)(*[float(v) for v in np.asarray(ground_plane, np.float64).reshape(-1)])
    _lib.check(L.svs_cloud_plane_side(_ptr(stl_d), stl_d.shape[0], plane, _ptr(above), _stream()), "svs_cloud_plane_side")
    stl_above = compact(stl_d, above)
    dist_s2d = nearest_neighbor(data_in, stl_above, max_dist)
    mean_s2d = mean_below(dist_s2d, max_dist)
    over_all = (mean_d2s + mean_s2d) / 2
    if details:
        return (mean_d2s, mean_s2d, over_all), dict(data_pcd=data_pcd, keep=keep, data_down=data_down, data_in=data_in,
                                                     data_in_obs=data_in_obs, dist_d2s=dist_d2s, stl_above=stl_above,
                                                     dist_s2d=dist_s2d)
    return mean_d2s, mean_s2d, over_all


def evaluate_scan_files(scan, datadir, dataset_dir, mode="pcd", **kw):
    """The reference's file layout (eval_dtu.py:59,121,139,158-161): {datadir}/mvsnet{scan:03}_l3.ply against
    {dataset_dir}/ObsMask/ObsMask{scan}_10.mat, Plane{scan}.mat (scan 82 uses Plane83) and Points/stl/stl{scan:03}_total.ply.
    mode 'mesh': the prediction is a triangle mesh, sampled at the down-sampling density first (:62-90)."""
    from scipy.io import loadmat
    from svs_hip.fusion import read_ply_mesh, read_ply_points
    pred = os.path.join(datadir, "mvsnet{:0>3}_l3.ply".format(scan))
    if mode == "mesh":
        vertices, triangles = read_ply_mesh(pred)
        data_pcd = sample_mesh(vertices, triangles, kw.get("downsample_density", 0.2)).cpu().numpy()
    else:
        data_pcd, _ = read_ply_points(pred)
    obs = loadmat(f"{dataset_dir}/ObsMask/ObsMask{scan}_10.mat")
    stl, _ = read_ply_points(f"{dataset_dir}/Points/stl/stl{scan:03}_total.ply")
    plane = loadmat(f"{dataset_dir}/ObsMask/Plane{83 if scan == 82 else scan}.mat")["P"]
    return evaluate_scan(data_pcd, stl, obs["ObsMask"], obs["BB"], obs["Res"], plane, **kw)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--datadir', type=str, default='', help='pred point cloud')
    parser.add_argument('--data_dir_root', type=str, default='data_s_volsdf', help='GT data dir')
    parser.add_argument('--scan', type=int, default=-1)
    parser.add_argument('--mode', type=str, default='pcd', choices=['mesh', 'pcd'])
    parser.add_argument('--downsample_density', type=float, default=0.2)
    parser.add_argument('--patch_size', type=float, default=60)
    parser.add_argument('--max_dist', type=float, default=20)
    args = parser.parse_args(argv)
    dataset_dir = os.path.join(args.data_dir_root, 'DTU', 'DTU_MVS_Data')
    scans = [21, 34, 38, 82, 24, 37, 40, 106, 110, 114, 118]
    if args.scan in scans:
        scans = [args.scan]
    results = []
    print("ply_name, accuracy(mm), completeness(mm), overall(mm)")
    for scan in scans:
        try:
            r = evaluate_scan_files(scan, args.datadir, dataset_dir, mode=args.mode, downsample_density=args.downsample_density,
                                    patch_size=args.patch_size, max_dist=args.max_dist)
        except (OSError, ValueError):
            r = (10000., 10000., 10000.)                      # the reference's fallback row (:163-167)
        print('scan{:0>3} {:.2f} {:.2f} {:.2f}'.format(scan, trun_n_d(r[0], 2), trun_n_d(r[1], 2), trun_n_d(r[2], 2)))
        results.append(list(r))
    results = np.array(results).mean(0)
    print('mean_err {:.3f} {:.3f} {:.3f}'.format(results[0], results[1], results[2]))
    return results


if __name__ == '__main__':
    main()

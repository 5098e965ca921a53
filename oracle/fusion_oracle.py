"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the depth-map fusion path, SURVEY.md section 8 row f3.

Follows helpers/utils.py:75-132 (reproject_with_depth, check_geometric_consistency) and runner.py:301-386
(filter_depth: mask aggregation, depth averaging, back-projection, vertex / colour arrays), keeping numpy's type
promotions (integer pixel grids x float32 depth -> float64; float32 / int32 -> float64; float32 comparisons against
python floats stay float32).

Pinning status:
  * check_geometric_consistency: PINNED by tests/golden/fusion_geo.npz, produced by running the reference's own
    function (tests/golden/make_fixtures.py::fx_fusion) -- with ONE substitution: cv2 is not installed in the image,
    so `cv2.remap` was bound to `remap_linear` below.
  * remap_linear (cv2.remap, INTER_LINEAR, BORDER_CONSTANT 0, float32 image): PARITY UNPINNED.  Restated from
    OpenCV's documented algorithm (imgproc remap: float maps are converted to fixed point with INTER_BITS = 5
    fractional bits via cvRound, the four weights come from the float bilinear table, border taps read 0).
  * filter_depth (runner.py:301-404: the loop, mask aggregation, depth averaging, back-projection, vertex / colour
    arrays) and the PLY vertex layout: PINNED by tests/golden/filter_depth.npz -- the reference's own function, its source
    taken from runner.py with `ast` (the module cannot be imported: hydra's get_config() runs at import) and executed
    unmodified on a synthetic scan folder, with cv2.remap bound as above and plyfile's PlyData / PlyElement replaced by a
    stub that captures the structured vertex array (make_fixtures.py::fx_filter_depth); the oracle reproduces every
    mask, vertex and colour exactly.  The PFM codec is pinned (tests/golden/pfm_codec.npz, reference datasets/data_io.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

F32 = np.float32


def remap_linear(img, mapx, mapy):
    """cv2.remap(img, mapx, mapy, interpolation=cv2.INTER_LINEAR): float32 image (H,W), float32 maps."""
    img = np.asarray(img, F32)
    H, W = img.shape
    fx = (np.asarray(mapx, F32) * F32(32.0)).astype(F32)
    fy = (np.asarray(mapy, F32) * F32(32.0)).astype(F32)
    bad = ~((fx > -2.1e9) & (fx < 2.1e9) & (fy > -2.1e9) & (fy < 2.1e9))
    sx = np.rint(np.where(bad, 0, fx)).astype(np.int64)           # cvRound: half to even
    sy = np.rint(np.where(bad, 0, fy)).astype(np.int64)
    ix = np.clip(sx >> 5, -32768, 32767)
    iy = np.clip(sy >> 5, -32768, 32767)
    ax = ((sx & 31).astype(F32) * F32(1.0 / 32.0)).astype(F32)
    ay = ((sy & 31).astype(F32) * F32(1.0 / 32.0)).astype(F32)
    one = F32(1.0)
    w = [(one - ay) * (one - ax), (one - ay) * ax, ay * (one - ax), ay * ax]
    out = np.zeros(fx.shape, F32)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        xx, yy = ix + dx, iy + dy
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        s = np.where(ok, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], F32(0.0)).astype(F32)
        out = (out + (s * w[k]).astype(F32)).astype(F32) if k else (s * w[k]).astype(F32)
    return np.where(bad, F32(0.0), out).astype(F32)


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """helpers/utils.py:75-113."""
    H, W = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, W), np.arange(0, H))
    x_ref, y_ref = x_ref.reshape([-1]), y_ref.reshape([-1])
    xyz_ref = np.matmul(np.linalg.inv(K_ref), np.vstack((x_ref, y_ref, np.ones_like(x_ref))) * depth_ref.reshape([-1]))
    xyz_src = np.matmul(np.matmul(E_src, np.linalg.inv(E_ref)), np.vstack((xyz_ref, np.ones_like(x_ref))))[:3]
    K_xyz_src = np.matmul(K_src, xyz_src)
    with np.errstate(all="ignore"):
        xy_src = K_xyz_src[:2] / K_xyz_src[2:3]
    x_src = xy_src[0].reshape([H, W]).astype(F32)
    y_src = xy_src[1].reshape([H, W]).astype(F32)
    sampled = remap_linear(depth_src, x_src, y_src)
    xyz_src = np.matmul(np.linalg.inv(K_src), np.vstack((xy_src, np.ones_like(x_ref))) * sampled.reshape([-1]))
    xyz_rep = np.matmul(np.matmul(E_ref, np.linalg.inv(E_src)), np.vstack((xyz_src, np.ones_like(x_ref))))[:3]
    depth_rep = xyz_rep[2].reshape([H, W]).astype(F32)
    K_xyz_rep = np.matmul(K_ref, xyz_rep)
    with np.errstate(all="ignore"):
        xy_rep = K_xyz_rep[:2] / K_xyz_rep[2:3]
    return depth_rep, xy_rep[0].reshape([H, W]).astype(F32), xy_rep[1].reshape([H, W]).astype(F32), x_src, y_src


def check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, filter_dist=1, filter_diff=0.01):
    """helpers/utils.py:115-132 -> (mask, depth_reprojected (0 where rejected), x2d_src, y2d_src)."""
    H, W = depth_ref.shape
    x_ref, y_ref = np.meshgrid(np.arange(0, W), np.arange(0, H))
    depth_rep, x_rep, y_rep, x_src, y_src = reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    with np.errstate(all="ignore"):
        dist = np.sqrt((x_rep - x_ref) ** 2 + (y_rep - y_ref) ** 2)
        rel = np.abs(depth_rep - depth_ref) / depth_ref
        mask = np.logical_and(dist < filter_dist, rel < filter_diff)
    depth_rep[~mask] = 0
    return mask, depth_rep, x_src, y_src


def fuse_view(ref, srcs, conf=0.0, filter_dist=1, filter_diff=0.01, thres_view=1, extra_mask=None):
    """One iteration of filter_depth's loop (runner.py:312-386).  ref / srcs[i]: dict(K, E, depth[, confidence, img]).
    -> dict(depth_avg float64, photo_mask, geo_mask, final_mask, xyz (n,3) float32, rgb (n,3) uint8)."""
    geo_sum = 0
    reps = []
    for s in srcs:
        m, d, _, _ = check_geometric_consistency(ref["depth"], ref["K"], ref["E"], s["depth"], s["K"], s["E"],
                                                 filter_dist, filter_diff)
        geo_sum = geo_sum + m.astype(np.int32)
        reps.append(d)
    depth_avg = (sum(reps) + ref["depth"]) / (geo_sum + 1)
    photo = ref["confidence"] > conf
    geo = (geo_sum >= thres_view) if len(srcs) else np.full(ref["depth"].shape, 0 >= thres_view)
    final = np.logical_and(photo, geo)
    if extra_mask is not None:
        final = np.logical_and(final, extra_mask > 0)
    H, W = depth_avg.shape[:2]
    x, y = np.meshgrid(np.arange(0, W), np.arange(0, H))
    x, y, depth = x[final], y[final], depth_avg[final]
    xyz_ref = np.matmul(np.linalg.inv(ref["K"]), np.vstack((x, y, np.ones_like(x))) * depth)
    xyz_world = np.matmul(np.linalg.inv(ref["E"]), np.vstack((xyz_ref, np.ones_like(x))))[:3]
    out = dict(depth_avg=np.asarray(depth_avg, np.float64), photo_mask=photo, geo_mask=geo, final_mask=final,
               xyz=xyz_world.transpose((1, 0)).astype(F32))
    if "img" in ref:
        out["rgb"] = (ref["img"][final] * 255).astype(np.uint8)
    return out


def ply_bytes(xyz, rgb):
    """What plyfile's PlyData([PlyElement.describe(vertex_all, 'vertex')]).write(f) emits for runner.py:389-400:
    binary little-endian, properties float x,y,z and uchar red,green,blue."""
    n = len(xyz)
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, k in enumerate("xyz"):
        rec[k] = np.asarray(xyz, F32)[:, i]
    for i, k in enumerate(("red", "green", "blue")):
        rec[k] = np.asarray(rgb, np.uint8)[:, i]
    return head + rec.tobytes()

"""Depth-map fusion on the HIP path (csrc/svs_fusion.hip).

`fuse_view` is one iteration of the reference's filter_depth loop (runner.py:312-386): geometric consistency of a
reference view against its source views, photometric mask, depth averaging and the coloured world-space points of the
surviving pixels.  `filter_depth` is the whole function on in-memory views or on a scan folder, ending in the same
binary PLY (`write_ply`).  The small camera matrices are formed on the host exactly as the reference forms them
(float32 inverses / products), everything per pixel runs on the GPU.
"""
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib
from .ops import _ptr, _ptr_array, _stream


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a, dtype):
    if torch.is_tensor(a):
        return a.detach().to(device=_dev(), dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.dtype(str(dtype).replace("torch.", "")))).to(_dev())


def pair_matrices(K_ref, E_ref, K_src, E_src):
    """The six matrices of reproject_with_depth (helpers/utils.py:82-111) for one (ref, src) pair, float64 (68,):
    inv(K_ref), E_src @ inv(E_ref), K_src, inv(K_src), E_ref @ inv(E_src), K_ref.  Inverses and products are taken
    in the dtype of the inputs (float32 from read_camera_parameters), as numpy does in the reference."""
    K_ref, E_ref, K_src, E_src = (np.asarray(m) for m in (K_ref, E_ref, K_src, E_src))
    mats = (np.linalg.inv(K_ref), np.matmul(E_src, np.linalg.inv(E_ref)), K_src, np.linalg.inv(K_src),
            np.matmul(E_ref, np.linalg.inv(E_src)), K_ref)
    return np.concatenate([np.asarray(m, np.float64).reshape(-1) for m in mats])


def fuse_view(ref, srcs, conf=0.0, filter_dist=1, filter_diff=0.01, thres_view=1, extra_mask=None, per_source=False,
              points=True):
    """ref / srcs[i]: dict(K (3,3), E (4,4), depth (H,W) [, confidence (H,W), img (H,W,3) float32 in [0,1]]); arrays or
    device tensors.  Returns device tensors: depth_avg (H,W) float64, photo_mask / geo_mask / final_mask (H,W) uint8,
    and with points=True xyz (n,3) float32, rgb (n,3) uint8 (row-major order of the surviving pixels); with
    per_source=True also src_mask, src_depth_reproj, src_x, src_y (n_src,H,W)."""
    L = _lib.load()
    dev = _dev()
    depth = _to_dev(ref["depth"], torch.float32)
    H, W = depth.shape
    confidence = _to_dev(ref["confidence"], torch.float32) if "confidence" in ref else torch.full((H, W), float("inf"), device=dev)
    n_src = len(srcs)
    src_depths = [_to_dev(s["depth"], torch.float32) for s in srcs]
    for d in src_depths:
        if tuple(d.shape) != (H, W):
            raise AssertionError("source depth map shape differs from the reference view's")     # runner.py:334
    per = L.svs_fuse_mats_per_src()
    mats = np.zeros((max(n_src, 1), per), np.float64)
    for v, s in enumerate(srcs):
        mats[v] = pair_matrices(ref["K"], ref["E"], s["K"], s["E"])
    mats_d = torch.from_numpy(mats).to(dev)
    out = dict(depth_avg=torch.empty(H, W, dtype=torch.float64, device=dev))
    for k in ("photo_mask", "geo_mask", "final_mask"):
        out[k] = torch.empty(H, W, dtype=torch.uint8, device=dev)
    if per_source:
        out["src_mask"] = torch.empty(n_src, H, W, dtype=torch.uint8, device=dev)
        for k in ("src_depth_reproj", "src_x", "src_y"):
            out[k] = torch.empty(n_src, H, W, dtype=torch.float32, device=dev)
    em = None if extra_mask is None else (_to_dev(extra_mask, torch.float32) > 0).to(torch.uint8).contiguous()
    _lib.check(L.svs_fuse_view(_ptr(depth), _ptr(confidence), _ptr_array(src_depths), _ptr(mats_d), n_src, H, W,
                               float(conf), float(filter_dist), float(filter_diff), int(thres_view), _ptr(em),
                               _ptr(out["depth_avg"]), _ptr(out["photo_mask"]), _ptr(out["geo_mask"]), _ptr(out["final_mask"]),
                               _ptr(out.get("src_mask")), _ptr(out.get("src_depth_reproj")), _ptr(out.get("src_x")),
                               _ptr(out.get("src_y")), _stream()), "svs_fuse_view")
    if points:
        K, E = np.asarray(ref["K"]), np.asarray(ref["E"])
        pm = np.concatenate([np.asarray(np.linalg.inv(K), np.float64).reshape(-1), np.asarray(np.linalg.inv(E), np.float64).reshape(-1)])
        pm_d = torch.from_numpy(pm).to(dev)
        img = _to_dev(ref["img"], torch.float32) if "img" in ref else None
        if img is not None and tuple(img.shape) != (H, W, 3):
            raise AssertionError("reference image shape differs from its depth map's")            # runner.py:322
        ws = torch.empty(H * W, dtype=torch.int32, device=dev)
        xyz = torch.empty(H * W, 3, dtype=torch.float32, device=dev)
        rgb = torch.empty(H * W, 3, dtype=torch.uint8, device=dev) if img is not None else None
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(L.svs_fuse_points(_ptr(out["depth_avg"]), _ptr(out["final_mask"]), _ptr(img), _ptr(pm_d), H, W, _ptr(ws),
                                     _ptr(xyz), _ptr(rgb), _ptr(count), _stream()), "svs_fuse_points")
        n = int(count.item())
        out["xyz"] = xyz[:n]
        if rgb is not None:
            out["rgb"] = rgb[:n]
    return out


def write_ply(filename, xyz, rgb):
    """The vertex-only binary PLY that runner.py:389-400 writes through plyfile: little-endian records of
    float x,y,z + uchar red,green,blue."""
    xyz = np.asarray(xyz.cpu() if torch.is_tensor(xyz) else xyz, np.float32)
    rgb = np.asarray(rgb.cpu() if torch.is_tensor(rgb) else rgb, np.uint8)
    rec = np.empty(len(xyz), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, k in enumerate("xyz"):
        rec[k] = xyz[:, i]
    for i, k in enumerate(("red", "green", "blue")):
        rec[k] = rgb[:, i]
    with open(filename, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\n"
                 "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(rec)).encode("ascii"))
        rec.tofile(f)


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1", "char": "i1",
              "int8": "i1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4"}


def read_ply_points(filename):
    """Vertex positions (n,3) float64 [and colours (n,3) uint8 or None] of an ascii / binary PLY -- what the Chamfer
    evaluator takes from open3d.io.read_point_cloud (evals/eval_dtu.py:96-97,139-140)."""
    with open(filename, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, n, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n, ndmin=2)
            cols = {name: rows[:, i] for i, (name, _) in enumerate(props)}
        else:
            order = "<" if fmt == "binary_little_endian" else ">"
            rec = np.fromfile(f, dtype=[(name, order + t) for name, t in props], count=n)
            cols = {name: rec[name] for name, _ in props}
    pts = np.stack([np.asarray(cols[k], np.float64) for k in "xyz"], 1)
    rgb = np.stack([np.asarray(cols[k], np.uint8) for k in ("red", "green", "blue")], 1) if "red" in cols else None
    return pts, rgb


def read_ply_mesh(filename):
    """(vertices (n,3) float64, triangles (m,3) int64) of an ascii / binary PLY with a face element -- what the Chamfer
    evaluator's mesh mode takes from open3d.io.read_triangle_mesh (evals/eval_dtu.py:65-68).  Faces with more than three
    corners are fanned around their first corner, as open3d's reader does."""
    with open(filename, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, elements = None, []                       # [name, count, [(property name, type) or (name, count type, item type)]]
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PLY header without end_header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), []])
            elif tok[0] == "property":
                elements[-1][2].append((tok[4], _PLY_TYPES[tok[2]], _PLY_TYPES[tok[3]]) if tok[1] == "list"
                                       else (tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        order = "<" if fmt == "binary_little_endian" else ">"
        vertices, faces = None, []
        for name, n, props in elements:
            has_list = any(len(q) == 3 for q in props)
            if not has_list:
                if fmt == "ascii":
                    rows = np.loadtxt(f, max_rows=n, ndmin=2) if n else np.zeros((0, len(props)))
                    cols = {q[0]: rows[:, i] for i, q in enumerate(props)}
                else:
                    rec = np.fromfile(f, dtype=[(q[0], order + q[1]) for q in props], count=n)
                    cols = {q[0]: rec[q[0]] for q in props}
                if name == "vertex":
                    vertices = np.stack([np.asarray(cols[k], np.float64) for k in "xyz"], 1)
                continue
            for _ in range(n):                           # elements with list properties: record by record
                if fmt == "ascii":
                    tok = f.readline().split()
                    at = 0
                for q in props:
                    if len(q) == 2:
                        if fmt == "ascii":
                            at += 1
                        else:
                            f.read(np.dtype(q[1]).itemsize)
                        continue
                    if fmt == "ascii":
                        k = int(tok[at]); items = [int(float(v)) for v in tok[at + 1:at + 1 + k]]; at += 1 + k
                    else:
                        k = int(np.frombuffer(f.read(np.dtype(q[1]).itemsize), order + q[1])[0])
                        items = np.frombuffer(f.read(k * np.dtype(q[2]).itemsize), order + q[2]).astype(np.int64).tolist()
                    if name == "face" and q[0] in ("vertex_indices", "vertex_index"):
                        faces.extend((items[0], items[i], items[i + 1]) for i in range(1, k - 1))
    if vertices is None:
        raise ValueError("PLY file without a vertex element")
    return vertices, np.asarray(faces, np.int64).reshape(-1, 3)


def filter_depth(views, pairs, conf=0.0, filter_dist=1, filter_diff=0.01, thres_view=1, plyfilename=None, eval_masks=None,
                 mask_dir=None):
    """filter_depth (runner.py:301-401) on in-memory views.  views: {view_id: dict(K, E, img, depth, confidence)},
    pairs: [(ref_view, [src_view, ...]), ...] (runner.py:303-305 builds all-vs-all pairs of the training views),
    eval_masks: optional {view_id: (H,W) mask} already dilated / resized (runner.py:349-368 is image preprocessing).
    Returns (vertices (n,3) float32, colours (n,3) uint8, per-view stats); writes the PLY if plyfilename is given and
    the three masks per view under mask_dir."""
    from helpers.utils import save_mask
    vertexs, colors, stats = [], [], []
    for ref_view, src_views in pairs:
        ref = views[ref_view]
        out = fuse_view(ref, [views[s] for s in src_views], conf=conf, filter_dist=filter_dist, filter_diff=filter_diff,
                        thres_view=thres_view, extra_mask=None if eval_masks is None else eval_masks[ref_view])
        vertexs.append(out["xyz"])
        colors.append(out["rgb"])
        m = torch.stack([out[k].float().mean() for k in ("photo_mask", "geo_mask", "final_mask")]).cpu().numpy()
        stats.append((ref_view, float(m[0]), float(m[1]), float(m[2])))
        if mask_dir is not None:
            os.makedirs(mask_dir, exist_ok=True)
            for k, tag in (("photo_mask", "photo"), ("geo_mask", "geo"), ("final_mask", "final")):
                save_mask(os.path.join(mask_dir, "{:0>8}_{}.png".format(ref_view, tag)), out[k].cpu().numpy().astype(bool))
    xyz = torch.cat(vertexs, 0).cpu().numpy() if vertexs else np.zeros((0, 3), np.float32)
    rgb = torch.cat(colors, 0).cpu().numpy() if colors else np.zeros((0, 3), np.uint8)
    if plyfilename is not None:
        write_ply(plyfilename, xyz, rgb)
    return xyz, rgb, stats


def filter_depth_folder(scan_folder, out_folder, plyfilename, view_ids, conf=0.0, filter_dist=1, filter_diff=0.01,
                        thres_view=1):
    """The file-level form (runner.py:301-332): cams/{id:08}_cam.txt and images/{id:08}.jpg under scan_folder,
    depth_est/ and confidence/ PFMs under out_folder; all-vs-all pairs of view_ids."""
    from datasets.data_io import read_pfm
    from helpers.utils import read_camera_parameters, read_img
    views = {}
    for v in view_ids:
        K, E = read_camera_parameters(os.path.join(scan_folder, "cams/{:0>8}_cam.txt".format(v)))
        views[v] = dict(K=K, E=E, img=read_img(os.path.join(scan_folder, "images/{:0>8}.jpg".format(v))),
                        depth=np.ascontiguousarray(read_pfm(os.path.join(out_folder, "depth_est/{:0>8}.pfm".format(v)))[0]),
                        confidence=np.ascontiguousarray(read_pfm(os.path.join(out_folder, "confidence/{:0>8}.pfm".format(v)))[0]))
    pairs = [(v, [x for x in view_ids if x != v]) for v in view_ids]
    return filter_depth(views, pairs, conf, filter_dist, filter_diff, thres_view, plyfilename,
                        mask_dir=os.path.join(out_folder, "mask"))

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the DTU Chamfer protocol, SURVEY.md section 8 row f4.

The neighbour searches of the reference live in a third-party dependency that IS installed here: scikit-learn
(sklearn.neighbors.NearestNeighbors, kd_tree; 1.7.2 in this image, unpinned in the reference's requirements).  The
oracle calls it exactly as evals/eval_dtu.py:104-176 does and restates the numpy steps around it.
PINNED by tests/golden/chamfer_ref.npz: the reference script itself run end to end on a synthetic scan
(tests/golden/make_fixtures.py::fx_chamfer; open3d's PLY reader is replaced by a numpy reader, the unseeded shuffle
by a seeded one); the mesh-mode sampler below by tests/golden/chamfer_mesh_ref.npz (the script run with --mode mesh,
fx_chamfer_mesh).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import sklearn.neighbors as skln


def sample_single_tri(n1, n2, v1, v2, tri_vert):
    """eval_dtu.py:14-23: the grid points of one triangle."""
    c = np.mgrid[:n1 + 1, :n2 + 1]
    c += 0.5
    c[0] /= max(n1, 1e-7)
    c[1] /= max(n2, 1e-7)
    c = np.transpose(c, (1, 2, 0))
    k = c[c.sum(axis=-1) < 1]
    return v1 * k[:, :1] + v2 * k[:, 1:] + tri_vert


def sample_mesh(vertices, triangles, thresh):
    """eval_dtu.py:65-90 -> (data_pcd = [vertices ; sampled points], points per triangle of non-zero area)."""
    tri_vert = vertices[triangles]
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
    new_pts = [sample_single_tri(n1[i, 0], n2[i, 0], v1[i:i + 1], v2[i:i + 1], tri_vert[i:i + 1, 0]) for i in range(len(n1))]
    per_tri = np.asarray([len(q) for q in new_pts], np.int64)
    new_pts = np.concatenate(new_pts, axis=0) if new_pts else np.zeros((0, 3))
    return np.concatenate([vertices, new_pts], axis=0), per_tri


def radius_downsample(data_pcd, thresh, n_jobs=-1):
    """eval_dtu.py:104-118 -> keep mask."""
    nn_engine = skln.NearestNeighbors(n_neighbors=1, radius=thresh, algorithm='kd_tree', n_jobs=n_jobs)
    nn_engine.fit(data_pcd)
    rnn_idxs = nn_engine.radius_neighbors(data_pcd, radius=thresh, return_distance=False)
    mask = np.ones(data_pcd.shape[0], dtype=np.bool_)
    for curr, idxs in enumerate(rnn_idxs):
        if mask[curr]:
            mask[idxs] = 0
            mask[curr] = 1
    return mask


def nn_distance(ref, query, n_jobs=-1):
    """eval_dtu.py:150-152 / :174-175 -> (dist (nq,), idx (nq,))."""
    nn_engine = skln.NearestNeighbors(n_neighbors=1, algorithm='kd_tree', n_jobs=n_jobs)
    nn_engine.fit(ref)
    d, i = nn_engine.kneighbors(query, n_neighbors=1, return_distance=True)
    return d[:, 0], i[:, 0]


def obs_filter(data_down, ObsMask, BB, Res, patch):
    """eval_dtu.py:124-135 -> (inbound mask over data_down, in_obs mask over data_down)."""
    BB = BB.astype(np.float32)
    inbound = ((data_down >= BB[:1] - patch) & (data_down < BB[1:] + patch * 2)).sum(axis=-1) == 3
    data_in = data_down[inbound]
    data_grid = np.around((data_in - BB[:1]) / Res).astype(np.int32)
    grid_inbound = ((data_grid >= 0) & (data_grid < np.expand_dims(ObsMask.shape, 0))).sum(axis=-1) == 3
    data_grid_in = data_grid[grid_inbound]
    in_obs = ObsMask[data_grid_in[:, 0], data_grid_in[:, 1], data_grid_in[:, 2]].astype(np.bool_)
    full = np.zeros(len(data_down), np.bool_)
    full[np.where(inbound)[0][grid_inbound][in_obs]] = True
    return inbound, full


def evaluate_scan(data_pcd, stl, ObsMask, BB, Res, ground_plane, thresh=0.2, patch=60, max_dist=20, n_jobs=-1):
    """eval_dtu.py:104-192 after the shuffle (data_pcd is taken in the given order)."""
    keep = radius_downsample(data_pcd, thresh, n_jobs)
    data_down = data_pcd[keep]
    inbound, in_obs = obs_filter(data_down, ObsMask, BB, Res, patch)
    data_in, data_in_obs = data_down[inbound], data_down[in_obs]
    dist_d2s, _ = nn_distance(stl, data_in_obs, n_jobs)
    mean_d2s = dist_d2s[dist_d2s < max_dist].mean()
    stl_hom = np.concatenate([stl, np.ones_like(stl[:, :1])], -1)
    above = (ground_plane.reshape((1, 4)) * stl_hom).sum(-1) > 0
    stl_above = stl[above]
    dist_s2d, _ = nn_distance(data_in, stl_above, n_jobs)
    mean_s2d = dist_s2d[dist_s2d < max_dist].mean()
    return (mean_d2s, mean_s2d, (mean_d2s + mean_s2d) / 2), dict(keep=keep, data_down=data_down, data_in=data_in,
                                                                 data_in_obs=data_in_obs, dist_d2s=dist_d2s,
                                                                 stl_above=stl_above, dist_s2d=dist_s2d)

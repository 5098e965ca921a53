"""The runner-side helpers of the reference (helpers/utils.py:11-132) that feed the depth-fusion path, with the same
names and return conventions.  `check_geometric_consistency` runs on the HIP kernel (csrc/svs_fusion.hip); the file
readers / writers are host-side parsing.  With a reference checkout importable every other helper of its
helpers/utils.py (`tocuda`, `tensor2numpy`, `load_K_Rt_from_P`, `glob_imgs`, ... -- runner.py:36 star-imports them) is
re-exported unchanged (svs_hip/refpath.py); only `check_geometric_consistency` and the readers below are replaced."""
import numpy as np

from svs_hip.refpath import overlay

overlay(globals(), __name__)


def read_camera_parameters(filename):
    """cams/*_cam.txt -> (intrinsics (3,3) float32, extrinsics (4,4) float32)   (helpers/utils.py:13-21)."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape((4, 4))
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape((3, 3))
    return intrinsics, extrinsics


def write_cam(file, cam, cam_near_far=None):
    """cam: (2,4,4) [extrinsic, intrinsic (+ depth range in row 3)]   (helpers/utils.py:54-73)."""
    with open(file, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
        if cam_near_far is not None:
            f.write("\n%.4f %.4f %.4f %.4f\n" % tuple(cam_near_far[:4]))
        else:
            f.write("\n" + " ".join(str(cam[1][3][j]) for j in range(4)) + "\n")


def read_img(filename):
    """image -> float32 in [0,1]   (helpers/utils.py:24-28)."""
    from PIL import Image
    return np.array(Image.open(filename), dtype=np.float32) / 255.


def read_mask(filename):
    return read_img(filename) > 0.5


def save_mask(filename, mask):
    from PIL import Image
    assert mask.dtype == np.bool_
    Image.fromarray(mask.astype(np.uint8) * 255).save(filename)


def read_pair_file(filename):
    """pair.txt -> [(ref_view, [src_view, ...]), ...]   (helpers/utils.py:41-51)."""
    data = []
    with open(filename) as f:
        for _ in range(int(f.readline())):
            ref_view = int(f.readline().rstrip())
            src_views = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if len(src_views) > 0:
                data.append((ref_view, src_views))
    return data


def check_geometric_consistency(depth_ref, intrinsics_ref, extrinsics_ref, depth_src, intrinsics_src, extrinsics_src,
                                filter_dist=1, filter_diff=0.01):
    """helpers/utils.py:115-132 on the GPU -> (mask bool (H,W), depth_reprojected float32 (0 where rejected),
    x2d_src, y2d_src float32), numpy arrays like the reference's."""
    from svs_hip import fusion
    out = fusion.fuse_view(dict(depth=depth_ref, K=intrinsics_ref, E=extrinsics_ref),
                           [dict(depth=depth_src, K=intrinsics_src, E=extrinsics_src)],
                           filter_dist=filter_dist, filter_diff=filter_diff, per_source=True, points=False)
    return (out["src_mask"][0].cpu().numpy().astype(bool), out["src_depth_reproj"][0].cpu().numpy(),
            out["src_x"][0].cpu().numpy(), out["src_y"][0].cpu().numpy())

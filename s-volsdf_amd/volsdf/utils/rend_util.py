"""`volsdf.utils.rend_util`: the two functions of the reference module that are on the hot path run here
(volsdf/utils/rend_util.py:60-95,143-156 `get_camera_params` + `lift` -> HIP ray kernel; :200-216
`get_sphere_intersections`), `get_psnr` (:14-22) is kept so that the path runs without a checkout.  Every other name
of the reference module (`load_rgb`, `load_K_Rt_from_P`, `get_uv`, `quat_to_rot`, ... -- image IO and camera prep its
datasets and plots use) is re-exported from a reference checkout when one is importable (svs_hip/refpath.py).
"""
import torch

from svs_hip import ops
from svs_hip.refpath import overlay

_reference = overlay(globals(), __name__)


def get_psnr(img1, img2, normalize_rgb=False):
    if normalize_rgb:
        img1, img2 = (img1 + 1.) / 2., (img2 + 1.) / 2.
    mse = torch.mean((img1 - img2) ** 2)
    return -10. * torch.log(mse) / torch.log(torch.tensor([10.], device=mse.device))


def get_camera_params(uv, pose, intrinsics):
    """uv (B,N,2), pose (B,4,4), intrinsics (B,4,4) -> ray_dirs (B,N,3), cam_loc (B,3).  The trainer's case (one view,
    4x4 pose: runner.py:164-169) runs on the ray kernel; quaternion poses and B > 1 are not on the hot path and go to
    the reference's implementation when its checkout is importable."""
    if pose.shape[1] == 7 or uv.shape[0] != 1 or not uv.is_cuda:
        if _reference is not None:
            return _reference.get_camera_params(uv, pose, intrinsics)
        raise NotImplementedError("only one view with a 4x4 pose on the device (runner.py:166) without a reference checkout")
    dirs, cam, _ = ops.rays_from_uv(uv[0], pose[0], intrinsics[0])
    return dirs[None], cam[None]


def get_sphere_intersections(cam_loc, ray_directions, r=1.0):
    """(n,3),(n,3) -> (n,2) near/far clamped at 0; raises instead of the reference's exit() on a miss."""
    dot = (ray_directions * cam_loc).sum(-1, keepdim=True)
    under = dot ** 2 - (cam_loc.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    if (under <= 0).any():
        raise RuntimeError("BOUNDING SPHERE PROBLEM!")
    s = torch.sqrt(under)
    return torch.cat([-s - dot, s - dot], -1).clamp_min(0.0)

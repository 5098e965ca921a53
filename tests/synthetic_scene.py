"""In-memory stand-in for the reference's SceneDataset (volsdf/datasets/scene_dataset.py:96-300) with the interface
VolOpt uses: `__len__`, `__getitem__ -> (idx, sample, ground_truth)`, `collate_fn`, `change_sampling_idx`, `trains_ids`,
`total_pixels`, `img_res`, `scale_factor`, `intrinsics_all`, `pose_all`, `mode`.  Test infrastructure only."""
import random

import numpy as np
import torch

import synth


class SyntheticSceneDataset(torch.utils.data.Dataset):
    def __init__(self, data_dir_root=None, data_dir="DTU", img_res=(24, 32), scan_id=24, num_views=3, scale_factor=1.5, **_):
        self.data_dir, self.scan_id, self.num_views = data_dir, scan_id, num_views
        self.img_res = list(img_res)
        self.total_pixels = img_res[0] * img_res[1]
        self.mode, self.plot_id = 'train', 0
        self.sampling_idx = None
        self.n_images = 5
        self.scale_factor = float(scale_factor)
        rng = np.random.default_rng(scan_id)
        H, W = img_res
        self.rgb_images, self.rgb_smooth, self.masks, self.intrinsics_all, self.pose_all = [], [], [], [], []
        for v in range(self.n_images):
            K, pose = synth.make_camera(center=(0.25 * (v - 2), 0.05 * v, -2.5), tilt=-0.1 * (v - 2))
            K = np.array(K, np.float32).copy()
            K[0, :3] *= W / 768.0
            K[1, :3] *= H / 576.0
            self.intrinsics_all.append(torch.from_numpy(K).float())
            self.pose_all.append(torch.from_numpy(np.array(pose, np.float32)).float())
            img = rng.uniform(0, 1, (self.total_pixels, 3)).astype(np.float32)
            self.rgb_images.append(torch.from_numpy(img))
            self.rgb_smooth.append(torch.from_numpy(0.5 * img + 0.25))
            self.masks.append(torch.ones(self.total_pixels, 3))

    def __len__(self):
        return self.n_images

    def trains_ids(self):
        return [0, 2, 4][:self.num_views]

    def __getitem__(self, idx):
        if self.mode == 'train':
            idx = self.trains_ids()[random.randint(0, self.num_views - 1)]
        elif self.mode == 'plot':
            idx = [1, 3][self.plot_id]
            self.plot_id = (self.plot_id + 1) % 2
        uv = np.mgrid[0:self.img_res[0], 0:self.img_res[1]].astype(np.int32)
        uv = torch.from_numpy(np.flip(uv, axis=0).copy()).float().reshape(2, -1).transpose(1, 0)
        sample = {"uv": uv, "intrinsics": self.intrinsics_all[idx], "pose": self.pose_all[idx]}
        if self.data_dir == "BlendedMVS":        # scene_dataset.py:239-240: the pose of a neighbouring view (eval-mode bg colours)
            sample["near_pose"] = self.pose_all[(idx + 1) % self.n_images]
        gt = {"rgb": self.rgb_images[idx], "rgb_smooth": self.rgb_smooth[idx], "mask": self.masks[idx]}
        if self.sampling_idx is not None:
            gt["rgb"] = self.rgb_images[idx][self.sampling_idx, :]
            gt["rgb_smooth"] = self.rgb_smooth[idx][self.sampling_idx, :]
            sample["uv"] = uv[self.sampling_idx, :]
        return idx, sample, gt

    def collate_fn(self, batch_list):
        out = []
        for entry in zip(*batch_list):
            if isinstance(entry[0], dict):
                out.append({k: torch.stack([o[k] for o in entry]) for k in entry[0]})
            else:
                out.append(torch.LongTensor(entry))
        return tuple(out)

    def change_sampling_idx(self, sampling_size):
        self.sampling_idx = None if sampling_size == -1 else torch.randperm(self.total_pixels)[:sampling_size]


class AnalyticSceneDataset(SyntheticSceneDataset):
    """The same interface over images of the analytic scene of synth.render_analytic_view (sphere + box, Lambertian):
    a scan with KNOWN geometry, for the Chamfer parity runs (tools/chamfer_parity.py).  `rgb_smooth` is a 5 x 5 box blur of
    the image (the reference smooths with a Gaussian, scene_dataset.py:150-170; only the annealing phase of a run with an MVS
    prior reads it).  scale_factor: world units (mm) per normalised scene unit, as the DTU scans' scale matrices give."""

    def __init__(self, data_dir_root=None, data_dir="DTU", img_res=(192, 256), scan_id=24, num_views=3, scale_factor=200.0, **kw):
        super().__init__(data_dir_root, data_dir, img_res, scan_id, num_views, scale_factor, **kw)
        H, W = img_res
        self.renders = []
        for v in range(self.n_images):
            r = synth.render_analytic_view(self.intrinsics_all[v].numpy(), self.pose_all[v].numpy(), (H, W))
            self.renders.append(r)
            img = r["rgb"]
            pad = np.pad(img, ((2, 2), (2, 2), (0, 0)), mode="edge")
            smooth = sum(pad[i:i + H, j:j + W] for i in range(5) for j in range(5)) / 25.0
            self.rgb_images[v] = torch.from_numpy(img.reshape(-1, 3).copy())
            self.rgb_smooth[v] = torch.from_numpy(smooth.reshape(-1, 3).astype(np.float32))
            self.masks[v] = torch.from_numpy(np.repeat(r["mask"].reshape(-1, 1), 3, 1).astype(np.float32))

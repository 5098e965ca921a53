"""Train batches drawn on the device (opt-in: `VolOpt(..., device_batches=True)` or SVS_DEVICE_BATCHES=1).

The reference feeds `VolOpt.train_step` from `SceneDataset` through a DataLoader (volsdf/vsdf.py:351-357,
volsdf/datasets/scene_dataset.py:211-253,275-279): every step builds the full pixel grid of the image with numpy
(`np.mgrid`, flip, copy: 442 368 x 2 values at 576 x 768), draws `torch.randperm(total_pixels)` on the CPU -- twice, once in
`run` / `train_step` each -- and collates the whole mask image.  Measured with the reference's one torch thread
(vsdf.py:21): ~40 ms of host work per step, against 3.9 ms for the step itself on the GPU.  That loop is the reference's
and stays the default (same batches for the same seeds); this module is the alternative for a run that wants the hot path's
speed end to end:

  * the train views' `rgb` / `rgb_smooth` images live on the device (two float32 (P,3) tensors per view);
  * a batch = a random train view (Python's `random.randint`, the reference's own draw) + `num_pixels` distinct pixels drawn
    by `torch.randperm` ON THE DEVICE (same distribution as the reference's CPU draw, a different random stream) +
    uv = (index % W, index // W) [+ 0.5 with use_pixel_centers] -- no image-sized temporary, no host-to-device copy;
  * the batch has the layout `SceneDataset.collate_fn` produces for batch size 1: `(indices (1,), {uv (1,R,2), intrinsics
    (1,4,4), pose (1,4,4)[, near_pose]}, {rgb (1,R,3), rgb_smooth (1,R,3)})`.  `mask` is not part of a train batch here
    (train_step does not read it).
"""
import random
import sys

import torch


class DeviceBatches:
    def __init__(self, dataset, num_pixels, device):
        self.ds, self.num_pixels, self.device = dataset, int(num_pixels), torch.device(device)
        self.W = int(dataset.img_res[1])
        self.total_pixels = int(dataset.total_pixels)
        # scene_dataset.py:215-219: with num_views >= 1 a train item is a random one of the first num_views training views;
        # otherwise (num_views < 1: every image is a training image) the DataLoader's shuffled index over all images
        self.num_views = int(getattr(dataset, "num_views", 1))
        if self.num_views >= 1:
            self.train_ids = list(dataset.trains_ids())[:self.num_views]
            if len(self.train_ids) != self.num_views:
                raise ValueError(f"dataset.trains_ids() has {len(self.train_ids)} entries, num_views = {self.num_views}")
        else:
            self.train_ids = list(range(int(dataset.n_images)))
        self._order = []
        self.centers = bool(getattr(dataset, "use_pixel_centers", False))
        up = lambda t: t.to(self.device, dtype=torch.float32).contiguous()
        self.rgb = {i: up(dataset.rgb_images[i]) for i in self.train_ids}
        self.rgb_smooth = {i: up(dataset.rgb_smooth[i]) for i in self.train_ids}
        self.K = {i: up(dataset.intrinsics_all[i])[None] for i in self.train_ids}
        self.pose = {i: up(dataset.pose_all[i])[None] for i in self.train_ids}
        self.near = None
        if getattr(dataset, "data_dir", None) in ("BlendedMVS",):
            # the reference adds the pose of a neighbouring view (scene_dataset.py:239-240) through a module-level helper
            get_near_id = getattr(sys.modules.get(type(dataset).__module__), "get_near_id", None)
            if get_near_id is None:
                raise NotImplementedError("device batches for BlendedMVS need the dataset module's get_near_id")
            self.near = {i: up(dataset.pose_all[get_near_id(data_dir=dataset.data_dir, scan_id=dataset.scan_id, idx=i)])[None]
                         for i in self.train_ids}
        self.idx_t = {i: torch.tensor([i], dtype=torch.long) for i in self.train_ids}

    def __len__(self):
        return len(self.ds)            # an epoch is len(dataset) steps, as with the reference's DataLoader

    def batch(self):
        if self.num_views >= 1:
            view = self.train_ids[random.randint(0, self.num_views - 1)]
        else:
            if not self._order:                      # a shuffled pass over all images, like DataLoader(shuffle=True)
                self._order = torch.randperm(len(self.train_ids)).tolist()
            view = self.train_ids[self._order.pop()]
        idx = torch.randperm(self.total_pixels, device=self.device)[:self.num_pixels]
        uv = torch.stack([idx % self.W, idx // self.W], -1).to(torch.float32)
        if self.centers:
            uv = uv + 0.5
        sample = {"uv": uv[None], "intrinsics": self.K[view], "pose": self.pose[view]}
        if self.near is not None:
            sample["near_pose"] = self.near[view]
        gt = {"rgb": self.rgb[view][idx][None], "rgb_smooth": self.rgb_smooth[view][idx][None]}
        return self.idx_t[view], sample, gt

    def __iter__(self):
        """One epoch of batches.  Batch i+1 is drawn on a side stream while step i runs (the device `randperm` and the gathers
        are 0.2 ms of small kernels that would otherwise sit in front of every step); the consumer's stream waits for it."""
        if self.device.type != "cuda":
            for _ in range(len(self)):
                yield self.batch()
            return
        side = getattr(self, "_side", None)
        if side is None:
            side = self._side = torch.cuda.Stream(device=self.device)

        def draw():
            main = torch.cuda.current_stream(self.device)
            side.wait_stream(main)                       # (the images were uploaded on the consumer's stream)
            with torch.cuda.stream(side):
                b = self.batch()
                ev = torch.cuda.Event(); ev.record(side)
            for d in b[1:]:
                for t in d.values():
                    if torch.is_tensor(t) and t.is_cuda:
                        t.record_stream(main)            # allocated on the side stream, consumed on the main one
            return b, ev
        nxt = draw()
        for i in range(len(self)):
            cur, ev = nxt
            if i + 1 < len(self):
                nxt = draw()
            torch.cuda.current_stream(self.device).wait_event(ev)
            yield cur

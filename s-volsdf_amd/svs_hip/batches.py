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


class CachedItems(torch.utils.data.Dataset):
    """The reference's train items without their per-item image-sized work -- the SAME batches and the same use of the
    random generators, for the DataLoader loop that stays the default (`VolOpt.gen_dataset` wraps the train dataset in this;
    SVS_CACHED_ITEMS=0 or `cached_items=False` turns it off).

    `SceneDataset.__getitem__` (volsdf/datasets/scene_dataset.py:211-253) rebuilds the whole pixel grid with numpy for
    every item (`np.mgrid`, flip, copy, transpose: 442 368 x 2 values at 576 x 768), gathers `num_pixels` rows of it and of the
    two images, and the collate stacks -- copies -- the view's full mask image (5.3 MB): 1.2 ms of host work per step, as
    much as a 256-ray step takes on the GPU (config 4 over 8 GPUs).  Here the grid is built ONCE, a view's static parts are
    taken from the first item the dataset's OWN method returns for it, and later items of that view are assembled by
    gathering the `num_pixels` rows from the cached grid and the dataset's image tensors; a batch of one item is collated by
    `unsqueeze(0)` views.

    Nothing is assumed about the dataset class; it is OBSERVED.  The first time a view comes up the dataset's own
    `__getitem__` builds the item (the reference's path, unchanged) and the assembly is checked against it: same keys, every
    tensor equal, the same view index, and Python's `random` generator in the same state as after the single
    `random.randint(0, num_views - 1)` the fast path makes (scene_dataset.py:219).  Any difference -- another dataset class,
    another mode, missing attributes -- and this object hands every item to the dataset's own method from then on (`reason`
    says why).  The train-mode item of a BlendedMVS scan carries `near_pose` (scene_dataset.py:239-240): taken from the
    observed item of the view."""

    def __init__(self, dataset):
        self.ds = dataset
        self.reason = None              # why the fast path is off (None: it is on)
        self._views = {}                # view index -> dict(static sample entries, mask)
        self._uv = None
        self.fast_items = self.own_items = 0
        need = ("rgb_images", "rgb_smooth", "masks", "intrinsics_all", "pose_all", "img_res", "num_views", "trains_ids", "mode")
        missing = [a for a in need if not hasattr(dataset, a)]
        if missing:
            self.reason = "dataset has no " + ", ".join(missing)
        elif int(dataset.num_views) < 1:
            self.reason = "num_views < 1: the item index is the DataLoader's"

    def __len__(self):
        return len(self.ds)

    def __getattr__(self, name):        # (everything else -- trains_ids, total_pixels, change_sampling_idx ... -- is the dataset's)
        ds = self.__dict__.get("ds")
        if ds is None:                  # being unpickled / deep-copied (DataLoader workers): no attribute yet, not a KeyError
            raise AttributeError(name)
        return getattr(ds, name)

    # ---- the cached parts
    def _grid(self):
        if self._uv is None:
            import numpy as np
            H, W = int(self.ds.img_res[0]), int(self.ds.img_res[1])
            uv = np.mgrid[0:H, 0:W].astype(np.int32)                                  # scene_dataset.py:228-233, once
            uv = torch.from_numpy(np.flip(uv, axis=0).copy()).float()
            uv = uv.reshape(2, -1).transpose(1, 0)
            if getattr(self.ds, "use_pixel_centers", False):
                uv = uv + 0.5
            self._uv = uv.contiguous()
        return self._uv

    def _assemble(self, idx, view):
        ds, sel = self.ds, self.ds.sampling_idx
        sample = dict(view["sample"])
        gt = {"rgb": ds.rgb_images[idx], "rgb_smooth": ds.rgb_smooth[idx], "mask": ds.masks[idx]}
        if sel is not None:
            gt["rgb"] = ds.rgb_images[idx][sel, :]
            gt["rgb_smooth"] = ds.rgb_smooth[idx][sel, :]
            sample["uv"] = self._grid()[sel, :]
        else:
            sample["uv"] = self._grid()
        # the order of the keys is the reference's: uv, intrinsics, pose[, near_pose]
        return idx, {k: sample[k] for k in view["order"]}, gt

    @staticmethod
    def _same(a, b):
        if torch.is_tensor(a) != torch.is_tensor(b):
            return False
        if torch.is_tensor(a):
            return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b))
        return a == b

    def _learn(self, item_index, state):
        """the dataset's own item for a view not seen yet; the assembly must reproduce it"""
        random.setstate(state)
        item = self.ds[item_index]
        after = random.getstate()
        self.own_items += 1
        try:
            idx, sample, gt = item
            random.setstate(state)
            mine_idx = self.ds.trains_ids()[random.randint(0, int(self.ds.num_views) - 1)]
            ok = random.getstate() == after and int(mine_idx) == int(idx) and isinstance(sample, dict) and isinstance(gt, dict)
            if ok:
                static = {k: v for k, v in sample.items() if k != "uv"}
                view = dict(sample=static, order=list(sample.keys()))
                mine = self._assemble(int(idx), view)
                ok = (list(mine[1]) == list(sample) and list(mine[2]) == list(gt)
                      and all(self._same(mine[1][k], sample[k]) for k in sample) and all(self._same(mine[2][k], gt[k]) for k in gt)
                      and self._same(static.get("intrinsics"), self.ds.intrinsics_all[int(idx)])
                      and self._same(static.get("pose"), self.ds.pose_all[int(idx)]))
                if ok:
                    self._views[int(idx)] = view
            if not ok:
                self.reason = "the dataset's own item differs from the cached assembly"
        except Exception as e:                       # (an item of another shape: not ours to interpret)
            self.reason = f"could not check the dataset's item: {e!r}"
        random.setstate(after)
        return item

    def __getitem__(self, item_index):
        ds = self.ds
        if self.reason is not None or ds.mode != 'train':
            self.own_items += 1
            return ds[item_index]
        state = random.getstate()
        idx = ds.trains_ids()[random.randint(0, int(ds.num_views) - 1)]          # scene_dataset.py:216-219
        view = self._views.get(int(idx))
        if view is None:
            return self._learn(item_index, state)
        self.fast_items += 1
        return self._assemble(int(idx), view)

    def collate_fn(self, batch_list):
        """`SceneDataset.collate_fn` (scene_dataset.py:258-273) for a batch of ONE item without its copies: `torch.stack` of
        one tensor is `unsqueeze(0)` of it (the mask image alone is 5.3 MB per step); larger batches go to the dataset's."""
        if len(batch_list) != 1 or self.reason is not None:
            return self.ds.collate_fn(batch_list)
        out = []
        # Small tensors (intrinsics, pose, near_pose: 64 bytes each) are copied like torch.stack copies them; the large ones
        # (the mask image, the pixel grid when nothing is sampled, the images) stay VIEWS of the dataset's tensors -- READ-ONLY
        # by contract: an in-place operation on a batch would change the dataset for every later item (the reference's loop
        # only uploads them: volsdf/vsdf.py:331-335)
        for entry in batch_list[0]:
            if isinstance(entry, dict):
                out.append({k: (v.unsqueeze(0).clone() if v.numel() <= 64 else v.unsqueeze(0)) for k, v in entry.items()})
            else:
                out.append(torch.LongTensor([entry]))
        return tuple(out)

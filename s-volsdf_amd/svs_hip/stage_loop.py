"""The MVS side of a scan's stage loop (runner.py:178-243): feature extraction + cost volume per stage and per
reference view, and the hand-off of the rendered depths into the next stage.

As written in the reference every (stage, reference view) pair re-extracts the FPN features of all its views:
3 stages x 3 reference views x 3 views = 27 `model.feature(img)` calls per scan, of which only 3 are distinct (the
images do not change between stages and every view appears in every sample).  Here an image's features are computed
once and kept on the device for the scan (3 x 1.9 MB at 640x512 of 288 GB): `feature_calls` counts the real
extractions.  Images are recognised by content (shape + two float64 checksums), not by identity, because the data
loader hands out fresh tensors every time.
"""
import torch


class StageLoop:
    def __init__(self, model, cache_features=True):
        self.model = model
        self.cache_features = cache_features
        self._features = {}
        self.feature_calls = 0

    def clear(self):
        """forget the cached features (new scan) and what the cost-volume / prior look-up wrappers cached for the old one"""
        self._features.clear()
        from . import costvol
        costvol.clear_caches()

    @staticmethod
    def _fingerprint(img):
        x = img.detach().double().reshape(-1)
        ramp = torch.arange(x.numel(), device=x.device, dtype=torch.float64) % 8191.0
        s = torch.stack([x.sum(), (x * ramp).sum()]).cpu()
        return (tuple(img.shape), float(s[0]), float(s[1]))

    @torch.no_grad()
    def features(self, imgs):
        """imgs (B, N, C, H, W) -> list over views of {'stage1','stage2','stage3'} (runner.py:188-196)"""
        out = []
        for v in range(imgs.size(1)):
            img = imgs[:, v]
            key = self._fingerprint(img) if self.cache_features else None
            if key is None or key not in self._features:
                f = self.model.feature(img)
                self.feature_calls += 1
                if key is None:
                    out.append(f)
                    continue
                self._features[key] = f
            out.append(self._features[key])
        return out

    @torch.no_grad()
    def cost_volumes(self, stage_idx, samples, outs_samples, view_extra_samples=None, int_r=None, inverse_depth=False,
                     prevent_oom=False):
        """runner.py:182-207 for one stage: samples = the (device) samples of the scan's reference views;
        outs_samples[i] = the previous stage's outputs of view i (None at stage 0).  Returns (outs, view_extras)."""
        if int_r is None:
            int_r = self.model.depth_interals_ratio[stage_idx]
        n = len(samples)
        view_extra_samples = view_extra_samples or [None] * n
        outs, view_extras = [None] * n, [None] * n
        for i, sample in enumerate(samples):
            feats = self.features(sample["imgs"])
            outs[i], view_extras[i] = self.model(stage_idx, sample, features=feats, extra=view_extra_samples[i],
                                                 outputs=outs_samples[i], int_r=int_r, prevent_oom=prevent_oom,
                                                 inverse_depth=inverse_depth)
        return outs, view_extras

    @staticmethod
    def hand_off_depth(outs, stage_idx, depths):
        """runner.py:238-243: the volume-rendered depth replaces the MVS depth as the centre of the next stage's
        hypothesis window."""
        for i, d in enumerate(depths):
            outs[i][f"stage{stage_idx + 1}"]["depth"] = d
            outs[i]["depth"] = d
        return outs

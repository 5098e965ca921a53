"""CascadeMVSNet with the reference's constructor, parameter names and forward() contract
(models/CasMVSNet.py:667-761), whose cost-volume build runs on the HIP kernels of csrc/svs_costvol.hip:
fused homography warp + variance, 3-D U-Net with folded BatchNorm, softmax / depth regression / confidence,
depth hypotheses.  Checkpoints load with strict=True (`feature.*`, `cost_regularization.{0,1,2}.*`).

The 2-D feature pyramid (`FeatureNet`) keeps torch modules as parameter containers so that
the reference's checkpoint loads; on the device in eval mode the whole pyramid is one call into csrc/svs_conv2d.hip
(svs_featurenet_fpn: BatchNorm folded, the FPN's nearest up-sampling fused into the lateral convolutions).  Inference only, like the reference
(`@torch.no_grad()` forward).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from svs_hip import costvol

Align_Corners_Range = False


# ---------------------------------------------------------------------------------------------------------
# FeatureNet (parameter names of models/CasMVSNet.py:24-55,338-439, arch_mode 'fpn')
# ---------------------------------------------------------------------------------------------------------
class Conv2d(nn.Module):
    """conv + BatchNorm2d + ReLU with the reference's parameter names (`conv.weight`, `bn.*`).  On the device, in eval
    mode, the block is ONE launch of svs_conv2d with the BatchNorm folded into the weights."""

    def __init__(self, cin, cout, k, stride=1, relu=True, bn=True, **kw):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, bias=not bn, **kw)
        self.bn = nn.BatchNorm2d(cout) if bn else None
        self.relu, self.stride = relu, stride
        self._folded, self._key = None, None

    def folded(self):
        ts = [self.conv.weight] + ([self.conv.bias] if self.conv.bias is not None else [])
        if self.bn is not None:
            ts += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._key != key:
            w = self.conv.weight.detach().float()
            b = self.conv.bias.detach().float() if self.conv.bias is not None else None
            if self.bn is not None:
                scale = (self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)).detach().float()
                shift = (self.bn.bias - self.bn.running_mean * scale).detach().float()
                w = w * scale.view(-1, 1, 1, 1)
                b = shift if b is None else b * scale + shift
            self._folded, self._key = (w.contiguous(), b.contiguous() if b is not None else None), key
        return self._folded

    def forward(self, x):
        if x.is_cuda and not self.training:
            w, b = self.folded()
            return _conv_batch(x, w, b, stride=self.stride, relu=self.relu)
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x) if self.relu else x


def _conv_batch(x, w, b, add=None, add_upsample2=False, stride=1, relu=False):
    """(B,Cin,H,W) -> (B,Cout,Ho,Wo): one launch per image, written straight into the batch tensor."""
    k = w.shape[-1]
    Ho, Wo = (x.shape[2] + 2 * (k // 2) - k) // stride + 1, (x.shape[3] + 2 * (k // 2) - k) // stride + 1
    out = torch.empty(x.shape[0], w.shape[0], Ho, Wo, device=x.device)
    for i in range(x.shape[0]):
        costvol.conv2d(x[i], w, b, add=None if add is None else add[i], add_upsample2=add_upsample2, stride=stride, relu=relu,
                       out=out[i])
    return out


class FeatureNet(nn.Module):
    def __init__(self, base_channels, num_stage=3, stride=4, arch_mode="fpn"):
        super().__init__()
        if arch_mode != "fpn" or num_stage != 3:
            raise NotImplementedError("only the default FPN / 3-stage feature net is declared")
        b = base_channels
        self.arch_mode, self.stride, self.base_channels, self.num_stage = arch_mode, stride, b, num_stage
        self.conv0 = nn.Sequential(Conv2d(3, b, 3, 1, padding=1), Conv2d(b, b, 3, 1, padding=1))
        self.conv1 = nn.Sequential(Conv2d(b, 2 * b, 5, stride=2, padding=2), Conv2d(2 * b, 2 * b, 3, 1, padding=1),
                                   Conv2d(2 * b, 2 * b, 3, 1, padding=1))
        self.conv2 = nn.Sequential(Conv2d(2 * b, 4 * b, 5, stride=2, padding=2), Conv2d(4 * b, 4 * b, 3, 1, padding=1),
                                   Conv2d(4 * b, 4 * b, 3, 1, padding=1))
        self.out1 = nn.Conv2d(4 * b, 4 * b, 1, bias=False)
        self.inner1 = nn.Conv2d(2 * b, 4 * b, 1, bias=True)
        self.inner2 = nn.Conv2d(b, 4 * b, 1, bias=True)
        self.out2 = nn.Conv2d(4 * b, 2 * b, 3, padding=1, bias=False)
        self.out3 = nn.Conv2d(4 * b, b, 3, padding=1, bias=False)
        self.out_channels = [4 * b, 2 * b, b]
        self._fpn = None

    def _layers(self):
        """(weight, bias) of the 13 convolutions in svs_featurenet_fpn's order, BatchNorm folded."""
        blocks = list(self.conv0) + list(self.conv1) + list(self.conv2)
        plain = [self.out1, self.inner1, self.out2, self.inner2, self.out3]
        return [blk.folded() for blk in blocks] + [(c.weight.detach(), None if c.bias is None else c.bias.detach()) for c in plain]

    def forward(self, x):
        if x.is_cuda and not self.training:
            # the whole pyramid from one library call per image (csrc/svs_conv2d.hip: svs_featurenet_fpn); like the
            # reference's top-down additions (models/CasMVSNet.py:413-431) it needs H and W to be multiples of 4
            if self._fpn is None:
                self._fpn = costvol.FeatureNetFpn(self.base_channels)
            layers = self._layers()
            per_image = [self._fpn(xi, layers) for xi in x]
            return {f"stage{j + 1}": torch.stack([o[j] for o in per_image]) if len(per_image) > 1 else per_image[0][j][None]
                    for j in range(3)}
        c0 = self.conv0(x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        out = {"stage1": self.out1(c2)}
        f = F.interpolate(c2, scale_factor=2, mode="nearest") + self.inner1(c1)
        out["stage2"] = self.out2(f)
        f = F.interpolate(f, scale_factor=2, mode="nearest") + self.inner2(c0)
        out["stage3"] = self.out3(f)
        return out


# ---------------------------------------------------------------------------------------------------------
# 3-D regularisation network on the HIP conv kernels
# ---------------------------------------------------------------------------------------------------------
class _Block3d(nn.Module):
    """conv (or transposed conv) + BatchNorm3d + ReLU with the reference's parameter names (`conv.weight`, `bn.*`).
    In eval mode BN is an affine map: its scale is folded into the packed weights, its shift becomes the bias."""

    def __init__(self, cin, cout, stride=1, transposed=False):
        super().__init__()
        self.stride, self.transposed = stride, transposed
        if transposed:
            self.conv = nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False)
        else:
            self.conv = nn.Conv3d(cin, cout, 3, stride=stride, padding=1, bias=False)
        self.bn = nn.BatchNorm3d(cout)
        self._folded, self._key = None, None

    def folded(self):
        ts = [self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        key = tuple((t.data_ptr(), t._version) for t in ts)
        if self._key != key:
            scale = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
            shift = self.bn.bias - self.bn.running_mean * scale
            w = self.conv.weight.detach()
            if self.transposed:          # (Cin,Cout,3,3,3) -> [Cin][27][Cout]
                w = w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], 27, w.shape[1])
            else:                        # (Cout,Cin,3,3,3) -> [Cin][27][Cout]
                w = w.permute(1, 2, 3, 4, 0).reshape(w.shape[1], 27, w.shape[0])
            self._folded = ((w * scale.view(1, 1, -1)).contiguous().float(), shift.detach().contiguous().float())
            self._key = key
        return self._folded

    def forward(self, x, skip=None, split_out=False):
        if self.training:
            raise NotImplementedError("CasMVSNet is inference-only in S-VolSDF (runner.py:153); call .eval()")
        w, b = self.folded()
        return costvol.conv3d(x, w, b, skip=skip, stride=self.stride, transposed=self.transposed, relu=True,
                              split_out=split_out)


class Conv3d(_Block3d):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, relu=True, bn=True, padding=1, **kw):
        assert kernel_size == 3 and relu and bn and padding == 1
        super().__init__(in_channels, out_channels, stride=stride, transposed=False)


class Deconv3d(_Block3d):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, relu=True, bn=True, padding=1,
                 output_padding=1, **kw):
        assert kernel_size == 3 and stride == 2 and relu and bn and padding == 1 and output_padding == 1
        super().__init__(in_channels, out_channels, stride=2, transposed=True)


def x_is_device(t):
    return torch.is_tensor(t) and t.is_cuda


class CostRegNet(nn.Module):
    """models/CasMVSNet.py:441-472."""

    def __init__(self, in_channels, base_channels):
        super().__init__()
        b = base_channels
        self.conv0 = Conv3d(in_channels, b)
        self.conv1, self.conv2 = Conv3d(b, 2 * b, stride=2), Conv3d(2 * b, 2 * b)
        self.conv3, self.conv4 = Conv3d(2 * b, 4 * b, stride=2), Conv3d(4 * b, 4 * b)
        self.conv5, self.conv6 = Conv3d(4 * b, 8 * b, stride=2), Conv3d(8 * b, 8 * b)
        self.conv7, self.conv9, self.conv11 = Deconv3d(8 * b, 4 * b), Deconv3d(4 * b, 2 * b), Deconv3d(2 * b, b)
        self.prob = nn.Conv3d(b, 1, 3, stride=1, padding=1, bias=False)
        self._prob_w, self._prob_key = None, None

    def forward(self, x):
        """x (1,C,D,H,W), or the same volume as a costvol.SplitVolume (conv0 fused with its producer) -> (1,1,D,H,W)"""
        if x.shape[0] != 1:
            raise NotImplementedError("batch size 1 (runner.py:122)")
        for n in x.shape[2:]:
            if n % 8:
                raise ValueError("CostRegNet needs D, H, W divisible by 8 (three stride-2 levels)")
        if not isinstance(x, costvol.SplitVolume):
            x = x[0]
        c0 = self.conv0(x)
        # conv1 hands its output to conv2 as fp16 hi / mid pieces where both have the fused form (svs_conv3d_s2c8 -> _rows)
        fused12 = (x_is_device(c0) and costvol.rows_supported(self.conv1.conv.out_channels, self.conv2.conv.out_channels)
                   and self.conv1.conv.in_channels == 8 and c0.shape[-1] % 2 == 0)
        c2 = self.conv2(self.conv1(c0, split_out=fused12))
        c4 = self.conv4(self.conv3(c2))
        y = self.conv6(self.conv5(c4))
        y = self.conv7(y, skip=c4)
        y = self.conv9(y, skip=c2)
        y = self.conv11(y, skip=c0)
        key = (self.prob.weight.data_ptr(), self.prob.weight._version)
        if self._prob_key != key:
            w = self.prob.weight.detach()
            self._prob_w = w.permute(1, 2, 3, 4, 0).reshape(w.shape[1], 27, 1).contiguous().float()
            self._prob_key = key
        return costvol.conv3d(y, self._prob_w, None, stride=1, relu=False)[None]


def homo_warping(src_fea, src_proj, ref_proj, depth_values):
    """models/CasMVSNet.py:280-315: src_fea (1,C,H,W), projections (1,4,4) (already K@[R|t]), depth_values (1,D,H,W)
    -> (1,C,D,H,W)."""
    import numpy as np
    rel = np.asarray(src_proj[0].detach().cpu(), np.float64) @ np.linalg.inv(np.asarray(ref_proj[0].detach().cpu(), np.float64))
    return costvol.homo_warp(src_fea[0], list(rel[:3, :3].reshape(-1)) + list(rel[:3, 3]), depth_values[0])[None]


class DepthNet(nn.Module):
    """models/CasMVSNet.py:597-663."""

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, prob_volume_init=None,
                prevent_oom=False):
        assert len(features) == proj_matrices.shape[1], "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth
        # steps 1-2, fused; where conv0 has the fused form the volume goes to it as fp16 hi / mid pieces (SplitVolume)
        conv0 = getattr(cost_regularization, "conv0", None)
        split = (isinstance(conv0, Conv3d) and not conv0.training and features[0].is_cuda
                 and costvol.pair_supported(features[0].shape[1], conv0.conv.out_channels))
        variance = costvol.warp_variance(features, proj_matrices, depth_values, split=split)
        reg = cost_regularization(variance)[0, 0]                                    # step 3
        if prob_volume_init is not None:
            reg = reg + prob_volume_init[0]
        prob, depth, conf, _ = costvol.prob_depth_conf(reg, depth_values[0])
        return {"depth": depth[None], "photometric_confidence": conf[None], "prob_volume": prob[None],
                "depth_values": depth_values}


class CascadeMVSNet(nn.Module):
    def __init__(self, refine=False, ndepths=[48, 32, 8], depth_interals_ratio=[4, 2, 1], share_cr=False,
                 grad_method="detach", arch_mode="fpn", cr_base_chs=[8, 8, 8]):
        super().__init__()
        if refine:
            raise NotImplementedError("refine=False everywhere in S-VolSDF (runner.py:131)")
        self.refine, self.share_cr, self.ndepths = refine, share_cr, ndepths
        self.depth_interals_ratio, self.grad_method, self.arch_mode = depth_interals_ratio, grad_method, arch_mode
        self.cr_base_chs, self.num_stage = cr_base_chs, len(ndepths)
        assert len(ndepths) == len(depth_interals_ratio)
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.feature = FeatureNet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode=arch_mode)
        if share_cr:
            self.cost_regularization = CostRegNet(in_channels=self.feature.out_channels, base_channels=8)
        else:
            self.cost_regularization = nn.ModuleList(
                [CostRegNet(in_channels=self.feature.out_channels[i], base_channels=cr_base_chs[i])
                 for i in range(self.num_stage)])
        self.DepthNet = DepthNet()

    @torch.no_grad()
    def forward(self, stage_idx, sample_cuda, features, extra, outputs, int_r, depth=None, prevent_oom=False,
                inverse_depth=False):
        imgs, proj_matrices, depth_values = sample_cuda["imgs"], sample_cuda["proj_matrices"], sample_cuda["depth_values"]
        if depth is None:
            depth = outputs['depth'] if stage_idx > 0 else None
        outputs = {} if outputs is None else outputs
        dv = costvol.host_copy(depth_values)[0]
        depth_min, depth_max = float(dv[0]), float(dv[-1])
        depth_interval = (depth_max - depth_min) / depth_values.size(1)
        H_img, W_img = imgs.shape[-2], imgs.shape[-1]
        key = "stage{}".format(stage_idx + 1)
        features_stage = [feat[key] for feat in features]
        scale = int(self.stage_infos[key]["scale"])
        nd = self.ndepths[stage_idx]
        dev = features_stage[0].device
        if depth is not None:
            if inverse_depth:
                pass    # stages 2,3 of the inverse variant use the same window (models/CasMVSNet.py:548-554)
            hyp = costvol.depth_hypotheses(depth[0], (H_img, W_img), nd, scale, depth_min, depth_max,
                                           int_r * depth_interval, False, dev)
        else:
            hyp = costvol.depth_hypotheses(None, (H_img, W_img), nd, scale, float(dv[0]), float(dv[-1]), 0.0,
                                           inverse_depth, dev)
        cr = self.cost_regularization if self.share_cr else self.cost_regularization[stage_idx]
        outputs_stage = self.DepthNet(features_stage, proj_matrices[key], depth_values=hyp[None], num_depth=nd,
                                      cost_regularization=cr, prevent_oom=prevent_oom)
        outputs[key] = outputs_stage
        outputs.update(outputs_stage)
        return outputs, None

"""Decoder building blocks of MGNet -- host-side mirror of mgnet/modeling/layers.py (same class names, ctor
arguments, attribute names => same state-dict keys), computing through mgnet_amd.modeling.ops."""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from ..registry import BACKBONE_REGISTRY, ShapeSpec
from . import ops

__all__ = ["Conv2d", "InPlaceABNSync", "MGNetDecoder", "MGNetHead", "PoseCNN", "FastGlobalAvgPool2d",
           "GlobalContextModule", "AttentionRefinementModule", "FeatureFusionModule", "mgnet_xavier_fill"]


class InPlaceABNSync(nn.Module):
    """Activated batch norm with cross-rank statistics (replaces inplace_abn.InPlaceABNSync, see ops.iabn).
    Parameters/buffers are named like nn.BatchNorm2d: weight, bias, running_mean, running_var."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, activation="leaky_relu",
                 activation_param=0.01, group=None):
        super().__init__()
        assert affine
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.activation, self.activation_param, self.group = activation, activation_param, group
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        return ops.iabn(x, self.weight, self.bias, self.running_mean, self.running_var, self.training, self.momentum,
                        self.eps, self.activation, self.activation_param, self.group)

    def train(self, mode=True):
        self.__dict__.pop("_mgn_eval_fold", None)   # (ops.conv_abn_eval's folded weights: the running statistics are about to move)
        return super().train(mode)

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, activation={self.activation}"


def _world():
    return dist.group.WORLD if (dist.is_available() and dist.is_initialized()) else None


def _abn(ch, activation="leaky_relu"):
    return InPlaceABNSync(ch, momentum=0.01, activation=activation, group=_world())


class Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d equivalent: conv -> optional `norm` -> optional `activation` (submodule names kept)."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x, with_skip=False, full=None):
        skip = None
        sf = self.norm if isinstance(self.norm, InPlaceABNSync) else None
        if sf is not None and not sf.training and full is None and not torch.is_grad_enabled():
            y = ops.conv_abn_eval(x, self)      # inference: the norm folded into the convolution (one launch)
            if y is not None:
                if self.activation is not None:
                    y = self.activation(y)
                return (y, x[:, :, ::2, ::2] if with_skip == 2 else x) if with_skip else y
        if full is not None:   # a block's 1x1 / stride-2 shortcut conv fed with conv1's `xsub` (ops._ShortcutS2Fn)
            x = ops.conv2d_shortcut_s2(x, full, self.weight, stats_for=sf)
        elif with_skip:   # also hand back the input for a second consumer (its gradient is fused into the conv's backward)
            x, skip = ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, with_skip=with_skip, stats_for=sf)
        else:
            x = ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, stats_for=sf)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return (x, skip) if with_skip else x


class PlainConv2d(nn.Conv2d):
    """nn.Conv2d whose forward goes through ops.conv2d (predictors, PoseCNN decoder convs)."""

    def forward(self, x, relu=False, keep_pad=False):
        """relu: `torch.relu_(conv(x))` with the ReLU in the convolution's epilogue (and its mask in the backward) on the GPU path.
        keep_pad: a few-channel predictor may return its 32-channel-padded output as an ops.PaddedMap (training path: the fused
        losses / ops.head_activation read it in place and hand back a padded gradient)"""
        if relu and os.environ.get("MGN_NO_POSERELU"):
            return torch.relu_(ops.conv2d(x, self.weight, self.bias, self.stride, self.padding))
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, relu=relu, keep_pad=keep_pad)


def mgnet_xavier_fill(module):  # layers.py:325-328
    nn.init.kaiming_normal_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class FastGlobalAvgPool2d(nn.Module):  # layers.py:170-184
    def __init__(self, flatten=False):
        super().__init__()
        self.flatten = flatten

    def forward(self, x):
        y = ops.global_avg_pool(x)
        return y.flatten(1) if self.flatten else y


class GlobalContextModule(nn.Module):  # layers.py:187-218
    def __init__(self, in_channels, out_channels, init_method="default"):
        super().__init__()
        self.global_context = nn.Sequential(
            FastGlobalAvgPool2d(),
            Conv2d(in_channels, out_channels, kernel_size=1, padding=0, bias=False, norm=_abn(out_channels)))
        if init_method == "xavier":
            mgnet_xavier_fill(self.global_context[1])

    def forward(self, x):
        return ops.upsample_nearest(self.global_context(x), x.shape[2:])


class AttentionRefinementModule(nn.Module):  # layers.py:221-267
    def __init__(self, in_channels, out_channels, init_method="default"):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False, norm=_abn(out_channels))
        self.channel_attention = nn.Sequential(
            FastGlobalAvgPool2d(),
            Conv2d(out_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=False,
                   norm=_abn(out_channels, "identity")),
            nn.Sigmoid())
        if init_method == "xavier":
            mgnet_xavier_fill(self.conv)
            mgnet_xavier_fill(self.channel_attention[1])

    def forward(self, x, addend=None):
        """addend: a tensor added to the result (`arm(x) + last`, layers.py:87) inside the scaling pass on the GPU path"""
        # conv -> norm -> attention; on the GPU path the norm's and the attention's passes over the map are fused (ops._AbnAttentionFn)
        return ops.conv_abn_attention(self.conv, x, self.channel_attention, "arm", addend=addend)


class FeatureFusionModule(nn.Module):  # layers.py:270-322
    def __init__(self, in_channels, out_channels, init_method="default"):
        super().__init__()
        self.conv = Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=False, norm=_abn(out_channels))
        self.channel_attention = nn.Sequential(
            FastGlobalAvgPool2d(),
            Conv2d(out_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=False, activation=nn.ReLU(inplace=True)),
            PlainConv2d(out_channels, out_channels, kernel_size=(1, 1), bias=False),
            nn.Sigmoid())
        if init_method == "xavier":
            mgnet_xavier_fill(self.conv)
            mgnet_xavier_fill(self.channel_attention[1])
            mgnet_xavier_fill(self.channel_attention[2])

    def forward(self, fsp, fcp):
        return ops.conv_abn_attention(self.conv, (fsp, fcp), self.channel_attention, "ffm", residual=True)


class MGNetDecoder(nn.Module):  # layers.py:22-94
    def __init__(self, input_shape, common_stride, arm_channels, refine_channels, ffm_channels, init_method="default"):
        super().__init__()
        order = sorted(input_shape.items(), key=lambda kv: kv[1].stride, reverse=True)
        self.in_features = [k for k, _ in order]
        chans = [v.channels for _, v in order]
        self.common_stride = common_stride
        assert len(arm_channels) == 2, "arm_channels have to be a list of ints with length 2!"
        assert len(refine_channels) == 2, "refine_channels have to be a list of ints with length 2!"
        self.arms = nn.ModuleList([AttentionRefinementModule(chans[k], arm_channels[k], init_method=init_method) for k in range(2)])
        self.refines = nn.ModuleList([
            Conv2d(arm_channels[k], refine_channels[k], kernel_size=3, padding=1, bias=False, norm=_abn(refine_channels[k]))
            for k in range(2)])
        self.ffm = FeatureFusionModule(chans[2] + refine_channels[1], ffm_channels, init_method=init_method)
        if init_method == "xavier":
            for r in self.refines:
                mgnet_xavier_fill(r)

    def forward(self, features):
        fms = [features[k] for k in self.in_features]
        msc, last = [], features["global_context"]
        for k in range(2):
            fm = self.arms[k](fms[k], last)   # arm(x) + last
            msc.append(fm)
            last = self.refines[k](ops.upsample_nearest(fm, fms[k + 1].shape[2:]))
        return self.ffm(fms[2], last), msc


class MGNetHead(nn.Module):  # layers.py:97-127
    def __init__(self, in_channels, head_channels, num_classes, init_method="default"):
        super().__init__()
        self.head = Conv2d(in_channels, head_channels, kernel_size=3, padding=1, bias=False, norm=_abn(head_channels))
        self.predictor = PlainConv2d(head_channels, num_classes, kernel_size=(1, 1), bias=False)
        if init_method == "xavier":
            mgnet_xavier_fill(self.head)
            mgnet_xavier_fill(self.predictor)

    def forward(self, x, keep_pad=False, with_skip=False):
        """with_skip: also return the input as an alias for a second consumer (the instance head feeds two MGNetHeads from one decoder
        output): its gradient is then added inside this head's data-gradient kernel instead of by an accumulation pass"""
        if with_skip:
            h, skip = self.head(x, with_skip=True)
            return self.predictor(h, keep_pad=keep_pad), skip
        return self.predictor(self.head(x), keep_pad=keep_pad)


class PoseCNN(nn.Module):  # layers.py:130-167
    def __init__(self, cfg, num_context_images=2):
        super().__init__()
        self.num_context_images = num_context_images
        self.pose_encoder = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, ShapeSpec(channels=(num_context_images + 1) * 3))
        self.conv1 = PlainConv2d(512, 256, kernel_size=(1, 1))
        self.conv2 = PlainConv2d(256, 256, kernel_size=(3, 3), padding=1)
        self.conv3 = PlainConv2d(256, 256, kernel_size=(3, 3), padding=1)
        self.conv4 = PlainConv2d(256, 6 * num_context_images, kernel_size=(1, 1))
        for m in (self.conv1, self.conv2, self.conv3, self.conv4):
            mgnet_xavier_fill(m)

    def forward(self, image_list):
        return self.head(self.pose_encoder(image_list)["res5"])

    def head(self, out):
        """what follows the encoder (MGNet.forward calls it by itself when it issues the encoder block by block beside the backbone)"""
        out = self.conv1(out, relu=True)   # layers.py:158-163: relu_(conv(x))
        out = self.conv2(out, relu=True)
        out = self.conv3(out, relu=True)
        out = ops.mean_hw(self.conv4(out, keep_pad=self.training), 0.01)   # 0.01 * out.mean(3).mean(2), layers.py:165-167
        return out.view(out.size(0), self.num_context_images, 6)

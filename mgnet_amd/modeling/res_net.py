"""ResNet-18/34 with activated batch norm -- mirror of mgnet/modeling/res_net.py (+ the parts of detectron2's
`ResNet` container it relies on: `stem`, `res2..res5` stage names, `output_shape()`, `out_features`)."""
import torch
import torch.nn as nn

from ..registry import BACKBONE_REGISTRY, ShapeSpec
from . import ops
from .layers import Conv2d, _abn

__all__ = ["BasicBlock", "BasicStem", "ResNet", "build_resnet_iabn_backbone"]


def c2_msra_fill(module):
    """fvcore.nn.weight_init.c2_msra_fill (recalled): kaiming_normal(fan_out, relu), zero bias."""
    nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


class BasicBlock(nn.Module):  # res_net.py:11-79
    def __init__(self, in_channels, out_channels, *, stride=1):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.shortcut = None
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=_abn(out_channels, "identity"))
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False, norm=_abn(out_channels))
        self.conv2 = Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=False,
                            norm=_abn(out_channels, "identity"))
        for layer in (self.conv1, self.conv2, self.shortcut):
            if layer is not None:
                c2_msra_fill(layer)

    def forward(self, x):
        if not self.training and not torch.is_grad_enabled():
            # inference: three launches per block -- every norm folded into its convolution, the shortcut add and the ReLU in conv2's
            # epilogue (ops.conv_abn_eval; each call falls back to conv + norm on its own where no kernel has the epilogue)
            out = self.conv1(x)
            sc = x if self.shortcut is None else self.shortcut(x)
            c2 = self.conv2
            y = ops.conv_abn_eval(out, c2, residual=sc.contiguous(memory_format=torch.channels_last) if sc.is_cuda else sc, relu=True)
            return y if y is not None else ops.abn_add_relu(ops.conv2d(out, c2.weight, c2.bias, c2.stride, c2.padding), c2.norm, sc)
        if self.shortcut is not None and self.stride == 2 and ops.sub2_supported(x, self.conv1.weight, self.shortcut.weight):
            # down-sampling block: the shortcut conv's data gradient travels at the LOW resolution into conv1's data-gradient kernel
            out, xsub = self.conv1(x, with_skip=2)
            # (a conv2d that did not take the fused path hands back x itself: the ordinary shortcut then)
            sc = self.shortcut(xsub, full=x) if xsub.shape[2] == (x.shape[2] + 1) // 2 and xsub.shape[2] != x.shape[2] else self.shortcut(xsub)
        else:
            out, skip = self.conv1(x, with_skip=True)   # res_net.py:62-79; `skip` is x: the shortcut's gradient joins conv1's dgrad
            sc = skip if self.shortcut is None else self.shortcut(skip)
        c2 = self.conv2   # conv -> InPlaceABNSync(identity) -> + shortcut -> ReLU; norm, add and ReLU run as one fused op on the GPU
        return ops.abn_add_relu(ops.conv2d(out, c2.weight, c2.bias, c2.stride, c2.padding, stats_for=c2.norm), c2.norm, sc)


class BasicStem(nn.Module):  # res_net.py:82-110
    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, 4
        self.conv1 = Conv2d(in_channels, out_channels, kernel_size=7, stride=2, padding=3, bias=False, norm=_abn(out_channels))
        c2_msra_fill(self.conv1)

    def forward(self, x):
        c = self.conv1   # conv -> InPlaceABNSync(leaky) -> max pool; norm + pooling run as one fused op on the GPU
        return ops.abn_max_pool(ops.conv2d(x, c.weight, c.bias, c.stride, c.padding, stats_for=c.norm), c.norm)


class ResNet(nn.Module):
    """Container with detectron2's naming: `stem`, `res2`..`res5` (each an nn.Sequential of blocks)."""

    def __init__(self, stem, stages, out_features, freeze_at=0):
        super().__init__()
        self.stem = stem
        self._out_feature_strides = {"stem": stem.stride}
        self._out_feature_channels = {"stem": stem.out_channels}
        self.stage_names = []
        stride = stem.stride
        for i, blocks in enumerate(stages):
            name = f"res{i + 2}"
            self.add_module(name, nn.Sequential(*blocks))
            self.stage_names.append(name)
            stride *= int(torch.tensor([b.stride for b in blocks]).prod())
            self._out_feature_strides[name] = stride
            self._out_feature_channels[name] = blocks[-1].out_channels
        self._out_features = list(out_features)
        self.freeze(freeze_at)

    def freeze(self, freeze_at=0):
        """detectron2 `ResNet.freeze` (res_net.py:165 hands MODEL.BACKBONE.FREEZE_AT to it): 1 freezes the stem, k >= 2 also res2..res<k>.
        `CNNBlockBase.freeze` sets requires_grad = False on the block's parameters and converts BatchNorm modules to FrozenBatchNorm2d --
        InPlaceABNSync is not a BatchNorm subclass, so the norms of a frozen block keep normalising with the batch statistics and keep
        updating their running statistics in training mode; only their affine parameters stop training."""
        units = [(1, [self.stem])] + [(idx, list(getattr(self, name).children())) for idx, name in enumerate(self.stage_names, start=2)]
        for idx, blocks in units:
            if freeze_at >= idx:
                for blk in blocks:
                    for p in blk.parameters():
                        p.requires_grad = False
        return self

    @property
    def size_divisibility(self):
        return 0

    def output_shape(self):
        return {n: ShapeSpec(channels=self._out_feature_channels[n], stride=self._out_feature_strides[n]) for n in self._out_features}

    def forward(self, x):
        outputs = {}
        x = self.stem(x)
        if "stem" in self._out_features:
            outputs["stem"] = x
        for name in self.stage_names:
            x = getattr(self, name)(x)
            if name in self._out_features:
                outputs[name] = x
        return outputs


@BACKBONE_REGISTRY.register()
def build_resnet_iabn_backbone(cfg, input_shape):  # res_net.py:113-165
    r = cfg.MODEL.RESNETS
    assert r.RES2_OUT_CHANNELS == 64, "Must set MODEL.RESNETS.RES2_OUT_CHANNELS = 64 for R18/R34"
    assert not any(r.DEFORM_ON_PER_STAGE), "MODEL.RESNETS.DEFORM_ON_PER_STAGE unsupported for R18/R34"
    assert r.RES5_DILATION == 1, "Must set MODEL.RESNETS.RES5_DILATION = 1 for R18/R34"
    assert r.NUM_GROUPS == 1, "Must set MODEL.RESNETS.NUM_GROUPS = 1 for R18/R34"
    depth_blocks = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}[r.DEPTH]
    stem = BasicStem(in_channels=input_shape.channels, out_channels=r.STEM_OUT_CHANNELS)
    cin, cout, stages = r.STEM_OUT_CHANNELS, r.RES2_OUT_CHANNELS, []
    for idx, n in enumerate(depth_blocks):
        blocks = []
        for k in range(n):
            blocks.append(BasicBlock(cin, cout, stride=(2 if (k == 0 and idx > 0) else 1)))
            cin = cout
        stages.append(blocks)
        cout *= 2
    return ResNet(stem, stages, out_features=r.OUT_FEATURES, freeze_at=cfg.MODEL.BACKBONE.FREEZE_AT)

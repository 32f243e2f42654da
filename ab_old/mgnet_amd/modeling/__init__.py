from .layers import (AttentionRefinementModule, FastGlobalAvgPool2d, FeatureFusionModule, GlobalContextModule,
                     InPlaceABNSync, MGNetDecoder, MGNetHead, PoseCNN)
from .loss import DeepLabCE, MultiViewPhotometricLoss, OhemCE
from .mg_net import (DEPTH_HEADS_REGISTRY, INS_EMBED_HEADS_REGISTRY, MGNet, MGNetInsEmbedHead,
                     MGNetSelfSupervisedDepthHead, MGNetSemSegHead, build_depth_head, build_ins_embed_head)
from .res_net import build_resnet_iabn_backbone

__all__ = [k for k in globals().keys() if not k.startswith("_")]

"""CPU: boundary (configs, registries, state-dict keys, optimizer groups, schedule) and the product network's host
logic against oracle/network_oracle.py on identical weights.  (On CPU the product runs its torch-staging ops; ops that
already have a HIP kernel are exercised by the -m gpu tests.)"""
import glob
import math
import os

import numpy as np
import pytest
import torch

REF_CFG = "/root/reference/configs"


def make_cfg(path=None, **over):
    from mgnet_amd import add_mgnet_config, get_cfg

    cfg = get_cfg()
    add_mgnet_config(cfg)
    if path:
        cfg.merge_from_file(path)
    opts = []
    for k, v in over.items():
        opts += [k, v]
    cfg.merge_from_list(opts)
    return cfg


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference checkout not present (GPU box)")
def test_reference_yamls_load_unchanged():
    files = sorted(glob.glob(os.path.join(REF_CFG, "MGNet-*.yaml")))
    assert len(files) == 5
    seen = {}
    for f in files:
        cfg = make_cfg(f)
        cfg.freeze()
        assert cfg.MODEL.META_ARCHITECTURE == "MGNet" and cfg.MODEL.BACKBONE.NAME == "build_resnet_iabn_backbone"
        assert cfg.MODEL.SEM_SEG_HEAD.NAME == "MGNetSemSegHead" and cfg.MODEL.INS_EMBED_HEAD.NAME == "MGNetInsEmbedHead"
        assert cfg.MODEL.DEPTH_HEAD.NAME == "MGNetSelfSupervisedDepthHead"
        seen[os.path.basename(f)] = cfg
        with pytest.raises(AttributeError):
            cfg.WITH_DEPTH = False  # frozen
    assert seen["MGNet-KITTI-Eigen-Zhou.yaml"].MODEL.SEM_SEG_HEAD.NUM_CLASSES == 19          # own override
    assert seen["MGNet-KITTI-Eigen-Zhou.yaml"].MODEL.SEM_SEG_HEAD.OHEM_N_MIN == 262143       # inherited via _BASE_
    assert seen["MGNet-KITTI-Eigen-PseudoLabelGeneration.yaml"].SOLVER.IMS_PER_BATCH == 24   # 2-level _BASE_ chain
    assert seen["MGNet-KITTI-Eigen-PseudoLabelGeneration.yaml"].WITH_DEPTH is False
    assert seen["MGNet-Cityscapes-Fine.yaml"].DATASETS.TRAIN == ("cityscapes_fine_scene_seg_train",)
    assert seen["MGNet-Cityscapes-Fine.yaml"].INPUT.IGNORED_CATEGORIES_IN_DEPTH == ["ego vehicle", "sky"]


def test_cfg_semantics():
    cfg = make_cfg()
    with pytest.raises(KeyError):
        cfg.merge_from_list(["MODEL.NOT_A_KEY", 1])
    with pytest.raises(ValueError):
        cfg.merge_from_list(["WITH_DEPTH", "yes"])
    cfg.merge_from_list(["SOLVER.BASE_LR", "0.01", "MODEL.SEM_SEG_HEAD.ARM_CHANNELS", "[64, 64]"])
    assert cfg.SOLVER.BASE_LR == 0.01 and cfg.MODEL.SEM_SEG_HEAD.ARM_CHANNELS == [64, 64]
    c2 = cfg.clone()
    c2.SOLVER.BASE_LR = 1.0
    assert cfg.SOLVER.BASE_LR == 0.01
    assert "DEPTH_HEAD" in cfg.dump()


def small_model(with_depth=True, with_panoptic=True, seed=0, **over):
    from mgnet_amd.registry import build_model

    torch.manual_seed(seed)
    cfg = make_cfg(os.path.join(os.path.dirname(os.path.dirname(__file__)), "configs", "bench-c4-cityscapes-videosequence.yaml"),
                   **{"MODEL.DEVICE": "cpu", "SOLVER.AMP.ENABLED": False, "WITH_DEPTH": with_depth,
                      "WITH_PANOPTIC": with_panoptic, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN": 1500, **over})
    return cfg, build_model(cfg)


def test_freeze_at_follows_detectron2():
    """MODEL.BACKBONE.FREEZE_AT (res_net.py:126,165 -> detectron2 ResNet.freeze): 1 = stem, k = stem + res2..res<k>; the backbone AND the
    pose encoder (layers.py:141-144 builds it from the same registry entry and config); InPlaceABNSync is no BatchNorm subclass, so its
    buffers stay live and only its affine parameters freeze.  The optimizer's groups still list every parameter (solver/build.py:85-116
    does not filter); the gradient buckets and the fused optimizer hold the trainable ones."""
    from mgnet_amd.engine.reducer import GradReducer
    from mgnet_amd.solver import get_mgnet_optimizer_params

    _, m0 = small_model()
    _, m = small_model(**{"MODEL.BACKBONE.FREEZE_AT": 3})
    for trunk in (m.backbone, m.pose_net.pose_encoder):
        for name, p in trunk.named_parameters():
            frozen = name.startswith(("stem.", "res2.", "res3."))
            assert p.requires_grad == (not frozen), name
    assert all(p.requires_grad for n, p in m.named_parameters() if "backbone" not in n and "pose_encoder" not in n)
    assert m.backbone.res2[0].conv1.norm.training and m.backbone.res2[0].conv1.norm.running_mean.requires_grad is False
    groups = get_mgnet_optimizer_params(m, 1e-4)
    assert len(groups) == len(get_mgnet_optimizer_params(m0, 1e-4))
    red = GradReducer([p for g in groups for p in (g["params"] if isinstance(g["params"], list) else [g["params"]])])
    held = {id(p) for b in red.buckets for p in b["params"]}
    assert held == {id(p) for p in m.parameters() if p.requires_grad}
    _, m1 = small_model(**{"MODEL.BACKBONE.FREEZE_AT": 1})
    assert [n for n, p in m1.backbone.named_parameters() if not p.requires_grad] == ["stem.conv1.weight", "stem.conv1.norm.weight", "stem.conv1.norm.bias"]


def test_state_dict_keys_and_param_counts():
    """SURVEY 8(b) state-dict layout + Appendix A parameter counts."""
    cfg, m = small_model()
    keys = set(m.state_dict().keys())
    for k in ["backbone.stem.conv1.weight", "backbone.stem.conv1.norm.running_var", "backbone.res3.0.shortcut.norm.weight",
              "backbone.res5.1.conv2.norm.bias", "global_context.global_context.1.weight",
              "sem_seg_head.arms.0.conv.weight", "sem_seg_head.arms.1.channel_attention.1.norm.weight",
              "sem_seg_head.refines.0.norm.running_mean", "sem_seg_head.ffm.conv.weight",
              "sem_seg_head.ffm.channel_attention.1.weight", "sem_seg_head.ffm.channel_attention.2.weight",
              "sem_seg_head.head.head.weight", "sem_seg_head.head.predictor.weight",
              "ins_embed_head.center_head.head.norm.weight", "ins_embed_head.offset_head.predictor.weight",
              "depth_head.heads.0.head.weight", "depth_head.heads.2.predictor.weight",
              "pose_net.pose_encoder.stem.conv1.weight", "pose_net.conv1.bias", "pose_net.conv4.weight", "log_vars"]:
        assert k in keys, k
    assert m.state_dict()["pose_net.pose_encoder.stem.conv1.weight"].shape == (64, 9, 7, 7)
    n = sum(p.numel() for p in m.parameters())
    assert abs(n - 30.95e6) < 0.02e6, n
    _, mp = small_model(with_depth=False)
    assert abs(sum(p.numel() for p in mp.parameters()) - 15.83e6) < 0.02e6
    n_abn = sum(1 for mod in m.modules() if type(mod).__name__ == "InPlaceABNSync")
    assert n_abn == 68


def test_optimizer_groups_and_schedule():
    from mgnet_amd.solver import build_lr_scheduler, build_optimizer

    cfg, m = small_model()
    opt = build_optimizer(cfg, m)
    lrs = {id(p): g["lr"] for g in opt.param_groups for p in g["params"]}
    assert lrs[id(m.backbone.stem.conv1.weight)] == pytest.approx(1e-4)
    assert lrs[id(m.pose_net.conv1.weight)] == pytest.approx(1e-4)            # "head" not in "pose_net"
    assert lrs[id(m.global_context.global_context[1].weight)] == pytest.approx(1e-4)
    assert lrs[id(m.sem_seg_head.head.predictor.weight)] == pytest.approx(1e-3)  # HEAD_LR_FACTOR 10
    assert lrs[id(m.depth_head.heads[1].head.norm.bias)] == pytest.approx(1e-3)
    assert lrs[id(m.log_vars)] == pytest.approx(1e-4)
    assert len(lrs) == len(list(m.parameters()))
    sched = build_lr_scheduler(cfg, opt)
    # WarmupPolyLR: linear warm-up from 0.1 over 1000 iters, then (1 - it/60000)^0.9
    assert sched.get_last_lr()[0] == pytest.approx(1e-4 * 0.1)
    for _ in range(500):
        opt.step(); sched.step()
    assert sched.get_last_lr()[0] == pytest.approx(1e-4 * (0.1 * 0.5 + 0.5) * math.pow(1 - 500 / 60000, 0.9), rel=1e-6)


@pytest.mark.parametrize("with_depth", [False, True])
def test_product_network_matches_oracle_on_cpu(with_depth):
    """Same weights, same batch: loss dict and parameter gradients of mgnet_amd.MGNet vs oracle/network_oracle.py.
    fp32, tolerance rel 1e-4 on losses (SURVEY 8d 'network blocks fp32: rel 1e-4')."""
    from mgnet_amd.data import synthetic_batch
    from oracle import network_oracle as NO

    cfg, m = small_model(with_depth=with_depth, seed=3)
    m.train()
    with torch.no_grad():  # make the norm layers non-trivial
        for mod in m.modules():
            if type(mod).__name__ == "InPlaceABNSync":
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
        m.log_vars.uniform_(-0.3, 0.3)
    batch = synthetic_batch(2, 64, 96, "cpu", seed=5, with_depth=with_depth)
    if with_depth:
        # no GPU here: route the depth loss of the PRODUCT model through the pinned C oracle for this host-logic test only
        class _OracleLoss(torch.nn.Module):
            def forward(self, pred, tgt):
                r = NO._ReprojOracle.apply(tgt["image_orig"], tgt["image_prev_orig"], tgt["image_next_orig"],
                                           tgt["reprojection_mask"], tgt["camera_matrix"], pred["poses"], *pred["depth"])
                return {"loss_photometric": r[0], "loss_smoothness": r[1]}

        m.depth_head.loss = _OracleLoss()
    got = m(batch)
    sum(got.values()).backward()
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    ref = NO.mgnet_losses(sd, batch, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, with_depth=with_depth,
                          ohem_n_min=1500)
    sum(ref.values()).backward()
    assert list(got.keys()) == list(ref.keys())
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=1e-4, abs=1e-6), k
    named = dict(m.named_parameters())
    worst, errs = 0.0, []
    for k, p in named.items():
        g_ref = sd[k].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        denom = float(g_ref.abs().max()) + 1e-12
        err = float((p.grad - g_ref).abs().max()) / denom
        worst = max(worst, err)
        errs.append(err)
        # fp32, ~70 layers deep, piecewise ops (relu/leaky/maxpool/OHEM/L1 sign/arg-min): isolated flips cost ~1e-2 on a
        # max-normalised scale (up to 6e-2 on single tensors, depending on the batch); a wrong backward (e.g. the channels_last
        # CPU issue found with this test) costs > 2e-2 EVERYWHERE, which the median below catches
        assert err < 1e-1, (k, err)
    assert float(np.median(errs)) < 2e-3, float(np.median(errs))

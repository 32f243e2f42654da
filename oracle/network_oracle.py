"""oracle/network_oracle.py -- plain-torch fp32 CPU restatement of MGNet's network forward + losses (group N).

TEST INFRASTRUCTURE ONLY (never imported by mgnet_amd/).

Parity status: **PINNED against outputs of the reference's own code, with stand-ins for three absent third-party packages**.
  * modules -- tests/golden/net_*.npz (tests/test_network_golden.py): BasicBlock (stride 1/2), BasicStem,
    GlobalContextModule, AttentionRefinementModule, FeatureFusionModule, MGNetDecoder, MGNetHead;
  * the full training step -- tests/golden/model_step.npz (tests/test_model_golden.py): MGNet.forward in training mode
    (input normalisation, PoseCNN, both ResNet-18, the three decoders/heads, target assembly, OhemCE, centre/offset losses,
    MultiViewPhotometricLoss, uncertainty weighting) -> loss dictionary, gradients of log_vars / pose_net.conv4.bias and the
    gradient norm of every top-level submodule.
  The fixtures are outputs of mgnet/modeling/{mg_net,layers,res_net,loss}.py + mgnet/geometry imported UNMODIFIED in the
  build container (tests/golden/make_golden_network.py, make_golden_model.py).  detectron2, inplace_abn and fvcore are
  absent from the image and from /root/reference, so those harnesses supply stand-ins for exactly the names the files
  use; the fixtures therefore pin the reference's wiring and arithmetic (state-dict keys, hyper-parameters and their
  config keys, order of operations, interpolation modes, concat order, loss weights, task order of the uncertainty
  weighting), while the internals of the third-party pieces stay restated from their published behaviour:
      - detectron2.layers.Conv2d             = conv -> norm -> activation
      - inplace_abn.InPlaceABNSync (>=1.1.0)  = batch_norm with gamma := |weight| + eps, leaky_relu(0.01) or identity
      - detectron2 ResNet container / ImageList.from_tensors = stem -> res2..res5 / zero-pad to a multiple of 32 and stack
  * The loss functions (OhemCE/DeepLabCE, MultiViewPhotometricLoss) are additionally pinned on their own by
    tests/golden/{ce_losses,reproj_*}.npz.
"""
import numpy as np
import torch
import torch.nn.functional as F

EPS, SLOPE = 1e-5, 0.01


def abn(sd, p, x, act="leaky_relu", training=True):
    """InPlaceABNSync(momentum=0.01) forward in training mode (batch statistics, single process)."""
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    y = F.batch_norm(x, None if training else sd[p + ".running_mean"], None if training else sd[p + ".running_var"],
                     w.abs() + EPS, b, training, 0.0, EPS)
    return F.leaky_relu(y, SLOPE) if act == "leaky_relu" else y


def conv_abn(sd, p, x, stride=1, padding=0, act="leaky_relu"):
    """detectron2 Conv2d with norm=InPlaceABNSync (e.g. res_net.py:42-50)."""
    return abn(sd, p + ".norm", F.conv2d(x, sd[p + ".weight"], None, stride, padding), act)


def basic_block(sd, p, x, stride):  # res_net.py:68-79
    out = conv_abn(sd, p + ".conv1", x, stride, 1)
    out = conv_abn(sd, p + ".conv2", out, 1, 1, act="identity")
    sc = conv_abn(sd, p + ".shortcut", x, stride, 0, act="identity") if (p + ".shortcut.weight") in sd else x
    return F.relu(out + sc)


def resnet18(sd, p, x):  # res_net.py:107-110, 113-165
    x = conv_abn(sd, p + ".stem.conv1", x, 2, 3)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = {}
    for s in range(2, 6):   # blocks per stage as the state dict has them: [2, 2, 2, 2] for depth 18, [3, 4, 6, 3] for 34 (res_net.py:137-146)
        k = 0
        while f"{p}.res{s}.{k}.conv1.weight" in sd:
            x = basic_block(sd, f"{p}.res{s}.{k}", x, 2 if (k == 0 and s > 2) else 1)
            k += 1
        assert k >= 2
        feats[f"res{s}"] = x
    return feats


def gap(x):  # layers.py:184
    return x.view(x.size(0), x.size(1), -1).mean(-1).view(x.size(0), x.size(1), 1, 1)


def gcm(sd, x):  # layers.py:215-218
    y = conv_abn(sd, "global_context.global_context.1", gap(x))
    return F.interpolate(y, x.shape[2:], mode="nearest")


def arm(sd, p, x):  # layers.py:262-267
    fm = conv_abn(sd, p + ".conv", x, 1, 1)
    att = torch.sigmoid(conv_abn(sd, p + ".channel_attention.1", gap(fm), act="identity"))
    return fm * att


def ffm(sd, p, fsp, fcp):  # layers.py:315-322
    fm = conv_abn(sd, p + ".conv", torch.cat([fsp, fcp], 1))
    a = F.relu(F.conv2d(gap(fm), sd[p + ".channel_attention.1.weight"]))
    a = torch.sigmoid(F.conv2d(a, sd[p + ".channel_attention.2.weight"]))
    return fm + fm * a


def decoder(sd, p, feats):  # layers.py:82-94
    fl = [feats["res5"], feats["res4"], feats["res3"]]
    msc, last = [], feats["global_context"]
    for i in range(2):
        fm = arm(sd, f"{p}.arms.{i}", fl[i]) + last
        msc.append(fm)
        last = F.interpolate(fm, size=fl[i + 1].shape[2:], mode="nearest")
        last = conv_abn(sd, f"{p}.refines.{i}", last, 1, 1)
    return ffm(sd, p + ".ffm", fl[2], last), msc


def head(sd, p, x):  # layers.py:124-127
    return F.conv2d(conv_abn(sd, p + ".head", x, 1, 1), sd[p + ".predictor.weight"])


def up(x, s):
    return F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=True)


def pose_cnn(sd, x):  # layers.py:155-167
    f = resnet18(sd, "pose_net.pose_encoder", x)["res5"]
    o = F.relu(F.conv2d(f, sd["pose_net.conv1.weight"], sd["pose_net.conv1.bias"]))
    o = F.relu(F.conv2d(o, sd["pose_net.conv2.weight"], sd["pose_net.conv2.bias"], padding=1))
    o = F.relu(F.conv2d(o, sd["pose_net.conv3.weight"], sd["pose_net.conv3.bias"], padding=1))
    o = F.conv2d(o, sd["pose_net.conv4.weight"], sd["pose_net.conv4.bias"])
    o = o.mean(3).mean(2)
    return 0.01 * o.view(o.size(0), 2, 6)


def ohem_ce(logits, labels, weights, ignore, thr, n_min):  # loss.py:67-81 (with the full sort, as written)
    pl = (F.cross_entropy(logits, labels, ignore_index=ignore, reduction="none") * weights).contiguous().view(-1)
    pl, _ = torch.sort(pl, descending=True)
    t = -torch.log(torch.tensor(thr, dtype=torch.float, device=pl.device))
    pl = pl[pl > t] if pl[n_min] > t else pl[:n_min]
    return pl.mean()


class _ReprojOracle(torch.autograd.Function):
    """MultiViewPhotometricLoss through the pinned C oracle (forward + its hand-derived backward)."""

    @staticmethod
    def forward(ctx, img, prev, nxt, mask, K, poses, *inv):
        import oracle
        dev = img.device   # (tensors of any device / dtype: evaluated on the host in fp32, results returned where they came from)
        npf = lambda t: t.detach().cpu().numpy() if t.dtype == torch.bool else t.detach().float().cpu().numpy()
        r = oracle.reproj_loss([npf(x) for x in inv], npf(img), npf(prev), npf(nxt), None if mask is None else npf(mask),
                               npf(K), npf(poses), g_photo=1.0, g_smooth=0.0)
        r2 = oracle.reproj_loss([npf(x) for x in inv], npf(img), npf(prev), npf(nxt), None if mask is None else npf(mask),
                                npf(K), npf(poses), g_photo=0.0, g_smooth=1.0)
        ctx.gp = ([torch.from_numpy(a).to(dev) for a in r["d_inv"]], torch.from_numpy(r["d_poses"]).to(dev))
        ctx.gs = [torch.from_numpy(a).to(dev) for a in r2["d_inv"]]
        ctx.dt = [x.dtype for x in inv] + [poses.dtype]
        return torch.tensor([float(r["loss_photometric"]), float(r["loss_smoothness"])], device=dev)

    @staticmethod
    def backward(ctx, g):
        d_inv = [(g[0] * a + g[1] * b).to(dt) for a, b, dt in zip(ctx.gp[0], ctx.gs, ctx.dt)]
        return (None, None, None, None, None, (g[0] * ctx.gp[1]).to(ctx.dt[-1])) + tuple(d_inv)


def pad32(t):
    H, W = t.shape[-2:]
    return F.pad(t, (0, (-W) % 32, 0, (-H) % 32))


def mgnet_losses(sd, batch, *, pixel_mean, pixel_std, with_panoptic=True, with_depth=True, with_uncertainty=True,
                 ohem_threshold=0.7, ohem_n_min=100000, ignore_value=255, sem_weight=1.0, center_weight=200.0,
                 offset_weight=0.01):
    """MGNet.forward, training branch (mg_net.py:249-373) on CPU tensors.  sd: name -> fp32 tensor (requires_grad ok)."""
    dev = batch[0]["image"].device
    mean = torch.tensor([m / 255.0 for m in pixel_mean], device=dev).view(-1, 1, 1)
    std = torch.tensor([s / 255.0 for s in pixel_std], device=dev).view(-1, 1, 1)
    stack = lambda key, f=lambda t: t: torch.stack([pad32(f(x[key])) for x in batch], 0)
    net_in = lambda key: (stack(key, lambda t: t.float() / 255.0) - mean) / std
    out = {}
    img = net_in("image")
    if with_depth:
        out["poses"] = pose_cnn(sd, torch.cat([img, net_in("image_prev"), net_in("image_next")], 1))
    feats = resnet18(sd, "backbone", img)
    feats["global_context"] = gcm(sd, feats["res5"])
    losses = {}
    if with_panoptic:
        y, _ = decoder(sd, "sem_seg_head", feats)
        sem = up(head(sd, "sem_seg_head.head", y), 8)
        y, _ = decoder(sd, "ins_embed_head", feats)
        center = up(torch.sigmoid(head(sd, "ins_embed_head.center_head", y)), 8)
        offset = up(head(sd, "ins_embed_head.offset_head", y), 8) * 8
        losses["loss_sem_seg"] = ohem_ce(sem, stack("sem_seg"), stack("sem_seg_weights"), ignore_value, ohem_threshold,
                                         ohem_n_min) * sem_weight
        cw, ow = stack("center_weights"), stack("offset_weights")
        lc = (center - stack("center").unsqueeze(1)) ** 2 * cw      # mg_net.py:697-715
        lc = lc.sum() / cw.sum() if cw.sum() > 0 else lc.sum() * 0
        lo = (offset - stack("offset")).abs() * ow
        lo = lo.sum() / ow.sum() if ow.sum() > 0 else lo.sum() * 0
        losses["loss_center"], losses["loss_offset"] = lc * center_weight, lo * offset_weight
    if with_depth:
        y, msc = decoder(sd, "depth_head", feats)
        inv = [up(torch.sigmoid(head(sd, f"depth_head.heads.{k}", f)) / 0.5, s)
               for k, (f, s) in enumerate(zip([y, msc[1], msc[0]], [8, 16, 32]))]
        f255 = lambda t: t.float() / 255.0
        K = torch.stack([x["camera_matrix"] for x in batch], 0)
        mask = stack("reprojection_mask").unsqueeze(1)
        lr = _ReprojOracle.apply(stack("image_orig", f255), stack("image_prev_orig", f255), stack("image_next_orig", f255),
                                 mask, K, out["poses"], *inv)
        losses["loss_photometric"], losses["loss_smoothness"] = lr[0], lr[1]
    if with_uncertainty:  # mg_net.py:360-372
        lv = sd["log_vars"]
        for idx, key in enumerate(list(losses)):
            tau = 1.0 if key == "loss_sem_seg" else 0.5
            losses[key] = tau * torch.exp(-lv[idx]) * losses[key] + 0.5 * lv[idx]
    return losses

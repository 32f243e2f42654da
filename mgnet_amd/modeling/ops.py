"""Functional ops of the network path.  CUDA tensors have ONE implementation per op: the hand-written HIP kernel behind the
C-ABI (marked [HIP]); the few ops without a kernel are torch GPU ops marked [torch-staging] and a
convolution without a kernel (fp32 activations) raises unless MGNET_ALLOW_TORCH_STAGING=1.  CPU tensors (the host-logic tests
of `-m "not gpu"`: config / registry / state-dict / gloo reducer plumbing) run torch restatements of `iabn` and `conv2d`; they
are never taken for a CUDA tensor and the oracle is never used here."""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

STAGING_USED = set()   # names of torch staging ops that actually ran on CUDA tensors (bench.py prints it)
SYNCBN_COLLECTIVES = [0]   # cross-rank statistics exchanges issued through torch.distributed so far (bench.py reports the count per step)
SYNCBN_P2P = [0]           # ... through the peer-to-peer mailbox kernels (engine/peer.py)


# ---------------------------------------------------------------------------------------------------------------
# InPlaceABNSync (inplace_abn >= 1.1.0, not vendored in the reference; semantics recalled):
#   y = act( (|gamma| + eps) * (x - mean) / sqrt(var + eps) + beta ),  act = leaky_relu(0.01) | identity
#   batch statistics over (N, H, W) synchronised over `group`; biased var for normalisation, unbiased for running_var;
#   running = (1 - momentum) * running + momentum * batch   (momentum 0.01 at all 68 call sites)
# ---------------------------------------------------------------------------------------------------------------
class _SyncStats(torch.autograd.Function):
    """all-reduce of per-channel [sum, sumsq, count] across ranks (forward) and of the matching gradients (backward)."""

    @staticmethod
    def forward(ctx, packed, group):
        ctx.group = group
        out = packed.clone()
        dist.all_reduce(out, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


def _dist_active(group):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def _peer_exchange(t, group=None):
    """the mailbox exchange of engine/peer.py when it is active for CUDA tensors AND spans exactly the layer's process group AND has a
    channel / slot for this payload -- else None: torch.distributed for this call (every rank decides the same: same shapes, same
    streams, same groups)"""
    if not t.is_cuda:
        return None
    from ..engine import peer
    ex = peer.exchange()
    if ex is None or not ex.can(t):
        return None
    if group is not None and group is not dist.group.WORLD and group is not ex.group:
        if dist.get_world_size(group) != ex.world or dist.get_process_group_ranks(group) != dist.get_process_group_ranks(ex.group):
            return None
    return ex


def _allreduce_sums(sums, group):
    """backward sums of a SyncBN layer over the ranks: [HIP] mailbox exchange (a new tensor), or dist.all_reduce in place"""
    ex = _peer_exchange(sums, group)
    if ex is not None:
        SYNCBN_P2P[0] += 1
        return ex.all_reduce(sums)
    SYNCBN_COLLECTIVES[0] += 1
    dist.all_reduce(sums, group=group)
    return sums


def _gather_stats(stats, world, group):
    """[world, 3, C] statistics of all ranks.  [HIP] mailbox exchange when active; RCCL: one all_gather_into_tensor (no per-rank output
    list and its copies); other backends (gloo in the CPU tests): the list form."""
    ex = _peer_exchange(stats, group)
    if ex is not None:
        SYNCBN_P2P[0] += 1
        return ex.all_gather(stats)
    gathered = torch.empty((world,) + tuple(stats.shape), dtype=stats.dtype, device=stats.device)
    SYNCBN_COLLECTIVES[0] += 1
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(gathered.view(-1), stats.contiguous().view(-1), group=group)
    else:
        dist.all_gather(list(gathered.unbind(0)), stats, group=group)
    return gathered


def _train_coef(x, M, C, w32, b32, eps, momentum, running_mean, running_var, world, group, pstats=None):
    """Coefficient block [scale, offset, mean, rstd] of a training-mode InPlaceABNSync over x [M, C] (+ running statistics).
    pstats = (partials, shift) from the producing convolution (csrc/conv_win.hip): no statistics pass over x."""
    from .. import _C
    if pstats is not None:
        part, shift = pstats
        if world > 1:
            stats = _C.iabn_from_partials(part, C, M, shift, stats_only=True)
            return _C.iabn_combine(_gather_stats(stats, world, group), w32, b32, eps, momentum, running_mean, running_var)
        return _C.iabn_from_partials(part, C, M, shift, w32, b32, eps, momentum, running_mean, running_var)
    if world > 1:
        return _C.iabn_combine(_gather_stats(_C.iabn_stats(x, M, C), world, group), w32, b32, eps, momentum, running_mean, running_var)
    return _C.iabn_train_coeffs(x, M, C, w32, b32, eps, momentum, running_mean, running_var)   # statistics + coefficients in one launch


def _pstats(x):
    """partial statistics the producing conv attached to its output (ops.conv2d(..., stats_for=norm)), consumed once"""
    return x.__dict__.pop("_mgn_stats", None) if isinstance(x, torch.Tensor) and hasattr(x, "__dict__") else None


class _IABNFn(torch.autograd.Function):
    """[HIP] mgnet_amd/csrc/iabn.hip through the C-ABI.  In place: the output overwrites the input's storage (the
    producing conv does not need its output for its own backward) and the backward re-derives x_hat from y."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, activation, slope, group, pstats=None):
        from .. import _C

        N, C, H, W = x.shape
        M = N * H * W
        inplace = x.is_contiguous(memory_format=torch.channels_last)
        xs = x if inplace else x.contiguous(memory_format=torch.channels_last)
        act = {"identity": 0, "leaky_relu": 1}[activation]
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        world = dist.get_world_size(group) if _dist_active(group) else 1
        if training:
            coef = _train_coef(xs, M, C, w32, b32, eps, momentum, running_mean, running_var, world, group, pstats)
            # every rank holds the same number of pixels (same per-GPU batch shape), as in the reference's DDP recipe;
            # the forward statistics themselves are combined with the true per-rank counts (Chan), this is only the
            # 1/n of the backward and avoids a host sync per layer
            total = float(M) * world
        else:
            coef = _C.iabn_eval_coeffs(w32, b32, running_mean, running_var, eps)
            total = float(M)
        _C.iabn_apply(xs, xs, M, C, coef[0], coef[1], act, slope)
        if inplace:
            ctx.mark_dirty(x)
        ctx.save_for_backward(xs, w32, b32, coef)
        ctx.cfg = (M, C, eps, act, slope, group, world, training, total, weight.dtype)
        return xs

    @staticmethod
    def backward(ctx, dy):
        from .. import _C

        y, w32, b32, coef = ctx.saved_tensors
        M, C, eps, act, slope, group, world, training, total, wdtype = ctx.cfg
        if dy.dtype != y.dtype:
            dy = dy.to(y.dtype)
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(y)
        if not training:  # eval: plain affine + activation
            raise NotImplementedError("backward through eval-mode InPlaceABNSync is not on the training path")
        sums, d_weight, d_bias = _C.iabn_bwd_reduce(y, dy, M, C, w32, b32, eps, act, slope)
        if world > 1:
            sums = _allreduce_sums(sums, group)
        _C.iabn_bwd_apply(y, dy, dx, M, C, w32, b32, coef[2:], sums, total, eps, act, slope)
        return dx, d_weight.to(wdtype), d_bias.to(wdtype), None, None, None, None, None, None, None, None, None


class _AbnPoolFn(torch.autograd.Function):
    """[HIP] csrc/pool.hip abn_maxpool_*: InPlaceABNSync + 3x3/s2 max pooling of the ResNet stems without materialising
    the normalised map (the conv output is kept instead; see the header of pool.hip for the traffic accounting)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, activation, slope, group, pstats=None):
        from .. import _C

        N, C, H, W = x.shape
        M = N * H * W
        act = {"identity": 0, "leaky_relu": 1}[activation]
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        world = dist.get_world_size(group) if _dist_active(group) else 1
        if training:   # statistics exactly as _IABNFn
            coef = _train_coef(x, M, C, w32, b32, eps, momentum, running_mean, running_var, world, group, pstats)
        else:
            coef = _C.iabn_eval_coeffs(w32, b32, running_mean, running_var, eps)
        y, arg = _C.abn_maxpool_fwd(x, coef[0], coef[1], act, slope)
        ctx.save_for_backward(x, y, arg, w32, b32, coef)
        ctx.cfg = (M, C, eps, act, slope, group, world, training, float(M) * world, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _C

        x, y, arg, w32, b32, coef = ctx.saved_tensors
        M, C, eps, act, slope, group, world, training, total, wdtype = ctx.cfg
        if not training:
            raise NotImplementedError("backward through eval-mode InPlaceABNSync is not on the training path")
        dy = dy.to(y.dtype).contiguous(memory_format=torch.channels_last)
        # d y is zero away from the arg-max positions and y there is the pooled value: the channel sums over the full map
        # equal the sums over the pooled tensors
        sums, d_weight, d_bias = _C.iabn_bwd_reduce(y, dy, y.numel() // C, C, w32, b32, eps, act, slope)
        if world > 1:
            sums = _allreduce_sums(sums, group)
        dx = _C.abn_maxpool_bwd(x, dy, arg, coef, w32, b32, sums, total, eps, act, slope)
        return dx, d_weight.to(wdtype), d_bias.to(wdtype), None, None, None, None, None, None, None, None, None


class _AbnAddReluFn(torch.autograd.Function):
    """[HIP] BasicBlock tail: relu(InPlaceABNSync_identity(x) + shortcut) in one pass over x; the normalised map is not
    written (x, the conv output, is kept and the norm's backward recomputes z = scale * x + offset)."""

    @staticmethod
    def forward(ctx, x, shortcut, weight, bias, running_mean, running_var, training, momentum, eps, group, pstats=None):
        from .. import _C

        N, C, H, W = x.shape
        M = N * H * W
        w32, b32 = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        world = dist.get_world_size(group) if _dist_active(group) else 1
        if training:
            coef = _train_coef(x, M, C, w32, b32, eps, momentum, running_mean, running_var, world, group, pstats)
        else:
            coef = _C.iabn_eval_coeffs(w32, b32, running_mean, running_var, eps)
        # [HIP] training, 16-bit: the ReLU mask leaves as one byte per 8 outputs beside y -- the backward reads it instead of y
        bits = None
        if training and x.dtype in _C.H16 and not os.environ.get("MGN_NO_TAILBITS") and not os.environ.get("MGN_NO_TAILMASK_FUSE"):
            y, bits = _C.abn_add_relu_fwd(x, coef, shortcut, want_bits=True)
        else:
            y = _C.abn_add_relu_fwd(x, coef, shortcut)
        ctx.has_bits = bits is not None
        ctx.save_for_backward(x, y, w32, b32, coef, *(() if bits is None else (bits,)))
        ctx.cfg = (M, C, eps, group, world, training, float(M) * world, weight.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _C

        x, y, w32, b32, coef = ctx.saved_tensors[:5]
        bits = ctx.saved_tensors[5] if ctx.has_bits else None
        M, C, eps, group, world, training, total, wdtype = ctx.cfg
        if not training:
            raise NotImplementedError("backward through eval-mode InPlaceABNSync is not on the training path")
        g = _cl(g, y)
        if x.dtype in _C.H16 and not os.environ.get("MGN_NO_TAILMASK_FUSE"):
            # [HIP] ReLU mask + reduction in one pass: dm = g * (y > 0) is written (gradient of both summands) while it is reduced
            dm, sums, d_weight, d_bias = _C.iabn_bwd_reduce_x_relu(x, g, y, M, C, w32, b32, coef, eps, relu_bits=bits)
        else:
            dm = _C.relu_mask_bwd(g, y)   # gradient of both summands
            sums, d_weight, d_bias = _C.iabn_bwd_reduce_x(x, dm, M, C, w32, b32, coef, eps, 0, 0.01)
        if world > 1:
            sums = _allreduce_sums(sums, group)
        dx = torch.empty_like(x)
        _C.iabn_bwd_apply_x(x, dm, dx, M, C, w32, b32, coef, sums, total, eps, 0, 0.01)
        return dx, dm, d_weight.to(wdtype), d_bias.to(wdtype), None, None, None, None, None, None, None


def conv_abn_eval(x, conv, residual=None, relu=False):
    """Inference form of `conv -> InPlaceABNSync` (and of the residual block's `-> + shortcut -> ReLU`): in eval mode the norm is a fixed
    per-channel affine + activation, so its scale is folded into the convolution's weights, its shift becomes the bias and the activation
    (after the optional residual) runs in the convolution's epilogue -- one launch, no pass over the activation for the norm (the
    reference swaps InPlaceABN for a foldable ABN for deployment the same way, tools/onnx_trt_export.py:19).  The folded 16-bit weights
    are cached on the norm module until a parameter / buffer changes or the module goes back to training mode.
    Returns None when the call is not such a case (training, gradients enabled, CPU, fp32 trunk, a shape without the epilogue): the
    caller then runs conv and norm as before."""
    from .. import _C
    norm = getattr(conv, "norm", None)
    if (norm is None or type(norm).__name__ != "InPlaceABNSync" or norm.training or torch.is_grad_enabled() or not x.is_cuda
            or x.dtype not in _C.H16 or x.dim() != 4 or conv.weight.shape[1] != x.shape[1] or x.shape[1] % 32 or conv.groups != 1
            or tuple(conv.dilation) != (1, 1) or conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1]
            or norm.activation not in ("identity", "leaky_relu") or (relu and norm.activation != "identity")
            or norm.running_mean.dtype != torch.float32 or os.environ.get("MGN_NO_EVALFOLD")):
        return None
    w = conv.weight
    Cout, Cin, KH, KW = w.shape
    stride, pad = conv.stride[0], conv.padding[0]
    N, _, IH, IW = x.shape
    OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
    if residual is not None and (tuple(residual.shape) != (N, Cout, OH, OW) or residual.dtype != x.dtype
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        return None
    key = (w.data_ptr(), w._version, getattr(_C.weight_cache, "epoch", 0), norm.weight._version, norm.bias._version, norm.running_mean._version,
           norm.running_var._version, norm.running_mean.data_ptr(), x.dtype, None if conv.bias is None else conv.bias._version)
    fold = norm.__dict__.get("_mgn_eval_fold")
    if fold is None or fold[0] != key:
        coef = _C.iabn_eval_coeffs(norm.weight.detach().float().contiguous(), norm.bias.detach().float().contiguous(), norm.running_mean,
                                   norm.running_var, norm.eps)          # [2, C]: y = coef[0] * x + coef[1]
        w2 = (w.detach().float() * coef[0].view(-1, 1, 1, 1)).contiguous()
        b2 = (coef[1] if conv.bias is None else coef[1] + coef[0] * conv.bias.detach().float()).contiguous()
        fold = (key, _C._weight_layout_now(w2, 0, 0, None, 0, x.dtype), b2)
        norm.__dict__["_mgn_eval_fold"] = fold
    act, slope = (1, 0.0) if relu else ((2, float(norm.activation_param)) if norm.activation == "leaky_relu" else (0, 0.0))
    return _C.conv_igemm_act(x.contiguous(memory_format=torch.channels_last), fold[1], (OH, OW), fold[2], stride, pad, act, slope, residual)


def abn_add_relu(x, norm, shortcut):
    """`relu_(norm(x) + shortcut)` for an identity-activation InPlaceABNSync (res_net.py:62-79); fused on the GPU path."""
    from .. import _C
    if (_C.elt_supported(x) and _C.elt_supported(shortcut) and x.shape == shortcut.shape and norm.activation == "identity"
            and not os.environ.get("MGN_NO_TAILFUSE")):
        return _AbnAddReluFn.apply(x, shortcut, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.training,
                                   norm.momentum, norm.eps, norm.group, _pstats(x))
    return add_relu(norm(x), shortcut)


def abn_max_pool(x, norm):
    """`max_pool_3x3_s2(norm(x))` for an InPlaceABNSync `norm` (BasicStem); fused on the GPU path."""
    from .. import _C
    if (x.is_cuda and x.dtype in _C.H16 and _C.elt_supported(x) and not os.environ.get("MGN_NO_STEMFUSE")
            and norm.activation in ("identity", "leaky_relu")):
        return _AbnPoolFn.apply(x, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.training, norm.momentum,
                                norm.eps, norm.activation, norm.activation_param, norm.group, _pstats(x))
    return max_pool_3x3_s2(norm(x))


def iabn(x, weight, bias, running_mean, running_var, training, momentum, eps, activation, slope, group=None):
    """Fused batch-norm + activation with cross-rank statistics.  CUDA tensors: [HIP]; CPU tensors (host-logic tests
    only): torch restatement below."""
    if x.is_cuda:
        return _IABNFn.apply(x, weight, bias, running_mean, running_var, training, momentum, eps, activation, slope, group, _pstats(x))
    xf = x.float()
    C = x.shape[1]
    gamma = weight.abs() + eps
    if training:
        # two-pass statistics: E[x^2]-mean^2 loses the variance of the 1x1-spatial layers (GCM, channel attention: only
        # N values per channel) to cancellation -- measured 8 % gradient error at N=2 in fp32
        n_local = x.numel() // C
        packed = torch.cat([xf.sum((0, 2, 3)), xf.new_tensor([float(n_local)])])
        if _dist_active(group):
            packed = _SyncStats.apply(packed, group)
        n = packed[-1]
        mean = packed[:C] / n
        m2 = ((xf - mean.view(1, C, 1, 1)) ** 2).sum((0, 2, 3))
        if _dist_active(group):
            m2 = _SyncStats.apply(m2, group)
        var = m2 / n
        with torch.no_grad():
            running_mean.mul_(1 - momentum).add_(mean.detach(), alpha=momentum)
            running_var.mul_(1 - momentum).add_(var.detach() * (n / (n - 1).clamp_min(1.0)), alpha=momentum)
    else:
        mean, var = running_mean, running_var
    scale = gamma * torch.rsqrt(var + eps)
    y = xf * scale.view(1, C, 1, 1) + (bias - mean * scale).view(1, C, 1, 1)
    if activation == "leaky_relu":
        y = F.leaky_relu(y, slope)
    elif activation != "identity":
        raise ValueError(activation)
    return y.to(x.dtype)


class _ConvFn(torch.autograd.Function):
    """[HIP] mgnet_amd/csrc/conv.hip: implicit-GEMM forward, data gradient (same kernel, flipped weights) and weight
    gradient on the bf16 matrix cores.  Master weights stay fp32 OIHW; the kernel layouts are derived per call."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu, with_skip=False, cout_pad=0, stats=None, keep_pad=False):
        from .. import _C

        N, Cin, IH, IW = x.shape   # Cin of the ACTIVATION (4/8/16 = channel-padded stem input)
        Cout, _, KH, KW = weight.shape
        OH, OW = (IH + 2 * pad - KH) // stride + 1, (IW + 2 * pad - KW) // stride + 1
        xs = x.contiguous(memory_format=torch.channels_last)
        b = None if bias is None else bias.detach().float().contiguous()
        packed = Cin in (4, 8, 16)
        if packed:  # stem on a channel-padded input: k = tap*Cin + c, row padded to a multiple of 32
            out = _C.conv_igemm(xs, _C.weight_layout(weight, 2, Cin, dtype=xs.dtype), (OH, OW), b, stride, pad, 1, relu, khw=(KH, KW), stats=stats)
        else:
            if cout_pad and b is not None:
                b = torch.cat([b, b.new_zeros(cout_pad - Cout)])
            out = _C.conv_igemm(xs, _C.weight_layout(weight, 0, 0, cout_pad, dtype=xs.dtype), (OH, OW), b, stride, pad, 1, relu, stats=stats)
        ctx.save_for_backward(xs, weight, out if relu else None)
        ctx.cfg = (stride, pad, relu, bias is not None, cout_pad, keep_pad)
        if cout_pad and keep_pad:   # the caller works on the padded map itself (ops.head_activation, the fused losses) and hands back a
            return out              # padded gradient with zero padding channels: no slice, no zero-fill + strided copy in the backward
        if cout_pad:    # few-class predictors: the kernels work on 32-padded output channels, the caller sees the real ones
            return out[:, :Cout]
        if with_skip == 2:
            # second output: the input at the pixels a stride-2 conv with no padding looks at (a strided VIEW: never read, it only carries
            # the autograd edge).  The block's 1x1 / stride-2 shortcut conv takes it as its differentiable input (_ShortcutS2Fn), so what
            # comes back into backward is that conv's data gradient as the plain LOW-resolution 1x1 product -- not its full-resolution
            # form, three quarters zeros, written by one kernel and read back by the next
            return out, x[:, :, ::2, ::2]
        if with_skip:   # second output: the input itself (autograd makes it an alias); its gradient comes back into backward
            return out, x
        return out

    @staticmethod
    def backward(ctx, dy, dskip=None):
        from .. import _C

        xs, weight, out = ctx.saved_tensors
        stride, pad, relu, has_bias, cout_pad, keep_pad = ctx.cfg
        Cout, Cin, KH, KW = weight.shape
        Cx = xs.shape[1]
        dy = dy.to(xs.dtype)
        if cout_pad and not keep_pad:    # zero gradient for the padding channels
            dyp = torch.zeros((dy.shape[0], cout_pad) + tuple(dy.shape[2:]), dtype=dy.dtype, device=dy.device).contiguous(memory_format=torch.channels_last)
            dyp[:, :Cout] = dy
            dy = dyp
        dy = dy.contiguous(memory_format=torch.channels_last)
        if relu:   # [HIP] dy * (out > 0) in one pass (csrc/eltwise.hip relu_mask_bwd)
            dy = _C.relu_mask_bwd(dy, out) if _C.elt_supported(out) and dy.shape == out.shape else dy * (out > 0)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            assert Cx == Cin, "no data gradient for the channel-padded stem input"
            # the gradient of the skip branch is added in the kernel's epilogue instead of by a separate accumulate pass
            res = None if dskip is None else dskip.to(xs.dtype).contiguous(memory_format=torch.channels_last)
            wl = _C.weight_layout(weight, 1, 0, cout_pad, dtype=xs.dtype)
            if res is not None and tuple(res.shape[2:]) != tuple(xs.shape[2:]):
                # with_skip = 2: the shortcut's gradient at the low resolution, added to the even pixels inside the window kernel
                dx = _C.conv_up2(dy, wl, xs.shape[2:], residual=res, residual_lowres=True)
                if dx is None:   # (a shape the window kernel does not take: the generic kernel + a strided accumulate)
                    dx = _C.conv_igemm(dy, wl, xs.shape[2:], None, 1, KH - 1 - pad, up=stride)
                    dx[:, :, ::2, ::2] += res
            else:
                dx = _C.conv_igemm(dy, wl, xs.shape[2:], None, 1, KH - 1 - pad, up=stride, residual=res)
        if ctx.needs_input_grad[1]:
            # (lazy: under a gradient reducer the split-K sums of a whole bucket run as one launch, engine/reducer.py)
            dw = _C.conv_wgrad(dy, xs, KH, KW, stride, pad, cin_real=Cin, lazy=not cout_pad)[:Cout]
        if has_bias and ctx.needs_input_grad[2]:
            # [HIP] per-channel sum over N*H*W: the column-sum kernel with the batch folded into the rows (deterministic two-stage sum)
            db = (_C.colsum_all(dy)[:Cout] if _C.elt_supported(dy) else dy.float().sum((0, 2, 3))[:Cout])
        return dx, dw, db, None, None, None, None, None, None, None


class _ConvCatFn(torch.autograd.Function):
    """[HIP] conv1x1(torch.cat([a, b], 1), weight) (FeatureFusionModule, layers.py:316-317) without the concatenated map: the streaming
    1x1 kernel reads its pixel rows as two half rows (mgn_conv1x1_cat), its data gradient writes the two halves of its output channels to
    two maps (mgn_conv1x1_split), the weight gradient gathers from both (mgn_conv_wgrad_cat).  Bit-identical to `_CatFn` + `_ConvFn`
    (same kernels, same summation order); saves the copy into the concatenation and the split of its gradient, 2 x 134 MB per decoder."""

    @staticmethod
    def forward(ctx, a, b, weight):
        from .. import _C
        out = _C.conv1x1_cat(a, b, _C.weight_layout(weight, 0, 0, 0, dtype=a.dtype))
        assert out is not None   # (conv_cat_supported is the gate)
        ctx.save_for_backward(a, b, weight)
        return out

    @staticmethod
    def backward(ctx, dy):
        from .. import _C
        a, b, weight = ctx.saved_tensors
        dy = dy.to(a.dtype).contiguous(memory_format=torch.channels_last)
        da = db = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            r = _C.conv1x1_split(dy, _C.weight_layout(weight, 1, 0, 0, dtype=a.dtype))
            if r is None:    # (not reachable for the shapes the forward accepted; kept as the definition)
                dcat = _C.conv_igemm(dy, _C.weight_layout(weight, 1, 0, 0, dtype=a.dtype), a.shape[2:], None, 1, 0)
                r = dcat[:, :a.shape[1]], dcat[:, a.shape[1]:]
            da, db = r
        if ctx.needs_input_grad[2]:
            dw = _C.conv_wgrad_cat(dy, a, b, lazy=True)
        return da, db, dw


def conv_cat_supported(a, b, weight, conv=None):
    from .. import _C
    if os.environ.get("MGN_NO_CONVCAT") or not (a.is_cuda and a.dtype in _C.H16 and a.dtype == b.dtype and a.shape == b.shape and a.shape[1] == 128):
        return False
    if tuple(weight.shape[1:]) != (256, 1, 1) or weight.shape[0] % 256 or weight.dtype != torch.float32:
        return False
    if conv is not None and (conv.bias is not None or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (0, 0)):
        return False
    if not (a.is_contiguous(memory_format=torch.channels_last) and b.is_contiguous(memory_format=torch.channels_last)):
        return False
    N, _, H, W = a.shape
    return (N * H * W + 127) // 128 * (weight.shape[0] // 256) >= 512    # (the streaming kernel's own size gate: tiles x channel groups)


class _ShortcutS2Fn(torch.autograd.Function):
    """The 1x1 / stride 2 / pad 0 shortcut conv of a down-sampling BasicBlock (res_net.py:52-60 `downsample`) next to a conv1 that
    returned `xsub` (with_skip = 2).  Forward: the ordinary strided 1x1 conv on the full input.  Backward: the gradient goes to `xsub`
    -- [N, Cin, OH, OW], a plain 1x1 conv of dy with the transposed weights -- and from there into conv1's data-gradient kernel."""

    @staticmethod
    def forward(ctx, xsub, xfull, weight, stats):
        from .. import _C
        Cout = weight.shape[0]
        xs = xfull.contiguous(memory_format=torch.channels_last)
        out = _C.conv_igemm(xs, _C.weight_layout(weight, 0, 0, 0, dtype=xs.dtype), tuple(xsub.shape[2:]), None, 2, 0, 1, False, stats=stats)
        ctx.save_for_backward(xs, weight)
        assert out.shape[1] == Cout
        return out

    @staticmethod
    def backward(ctx, dy):
        from .. import _C
        xs, weight = ctx.saved_tensors
        dy = dy.to(xs.dtype).contiguous(memory_format=torch.channels_last)
        dsub = dw = None
        if ctx.needs_input_grad[0]:
            dsub = _C.conv_igemm(dy, _C.weight_layout(weight, 1, 0, 0, dtype=xs.dtype), tuple(dy.shape[2:]), None, 1, 0)
        if ctx.needs_input_grad[2]:
            dw = _C.conv_wgrad(dy, xs, 1, 1, 2, 0, cin_real=weight.shape[1], lazy=True)
        return dsub, None, dw, None


def sub2_supported(x, w3, w1):
    """conv1 (3x3 / s2 / p1) + shortcut (1x1 / s2 / p0) of a down-sampling block on the fused path: 16-bit CUDA activations that carry a
    gradient, channel counts the window kernel takes"""
    from .. import _C
    return (x.is_cuda and x.dtype in _C.H16 and torch.is_grad_enabled() and x.requires_grad and x.shape[1] % 64 == 0 and w3.shape[0] % 32 == 0
            and w3.shape[1] == x.shape[1] == w1.shape[1] and tuple(w3.shape[2:]) == (3, 3) and tuple(w1.shape[2:]) == (1, 1)
            and not os.environ.get("MGN_NO_SKIPFUSE") and not os.environ.get("MGN_NO_SUB2"))


def conv2d_shortcut_s2(xsub, xfull, weight, stats_for=None):
    """the shortcut conv of a down-sampling block when conv1 was called with with_skip=2 (see _ShortcutS2Fn)"""
    holder = []
    stats = None
    if (stats_for is not None and stats_for.training and torch.is_grad_enabled() and not os.environ.get("MGN_NO_STATFUSE")
            and stats_for.running_mean.dtype == torch.float32):
        stats = (stats_for.running_mean, holder)
    y = _ShortcutS2Fn.apply(xsub, xfull, weight, stats)
    if holder:
        y._mgn_stats = holder[0]
    return y


def conv2d(x, weight, bias=None, stride=1, padding=0, relu=False, with_skip=False, stats_for=None, keep_pad=False):
    """Convolution in the activation dtype (bf16 under AMP) from fp32 master weights.
    bf16 CUDA activations with Cin % 32 == 0: [HIP] implicit GEMM (Cout is zero-padded to a multiple of 32 for the
    few-class predictors).  Otherwise (fp32 activations, the 3/9-channel 7x7 stems, CPU tests): [torch-staging].
    with_skip: also return the input as a second output whose gradient is accumulated inside the data-gradient kernel
    (ResNet shortcut: `out, skip = conv2d(x, ..., with_skip=True)` then use `skip` wherever `x` would be re-used)."""
    stride = stride[0] if isinstance(stride, (tuple, list)) else stride
    padding = padding[0] if isinstance(padding, (tuple, list)) else padding
    from .. import _C
    if _C.conv_supported(x, weight):
        Cout = weight.shape[0]
        if Cout % 32:
            assert not with_skip
            # keep_pad: return the 32-padded map (a PaddedMap: tensor + real channel count) for consumers that read it in place
            keep_pad = bool(keep_pad) and not os.environ.get("MGN_NO_PADTAIL")
            y = _ConvFn.apply(x, weight, bias, stride, padding, relu, False, (Cout + 31) // 32 * 32, None, keep_pad)
            return PaddedMap(y, Cout) if keep_pad else y
        # stats_for: the InPlaceABNSync that consumes the output next; in training the conv kernel then leaves the partial sums of
        # the batch statistics behind (attached to the output, taken by ops.iabn / abn_add_relu / abn_max_pool)
        holder = []
        stats = None
        if (stats_for is not None and stats_for.training and torch.is_grad_enabled() and not os.environ.get("MGN_NO_STATFUSE")
                and stats_for.running_mean.dtype == torch.float32):
            stats = (stats_for.running_mean, holder)
        if with_skip and x.requires_grad and x.shape[1] == weight.shape[1] and not os.environ.get("MGN_NO_SKIPFUSE"):
            y, skip = _ConvFn.apply(x, weight, bias, stride, padding, relu, with_skip, 0, stats)
        else:
            y, skip = _ConvFn.apply(x, weight, bias, stride, padding, relu, False, 0, stats), x
        if holder:
            y._mgn_stats = holder[0]
        return (y, skip) if with_skip else y
    if x.is_cuda and x.dtype == torch.float32 and not os.environ.get("MGNET_ALLOW_TORCH_STAGING") and _fp32_split_supported(x, weight):
        if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
            y = _Conv32Fn.apply(x, weight, bias, stride, padding, relu)   # [HIP] fp32 TRAINING (SOLVER.AMP.ENABLED False, detectron2's default)
        else:
            y = _conv2d_fp32_split(x, weight, bias, stride, padding, relu)   # [HIP] inference with fp32 activations (AMP off)
        return (y, x) if with_skip else y
    if x.is_cuda and not os.environ.get("MGNET_ALLOW_TORCH_STAGING"):
        raise NotImplementedError(
            f"conv2d: no HIP kernel for {x.dtype} activations with {tuple(weight.shape)} weights (the conv kernels are bf16, Cin % 32 == 0 "
            "or the channel-padded stems): enable SOLVER.AMP.ENABLED, or set MGNET_ALLOW_TORCH_STAGING=1 to run this layer on "
            "torch's convolution (staging, not the product path)")
    STAGING_USED.add("conv2d:" + str(x.dtype).replace("torch.", ""))
    w = weight.to(x.dtype)
    b = None if bias is None else bias.to(x.dtype)
    y = F.conv2d(x, w, b, stride=stride, padding=padding)
    y = torch.relu_(y) if relu else y
    return (y, x) if with_skip else y


class PaddedMap:
    """a predictor output with its channels padded to a multiple of 32 (`t` [B,P,h,w] 16-bit channels_last, `C` real channels): the
    convolution kernels' own layout, handed to consumers that read it in place and return a padded gradient.  `real()` is the
    [B,C,h,w] tensor the reference's module returns."""

    def __init__(self, t, C):
        self.t, self.C = t, C

    def real(self):
        return self.t[:, :self.C]

    def tensors(self):
        return [self.t]


def real_channels(x):
    return x.real() if isinstance(x, PaddedMap) else x


class _HeadActFn(torch.autograd.Function):
    """[HIP] csrc/headact.hip: padded predictor output -> fp32 [B,C,h,w] = x | sigmoid(x) | sigmoid(x) / 0.5, and the gradient back
    into the padded 16-bit layout (zero padding channels) in one launch each"""

    @staticmethod
    def forward(ctx, xp, C, kind):
        from .. import _C
        y = _C.head_act_fwd(xp, C, kind)
        ctx.save_for_backward(y if kind != "none" else None)
        ctx.cfg = (tuple(xp.shape), C, kind, xp.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        (y,) = ctx.saved_tensors
        shape, C, kind, dtype = ctx.cfg
        g = g.float()
        if g.stride(3) * g.shape[3] != g.stride(2):
            g = g.contiguous()
        return _C.head_act_bwd(g, y, shape, C, kind, dtype), None, None


def head_activation(x, kind):
    """`act(x.float())` of a predictor output, act = none | sigmoid (mg_net.py:694) | sigmoid2 = sigmoid / 0.5 (:822).
    PaddedMap on the GPU: [HIP] one launch forward, one backward; tensors: torch ops"""
    if isinstance(x, PaddedMap):
        return _HeadActFn.apply(x.t, x.C, kind)
    y = x.float()
    if kind == "none":
        return y
    y = torch.sigmoid(y)
    return y / 0.5 if kind == "sigmoid2" else y


class _MeanHWFn(torch.autograd.Function):
    """[HIP] scale * mean over H, W of a padded map -> [N, C] fp32 (PoseCNN's `0.01 * out.mean(3).mean(2)`, layers.py:165-167): the column-sum
    kernels forward, the broadcast kernel backward -- which writes the padded 16-bit gradient the predictor's backward consumes"""

    @staticmethod
    def forward(ctx, xp, C, scale):
        from .. import _C
        N, P, H, W = xp.shape
        ctx.cfg = (tuple(xp.shape), C, scale / (H * W), xp.dtype)
        return _C.colsum(xp, None, scale / (H * W))[:, :C].contiguous()

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        shape, C, k, dtype = ctx.cfg
        gp = g.new_zeros((shape[0], shape[1]))
        gp[:, :C] = g
        return _C.bcast_rows(gp, shape, k, dtype), None, None


def mean_hw(x, scale=1.0):
    """scale * x.float().mean((2, 3)) -> [N, C] fp32"""
    from .. import _C
    if isinstance(x, PaddedMap) and _C.elt_supported(x.t):
        return _MeanHWFn.apply(x.t, x.C, float(scale))
    x = real_channels(x).float()
    return scale * x.mean(3).mean(2)


def _fp32_split_supported(x, weight):
    Cin = weight.shape[1]
    return x.dim() == 4 and (Cin % 32 == 0 or Cin in (3, 9)) and x.shape[1] in (Cin, 8, 16)


def _conv2d_fp32_split(x, weight, bias, stride, padding, relu):
    """fp32-accurate convolution for INFERENCE on fp32 activations (a config with SOLVER.AMP.ENABLED False; the reference's
    MGNet-*-PseudoLabelGeneration.yaml): the matrix cores have no fp32 mode worth using here, so x and w are split into bf16 high and low
    parts and the product is three bf16 MFMA passes accumulated in fp32,
        x w ~= x_hi w_hi + x_hi w_lo + x_lo w_hi          (the dropped x_lo w_lo term is 2^-16 relative),
    each pass the product's own implicit-GEMM kernel with fp32 output (mgn_conv_igemm, out_f32).  ~1e-5 relative to an fp32 convolution
    (tests/test_conv_gpu.py).  With gradients: _Conv32Fn (the same three-pass form for the data and weight gradients)."""
    from .. import _C
    with torch.no_grad():
        Cout, Cin, KH, KW = weight.shape
        N, Cx, IH, IW = x.shape
        OH, OW = (IH + 2 * padding - KH) // stride + 1, (IW + 2 * padding - KW) // stride + 1
        packed = Cin in (3, 9)
        if packed and Cx == Cin:       # the stems' channel-padded layout (8 / 16 channels, zeros beyond the real ones)
            x = F.pad(x, (0, 0, 0, 0, 0, (8 if Cin == 3 else 16) - Cin))
        Cp = x.shape[1]
        w32 = weight.detach().float()
        cout_pad = (Cout + 31) // 32 * 32 if Cout % 32 else 0
        xh = x.to(torch.bfloat16)
        xl = (x - xh.float()).to(torch.bfloat16)
        xh, xl = (t.contiguous(memory_format=torch.channels_last) for t in (xh, xl))
        wh32 = w32.to(torch.bfloat16).float()
        parts = []
        for xa, wa in ((xh, wh32), (xh, w32 - wh32), (xl, wh32)):
            wl = _C._weight_layout_now(wa.contiguous(), 2 if packed else 0, Cp if packed else 0, None, cout_pad)
            parts.append(_C.conv_igemm(xa, wl, (OH, OW), None, stride, padding, 1, False, out_dtype=torch.float32, khw=(KH, KW) if packed else None))
        y = (parts[0] + parts[1]) + parts[2]
        if cout_pad:
            y = y[:, :Cout]
        if bias is not None:
            y = y + bias.detach().float().view(1, -1, 1, 1)
        return torch.relu_(y) if relu else y


def _split16(t):
    """fp32 -> (hi, lo) bf16 parts with t ~= hi + lo to 2^-16 relative; channels_last"""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return hi.contiguous(memory_format=torch.channels_last), lo.contiguous(memory_format=torch.channels_last)


class _Conv32Fn(torch.autograd.Function):
    """[HIP] convolution on fp32 activations WITH gradients (a config with SOLVER.AMP.ENABLED False -- detectron2's default, which
    mgnet/config.py leaves; tools/train_net.py:37): forward, data gradient and weight gradient are each three bf16 MFMA passes over hi / lo
    splits of both operands with fp32 accumulation, a b ~= a_hi b_hi + a_hi b_lo + a_lo b_hi (the dropped term is 2^-16 relative), on the
    product's own kernels (mgn_conv_igemm with fp32 output, mgn_conv_wgrad).  ~1e-5 of an fp32 convolution; 3x the 16-bit cost, which is
    what fp32 training costs on matrix cores without an fp32 mode worth using."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, relu):
        y = _conv2d_fp32_split(x, weight, bias, stride, pad, relu)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.cfg = (stride, pad, relu, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _C
        x, weight, y = ctx.saved_tensors
        stride, pad, relu, has_bias = ctx.cfg
        Cout, Cin, KH, KW = weight.shape
        dy = dy.float()
        if relu:
            dy = dy * (y > 0)
        dx = dw = db = None
        packed = Cin in (3, 9)
        xp = x
        if packed and x.shape[1] == Cin:
            xp = F.pad(x, (0, 0, 0, 0, 0, (8 if Cin == 3 else 16) - Cin))
        cout_pad = (Cout + 31) // 32 * 32 if Cout % 32 else 0
        if cout_pad:     # the kernels work on 32-padded output channels: zero gradient for the padding
            dy = F.pad(dy, (0, 0, 0, 0, 0, cout_pad - Cout))
        dyh, dyl = _split16(dy)
        if ctx.needs_input_grad[0]:
            assert not packed, "no data gradient for the channel-padded stem input"
            w32 = weight.detach().float()
            wh = w32.to(torch.bfloat16).float()
            parts = []
            for ga, wa in ((dyh, wh), (dyh, w32 - wh), (dyl, wh)):
                wl = _C._weight_layout_now(wa.contiguous(), 1, 0, None, cout_pad)   # flipped / transposed layout of the data gradient
                parts.append(_C.conv_igemm(ga, wl, x.shape[2:], None, 1, KH - 1 - pad, up=stride, out_dtype=torch.float32))
            dx = (parts[0] + parts[1]) + parts[2]
        if ctx.needs_input_grad[1]:
            xh, xl = _split16(xp)
            dw = None
            for ga, xa in ((dyh, xh), (dyh, xl), (dyl, xh)):
                part = _C.conv_wgrad(ga, xa, KH, KW, stride, pad, cin_real=Cin)[:Cout]
                dw = part if dw is None else dw + part
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))[:Cout]
        return dx, dw, db, None, None, None


class _FanoutFn(torch.autograd.Function):
    """x -> (x, x, x) for a tensor three branches consume (the backbone features and the three heads, mg_net.py:290-311): the backward
    sums the three gradients in ONE pass ([HIP] csrc/eltwise.hip sum3) instead of autograd's two accumulation passes."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        from .. import _C
        gs = [g for g in (g0, g1, g2) if g is not None]
        if len(gs) <= 1:
            return gs[0] if gs else None
        same = all(g.dtype == gs[0].dtype and g.shape == gs[0].shape and g.stride() == gs[0].stride() for g in gs)
        if same and gs[0].is_cuda and gs[0].dtype in _C.H16 and gs[0].numel() % 8 == 0 and \
                (gs[0].is_contiguous() or gs[0].is_contiguous(memory_format=torch.channels_last)):
            return _C.sum3(*gs)
        out = gs[0] + gs[1]
        return out + gs[2] if len(gs) == 3 else out


def fanout3(x):
    """three aliases of x whose gradients are summed by one kernel"""
    if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled() and not os.environ.get("MGN_NO_FANOUT"):
        return _FanoutFn.apply(x)
    return x, x, x


class _MaxPoolFn(torch.autograd.Function):
    """[HIP] csrc/pool.hip"""

    @staticmethod
    def forward(ctx, x):
        from .. import _C
        y, arg = _C.maxpool_fwd(x.contiguous(memory_format=torch.channels_last))
        ctx.save_for_backward(arg)
        ctx.in_shape, ctx.dtype = tuple(x.shape), x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _C
        (arg,) = ctx.saved_tensors
        return _C.maxpool_bwd(dy.to(ctx.dtype).contiguous(memory_format=torch.channels_last), arg, ctx.in_shape)


def max_pool_3x3_s2(x):
    """F.max_pool2d(x, 3, stride 2, padding 1) (res_net.py:109).  bf16 CUDA: [HIP]; otherwise [torch-staging]."""
    if x.is_cuda and x.dtype in _C_H16() and x.shape[1] % 8 == 0:
        return _MaxPoolFn.apply(x)
    return F.max_pool2d(x, kernel_size=3, stride=2, padding=1)


def _C_H16():
    from .. import _C
    return _C.H16


def _cl(t, like):
    """gradient `t` in the 16-bit activation format of `like` (a tensor or a dtype), channels-last"""
    return t.to(like if isinstance(like, torch.dtype) else like.dtype).contiguous(memory_format=torch.channels_last)


class _GapFn(torch.autograd.Function):
    """[HIP] csrc/eltwise.hip: two-stage column mean over H*W and its broadcast adjoint"""

    @staticmethod
    def forward(ctx, x):
        from .. import _C
        ctx.shape, ctx.dtype = tuple(x.shape), x.dtype
        N, C, H, W = x.shape
        return _C.colsum(x, None, 1.0 / (H * W)).to(x.dtype).view(N, C, 1, 1)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        N, C, H, W = ctx.shape
        return _C.bcast_rows(g.float().reshape(N, C).contiguous(), ctx.shape, 1.0 / (H * W), ctx.dtype)


def global_avg_pool(x):
    """layers.py:170-184 FastGlobalAvgPool2d (flatten=False): [B,C,H,W] -> [B,C,1,1].  bf16 CUDA: [HIP]"""
    from .. import _C
    if _C.elt_supported(x):
        return _GapFn.apply(x)
    return x.float().mean((2, 3), keepdim=True).to(x.dtype)


class _NearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        from .. import _C
        ctx.hw, ctx.dtype = x.shape[2:], x.dtype
        return _C.nearest_fwd(x, H, W)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        g = _cl(g, ctx.dtype)
        if ctx.hw[0] == 1 and ctx.hw[1] == 1:  # broadcast of a pooled vector (GlobalContextModule): its adjoint is a column sum
            N, C = g.shape[:2]
            return _C.colsum(g, None, 1.0).to(g.dtype).view(N, C, 1, 1).contiguous(memory_format=torch.channels_last), None, None
        return _C.nearest_bwd(g, *ctx.hw), None, None


def upsample_nearest(x, size):
    """F.interpolate(mode='nearest') (layers.py:90, :217).  bf16 CUDA: [HIP]"""
    from .. import _C
    H, W = int(size[0]), int(size[1])
    if _C.elt_supported(x) and H >= x.shape[2] and W >= x.shape[3]:
        return _NearestFn.apply(x, H, W)
    return F.interpolate(x, size=(H, W), mode="nearest")


class _AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        from .. import _C
        y = _C.add_relu_fwd(a, b)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        (y,) = ctx.saved_tensors
        dx = _C.relu_mask_bwd(_cl(g, y), y)
        return dx, dx


def add_relu(a, b):
    """res_net.py:77-78: relu_(out + shortcut).  bf16 CUDA: [HIP]"""
    from .. import _C
    if _C.elt_supported(a) and _C.elt_supported(b) and a.shape == b.shape:
        return _AddReluFn.apply(a, b)
    return torch.relu_(a + b)


class _ScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, mode):
        from .. import _C
        s32 = s.float().reshape(x.shape[0], x.shape[1]).contiguous()
        ctx.save_for_backward(x, s32)
        ctx.mode, ctx.sdtype, ctx.sshape = mode, s.dtype, tuple(s.shape)
        return _C.scale_channels(x, s32, mode)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        x, s32 = ctx.saved_tensors
        g = _cl(g, x)
        dx = _C.scale_channels(g, s32, ctx.mode)
        ds = _C.colsum(g, x, 1.0).to(ctx.sdtype).view(ctx.sshape)
        return dx, ds, None


def scale_channels(x, s, residual=False):
    """x * s (AttentionRefinementModule, layers.py:262-267) or x + x * s (FeatureFusionModule, :315-322) with a
    per-(image, channel) factor s [B,C,1,1].  bf16 CUDA: [HIP]"""
    from .. import _C
    if _C.elt_supported(x):
        return _ScaleFn.apply(x, s, 1 if residual else 0)
    return x + x * s if residual else x * s


class _ChannelAttentionFn(torch.autograd.Function):
    """[HIP] y = x * s (ARM, layers.py:262-267) or x + x * s (FFM, :315-322) with s = attention(global_avg_pool(x)), the
    whole attention branch in a handful of launches (csrc/attention.hip) and ONE autograd node: the pooled branch's
    gradient is added inside the scale kernel of the backward (no broadcast tensor, no accumulate pass).
      kind "arm": s = sigmoid(IABN_identity(W p))        params: W, bn_weight, bn_bias (+ running stats buffers)
      kind "ffm": s = sigmoid(W2 relu(W1 p))             params: W1, W2"""

    @staticmethod
    def forward(ctx, x, kind, residual, w1, p2, p3, running_mean, running_var, training, momentum, eps, addend=None):
        from .. import _C
        N, C, H, W = x.shape
        pooled = _C.colsum(x, None, 1.0 / (H * W))
        w1c = w1.detach().reshape(w1.shape[0], -1)
        if kind == "arm":
            s, xhat, rstd = _C.vec_linear_fwd(pooled, w1c, "sigmoid", (p2.detach(), p3.detach(), running_mean, running_var, training, momentum, eps))
            ctx.save_for_backward(x, pooled, s, w1c, p2.detach(), xhat, rstd)
        else:
            w2c = p2.detach().reshape(p2.shape[0], -1)
            h, _, _ = _C.vec_linear_fwd(pooled, w1c, "relu")
            s, _, _ = _C.vec_linear_fwd(h, w2c, "sigmoid")
            ctx.save_for_backward(x, pooled, s, w1c, w2c, h)
        ctx.cfg = (kind, residual, eps, tuple(w1.shape), None if p2 is None else tuple(p2.shape), addend is not None)
        # addend: `arm(x) + last` of the decoder (layers.py:87) folded into the scaling pass (fp32 sum, one rounding)
        return _C.scale_channels(x, s, 1 if residual else 0, addt=addend)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        kind, residual, eps, w1s, p2s, has_addend = ctx.cfg
        g = _cl(g, ctx.saved_tensors[0])
        if kind == "arm":
            x, pooled, s, w1c, bnw, xhat, rstd = ctx.saved_tensors
            ds = _C.colsum(g, x, 1.0)
            dW1, dpool, dbw, dbb = _C.vec_linear_bwd(ds, s, pooled, w1c, "sigmoid", bnw, xhat, rstd, eps, dv_scale=1.0 / (x.shape[2] * x.shape[3]))
            d2, d3 = dbw, dbb
        else:
            x, pooled, s, w1c, w2c, h = ctx.saved_tensors
            ds = _C.colsum(g, x, 1.0)
            dW2, dh, _, _ = _C.vec_linear_bwd(ds, s, h, w2c, "sigmoid")
            dW1, dpool, _, _ = _C.vec_linear_bwd(dh, h, pooled, w1c, "relu", dv_scale=1.0 / (x.shape[2] * x.shape[3]))
            d2, d3 = dW2.view(p2s), None
        N, C, H, W = x.shape
        dx = _C.scale_channels(g, s, 1 if residual else 0, add=dpool)   # dpool already carries the pool's 1/(H*W)
        return dx, None, None, dW1.view(w1s), d2, d3, None, None, None, None, None, (g if has_addend else None)


def channel_attention(x, attention, kind, residual=False, addend=None):
    """x * s / x + x * s with s = `attention`(x), an nn.Sequential(FastGlobalAvgPool2d, Conv2d(+norm | +ReLU), [conv], Sigmoid)
    of the reference's decoder modules.  bf16 CUDA, one process: [HIP] fused path; otherwise the module itself."""
    from .. import _C
    conv = attention[1]
    fused = _C.elt_supported(x) and x.shape[0] <= 64 and conv.bias is None and not os.environ.get("MGN_NO_ATTN_FUSE")
    add_ok = addend is not None and not os.environ.get("MGN_NO_ARMADD") and addend.shape == x.shape and addend.dtype == x.dtype and addend.is_contiguous(memory_format=torch.channels_last)
    if kind == "arm":
        bn = conv.norm
        fused = fused and not _dist_active(bn.group) and (bn.training or not torch.is_grad_enabled())
        if fused:
            y = _ChannelAttentionFn.apply(x, "arm", residual, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                          bn.training, bn.momentum, bn.eps, addend if add_ok else None)
            return y if (addend is None or add_ok) else y + addend
    elif fused and attention[2].bias is None:
        y = _ChannelAttentionFn.apply(x, "ffm", residual, conv.weight, attention[2].weight, None, None, None, True, 0.0, 0.0,
                                      addend if add_ok else None)
        return y if (addend is None or add_ok) else y + addend
    y = scale_channels(x, attention(x), residual=residual)
    return y if addend is None else y + addend


class _AbnAttentionFn(torch.autograd.Function):
    """[HIP] conv output -> InPlaceABNSync -> channel attention (ARM layers.py:221-267 / FFM :270-322) as ONE autograd node whose passes over
    the activation are fused (csrc/eltwise.hip `abn_apply_pool`, `att_abn_bwd_*`): the pool is taken while the normalised map is written,
    and in the backward the five per-(image, channel) sums of one pass over (g, z) give both the pooled branch's input and the norm's two
    sums -- 2 R + 2 W forward and 4 R + 1 W backward of activation-sized traffic instead of 3 R + 2 W and 7 R + 2 W for the separate
    _IABNFn and _ChannelAttentionFn nodes.  Same formulas; the gradient g * base + dpool is no longer rounded to 16 bit in between."""

    @staticmethod
    def forward(ctx, y, pstats, nw, nb, nrm, nrv, training, momentum, eps, activation, slope, group,
                kind, residual, w1, p2, p3, a_rm, a_rv, a_training, a_momentum, a_eps, addend=None):
        from .. import _C
        N, C, H, W = y.shape
        M = N * H * W
        act = {"identity": 0, "leaky_relu": 1}[activation]
        w32, b32 = nw.detach().float().contiguous(), nb.detach().float().contiguous()
        world = dist.get_world_size(group) if _dist_active(group) else 1
        if training:
            coef = _train_coef(y, M, C, w32, b32, eps, momentum, nrm, nrv, world, group, pstats)
        else:
            coef = _C.iabn_eval_coeffs(w32, b32, nrm, nrv, eps)
        pooled = _C.abn_apply_pool(y, y, coef[0], coef[1], act, slope, 1.0 / (H * W))   # in place: y now holds z
        z = y
        w1c = w1.detach().reshape(w1.shape[0], -1)
        if kind == "arm":
            s, xhat, rstd = _C.vec_linear_fwd(pooled, w1c, "sigmoid", (p2.detach(), p3.detach(), a_rm, a_rv, a_training, a_momentum, a_eps))
            ctx.save_for_backward(z, pooled, s, w1c, w32, b32, coef, p2.detach(), xhat, rstd)
        else:
            w2c = p2.detach().reshape(p2.shape[0], -1)
            h, _, _ = _C.vec_linear_fwd(pooled, w1c, "relu")
            s, _, _ = _C.vec_linear_fwd(h, w2c, "sigmoid")
            ctx.save_for_backward(z, pooled, s, w1c, w32, b32, coef, w2c, h)
        # (y is overwritten through the raw kernel and never used again: the producing conv keeps its inputs, not its output)
        ctx.cfg = (kind, residual, eps, act, slope, group, world, training, float(M) * world, nw.dtype, a_eps, tuple(w1.shape),
                   None if p2 is None else tuple(p2.shape), addend is not None)
        return _C.scale_channels(z, s, 1 if residual else 0, addt=addend)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        kind, residual, eps, act, slope, group, world, training, total, wdtype, a_eps, w1s, p2s, has_addend = ctx.cfg
        if not training:
            raise NotImplementedError("backward through eval-mode InPlaceABNSync is not on the training path")
        z = ctx.saved_tensors[0]
        g = _cl(g, z)
        mode = 1 if residual else 0
        if kind == "arm":
            z, pooled, s, w1c, w32, b32, coef, bnw, xhat, rstd = ctx.saved_tensors
        else:
            z, pooled, s, w1c, w32, b32, coef, w2c, h = ctx.saved_tensors
        S = _C.att_abn_bwd_stats(g, z, w32, b32, eps, act, slope)
        ds = S[0]     # sum g z per (image, channel): the gradient of the attention factor
        hw = z.shape[2] * z.shape[3]
        if kind == "arm":
            dW1, dpool, dbw, dbb = _C.vec_linear_bwd(ds, s, pooled, w1c, "sigmoid", bnw, xhat, rstd, a_eps, dv_scale=1.0 / hw)
            d2, d3 = dbw, dbb
        else:
            dW2, dh, _, _ = _C.vec_linear_bwd(ds, s, h, w2c, "sigmoid")
            dW1, dpool, _, _ = _C.vec_linear_bwd(dh, h, pooled, w1c, "relu", dv_scale=1.0 / hw)
            d2, d3 = dW2.view(p2s), None
        sums, d_nw, d_nb = _C.att_abn_bwd_sums(S, s, dpool, mode, w32)
        if world > 1:
            sums = _allreduce_sums(sums, group)
        dy = _C.att_abn_bwd_apply(g, z, s, dpool, mode, w32, b32, coef[3], sums, total, eps, act, slope)
        return (dy, None, d_nw.to(wdtype), d_nb.to(wdtype), None, None, None, None, None, None, None, None,
                None, None, dW1.view(w1s), d2, d3, None, None, None, None, None, (g if has_addend else None))


def conv_abn_attention(conv, x, attention, kind, residual=False, addend=None):
    """`fm = conv(x)` (a layers.Conv2d with an InPlaceABNSync norm) followed by ops.channel_attention(fm, attention, kind, ...): on the GPU
    path one fused node behind the convolution (_AbnAttentionFn), otherwise exactly those two calls"""
    from .. import _C
    norm = getattr(conv, "norm", None)
    a1 = attention[1]
    pair = x if isinstance(x, tuple) else None      # (fsp, fcp) of the FeatureFusionModule: the conv reads both maps, no concatenation
    if pair is not None:
        x = pair[0]
    ok = (x.is_cuda and x.dtype in _C.H16 and norm is not None and type(norm).__name__ == "InPlaceABNSync" and conv.activation is None
          and conv.out_channels % 8 == 0 and conv.out_channels // 8 <= 256 and 256 % (conv.out_channels // 8) == 0 and x.shape[0] <= 64
          and a1.bias is None and (norm.training or not torch.is_grad_enabled()) and not os.environ.get("MGN_NO_ATTN_FUSE")
          and not os.environ.get("MGN_NO_ABN_ATTN"))
    add_ok = addend is None or (not os.environ.get("MGN_NO_ARMADD") and addend.dtype == x.dtype and addend.is_contiguous(memory_format=torch.channels_last))
    if kind == "arm":
        bn = a1.norm
        ok = ok and not _dist_active(bn.group) and (bn.training or not torch.is_grad_enabled())
    else:
        ok = ok and attention[2].bias is None
    if pair is not None and not (ok and add_ok and conv_cat_supported(pair[0], pair[1], conv.weight, conv)):
        x, pair = concat_channels(*pair), None
    if ok and add_ok:
        y = _ConvCatFn.apply(pair[0], pair[1], conv.weight) if pair is not None else conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, stats_for=norm)
        if _C.elt_supported(y) and (addend is None or addend.shape == y.shape):
            ps = _pstats(y)
            if kind == "arm":
                bn = a1.norm
                return _AbnAttentionFn.apply(y, ps, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.training, norm.momentum,
                                             norm.eps, norm.activation, norm.activation_param, norm.group, "arm", residual, a1.weight,
                                             bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, addend)
            return _AbnAttentionFn.apply(y, ps, norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.training, norm.momentum,
                                         norm.eps, norm.activation, norm.activation_param, norm.group, "ffm", residual, a1.weight,
                                         attention[2].weight, None, None, None, True, 0.0, 0.0, addend)
        fm = norm(y)   # (not a shape for the fused kernels: the ordinary norm consumes the conv's statistics)
        return channel_attention(fm, attention, kind, residual=residual, addend=addend)
    return channel_attention(conv(x), attention, kind, residual=residual, addend=addend)


class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        from .. import _C
        ctx.c, ctx.dtype = (a.shape[1], b.shape[1]), a.dtype
        return _C.concat2(a, b)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        return _C.split2(_cl(g, ctx.dtype), *ctx.c)


def concat_channels(a, b):
    """torch.cat([a, b], dim=1) (layers.py:316).  bf16 CUDA: [HIP]"""
    from .. import _C
    if _C.elt_supported(a) and _C.elt_supported(b) and a.shape[0] == b.shape[0] and a.shape[2:] == b.shape[2:]:
        return _CatFn.apply(a, b)
    return torch.cat([a, b], dim=1)


class _Up1Fn(torch.autograd.Function):
    """[HIP] single-channel fp32 bilinear upsampling (depth head) and its separable adjoint (csrc/headloss.hip)."""

    @staticmethod
    def forward(ctx, x, scale):
        from .. import _C
        ctx.hw = x.shape[2:]
        return _C.upsample1_fwd(x.contiguous(), x.shape[2] * scale, x.shape[3] * scale)

    @staticmethod
    def backward(ctx, g):
        from .. import _C
        return _C.upsample1_bwd(g.float().contiguous(), *ctx.hw), None


def upsample_bilinear(x, scale_factor):
    """F.interpolate(bilinear, align_corners=True) (mg_net.py:599, :678-687, :804-806).
    Single-channel fp32 CUDA maps with factor >= 8 (the depth head): [HIP]; everything else: [torch-staging]."""
    if x.is_cuda and x.dtype == torch.float32 and x.shape[1] == 1 and int(scale_factor) == scale_factor and scale_factor >= 8 \
            and x.shape[2] >= 2 and x.shape[3] >= 2:
        return _Up1Fn.apply(x, int(scale_factor))
    return F.interpolate(x.float(), scale_factor=scale_factor, mode="bilinear", align_corners=True)


# ---------------------------------------------------------------------------------------------------------------
# head losses fused with the bilinear upsampling of the low-resolution head outputs
# ---------------------------------------------------------------------------------------------------------------
class LazyUpsample:
    """`F.interpolate(lr, scale_factor=scale, bilinear, align_corners=True) * mult`, not materialised: the fused loss
    kernels interpolate on the fly.  `materialize()` gives the tensor the reference would have produced."""

    def __init__(self, lr, scale, mult=1.0):
        # lr: the low-resolution map, or a PaddedMap (channel-padded predictor output read in place by the fused loss kernels)
        self.padded = lr if isinstance(lr, PaddedMap) else None
        self.lr = lr.real() if self.padded is not None else lr
        self.scale, self.mult = scale, mult

    @property
    def shape(self):
        return tuple(self.lr.shape[:2]) + (self.lr.shape[2] * self.scale, self.lr.shape[3] * self.scale)

    def materialize(self):
        y = upsample_bilinear(self.lr, self.scale)
        return y * self.mult if self.mult != 1.0 else y

    def tensors(self):
        return [self.lr] if self.padded is None else [self.lr, self.padded.t]


def materialize(x):
    return x.materialize() if isinstance(x, LazyUpsample) else x


class _UpCEFn(torch.autograd.Function):
    """[HIP] mgnet_amd/csrc/headloss.hip: x`scale` upsampling + weighted per-pixel CE + OHEM / top-k / mean selection."""

    @staticmethod
    def forward(ctx, lr, labels, weights, H, W, ignore, mode, thr, n_sel, K=None):
        from .. import _C

        ctx.padded = None
        if K is not None:      # lr is the channel-padded predictor output [B,P,h,w]: read in place, gradient returned padded
            ctx.padded = (tuple(lr.shape), lr.dtype)
            lr = lr[:, :K]
        labels = labels.contiguous()
        weights = None if weights is None else weights.float().contiguous()
        ce, sums = _C.upce_fwd(lr, labels, weights, H, W, ignore, thr if mode == "ohem" else 3.0e38)
        n_px = ce.numel()
        if mode == "mean":          # DeepLabCE(top_k=1.0): mean over ALL pixels (ignored ones count with loss 0)
            loss = sums[2] / n_px
            sel = torch.cat([sums.new_full((1,), -1.0), sums.new_zeros(1), sums.new_full((1,), 1.0 / n_px)])  # (device fills: no H2D)
        else:
            # threshold vs top-n_sel branch (loss.py:76) decided ON THE DEVICE: no host sync, no sort, no torch.topk
            sel, loss = _C.ohem_select(ce, sums, thr, n_sel, mode != "ohem")
        ctx.save_for_backward(lr, labels, weights, ce, sel.float().contiguous())
        ctx.cfg = (H, W, ignore)
        return loss

    @staticmethod
    def backward(ctx, g):
        from .. import _C

        lr, labels, weights, ce, sel = ctx.saved_tensors
        H, W, ignore = ctx.cfg
        K = lr.shape[1]
        Kp = (K + 7) // 8 * 8
        dlg = _C.upce_bwd(lr, labels, weights, H, W, ignore, ce, sel, g.float().reshape(1).contiguous(), Kp)
        if ctx.padded is not None:   # [HIP] fp32 [B,h,w,Kp] table -> padded 16-bit gradient in one launch
            B, h, w = dlg.shape[:3]
            dpad = _C.head_act_bwd(dlg, None, ctx.padded[0], K, "none", ctx.padded[1], g_strides=(h * w * Kp, 1, Kp))
            return dpad, None, None, None, None, None, None, None, None, None
        return dlg[..., :K].permute(0, 3, 1, 2).to(lr.dtype), None, None, None, None, None, None, None, None, None


def upsampled_ce(lazy, labels, weights, ignore, mode, thr=0.0, n_sel=0):
    """loss of the semantic head on a LazyUpsample of the low-res logits (mode: ohem | topk | mean)."""
    from .. import _C

    lr = lazy.lr
    H, W = lazy.shape[2:]
    if mode != "mean" and n_sel >= labels.numel():
        raise IndexError(f"index {n_sel} is out of bounds for dimension 0 with size {labels.numel()}")
    assert lazy.mult == 1.0 and _C.upce_supported(lr)
    if lazy.padded is not None:
        return _UpCEFn.apply(lazy.padded.t, labels, weights, H, W, ignore, mode, float(thr), int(n_sel), lazy.padded.C)
    return _UpCEFn.apply(lr, labels, weights, H, W, ignore, mode, float(thr), int(n_sel))


class _InsLossFn(torch.autograd.Function):
    """[HIP] centre (weighted MSE) and offset (weighted L1) losses on the fly-upsampled low-res maps."""

    @staticmethod
    def forward(ctx, center_lr, offset_lr, ct, cw, ot, ow, H, W, oscale, offset_C=None):
        from .. import _C

        ctx.padded = None
        if offset_C is not None:   # offset_lr is the channel-padded predictor output: read in place, gradient returned padded
            ctx.padded = (tuple(offset_lr.shape), offset_lr.dtype)
            offset_lr = offset_lr[:, :offset_C]
        center_lr = center_lr.float().contiguous()
        ct, cw, ot, ow = ct.float().contiguous(), cw.float().contiguous(), ot.float().contiguous(), ow.float().contiguous()
        out4 = _C.ins_loss_fwd(center_lr, offset_lr, H, W, ct, cw, ot, ow, oscale)
        ctx.save_for_backward(center_lr, offset_lr, ct, cw, ot, ow, out4)
        ctx.cfg = (H, W, oscale)
        return out4[:2].clone()

    @staticmethod
    def backward(ctx, g):
        from .. import _C

        center_lr, offset_lr, ct, cw, ot, ow, out4 = ctx.saved_tensors
        H, W, oscale = ctx.cfg
        dco = _C.ins_loss_bwd(center_lr, offset_lr, H, W, ct, cw, ot, ow, oscale, out4, g.float().contiguous())
        d_center = dco[..., 0].unsqueeze(1)
        if ctx.padded is not None:
            B, h, w, Kp = dco.shape
            d_offset = _C.head_act_bwd(dco[..., 1:], None, ctx.padded[0], 2, "none", ctx.padded[1], g_strides=(h * w * Kp, 1, Kp))
        else:
            d_offset = dco[..., 1:3].permute(0, 3, 1, 2).to(offset_lr.dtype)
        return d_center, d_offset, None, None, None, None, None, None, None, None


def ins_losses_supported(center, offset):
    from .. import _C
    return (isinstance(center, LazyUpsample) and isinstance(offset, LazyUpsample) and center.lr.is_cuda
            and center.lr.dtype == torch.float32 and _C.upce_supported(offset.lr) and center.scale == offset.scale)


def upsampled_ins_losses(center, offset, targets):
    H, W = center.shape[2:]
    if offset.padded is not None:
        return _InsLossFn.apply(center.lr, offset.padded.t, targets["center"], targets["center_weights"], targets["offset"],
                                targets["offset_weights"], H, W, float(offset.mult), offset.padded.C)
    return _InsLossFn.apply(center.lr, offset.lr, targets["center"], targets["center_weights"], targets["offset"],
                            targets["offset_weights"], H, W, float(offset.mult))


# ---------------------------------------------------------------------------------------------------------------
# uncertainty weighting of the task losses (mg_net.py:360-372)
# ---------------------------------------------------------------------------------------------------------------
class _UncertaintyFn(torch.autograd.Function):
    """[HIP] csrc/scalars.hip: weighted_k = tau_k exp(-log_vars[k]) raw_k + 0.5 log_vars[k] for the tasks k0 .. k0 + n - 1 in one launch;
    one output per task (views of one buffer), one launch for the backward -- the launches carry the pointers of the loss scalars where
    they lie.  A call covers the tasks of ONE head (k0 = its first task), so that nothing in a head's forward -> loss -> backward chain
    reads another head's results."""

    @staticmethod
    def forward(ctx, log_vars, tau_mask, k0, *raws):
        from .. import _C
        raws = [r.detach().float().reshape(()).contiguous() for r in raws]
        lv = log_vars.detach().contiguous()
        weighted, unc = _C.uncertainty_fwd(raws, lv[k0:k0 + len(raws)], tau_mask >> k0)
        ctx.raws, ctx.lv, ctx.tau_mask, ctx.k0 = raws, lv, tau_mask, k0
        ctx.mark_non_differentiable(unc)
        return (unc,) + tuple(weighted.unbind(0))

    @staticmethod
    def backward(ctx, _g_unc, *gs):
        from .. import _C
        k0, n, nt = ctx.k0, len(ctx.raws), ctx.lv.numel()
        gs = [None if g is None else g.float().reshape(()).contiguous() for g in gs]
        # one launch over ALL of log_vars: the other tasks carry no gradient here (their d log_vars entries come out as exact zeros; any
        # finite scalar stands in for their raw loss)
        n_all = min(nt, _C.MGN_MAX_TASKS)
        raws = [ctx.raws[k - k0] if k0 <= k < k0 + n else ctx.raws[0] for k in range(n_all)]
        grads = [gs[k - k0] if k0 <= k < k0 + n else None for k in range(n_all)]
        d_raw, d_lv = _C.uncertainty_bwd(raws, grads, ctx.lv, ctx.tau_mask)
        return (d_lv, None, None) + tuple(d_raw[k0:k0 + n].unbind(0))


_TAU_CACHE = {}


def uncertainty_weighting(losses, log_vars, k0=0):
    """mg_net.py:360-372 for a dict of task losses (in task order, the first one being task `k0` of log_vars): -> (weighted dict, raw
    dict, uncertainty dict) of device scalars"""
    keys = list(losses)
    if log_vars.is_cuda and k0 + len(keys) <= 8 and all(v.is_cuda for v in losses.values()) and not os.environ.get("MGN_NO_UNC_FUSE"):
        mask = sum(1 << (k0 + i) for i, k in enumerate(keys) if k == "loss_sem_seg")
        out = _UncertaintyFn.apply(log_vars, mask, k0, *[losses[k] for k in keys])
        unc, weighted = out[0], out[1:]
        return ({k: weighted[i] for i, k in enumerate(keys)}, {k: losses[k].detach() for k in keys}, {k: unc[i] for i, k in enumerate(keys)})
    raw = torch.stack([losses[k].float().reshape(()) for k in keys])
    lv = log_vars[k0:k0 + len(keys)]
    ck = (tuple(keys), str(raw.device))
    tau = _TAU_CACHE.get(ck)   # (uploaded once: a copy from pageable host memory stalls the host until the queue has drained)
    if tau is None:
        tau = _TAU_CACHE[ck] = torch.tensor([1.0 if k == "loss_sem_seg" else 0.5 for k in keys], dtype=raw.dtype).to(raw.device)
    weighted = tau * torch.exp(-lv) * raw + 0.5 * lv
    unc = torch.exp(lv.detach())
    return ({k: weighted[i] for i, k in enumerate(keys)}, {k: raw[i].detach() for i, k in enumerate(keys)}, {k: unc[i] for i, k in enumerate(keys)})

"""is the bf16 gradient deviation noise or a bug?  (a) torch ops under bf16 autocast on the GPU vs the fp32 oracle, (b) the HIP path
with every fusion switched off vs the fused HIP path, (c) HIP vs torch-autocast."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["MGNET_ALLOW_TORCH_STAGING"] = "1"
import torch
from test_network_cpu import small_model
from test_network_gpu import _randomise
from mgnet_amd.data import synthetic_batch
from oracle import network_oracle as NO
H, W = int(sys.argv[1]), int(sys.argv[2])
def build():
    cfg, m = small_model(with_depth=False, seed=3); _randomise(m); m.train(); return cfg, m
cfg, m = build()
batch = synthetic_batch(2, H, W, "cpu", seed=5, with_depth=False)
kw = dict(pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, ohem_n_min=1500, with_depth=False)
def oracle(device, autocast):
    sd = {k: v.detach().clone().to(device).requires_grad_(v.dtype.is_floating_point) for k, v in m.state_dict().items()}
    b = [{k: (v.to(device) if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch]
    import oracle.network_oracle as N
    if device != "cpu":   # the helper builds mean/std on the CPU
        _t = torch.tensor
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        torch.set_default_device(device)
        try:
            ls = NO.mgnet_losses(sd, b, **kw)
        finally:
            torch.set_default_device("cpu")
    sum(ls.values()).backward()
    return {k: v.grad.detach().double().cpu().flatten() for k, v in sd.items() if v.grad is not None}
def hip(env):
    for k in ("MGN_NO_SKIPFUSE", "MGN_NO_TAILFUSE", "MGN_NO_STEMFUSE", "MGN_NO_ATTN_FUSE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    cfg2, m2 = build(); m2 = m2.cuda(); m2.amp_dtype = torch.bfloat16
    out = m2([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch])
    sum(out.values()).backward()
    return {n: p.grad.detach().double().cpu().flatten() for n, p in m2.named_parameters()}
def cmp(a, b, what):
    cs = []
    for n in a:
        if n in b and float(b[n].norm()) > 0:
            cs.append((float((a[n] @ b[n]) / (a[n].norm() * b[n].norm() + 1e-300)), n))
    cs.sort()
    print(f"{what:46s} median cos {cs[len(cs)//2][0]:.4f}  min {cs[0][0]:.4f} ({cs[0][1]})  n<0.999: {sum(c < 0.999 for c, _ in cs)}/{len(cs)}  stem: {dict((n, round(c, 4)) for c, n in cs).get('backbone.stem.conv1.weight')}")
ref = oracle("cpu", False)
ac = oracle("cuda", True)
f32gpu = oracle("cuda", False)
fused = hip({})
unf = hip({"MGN_NO_SKIPFUSE": "1", "MGN_NO_TAILFUSE": "1", "MGN_NO_STEMFUSE": "1", "MGN_NO_ATTN_FUSE": "1"})
cmp(f32gpu, ref, "torch fp32 GPU vs oracle fp32 CPU")
cmp(ac, ref, "torch bf16 autocast GPU vs oracle fp32")
cmp(fused, ref, "HIP bf16 vs oracle fp32")
cmp(unf, ref, "HIP bf16 unfused vs oracle fp32")
cmp(fused, unf, "HIP bf16 fused vs unfused")
cmp(fused, ac, "HIP bf16 vs torch bf16 autocast")

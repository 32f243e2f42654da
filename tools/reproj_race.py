"""Race screen of the reprojection-loss kernels (csrc/reproj_loss.hip: reproj_prep / reproj_march / fin1 / fin2 / reproj_bwd4) at the
C4 size (8 frames of 1024x2048, uint8 RGBX frames): forward + backward repeated on the same operands while two other streams keep the
chip busy must give the same bits (losses, pose gradient, the three inverse-depth gradients).  Usage: reproj_race.py [reps] [BxHxW]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C
from mgnet_amd.data import synthetic_batch
from mgnet_amd.modeling import MultiViewPhotometricLoss

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B, H, W = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "8x1024x2048").split("x")]
dev = torch.device("cuda:0")
batch = synthetic_batch(B, H, W, dev, seed=1234)
frames = [x[k] for k in ("image_orig", "image_prev_orig", "image_next_orig") for x in batch]
rgbx = _C.u8_frames_to_rgbx(frames)
tg = {"image_orig": rgbx[:B], "image_prev_orig": rgbx[B:2 * B], "image_next_orig": rgbx[2 * B:],
      "camera_matrix": torch.stack([x["camera_matrix"] for x in batch]).to(dev), "reprojection_mask": torch.stack([x["reprojection_mask"] for x in batch]).unsqueeze(1)}
g = torch.Generator(device="cuda").manual_seed(7)
# DEPTH=flat: what a freshly initialised head predicts (sigmoid(~0) / 0.5 = ~1 everywhere, the warp close to the identity: adjacent lanes'
# gathers share cache lines); default: smooth random inverse depths in (0.05, 1.95) and poses 0.01 N(0, 1)
FLAT = os.environ.get("DEPTH") == "flat"
inv0 = [torch.nn.functional.interpolate((torch.rand(B, 1, H // s, W // s, device=dev, generator=g) * (0.02 if FLAT else 1.9) + (0.99 if FLAT else 0.05)),
                                        size=(H, W), mode="bilinear", align_corners=True) for s in (8, 16, 32)]
poses0 = (0.0005 if FLAT else 0.01) * torch.randn(B, 2, 6, device=dev, generator=g)
crit = MultiViewPhotometricLoss(0.85, 1.0, 0.001, True, "min", "zeros")
side = [torch.cuda.Stream() for _ in range(2)]
big = torch.randn(64 << 20, device=dev)
mm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)


# BUSY=conv: the side streams run the windowed 3x3 kernels of the heads (what shares the chip with this kernel inside the training step)
cx = [torch.randn(8, c, 128, 256, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for c in (256, 128)]
cw = [(torch.randn(c, 3, 3, c, device=dev, generator=g) * 0.05).to(torch.bfloat16).contiguous() for c in (256, 128)]


def once():
    inv = [x.clone().requires_grad_(True) for x in inv0]
    poses = poses0.clone().requires_grad_(True)
    out = crit({"depth": inv, "poses": poses}, tg)
    (out["loss_photometric"] + out["loss_smoothness"]).backward()
    return [out["loss_photometric"].detach().clone(), out["loss_smoothness"].detach().clone(), poses.grad.clone()] + [x.grad.clone() for x in inv]


names = ["loss_photometric", "loss_smoothness", "d_pose", "d_inv0", "d_inv1", "d_inv2"]
ref = once()
torch.cuda.synchronize()
bad = gross = 0
for r in range(REPS):
    if os.environ.get("BUSY", "1") == "1":
        for st in side:
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                big.mul_(1.0001)
                torch.mm(mm, mm)
    elif os.environ.get("BUSY") == "conv":
        for k, st in enumerate(side):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(12):
                    _C.conv3x3_win(cx[k], cw[k], patch_rows=8)
    cur = once()
    torch.cuda.synchronize()
    d = [f"{n}: {int((a != b).sum())} elements" + (f" ({float(a):.9g} / {float(b):.9g})" if a.dim() == 0 else "") for n, a, b in zip(names, ref, cur) if not torch.equal(a, b)]
    bad += bool(d)
    # "grossly": a loss off by more than 1e-6 relative, or more than 1e-5 of a gradient's elements different (a race inside the row march
    # shifts whole strips; a last-bit flip of a few elements is the box, profiles/r06_determinism.txt)
    g = any((a.dim() == 0 and abs(float(a) - float(b)) > 1e-6 * abs(float(a))) or (a.dim() > 0 and int((a != b).sum()) > 1e-5 * a.numel())
            for a, b in zip(ref, cur))
    gross += g
    if d:
        print(f"[rep {r}] " + " | ".join(d) + (" -- GROSS" if g else ""), flush=True)
print(f"reprojection loss {B}x{H}x{W}: {bad} of {REPS} evaluations differ from the first, {gross} of {REPS} evaluations differ grossly   "
      f"lib={os.environ.get('MGNET_HIP_LIB', 'in-tree')}")

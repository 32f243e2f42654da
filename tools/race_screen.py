"""Race screen of the convolution kernels at the shapes of the C4 training step (8 frames of 1024x2048): every kernel family -- forward,
data gradient (windowed / stride-2 window / generic), weight gradient (3x3 row march, split-K tiles, stems), 64-channel row march,
streaming 1x1 -- is launched repeatedly on the SAME operands while two other streams keep the memory system and the matrix pipes busy,
and must return the same bits every time.  Counted-`vmcnt` LDS-DMA pipelines fail this way when a wait retires too little (round 6:
csrc/conv_win.hip, compiler-merged dummy loads).  Usage: race_screen.py [reps]   (MGNET_HIP_LIB=<.so> selects a build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mgnet_amd import _C

dev = torch.device("cuda:0")
g = torch.Generator(device="cuda").manual_seed(3)


def cl(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev, generator=g) * scale).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


def wl(cout, k, cin, scale=0.05):
    return (torch.randn(cout, k, k, cin, device=dev, generator=g) * scale).to(torch.bfloat16).contiguous()


def cases(B=8, div=1):
    """div: every spatial size divided by it (1: the C4 frame 1024 x 2048; 2: C2's 512 x 1024; 4: C1's 256 x 512)"""
    out = []
    FH, FW = 1024 // div, 2048 // div
    # forward / data gradient of the 3x3 stride-1 layers (windowed kernels, 64-channel row march, generic)
    for (cin, cout, h, w) in [(128, 128, 128, 256), (256, 256, 128, 256), (128, 128, 64, 128), (256, 256, 64, 128), (512, 512, 32, 64), (128, 256, 32, 64),
                              (512, 128, 32, 64), (64, 64, 256, 512), (256, 128, 64, 128)]:
        h, w = h // div, w // div
        x, wt = cl(B, cin, h, w), wl(cout, 3, cin)
        out.append((f"conv3x3 s1 {cin}->{cout} @{h}x{w}", lambda x=x, wt=wt, h=h, w=w: _C.conv_igemm(x, wt, (h, w), None, 1, 1),
                    lambda x=x, wt=wt: _ref_conv(x, wt, 1, 1)))
        shift = torch.zeros(cout, device=dev)
        out.append((f"conv3x3 s1 {cin}->{cout} @{h}x{w} + statistics rows",
                    lambda x=x, wt=wt, h=h, w=w, shift=shift: _stats(x, wt, (h, w), 1, 1, shift), None))
    # stride-2 forward (3x3 and the 1x1 shortcut) and their data gradients
    for (cin, cout, h, w) in [(64, 128, 256, 512), (128, 256, 128, 256), (256, 512, 64, 128)]:
        h, w = h // div, w // div
        x, w3, w1 = cl(B, cin, h, w), wl(cout, 3, cin), wl(cout, 1, cin)
        out.append((f"conv3x3 s2 {cin}->{cout} @{h}x{w}", lambda x=x, w3=w3, h=h, w=w: _C.conv_igemm(x, w3, (h // 2, w // 2), None, 2, 1),
                    lambda x=x, w3=w3: _ref_conv(x, w3, 2, 1)))
        out.append((f"conv1x1 s2 {cin}->{cout} @{h}x{w}", lambda x=x, w1=w1, h=h, w=w: _C.conv_igemm(x, w1, (h // 2, w // 2), None, 2, 0),
                    lambda x=x, w1=w1: _ref_conv(x, w1, 2, 0)))
        dy = cl(B, cout, h // 2, w // 2)
        wi = wl(cin, 3, cout)   # layout mode 1: [Cin_fwd][kh][kw][Cout_fwd]
        res = cl(B, cin, h, w)
        out.append((f"dgrad 3x3 s2 {cout}->{cin} to {h}x{w} (conv_up2 + residual)", lambda dy=dy, wi=wi, h=h, w=w, res=res: _up2(dy, wi, (h, w), res),
                    lambda dy=dy, wi=wi, res=res, h=h, w=w: _ref_dgrad_s2(dy, wi, res, (h, w))))
    # 1x1 layers (streaming kernel / generic)
    for (cin, cout, h, w) in [(256, 256, 128, 256), (256, 32, 128, 256), (32, 256, 128, 256), (512, 256, 32, 64), (128, 64, 128, 256)]:
        h, w = h // div, w // div
        x, w1 = cl(B, cin, h, w), wl(cout, 1, cin)
        out.append((f"conv1x1 {cin}->{cout} @{h}x{w}", lambda x=x, w1=w1, h=h, w=w: _C.conv_igemm(x, w1, (h, w), None, 1, 0),
                    lambda x=x, w1=w1: _ref_conv(x, w1, 1, 0)))
    # stems
    x4, x16 = cl(B, 4, FH, FW), cl(B, 16, FH, FW)
    w4 = (torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.05)
    w9 = (torch.randn(64, 9, 7, 7, device=dev, generator=g) * 0.05)
    out.append(("stem 7x7 s2 3(4)->64", lambda: _C.conv_igemm(x4, _C.weight_layout(w4, 2, 4, dtype=torch.bfloat16), (FH // 2, FW // 2), None, 2, 3, khw=(7, 7)),
                lambda: torch.nn.functional.conv2d(x4[:NREF, :3].float(), w4.to(torch.bfloat16).float(), stride=2, padding=3)))
    out.append(("stem 7x7 s2 9(16)->64", lambda: _C.conv_igemm(x16, _C.weight_layout(w9, 2, 16, dtype=torch.bfloat16), (FH // 2, FW // 2), None, 2, 3, khw=(7, 7)),
                lambda: torch.nn.functional.conv2d(x16[:NREF, :9].float(), w9.to(torch.bfloat16).float(), stride=2, padding=3)))
    dys = cl(B, 64, FH // 2, FW // 2)
    out.append(("wgrad stem 7x7 s2 (4-channel pixels)", lambda: _C.conv_wgrad(dys, x4, 7, 7, 2, 3, cin_real=3),
                lambda: _ref_wgrad(x4[:, :3], dys, 7, 2, 3)))
    out.append(("wgrad stem 7x7 s2 (16-channel pixels)", lambda: _C.conv_wgrad(dys, x16, 7, 7, 2, 3, cin_real=9),
                lambda: _ref_wgrad(x16[:, :9], dys, 7, 2, 3)))
    # weight gradients
    for (cin, cout, h, w, k, s) in [(64, 64, 256, 512, 3, 1), (128, 128, 128, 256, 3, 1), (256, 256, 128, 256, 3, 1), (512, 512, 32, 64, 3, 1),
                                    (64, 128, 256, 512, 3, 2), (128, 256, 128, 256, 3, 2), (256, 256, 128, 256, 1, 1), (256, 32, 128, 256, 1, 1),
                                    (64, 128, 256, 512, 1, 2), (512, 128, 32, 64, 3, 1)]:
        h, w = h // div, w // div
        x, dy = cl(B, cin, h, w), cl(B, cout, h // s, w // s)
        out.append((f"wgrad {k}x{k} s{s} {cin}->{cout} @{h}x{w}", lambda x=x, dy=dy, k=k, s=s: _C.conv_wgrad(dy, x, k, k, s, k // 2),
                    lambda x=x, dy=dy, k=k, s=s: _ref_wgrad(x, dy, k, s, k // 2)))
    return out


NREF = 2   # images of the batch the fp32 reference of a forward / data-gradient case covers (the weight gradients cover all of them)


def _ref_conv(x, w_ohwi, stride, pad):
    """fp32 torch convolution of the first NREF images on the same 16-bit operands (the yardstick: MIOpen fp32)"""
    return torch.nn.functional.conv2d(x[:NREF].float(), w_ohwi.float().permute(0, 3, 1, 2).contiguous(), stride=stride, padding=pad)


def _ref_dgrad_s2(dy, w_ihwo, res, hw):
    # layout mode 1 = [Cin_fwd][kh][kw][Cout_fwd] with the taps flipped: the forward weight is w[co][ci][KH-1-kh][KW-1-kw]
    wf = w_ihwo.float().flip(1, 2).permute(3, 0, 1, 2).contiguous()          # [Cout_fwd, Cin_fwd, kh, kw]
    g = torch.nn.grad.conv2d_input((NREF, wf.shape[1]) + tuple(hw), wf, dy[:NREF].float(), stride=2, padding=1)
    return g + res[:NREF].float()


def _ref_wgrad(x, dy, k, stride, pad):
    return torch.nn.grad.conv2d_weight(x.float(), (dy.shape[1], x.shape[1], k, k), dy.float(), stride=stride, padding=pad)


def _stats(x, wt, hw, stride, pad, shift):
    holder = []
    y = _C.conv_igemm(x, wt, hw, None, stride, pad, stats=(shift, holder))
    return torch.cat([y.float().flatten()[:: 64], holder[0][0].flatten()]) if holder else y


def _up2(dy, wi, hw, res):
    y = _C.conv_up2(dy, wi, hw, residual=res)
    return y if y is not None else _C.conv_igemm(dy, wi, hw, None, 1, 1, up=2, residual=res)


def _gross(y, ref):
    """elements of y that differ from ref by more than rounding noise: > 2 ulps of the 16-bit format (> 1e-5 relative for fp32 outputs).
    A pipeline race gives wrong TILES (errors of order one in thousands of elements); one box of round 6 flipped the last bit of a handful
    of elements of nearly every kernel under load (profiles/r06_determinism.txt), which is not what this screen is for."""
    yf, rf = y.float(), ref.float()
    tol = (2.0 ** -7 if y.dtype in (torch.bfloat16, torch.float16) else 1e-5) * torch.maximum(yf.abs(), rf.abs()) + 1e-30
    return int(((yf - rf).abs() > tol).sum())


def screen(reps=8, busy=True, B=8, verbose=True, strict=True, div=1):
    side = [torch.cuda.Stream() for _ in range(2)]
    big = torch.randn(64 << 20, device=dev)
    mm = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    failures = []
    for name, fn, _ref in cases(B, div):
        ref = fn()
        torch.cuda.synchronize()
        nbad, worst = 0, 0
        for r in range(reps):
            if busy:
                for st in side:
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        big.mul_(1.0001)
                        torch.mm(mm, mm)
            y = fn()
            n = int((y != ref).sum())
            if n and not strict:   # (tests: last-bit flips of a few elements are reported, not failed)
                g = _gross(y, ref)
                if g == 0 and n <= 1e-4 * y.numel():
                    if verbose:
                        print(f"{name}: {n} elements differ in the last bits (tolerated)", flush=True)
                    n = 0
            nbad += n > 0
            worst = max(worst, n)
        torch.cuda.synchronize()
        if nbad:
            failures.append((name, nbad, worst))
        if verbose:
            print(f"{name:62s} {'ok' if not nbad else f'{nbad} of {reps} launches differ (up to {worst} elements)'}", flush=True)
    return failures


def parity(B=8, verbose=True, tol=6e-3, div=1):
    """every case against an fp32 torch evaluation on the same 16-bit operands, at the full C4 shapes (first NREF images for the forward /
    data-gradient cases): max |difference| / max |reference| <= tol (16-bit outputs: 2^-8 rounding; fp32 weight gradients far below)"""
    failures = []
    for name, fn, ref in cases(B, div):
        if ref is None:
            continue
        y, r = fn(), ref()
        y = y[:r.shape[0]] if y.dim() == 4 and y.shape[0] != r.shape[0] and y.shape[1:] == r.shape[1:] else y
        assert y.shape == r.shape, (name, y.shape, r.shape)
        err = float((y.float() - r).abs().max() / r.abs().max().clamp_min(1e-20))
        if verbose:
            print(f"{name:62s} max |d| / max |ref| = {err:.2e}", flush=True)
        if not err <= tol:
            failures.append((name, err))
        del y, r
    return failures


if __name__ == "__main__":
    DIV = int(os.environ.get("DIV", 1))
    if "--parity" in sys.argv:
        f = parity(div=DIV)
        print(f"cases beyond the tolerance: {f}")
        sys.exit(1 if f else 0)
    f = screen(int(sys.argv[1]) if len(sys.argv) > 1 else 8, busy=os.environ.get("BUSY", "1") == "1", div=DIV)
    print(f"kernels that are not bit-reproducible under load: {len(f)}   lib={os.environ.get('MGNET_HIP_LIB', 'in-tree')}")

"""Host-side check of the step table of csrc/conv_up2.hip (no GPU): the nine (parity class, window shift, tap) triples against torch."""
def test_stride2_data_gradient_step_table_is_the_transposed_convolution():
    """csrc/conv_up2.hip computes the data gradient of a 3x3 / stride 2 / pad 1 conv as nine (parity class, window shift, flipped tap)
    MFMA steps on the low-resolution gradient.  The table is read from the kernel source and evaluated with plain tensor algebra
    against torch's conv2d_input: a wrong tap, shift or class in the table fails here, without a GPU."""
    import os
    import re
    import torch
    from conftest import ROOT
    src = open(os.path.join(ROOT, "mgnet_amd", "csrc", "conv_up2.hip")).read()
    body = re.search(r"STEPS\[9\] = \{(.*?)\};", src, re.S).group(1)
    steps = [tuple(int(v) for v in m) for m in re.findall(r"\{(\d+), (\d+), (\d+), (\d+)\}", body)]
    assert len(steps) == 9 and sorted(s[3] for s in steps) == list(range(9))
    torch.manual_seed(0)
    N, Cin, Cout, H, W = 2, 3, 4, 9, 12                     # forward conv: Cin -> Cout, input H x W (odd and even extents)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    w = torch.randn(Cout, Cin, 3, 3, dtype=torch.float64)
    dy = torch.randn(N, Cout, OH, OW, dtype=torch.float64)
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), w, dy, stride=2, padding=1)
    wd = w.flip(2, 3).permute(1, 2, 3, 0)                   # mgn_weight_layout mode 1: [Cin][kh'][kw'][Cout], taps flipped
    dyp = torch.nn.functional.pad(dy, (0, 1, 0, 1))        # the window's halo: out-of-range low-resolution pixels read zero
    got = torch.zeros(N, Cin, 2 * OH, 2 * OW, dtype=torch.float64)
    for cls, sy, sx, tap in steps:
        a, b = cls >> 1, cls & 1
        contrib = torch.einsum("nohw,io->nihw", dyp[:, :, sy:sy + OH, sx:sx + OW], wd[:, tap // 3, tap % 3, :])
        got[:, :, a::2, b::2] += contrib
    assert torch.allclose(got[:, :, :H, :W], ref, atol=1e-12)

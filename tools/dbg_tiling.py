import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_reproj_gpu import run_hip
B, H, W = 1, 128, 512
rs_ = np.random.RandomState(11)
K = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1)); K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 0.6 * W, 1.9 * H, 0.5 * W, 0.5 * H
c = dict(inv=[rs_.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)],
         img=rs_.uniform(0, 1, (B, 3, H, W)).astype(np.float32), prev=rs_.uniform(0, 1, (B, 3, H, W)).astype(np.float32),
         nxt=rs_.uniform(0, 1, (B, 3, H, W)).astype(np.float32), poses=(0.01 * rs_.randn(B, 2, 6)).astype(np.float32), mask=None, K=K)
a = run_hip(c, rows_per_wave=128, g=(1.0, 0.0))
for rows in (64, 32, 16, 8):
    b = run_hip(c, rows_per_wave=rows, g=(1.0, 0.0))
    print("rows", rows, "losses", a["losses"], b["losses"])
    for i in range(3):
        d = np.abs(a["d_inv"][i] - b["d_inv"][i])[0, 0]
        bad = d > 1e-5 * np.abs(a["d_inv"][i]).max()
        rr = np.nonzero(bad.any(1))[0]
        cc = np.nonzero(bad.any(0))[0]
        print("  scale", i, "bad px", int(bad.sum()), "rows", rr[:24], "cols", cc[:8], "...", cc[-4:] if len(cc) else "")

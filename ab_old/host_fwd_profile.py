#!/usr/bin/env python3
"""cProfile of MGNet.forward (training) alone, host side: top functions by own time and by cumulative time."""
import cProfile, os, pstats, sys, time, io
sys.path.insert(0, os.getcwd())
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model

B, H, W = 2, 512, 1024
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join("configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
model = build_model(cfg); tr = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
for _ in range(4):
    tr.run_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
n = 5
t_f = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    tr.model.train(); tr.reducer.zero_grad()
    with tr.storage:
        t0 = time.perf_counter()
        pr.enable()
        ld = tr.model(batch)
        pr.disable()
        t_f += time.perf_counter() - t0
        tr._backward(ld)
    tr.reducer.finish(); tr.optimizer.step()
print(f"forward host time {t_f / n * 1e3:.2f} ms (profiled)")
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(28)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])

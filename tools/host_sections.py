#!/usr/bin/env python3
"""Host time of the sections of a training step (forward / backward / reducer.finish / optimizer) with an empty GPU queue in front of each
step, and the GPU time of the step: finds host-side stalls (a hidden synchronisation shows up as a section whose host time tracks the GPU's)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model

B, H, W = int(os.environ.get("B", 2)), int(os.environ.get("H", 512)), int(os.environ.get("W", 1024))
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join("configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
model = build_model(cfg); tr = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
import gc
for _ in range(4):
    tr.run_step(batch)
torch.cuda.synchronize()
mode = os.environ.get("GCMODE", "default")
if mode == "off":
    gc.disable()
elif mode == "freeze":
    gc.collect(); gc.freeze()
print("gc mode", mode, "counts", gc.get_count(), "stats", [s["collections"] for s in gc.get_stats()], "objects", len(gc.get_objects()))
acc = {}
n = 6
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.model.train(); tr.reducer.zero_grad()
    with tr.storage:
        t1 = time.perf_counter()
        ld = tr.model(batch)
        t2 = time.perf_counter()
        tr._backward(ld)
        t3 = time.perf_counter()
    tr.reducer.finish()
    t4 = time.perf_counter()
    tr.optimizer.step(); tr.scheduler.step()
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    for k, v in (("zero", t1 - t0), ("forward", t2 - t1), ("backward", t3 - t2), ("finish", t4 - t3), ("optimizer", t5 - t4), ("issue", t5 - t0), ("total", t6 - t0)):
        acc[k] = acc.get(k, 0.0) + v / n
print("gc stats after", [s["collections"] for s in gc.get_stats()], "objects", len(gc.get_objects()))
print(f"B={B} {H}x{W}: " + "  ".join(f"{k} {v * 1e3:.2f}" for k, v in acc.items()) + " ms")

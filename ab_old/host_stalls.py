#!/usr/bin/env python3
"""Which HIP runtime calls the host spends its time in during a training step (torch.profiler, CPU side): a hidden synchronisation shows
up as hipStreamSynchronize / hipEventSynchronize / hipMemcpy* / hipMalloc / hipFree with milliseconds of duration."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
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
for i in range(6):
    tr.run_step(batch)
    torch.cuda.synchronize()
    st = torch.cuda.memory_stats()
    print(f"step {i}: reserved {st['reserved_bytes.all.current'] / 2**30:.2f} GiB  allocated {st['allocated_bytes.all.current'] / 2**30:.2f} GiB  "
          f"device allocs {st['num_device_alloc']}  frees {st['num_device_free']}  retries {st['num_alloc_retries']}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        tr.run_step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith(("hip", "cuda")) and e.self_cpu_time_total > 0]
rows.sort(key=lambda e: -e.self_cpu_time_total)
for e in rows[:25]:
    print(f"{e.self_cpu_time_total / 3e3:9.3f} ms/step  {e.count / 3:8.1f} calls/step  {e.key}")
slow = sorted((ev for ev in prof.events() if ev.name.startswith(("hip", "cuda")) and ev.cpu_time_total > 500), key=lambda ev: -ev.cpu_time_total)
for ev in slow[:20]:
    print(f"slow call: {ev.name} {ev.cpu_time_total / 1e3:.2f} ms")

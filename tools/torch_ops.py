#!/usr/bin/env python3
"""Which torch (non-HIP-library) ops are still in the training step: torch.profiler table by op with stacks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, H, W = 8, 1024, 2048
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
model = build_model(cfg); trainer = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
for _ in range(3):
    trainer.run_step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(2):
        trainer.run_step(batch)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12) if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows) / 2e3
print(f"# torch (aten) ops with device time in one training step: {tot:.3f} ms/step, {sum(e.count for e in rows) // 2} calls/step")
for e in rows[:int(os.environ.get("TOP", "80"))]:
    where = [f.split("/")[-1] for f in (e.stack or []) if "mgnet_amd" in f or "bench" in f][:3]
    print(f"{e.self_device_time_total / 2e3:8.3f} ms/step  {e.count / 2:5.1f} calls  {e.key:26s} {str(e.input_shapes)[:90]:90s} {' <- '.join(where)}")

"""Is one training step (forward + backward, no optimizer step) bit-reproducible from identical state?  Prints the losses and the
parameter gradients that differ between repeated evaluations, in registration order (to localise a race / atomic)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mgnet_amd import add_mgnet_config, get_cfg
from mgnet_amd.data import synthetic_batch
from mgnet_amd.engine import Trainer
from mgnet_amd.registry import build_model
B, H, W = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4x512x1024").split("x")]
dev = torch.device("cuda:0")
cfg = get_cfg(); add_mgnet_config(cfg)
cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
torch.manual_seed(0)
model = build_model(cfg); trainer = Trainer(cfg, model)
batch = synthetic_batch(B, H, W, dev, seed=1234)
for _ in range(2):
    trainer.run_step(batch)
model.train()
names = [n for n, p in model.named_parameters() if p.requires_grad]
params = dict(model.named_parameters())
def once():
    trainer.reducer.zero_grad()
    with trainer.storage:
        ld = model(batch)
        trainer._backward(ld)
    trainer.reducer.finish()
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in ld.items()}, {n: params[n].grad.detach().clone() for n in names}
ref_l, ref_g = once()
for rep in range(int(os.environ.get("DBG_REPS", "4"))):
    l, g = once()
    dl = {k: float((l[k] - ref_l[k]).abs()) for k in l if not torch.equal(l[k], ref_l[k])}
    bad = [(n, float((g[n] - ref_g[n]).abs().max()), float(ref_g[n].abs().max())) for n in names if not torch.equal(g[n], ref_g[n])]
    print(f"rep {rep}: losses differing {dl}; {len(bad)} / {len(names)} gradient tensors differ", flush=True)
    for n, d, m in bad[:12] + ([("...", 0, 0)] if len(bad) > 24 else []) + bad[-12:]:
        print(f"      {n:70s} max|d| {d:.3e}  (max|g| {m:.3e})")

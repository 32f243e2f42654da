"""bisect which part of the step breaks hipGraph capture: python tools/dbg_graph.py <stage>"""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from test_network_cpu import small_model
from mgnet_amd.engine import Trainer
from mgnet_amd.data import synthetic_batch
stage = sys.argv[1]
cfg, m = small_model(with_depth=(stage != "nodepth"), seed=1)
m = m.cuda(); m.amp_dtype = torch.bfloat16
tr = Trainer(cfg, m)
batch = synthetic_batch(2, 64, 96, "cuda", seed=2)
for _ in range(2):
    tr.run_step(batch)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
m.train()
if stage == "fwd":
    with torch.no_grad():
        with torch.cuda.graph(g):
            with tr.storage:
                out = m(batch)
elif stage == "fwd_grad":
    with torch.cuda.graph(g):
        with tr.storage:
            out = m(batch)
elif stage in ("fwdbwd", "nodepth"):
    with torch.cuda.graph(g):
        tr.reducer.zero_grad()
        with tr.storage:
            out = m(batch)
            sum(out.values()).backward()
elif stage == "bwd_finish":
    with torch.cuda.graph(g):
        tr.reducer.zero_grad()
        with tr.storage:
            out = m(batch)
            sum(out.values()).backward()
        tr.reducer.finish()
elif stage == "opt":
    tr.optimizer.prepare_step()
    with torch.cuda.graph(g):
        tr.optimizer.launch_step()
elif stage == "full":
    tr.capture_step(batch); g = tr._graph
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed OK")

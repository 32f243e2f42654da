"""GPU: the side streams of MGNet.forward (pose network | backbone, then the three heads with their losses on three streams; the
autograd engine replays each node on the stream of its forward; engine/reducer.py packs buckets whose gradients come from several
streams) must not change a single bit of the training trajectory.  The step has no order-dependent arithmetic (the bilinear-adjoint
kernels of csrc/headloss.hip sum their tile footprints in a fixed order; their float-atomic form is MGN_ADJOINT_ATOMICS=1), so
the comparison is exact."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT


def _trajectory(streams, steps, B, H, W):
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model

    os.environ["MGNET_STREAMS"] = streams
    dev = torch.device("cuda:0")
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", B * H * W // 4 - 1])
    torch.manual_seed(0)
    model = build_model(cfg)
    trainer = Trainer(cfg, model)
    batch = synthetic_batch(B, H, W, dev, seed=77)
    losses = []
    for _ in range(steps):
        out = trainer.run_step(batch)
        losses.append(torch.stack([out[k].detach().float().reshape(()) for k in sorted(out)]).clone())
    torch.cuda.synchronize()
    used = getattr(model, "_streams", None) is not None
    return torch.stack(losses).cpu(), [p.detach().clone().cpu() for p in model.parameters()], used


def test_side_streams_do_not_change_the_trajectory():
    try:
        l0, p0, used0 = _trajectory("0", 4, 2, 128, 256)
        l1, p1, used1 = _trajectory("1", 4, 2, 128, 256)
        l2, p2, _ = _trajectory("1", 4, 2, 128, 256)
    finally:
        os.environ.pop("MGNET_STREAMS", None)
    assert used1 and not used0
    assert torch.isfinite(l0).all()
    assert torch.equal(l1, l2) and all(torch.equal(a, b) for a, b in zip(p1, p2)), "side streams: not reproducible run to run"
    assert torch.equal(l0, l1), (l0, l1)
    assert all(torch.equal(a, b) for a, b in zip(p0, p1)), "parameters after 4 steps differ between one stream and three"

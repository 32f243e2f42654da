"""GPU: the product MGNet (HIP reprojection loss + the network ops as they currently run on the device) against
oracle/network_oracle.py (CPU fp32) on identical weights and batch."""
import numpy as np
import pytest
import torch

from test_network_cpu import small_model

pytestmark = pytest.mark.gpu


def _randomise(m):
    with torch.no_grad():
        for mod in m.modules():
            if type(mod).__name__ == "InPlaceABNSync":
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
        m.log_vars.uniform_(-0.3, 0.3)


@pytest.mark.parametrize("amp", [False, True])
def test_full_step_losses_match_oracle(amp, torch_staging):
    from mgnet_amd.data import synthetic_batch
    from oracle import network_oracle as NO

    cfg, m = small_model(with_depth=True, seed=3)
    _randomise(m)
    m.train()
    batch = synthetic_batch(2, 64, 96, "cpu", seed=5)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    ref = NO.mgnet_losses(sd, batch, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, ohem_n_min=1500)
    m = m.cuda()
    m.amp_dtype = torch.bfloat16 if amp else None
    got = m([{k: (v.cuda() if torch.is_tensor(v) else v) for k, v in x.items()} for x in batch])
    sum(got.values()).backward()
    assert list(got) == ["loss_sem_seg", "loss_center", "loss_offset", "loss_photometric", "loss_smoothness"]
    tol = 3e-2 if amp else 1e-3  # fp32 rel 1e-4 per block (accumulated over ~70 layers), bf16 rel 2e-2
    for k in ref:
        assert float(got[k]) == pytest.approx(float(ref[k]), rel=tol, abs=1e-4), (k, float(got[k]), float(ref[k]))
    for n, p in m.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_training_reduces_loss_on_fixed_batch(torch_staging):
    """A few optimizer steps on one fixed synthetic batch: the total loss must go down (Adam, clip 0.01, poly LR)."""
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer

    cfg, m = small_model(with_depth=True, seed=1)
    cfg.defrost() if cfg.is_frozen() else None
    m = m.cuda()
    tr = Trainer(cfg, m)
    batch = synthetic_batch(2, 64, 96, "cuda", seed=2)
    first = last = None
    for it in range(12):
        out = tr.run_step(batch)
        tot = float(sum(v.detach() for v in out.values()))
        first = tot if first is None else first
        last = tot
    assert np.isfinite(last) and last < first, (first, last)


def test_trainer_backward_takes_any_loss_dict_a_sum_would():
    """Trainer._backward starts the backward pass from the task losses themselves (one root per loss).  A registered model may
    also return a term without a gradient (a constant / detached diagnostic) or a shape-[1] loss: both worked with `sum(losses).backward()`
    and still do -- gradients equal the summed root's."""
    from mgnet_amd.engine import Trainer

    x = torch.randn(64, device="cuda")
    grads = []
    for variant in ("scalars", "with_detached", "shape_1"):
        w = torch.nn.Parameter(torch.linspace(-1, 1, 64, device="cuda"))
        t = Trainer.__new__(Trainer)
        t.optimizer = type("O", (), {})()
        t.model = None
        t.reducer = type("R", (), {"lazy_wgrad": lambda self, on: None, "abort": lambda self: None})()
        a, b = (w * x).sum(), (w * w).mean()
        if variant == "scalars":
            ld = {"a": a, "b": b}
        elif variant == "with_detached":
            ld = {"a": a, "b": b, "diag": (w * 3).sum().detach()}
        else:
            ld = {"a": a.reshape(1), "b": b.reshape(1)}
        t._backward(ld)
        grads.append(w.grad.clone())
    ref = x + 2 * torch.linspace(-1, 1, 64, device="cuda") / 64
    for g in grads:
        assert torch.allclose(g, ref, rtol=1e-6, atol=1e-7)

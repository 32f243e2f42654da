"""The full MGNet training step against tests/golden/model_step.npz, which was produced by the reference's own
mg_net.py / layers.py / res_net.py / loss.py (tests/golden/make_golden_model.py; third-party names on stand-ins):
loss dictionary (incl. the uncertainty weighting and its task order), selected gradients and per-submodule gradient norms.
Checked: the oracle (`oracle.network_oracle.mgnet_losses`), the product's host mirror on the CPU (strict state-dict keys)
and the product's HIP path on the GPU (bf16)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "golden", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


GN = _load("make_golden_network")
GM = _load("make_golden_model")
Z = np.load(os.path.join(HERE, "golden", "model_step.npz"))
ORDER = [str(k) for k in Z["loss_order"]]


def _cfg(device, amp):
    cfg = GM.make_cfg()
    cfg.merge_from_list(["MODEL.DEVICE", device, "SOLVER.AMP.ENABLED", amp])
    return cfg


def _model(device, amp):
    from mgnet_amd.registry import build_model
    torch.manual_seed(0)
    m = build_model(_cfg(device, amp))
    assert sorted(m.state_dict().keys()) == [str(k) for k in Z["keys"]], "state-dict keys differ from the reference model's"
    GN.fill_state(m, seed=2)
    return m.train()


def test_product_host_mirror_full_step_matches_reference_cpu():
    from oracle import network_oracle as O

    m = _model("cpu", False)

    class _OracleLoss(torch.nn.Module):   # no GPU here: the photometric loss (HIP only in the product) goes through the
        def forward(self, pred, tgt):     # pinned C oracle for this host-logic test, as in tests/test_network_cpu.py
            r = O._ReprojOracle.apply(tgt["image_orig"], tgt["image_prev_orig"], tgt["image_next_orig"],
                                      tgt["reprojection_mask"], tgt["camera_matrix"], pred["poses"], *pred["depth"])
            return {"loss_photometric": r[0], "loss_smoothness": r[1]}

    m.depth_head.loss = _OracleLoss()
    losses = m(GM.arrays_to_batch(Z))
    assert list(losses.keys()) == ORDER
    for k in ORDER:
        assert float(losses[k].detach()) == pytest.approx(float(Z["loss." + k]), rel=2e-3, abs=1e-5), k
    sum(losses.values()).backward()
    g = GM.grad_summary(m)
    assert np.allclose(g["g.log_vars"], Z["g.log_vars"], rtol=2e-3, atol=1e-5)
    assert np.allclose(g["g.pose_conv4_bias"], Z["g.pose_conv4_bias"], rtol=2e-2, atol=2e-6)
    for t in GM.TOP:
        assert float(g["gn." + t]) == pytest.approx(float(Z["gn." + t]), rel=2e-2), t


def test_network_oracle_full_step_matches_reference():
    from oracle import network_oracle as O

    cfg = _cfg("cpu", False)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in GN.fill_state(_model("cpu", False), seed=2).items()}
    losses = O.mgnet_losses(sd, GM.arrays_to_batch(Z), pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD,
                            ohem_threshold=cfg.MODEL.SEM_SEG_HEAD.OHEM_THRESHOLD, ohem_n_min=cfg.MODEL.SEM_SEG_HEAD.OHEM_N_MIN)
    assert list(losses.keys()) == ORDER
    for k in ORDER:
        assert float(losses[k].detach()) == pytest.approx(float(Z["loss." + k]), rel=2e-3, abs=1e-5), k
    sum(losses.values()).backward()
    assert np.allclose(sd["log_vars"].grad.numpy(), Z["g.log_vars"], rtol=2e-3, atol=1e-5)
    assert np.allclose(sd["pose_net.conv4.bias"].grad.numpy(), Z["g.pose_conv4_bias"], rtol=2e-2, atol=2e-6)


@pytest.mark.gpu
def test_product_hip_full_step_matches_reference():
    m = _model("cuda", True)
    losses = m(GM.arrays_to_batch(Z, "cuda"))
    assert list(losses.keys()) == ORDER
    # bf16 activations through 2-sample batch statistics at the 1x1 / 2x4 layers of this tiny input: loose bound
    for k in ORDER:
        assert float(losses[k].detach()) == pytest.approx(float(Z["loss." + k]), rel=6e-2, abs=2e-3), k
    sum(losses.values()).backward()
    g = GM.grad_summary(m)
    assert np.allclose(g["g.log_vars"], Z["g.log_vars"], rtol=6e-2, atol=2e-3)

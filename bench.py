#!/usr/bin/env python3
"""bench.py -- throughput of the MGNet training hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no torch.distributed environment the process
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (before anything touches the GPU) and
exits with its code; under a launcher (RANK/WORLD_SIZE set) it is one rank.  A step = one full MGNet training step
(2x ResNet-18 + 3 decoders/heads forward + backward, five losses incl. the photometric reprojection loss, gradient
all-reduce, clip, Adam) over one per-GPU batch of synthetic Cityscapes-shaped input that is already resident in HBM.
Rank 0 prints ONE JSON line; `config.workload` says exactly what is timed.  `--loss-only` / `--fwd-only` are diagnostics
that time the reprojection loss alone (not the benchmark).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
FWD_BYTES_PER_PX = 49 + 12     # march kernel on the reference's fp32 frames: reads 3 inv + 9 image floats + mask, writes 3 grad floats
BWD_BYTES_PER_PX = 37 + 12     # streaming backward: 3 inv + 3 img + mask + 3 g (read) ; 3 d_inv (write)
FWD_BYTES_PER_PX_U8 = 25 + 12  # ... on uint8 RGBX frames (frame_layout 2, what the training step uses): 3 inv floats + 3 packed pixels + mask
BWD_BYTES_PER_PX_U8 = 29 + 12


def reproj_roofline(kern_ms, npx, u8, traffic, extra=None):
    """roofline object of reproj_march<true>.  `achieved` counts the ALGORITHMIC bytes of the kernel that ran -- with uint8 RGBX frames
    that is 37 B/px, not the 61 B/px of the reference's fp32 tensors; the 61 B/px equivalent (comparable with the
    round-1/2 lines, which ran the fp32 layout) is reported next to it."""
    bpp = FWD_BYTES_PER_PX_U8 if u8 else FWD_BYTES_PER_PX
    achieved = bpp * npx / (kern_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": "reproj_march<true, %s> (fused reprojection loss + photometric gradient)" % ("uint8 RGBX frames" if u8 else "fp32 planar frames"),
         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
         "traffic": traffic, "traffic_source": "profiles/traffic.json: PMC measurement (FETCH_SIZE + WRITE_SIZE, calibrated in the same pass) of this kernel "
                                                 "in this frame layout on the current csrc/reproj_loss.hip (re-measured in round 6: profiles/r06_traffic.json), not re-measured in this run", "bytes_per_px": bpp, "bytes_per_launch": bpp * npx, "avg_launch_ms": round(kern_ms, 4),
         "achieved_at_61_B_per_px": round(FWD_BYTES_PER_PX * npx / (kern_ms * 1e-3) / 1e9, 1),
         "frac_at_61_B_per_px": round(FWD_BYTES_PER_PX * npx / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "limiter": "VALU issue + vector-memory instruction rate, not HBM (DESIGN.md 2.4: counted)"}
    r.update(extra or {})
    return r


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0   # wave64 instructions per second: 1024 SIMD-32 units, two cycles per wave instruction, 2.4 GHz


def valu_ceiling(kern_ms, tj):
    """the ceiling that binds reproj_march: vector-ALU issue.  Wave-instruction count of one launch from the PMC pass in
    profiles/traffic.json (a static property of the kernel on this workload shape), time live: `valu_frac` = issued wave instructions per
    second / (1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction) -- a lower bound of the pipe's occupancy (transcendentals, DPP and
    64-bit operations take more than one pass, and the sustained clock is below 2.4 GHz); `valu_active_frac` is the counted share."""
    v = (tj or {}).get("valu")
    if not v:
        return {}
    rate = v["wave_instructions_per_launch"] / (kern_ms * 1e-3)
    return {"valu_frac": round(rate / VALU_ISSUE_PEAK, 4), "valu_wave_instructions_per_launch": v["wave_instructions_per_launch"],
            "valu_lane_instructions_per_px": v["lane_instructions_per_px"], "valu_issue_peak_per_s": VALU_ISSUE_PEAK,
            "valu_active_frac": v.get("active_frac"), "valu_source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE, tools/pmc_traffic2.sh)"}


def synth_batch(B, H, W, seed, device):
    """SURVEY.md 8(d) synthetic inputs, built on the GPU with torch (plumbing)."""
    g = torch.Generator(device=device).manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32),
                            torch.arange(W + 16, device=device, dtype=torch.float32), indexing="ij")
    img = torch.zeros(B, 3, H, W + 16, device=device)
    for _ in range(8):
        f = torch.rand(B, 3, 2, device=device, generator=g) * 0.2 + 0.005
        ph = torch.rand(B, 3, 1, 1, device=device, generator=g) * 6.283
        img += torch.sin(f[..., 0, None, None] * xx + f[..., 1, None, None] * yy + ph)
    img = (img - img.amin((2, 3), keepdim=True)) / (img.amax((2, 3), keepdim=True) - img.amin((2, 3), keepdim=True))
    img = (0.95 * img + 0.05 * torch.rand(img.shape, device=device, generator=g)).clamp(0, 1)
    img = (img * 255).round() / 255  # uint8 frames / 255 like mg_net.py:320-335
    cur = img[..., 8:8 + W].contiguous()
    prev = torch.roll(img[..., 5:5 + W], 1, 2).contiguous()    # shifted by (+3,+1) px
    nxt = torch.roll(img[..., 11:11 + W], -1, 2).contiguous()  # shifted by (-3,-1) px
    inv = []
    for s in (8, 16, 32):  # the heads predict at /8,/16,/32 and upsample bilinearly (mg_net.py:804-807)
        lo = torch.rand(B, 1, H // s, W // s, device=device, generator=g) * 1.9 + 0.05
        inv.append(torch.nn.functional.interpolate(lo, size=(H, W), mode="bilinear", align_corners=True).contiguous())
    poses = 0.01 * torch.randn(B, 2, 6, device=device, generator=g)
    mask = torch.rand(B, 1, H, W, device=device, generator=g) < 0.9
    K = torch.eye(4, device=device).repeat(B, 1, 1)
    sx, sy = W / 2048.0, H / 1024.0  # camera_utils.py:15-21 scale_intrinsics of the Cityscapes camera
    K[:, 0, 0], K[:, 1, 1] = 2262.52 * sx, 2265.30 * sy
    K[:, 0, 2], K[:, 1, 2] = (1096.98 + 0.5) * sx - 0.5, (513.137 + 0.5) * sy - 0.5
    return dict(inv=inv, img=cur, prev=prev, nxt=nxt, poses=poses, mask=mask, K=K)


class HipEvents:
    """Raw hipEvent_t pair handed to the C-ABI (cfg.prof_begin/prof_end) so that the dominant kernel is timed on
    the stream it is launched on."""

    def __init__(self, n):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.pairs = []
        for _ in range(n):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))

    def elapsed_ms(self):
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
            out.append(ms.value)
        return out


def cpu_baseline(H, W, state_dict=None, cfg=None):
    """CPU baseline on a bounded sample (ONE frame of the workload), `kind: "port"`:
    with a state_dict: the full training step (network forward + five losses + backward) of oracle/network_oracle.py
    (plain-torch fp32 restatement of the reference network + the pinned C oracle of the reprojection loss);
    without: only the reprojection loss (used by --loss-only)."""
    import oracle

    oracle.build()
    if state_dict is not None:
        from mgnet_amd.data import synthetic_batch
        from oracle import network_oracle as NO

        torch.manual_seed(0)
        nb = 2  # F.batch_norm refuses a single value per channel (the 1x1-spatial GCM / attention layers need B >= 2)
        batch = synthetic_batch(nb, H, W, "cpu", seed=99)
        sd = {k: v.detach().float().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in state_dict.items()}
        t0 = time.time()
        losses = NO.mgnet_losses(sd, batch, pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD,
                                 ohem_n_min=min(cfg.MODEL.SEM_SEG_HEAD.OHEM_N_MIN, H * W // 4 - 1))
        sum(losses.values()).backward()
        dt = time.time() - t0
        return {"value": round(nb / dt, 4), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{nb} frames {H}x{W}: full MGNet training step fwd+bwd (no optimizer), oracle/network_oracle.py "
                          f"(torch fp32 CPU, {torch.get_num_threads()} threads) + oracle/reproj_oracle.c, {dt:.1f} s"}
    rs = np.random.RandomState(0)
    B = 1
    inv = [rs.uniform(0.05, 1.95, (B, 1, H, W)).astype(np.float32) for _ in range(3)]
    img, prev, nxt = [rs.uniform(0, 1, (B, 3, H, W)).astype(np.float32) for _ in range(3)]
    poses = (0.01 * rs.randn(B, 2, 6)).astype(np.float32)
    mask = rs.uniform(size=(B, 1, H, W)) > 0.1
    K = np.tile(np.eye(4, dtype=np.float32), (B, 1, 1))
    K[:, 0, 0], K[:, 1, 1], K[:, 0, 2], K[:, 1, 2] = 2262.52, 2265.30, 1096.98, 513.137
    t0 = time.time()
    oracle.reproj_loss(inv, img, prev, nxt, mask, K, poses)
    dt = time.time() - t0
    return {"value": round(B / dt, 4), "unit": "img/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": f"1 frame {H}x{W}, reprojection loss fwd+bwd, oracle/reproj_oracle.c fp32 OpenMP, {dt:.1f} s"}


def conv_roofline(dev, B):
    """Second roofline object (informative): the convolution kernel that takes the largest share of the step -- the 3x3
    256->256 head/refine layers at 1/8 resolution (conv3x3_win8f: the 8-row patch with two blocks per CU, csrc/conv_win.hip:360 picks it
    for pc * rows >= 1024) -- timed live with events on the launch stream."""
    from mgnet_amd import _C
    x = torch.randn(B, 256, 128, 256, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(256, 256, 3, 3, device=dev) * 0.02)
    wl = _C.weight_layout(w, 0)
    for _ in range(3):
        _C.conv_igemm(x, wl, (128, 256), None, 1, 1)
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _C.conv_igemm(x, wl, (128, 256), None, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = 2.0 * B * 128 * 256 * 256 * 256 * 9
    ach = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "conv3x3_win8f (windowed 3x3, 256->256 channels, 8x128x256 pixels: the layer shape with the largest share of the step)",
            "achieved": round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4),
            "flops_per_launch": flops, "avg_launch_ms": round(ms, 4)}


def kernel_census(trainer, batch, steps=2):
    """Every GPU kernel of `steps` training steps (torch.profiler, AFTER the timed region) by origin: this library's kernels, torch's own
    (at::native::*: what is left of torch on the path), the runtime's copy / fill kernels, RCCL's."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(steps):
            trainer.run_step(batch)
        torch.cuda.synchronize()
    cnt = {"mgn": [0, 0.0], "torch": [0, 0.0], "rocclr_copy_fill": [0, 0.0], "rccl": [0, 0.0]}
    for e in prof.key_averages():
        t = float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0))
        if t <= 0 or e.key.startswith("hip"):
            continue
        # (the runtime's copies show up as `Memcpy DtoD / HtoD (...)` / `Memset (...)` activities or as __amd_rocclr_* kernels depending on
        #  the path they take: both are counted -- round 3 dropped the former and reported 0 next to ~38 copyBuffer kernels per step)
        k = ("torch" if "at::" in e.key else "rocclr_copy_fill" if ("__amd_rocclr" in e.key or e.key.startswith(("Memcpy", "Memset")))
             else "rccl" if "nccl" in e.key.lower() else "mgn")
        cnt[k][0] += e.count
        cnt[k][1] += t
    out = {k: {"launches_per_step": round(v[0] / steps, 1)} for k, v in cnt.items()}
    out["torch"]["ms_per_step"] = round(cnt["torch"][1] / steps / 1e3, 3)   # (kernel durations under the profiler; the library's own are in profiles/)
    out["total_launches_per_step"] = round(sum(v[0] for v in cnt.values()) / steps, 1)
    return out


def fp16_leg(args, dev, B, H, W, steps=None):
    """BASELINE C5's dtype on the same workload: fp16 activations + dynamic loss scaling, measured after the bf16 run like the bf16 run
    (same number of timed steps, same warm-up, same execution mode) -- round 3 timed 5 steps right behind the other legs and reported
    +2.3 ms; the same comparison as two full runs on one box is +0.44 ms (profiles/r04_fp16_vs_bf16.txt).  A driver-run number for the
    reference's AMP format; the headline `value` stays the bf16 run."""
    steps = args.steps if steps is None else steps
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1),
                         "SOLVER.AMP.DTYPE", "float16"])
    torch.manual_seed(0)
    trainer = Trainer(cfg, build_model(cfg))
    batch = synthetic_batch(B, H, W, dev, seed=1234)
    for _ in range(4):
        trainer.run_step(batch)
    step, mode, warm = (lambda: trainer.run_step(batch)), "eager", 4
    if args.exec in ("auto", "plan"):
        try:   # the same execution mode as the headline measurement: launch-plan replay
            trainer.record_plan(batch)
            for _ in range(max(2, args.warmup)):
                trainer.replay_plan()
            step, mode, warm = trainer.replay_plan, "plan", 5 + max(2, args.warmup)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] fp16 leg: plan recording failed ({type(e).__name__}: {e}); eager steps timed", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    sc = [float(v) for v in trainer.optimizer.scaler.tolist()]
    return {"dtype": "fp16", "ms_per_step": round(dt * 1e3, 3), "value": round(B / dt, 2), "unit": "img/s", "steps": steps, "step_execution": mode,
            "loss_scale": sc[0], "optimizer_steps_taken": int(sc[2]), "steps_attempted": warm + steps,
            "losses_finite": bool(all(torch.isfinite(v.detach()).item() for v in last.values()))}


def eval_leg(args, dev, H, W):
    """Inference throughput of the same network (mg_net.py:375-520: eval-mode forward + the per-image post-processing -- panoptic
    fusion, depth scaling -- at full resolution): single-scale at B = 1 and B = 8, and the 14-pass multi-scale + flip protocol
    (TEST.MSC_FLIP_EVAL) at B = 1.  Informative (not the headline metric): wall clock around a few calls, device synchronised on both
    sides, inputs resident in HBM.  The norm layers run their eval form (one affine + activation pass per layer, no statistics)."""
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.registry import build_model
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev)])
    torch.manual_seed(0)
    model = build_model(cfg).eval()

    def timed(batch, n, warm):
        with torch.no_grad():
            for _ in range(warm):
                model(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = model(batch)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        assert len(out) == len(batch) and all(torch.isfinite(r["depth"][0]).all().item() for r in out)
        # the network alone (backbone, decoders, heads; the per-image post-processing replaced by a stub for these calls)
        model._inference = lambda bi, outputs: [None] * len(bi)
        try:
            with torch.no_grad():
                model(batch)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    model(batch)
                torch.cuda.synchronize()
        finally:
            del model._inference
        dn = (time.perf_counter() - t0) / n
        return {"ms_per_call": round(dt * 1e3, 2), "img_per_s": round(len(batch) / dt, 2), "calls": n,
                "network_only_ms_per_call": round(dn * 1e3, 2), "network_only_img_per_s": round(len(batch) / dn, 2)}
    res = {"what": f"model.eval()(batch): forward + panoptic / depth post-processing per image at {H}x{W}, bf16, random-init weights, synthetic frames",
           "post_processing": "on the device (csrc/postproc.hip), per image like the reference; its cost is data-dependent -- the centre grouping visits "
                              "every centre candidate per thing pixel, and random-init heads yield thousands of candidates (a trained model: tens) -- "
                              "so `network_only_*` (same calls with the post-processing stubbed out) is the figure that transfers",
           "norm_layers": "folded into the convolutions (ops.conv_abn_eval; MGN_NO_EVALFOLD=1 restores the separate eval pass)"}
    def frames(B):   # the inference fields of the dataset mapper: image, 3x3 camera matrix, camera height (mg_net.py:404-417)
        batch = synthetic_batch(B, H, W, dev, seed=77)
        for d in batch:
            d["camera_matrix"] = d["camera_matrix"][:3, :3].cpu()
            d["camera_height"] = torch.tensor([1.22])
        return batch
    for B, n in ((1, 10), (8, 4)):
        res[f"single_scale_B{B}"] = timed(frames(B), n, 2)
    model.msc_flip_eval = True
    res["msc_flip_14_passes_B1"] = timed(frames(1), 2, 1)
    return res


def fp16_leg_child(args):
    """The fp16 leg as a CHILD process running this script with --dtype fp16 (a fresh process, started -- not exec'ed -- after the
    bf16 measurement): inside the parent, behind its other legs, the same steps ran 1-1.5 ms slower than as a run of their own (so did the
    conv roofline leg: 289 vs 249 us for the same launch), which is what round 3 reported as the 'fp16 gap'.  Two stand-alone runs per dtype
    on one box differ by +0.44 ms (profiles/r04_fp16_vs_bf16.txt)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "fp16", "--no-fp16-leg", "--no-eval-leg", "--no-cpu-baseline", "--no-host-probe", "--steps",
           str(args.steps), "--warmup", str(args.warmup), "--batch", str(args.batch), "--height", str(args.height), "--width", str(args.width),
           "--exec", args.exec, "--timeout", str(min(args.timeout, 600.0))]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=min(args.timeout, 600.0) + 30)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"child run exited with code {r.returncode}", "stderr_tail": r.stderr.splitlines()[-5:]}
    d = json.loads(lines[-1])
    out = {"dtype": "fp16", "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "steps": d["steps"], "warmup": d["warmup"],
           "step_execution": d["config"]["step_execution"].split(" ")[0], "measured_as": "a run of this script of its own (child process), like the headline",
           "losses_finite": all(v == v and abs(v) != float("inf") for v in d["config"]["losses"].values())}
    out.update(d["config"].get("loss_scaling", {}))
    return out


def full_step_bench(args, world, rank, dev):
    """The benchmark: one full MGNet training step (SURVEY 3.1 hot loop) per per-GPU batch of synthetic frames."""
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model

    B, H, W = args.batch, args.height, args.width
    cfg = get_cfg()
    add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B * world,
                         "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1),
                         "SOLVER.AMP.DTYPE", {"bf16": "bfloat16", "fp16": "float16"}[args.dtype]])
    batch = synthetic_batch(B, H, W, dev, seed=1234 + rank)
    ev = HipEvents(args.steps)
    syncbn_note = None
    for attempt in range(2):
        torch.manual_seed(0)  # identical initial weights on every rank (DDP broadcasts rank 0's; same seed is equivalent)
        model = build_model(cfg)
        trainer = Trainer(cfg, model)
        if world == 1:
            break
        # Multi-rank: the SyncBN statistics go through the peer-to-peer mailbox kernels when their self-test passed (engine/peer.py).
        # That path has only ever run between processes sharing ONE GPU, so the bench does not stake its measurement on it: two probe
        # steps, then every rank reports whether a mailbox wait timed out (the wait budget of this process is MGNET_P2P_TIMEOUT_S =
        # 30 s); if any did, all ranks rebuild the model with the process group's collectives instead and say so in the JSON line.
        from mgnet_amd.engine import peer as _peer
        if _peer.exchange() is None:
            break
        probe_err = None
        try:   # (Trainer.run_step raises as soon as it sees the mailbox's time-out flag: that must reach the vote below, not end the rank)
            for _ in range(2):
                trainer.run_step(batch)
        except RuntimeError as e:
            probe_err = str(e)
        torch.cuda.synchronize()
        bad = torch.tensor([1.0 if (probe_err is not None or _peer.exchange().failed()) else 0.0], device=dev)
        torch.distributed.all_reduce(bad, op=torch.distributed.ReduceOp.MAX)
        if bad.item() == 0.0:
            break
        syncbn_note = "peer-to-peer SyncBN exchange timed out in the probe steps: rebuilt with torch.distributed collectives"
        print(f"[bench] {syncbn_note}", file=sys.stderr, flush=True)
        _peer.disable()
        os.environ["MGNET_SYNCBN"] = "rccl"
        del trainer, model
    depth_loss = model.depth_head.loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # Execution mode of the timed steps: "eager" (default) = the launches of a step issued from Python, with the independent
    # branches of MGNet.forward on side streams (pose network | backbone, three heads: concurrent short kernels, hidden dispatch
    # latency); "graph" (--graph on, one rank only) = the whole step captured once in a hipGraph on ONE stream and replayed.
    # Measured (end of round 2): eager + side streams 30.2 ms, graph 32.5 ms (capturing the side-stream branches crashes hipGraph on ROCm 7.0).
    use_graph = args.graph == "on" and world == 1
    # One rank: the recorded step is replayed (--exec auto / plan).  Multi-rank runs: `--exec auto` issues the step EAGERLY and only
    # `--exec plan` replays it.  The multi-rank replay exists and is tested -- the gradient all-reduces re-issued by the plan at their
    # place, the SyncBN statistics on the mailbox kernels, every rank or none (tests/test_dist_gpu.py: two ranks on one GPU, plan ==
    # eager bit for bit) -- but it has never run across GPUs (no multi-GPU box has been available to this repository in any round), and
    # a wrong cross-stream edge or a hang there would cost the driver's scaling run its numbers.  The eager step is GPU-bound too
    # (18-20 ms of host issue under a ~26 ms step: +0.2 ms at one rank, config.step_execution_probe), so the default gives up little.
    use_plan = (args.exec == "plan" or (args.exec == "auto" and world == 1)) and not use_graph
    mode, plan_note = "eager", None
    if world > 1 and args.exec == "auto":
        plan_note = ("multi-rank default: eager issue -- the launch-plan replay of a multi-rank step is validated on two ranks sharing one GPU "
                     "only (bit-identical to eager there), never across GPUs; --exec plan selects it")
    for _ in range(max(args.warmup, 3) if (use_graph or use_plan) else args.warmup):
        trainer.run_step(batch)
    if use_plan:
        # launch-plan replay (engine/plan.py): one eager step recorded, the timed steps replayed from C with the side streams kept
        plan = None
        try:
            # (one process: three recordings, the fastest kept -- a recording's replay time is fixed when its buffers are placed,
            #  Trainer.record_plan; the trial steps are training steps like the warm-up's)
            #  The two fastest recordings are then replayed from one state and must agree bit for bit (Trainer.record_plan
            #  verify_steps; on a difference the read-only declarations are dropped and the step is recorded again): config.plan_check
            plan = trainer.record_plan(batch, prof_slots=args.steps, best_of=3 if world == 1 else 1, verify_steps=args.plan_verify_steps)
        except Exception as e:  # noqa: BLE001 -- report and fall back to the eager step rather than lose the measurement
            if "replay to different results" in str(e):
                chk = getattr(trainer, "plan_check", None) or {}
                plan_note = ("launch plan refused by the recording check: two recordings of the step, replayed from one state, do not agree bit for bit "
                             f"(first difference at step {(chk.get('first_difference') or {}).get('step')}: {(chk.get('first_difference') or {}).get('losses')}; "
                             "with and without the read-only declarations; profiles/r06_determinism.txt section 4) -- eager steps timed; "
                             "--exec plan --plan-verify-steps 0 times the replay anyway")
            else:
                plan_note = f"plan recording failed ({type(e).__name__}: {e}); eager steps timed"
            print(f"[bench] {plan_note}", file=sys.stderr, flush=True)
        ok = torch.tensor([0.0 if plan is None else 1.0], device=dev)
        if world > 1:   # every rank replays, or none does (a rank issuing another launch sequence would leave its peers waiting)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if ok.item() == 1.0:
            for _ in range(2):
                trainer.replay_plan()
            mode = "plan"
        elif plan is not None:
            plan_note = "plan recording failed on another rank; eager steps timed"
    exec_probe = None
    if mode == "plan" and args.exec == "auto" and world == 1:
        # Which way of ISSUING the same step is faster on this box?  The replay takes the host from 15-25 ms to 2.5-3.5 ms per step, but
        # the step is GPU-bound either way, and the replay's minimal cross-stream dependencies let the branches overlap a little more
        # than the eager step's do -- which costs the one-block-per-CU kernels more than it hides: measured -0.1 ... +0.9 ms per step
        # depending on the box (profiles/r04_same_box_r3_vs_r4.txt).  A short untimed calibration (2 x 8 steps each, interleaved)
        # picks the mode of the timed steps; both figures and both host costs are reported (`config.step_execution_probe`).
        def probe(fn, n=8):
            fence()
            tp = time.perf_counter()
            for _ in range(n):
                fn()
            fence()
            return (time.perf_counter() - tp) / n * 1e3

        def issue(fn, n=3):
            ts = []
            for _ in range(n):
                torch.cuda.synchronize()
                ti = time.perf_counter()
                fn()
                ts.append((time.perf_counter() - ti) * 1e3)
            torch.cuda.synchronize()
            return float(np.median(ts))

        eager_fn, plan_fn = (lambda: trainer.run_step(batch)), (lambda: trainer.replay_plan())
        ms_p, ms_e = [], []
        for _ in range(2):
            ms_p.append(probe(plan_fn))
            ms_e.append(probe(eager_fn))
        exec_probe = {"plan_ms_per_step": round(min(ms_p), 3), "eager_ms_per_step": round(min(ms_e), 3),
                      "plan_host_issue_ms": round(issue(plan_fn), 2), "eager_host_issue_ms": round(issue(eager_fn), 2),
                      "how": "untimed calibration before the timed region: 2 x 8 steps per mode, interleaved, best of two; host issue = one step "
                             "into an empty queue, median of 3"}
        if min(ms_e) < 0.997 * min(ms_p):
            mode = "eager"
            plan_note = "launch-plan replay recorded and available (--exec plan); the calibration found the eager issue faster on this box"
        exec_probe["timed"] = mode
        if getattr(trainer, "plan_trials", None):
            exec_probe["plan_recordings_ms"] = trainer.plan_trials   # (6 replays each; the fastest recording is the one probed and timed)
    if use_graph:
        try:
            trainer.capture_step(batch)
            for _ in range(2):
                trainer.replay_step()
            mode = "graph"
        except Exception as e:  # noqa: BLE001 -- report and fall back to the eager step rather than lose the measurement
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing the eager step", file=sys.stderr, flush=True)
    fence()
    t0 = time.perf_counter()
    if mode == "graph":
        for k in range(args.steps):
            last = trainer.replay_step()
    elif mode == "plan":
        for k in range(args.steps):
            last = trainer.replay_plan(prof_slot=k)
    else:
        for k in range(args.steps):
            depth_loss.prof_events = ev.pairs[k]
            last = trainer.run_step(batch)
    t_issue = time.perf_counter() - t0   # host time to issue the K steps (no sync inside a step): ~dt means launch-bound
    fence()
    dt = time.perf_counter() - t0
    if mode == "graph":
        # events recorded by graph nodes cannot be read back with hipEventElapsedTime, so the dominant kernel is timed with
        # event pairs on its launch stream in eager steps of the same workload run right after the timed region
        for k in range(args.steps):
            depth_loss.prof_events = ev.pairs[k]
            trainer.run_step(batch)
        fence()
    # What the host needs to ISSUE one step at the benchmark size: the issue call timed with an EMPTY launch queue (device synchronised
    # before each step, none inside), so that it measures the host's own work and not the wait for queue slots -- over the K timed steps
    # the issue loop runs ahead of a GPU-bound step only until the hardware queues are full, and its wall time then equals the GPU's.
    issue_one = []
    for _ in range(5 if world == 1 else 3):   # (every rank takes part: a multi-rank step is a round of collectives)
        torch.cuda.synchronize()
        ti = time.perf_counter()
        if mode == "plan":
            trainer.replay_plan()
        elif mode == "graph":
            trainer.replay_step()
        else:
            trainer.run_step(batch)
        issue_one.append(time.perf_counter() - ti)
    torch.cuda.synchronize()
    host_issue_ms = float(np.median(issue_one)) * 1e3 if issue_one else t_issue / args.steps * 1e3
    # What the host needs to ISSUE a step, measured where the GPU cannot back-pressure the launch queue: the same model and launch
    # sequence on two 512x1024 frames (an eighth of the device work, the same host work).  `host_issue_ms_per_step` above is the
    # wall time of the launch loop at the benchmark size, which mostly waits for queue slots once the GPU is the bottleneck.
    host_unloaded = None
    if rank == 0 and world == 1 and mode in ("eager", "plan") and not args.no_host_probe:
        try:
            small = synthetic_batch(2, 512, 1024, dev, seed=7)
            for _ in range(2):
                trainer.run_step(small)
            torch.cuda.synchronize()
            th = time.perf_counter()
            for _ in range(5):
                trainer.run_step(small)
            host_unloaded = (time.perf_counter() - th) / 5 * 1e3
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001 -- informative only
            print(f"[bench] unloaded host-issue measurement skipped ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
    dist_info = None
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
        # what the exchange costs: a few more steps WITHOUT the gradient all-reduce (measurement only, after the timed region;
        # the ranks' weights drift apart from here on, nothing is timed afterwards)
        from mgnet_amd.modeling import ops as _ops
        from mgnet_amd.engine import peer as _peer
        n_ar, n_bn, n_p2p = trainer.reducer.collectives, _ops.SYNCBN_COLLECTIVES[0], _ops.SYNCBN_P2P[0]
        trainer.run_step(batch)
        n_ar, n_bn, n_p2p = trainer.reducer.collectives - n_ar, _ops.SYNCBN_COLLECTIVES[0] - n_bn, _ops.SYNCBN_P2P[0] - n_p2p
        p2p_failed = bool(_peer.exchange().failed()) if _peer.exchange() is not None else None
        trainer.reducer.enabled = False
        n_extra = max(3, min(10, args.steps))
        fence()
        t1 = time.perf_counter()
        for _ in range(n_extra):
            trainer.run_step(batch)
        fence()
        t_no = torch.tensor([(time.perf_counter() - t1) / n_extra], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t_no, op=torch.distributed.ReduceOp.MAX)
        dist_info = {"backend": torch.distributed.get_backend(), "rccl_world_size": torch.distributed.get_world_size(),
                     "grad_allreduce_calls_per_step": n_ar, "grad_bytes_per_step": trainer.reducer.grad_bytes(),
                     "syncbn_collectives_per_step": n_bn, "syncbn_p2p_exchanges_per_step": n_p2p,
                     "syncbn_exchange": dict(_peer.report(), wait_timed_out=p2p_failed, **({"note": syncbn_note} if syncbn_note else {})),
                     "ms_per_step_without_grad_allreduce": round(float(t_no.item()) * 1e3, 3),
                     "exposed_grad_allreduce_ms_per_step": round(dt / args.steps * 1e3 - float(t_no.item()) * 1e3, 3)}
    if rank == 0:
        state_dict_cpu = None if (args.no_cpu_baseline or world > 1) else {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        kern_ms = float(np.median([plan.prof_elapsed_ms(k) for k in range(args.steps)] if mode == "plan" else ev.elapsed_ms()))
        npx = B * H * W
        u8_frames = model._orig_frames_u8(batch) is not None     # the layout MGNet.forward hands to the loss for this batch
        traffic, tj_match = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if (tj.get("B"), tj.get("H"), tj.get("W")) == (B, H, W) and bool(tj.get("u8_frames", False)) == u8_frames:
                traffic, tj_match = tj.get("hbm_bytes_per_launch"), tj
        img_s = world * B * args.steps / dt
        # forward conv FLOPs per image scale with the pixel count; training ~ 3x forward
        gflop_fwd = 560.2 * (H * W) / (1024 * 2048)
        line = {
            "metric": "training img/s at 1024x2048 Cityscapes, 1/2/4/8 MI355X; reprojection-loss HBM GB/s",
            "value": round(img_s, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"MGNet-Cityscapes-VideoSequence recipe (BASELINE {'C5: fp16 + dynamic loss scaling' if args.dtype == 'fp16' else 'C4/C5'}): full multi-task training step "
                                   f"(2x ResNet-18 + 3 decoders/heads fwd+bwd, OHEM CE, centre/offset, photometric "
                                   f"reprojection + smoothness, uncertainty weighting, grad all-reduce, clip, Adam), "
                                   f"{B} frames/GPU of {H}x{W}",
                       "frames_per_gpu": B, "global_batch": B * world, "height": H, "width": W,
                       "parallelism": f"dp{world}", "step_execution": mode + (" (one hipGraph replay per step, one stream)" if mode == "graph" else
                                                  (" replay: one recorded step's launches re-issued from C (csrc/plan.hip), side streams kept; " + json.dumps(plan.report)) if mode == "plan" else
                                                  " (launches issued from Python; side streams for the independent branches: " +
                                                  ("on" if model._side_streams() is not None else "off") + ")" + (f"; {plan_note}" if plan_note else "")),
                       **({"step_execution_probe": exec_probe} if exec_probe else {}),
                       **({"plan_check": trainer.plan_check} if getattr(trainer, "plan_check", None) else {}),
                       "host_issue_ms_per_step": round(host_issue_ms, 2),
                       "host_issue_is": "wall time of issuing ONE step of this workload into an empty launch queue (median of 5 -- 3 on multi-rank runs -- after the timed region)",
                       "host_issue_loop_wall_ms_per_step": round(t_issue / args.steps * 1e3, 2),
                       "host_issue_loop_wall_is": "wall time of the K-step issue loop / K: includes waiting for launch-queue slots once the GPU is the bottleneck",
                       "host_issue_ms_per_step_unloaded": None if host_unloaded is None else round(host_unloaded, 2),
                       "host_issue_unloaded_is": "the EAGER issue of the same step (Python / autograd / ctypes), measured at an eighth of the device work", "conv_tflops_per_gpu": round(3 * gflop_fwd * img_s / world / 1e3, 1),
                       "losses": {k: round(float(v.detach()), 5) for k, v in last.items()},
                       "torch_staging_ops": sorted(__import__("mgnet_amd.modeling.ops", fromlist=["x"]).STAGING_USED)},
            "roofline": reproj_roofline(kern_ms, npx, u8_frames, traffic, {
                **valu_ceiling(kern_ms, tj_match),
                "timed_with": "hipEvent pairs around the kernel on its launch stream, " +
                              (f"{args.steps} eager steps run right after the graph-replayed timed region" if mode == "graph"
                               else "recorded by the replayed plan inside the timed steps" if mode == "plan" else "inside the timed steps")}),
        }
        if dist_info is not None:
            line["config"]["distributed"] = dist_info
        if world == 1:   # (extra steps on rank 0 alone would wait for the other ranks' SyncBN rows forever)
            try:
                line["config"]["kernels_per_step"] = kernel_census(trainer, batch)
            except Exception as e:  # noqa: BLE001 -- informative only
                line["config"]["kernels_per_step"] = f"unavailable ({type(e).__name__}: {e})"
        line["roofline_mfma"] = conv_roofline(dev, B)
        if args.dtype == "fp16" and getattr(trainer.optimizer, "scaler", None) is not None:
            sc = [float(v) for v in trainer.optimizer.scaler.tolist()]
            line["config"]["loss_scaling"] = {"loss_scale": sc[0], "optimizer_steps_taken": int(sc[2]), "steps_attempted": trainer.iter}
        if world == 1 and args.dtype == "bf16" and not args.no_fp16_leg:
            try:
                del trainer, model
                torch.cuda.empty_cache()
                line["fp16"] = fp16_leg_child(args)
            except Exception as e:  # noqa: BLE001
                line["fp16"] = {"error": f"{type(e).__name__}: {e}"}
            model = None
        if world == 1 and args.dtype == "bf16" and not args.no_eval_leg:
            try:
                line["eval"] = eval_leg(args, dev, H, W)
            except Exception as e:  # noqa: BLE001 -- informative only
                line["eval"] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(H, W, state_dict_cpu, cfg)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()   # rank 0 may still be timing the informative conv roofline: leave together
        torch.distributed.destroy_process_group()


def self_launch(n, timeout_s):
    """`python bench.py --gpus N` without a launcher: run the N ranks as a child job (one process per GPU over RCCL) and
    return its exit code.  Called before any HIP call of this process; the parent never initialises the GPU.  The child runs in its
    own process group: on a timeout exactly that group is stopped (never a pattern), the tail of its stderr (NCCL_DEBUG=WARN) goes into
    a JSON failure line, and the exit code is non-zero."""
    import signal
    import socket
    import subprocess
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    with tempfile.TemporaryFile(mode="w+") as err:
        proc = subprocess.Popen(cmd, env=env, stderr=err, start_new_session=True)
        try:
            rc = proc.wait(timeout=timeout_s)
            why = None
        except subprocess.TimeoutExpired:
            why = f"child job exceeded --timeout {timeout_s:.0f} s"
            for sig in (signal.SIGTERM, signal.SIGKILL):
                try:
                    os.killpg(proc.pid, sig)      # the session / process group this call created
                except ProcessLookupError:
                    break
                try:
                    proc.wait(timeout=15)
                    break
                except subprocess.TimeoutExpired:
                    continue
            rc = 124
        err.seek(0)
        tail = err.read()[-6000:]
        sys.stderr.write(tail)
        if rc != 0:
            print(json.dumps({"metric": "training img/s at 1024x2048 Cityscapes, 1/2/4/8 MI355X; reprojection-loss HBM GB/s", "value": None,
                              "n_gpus": n, "error": why or f"child job exited with code {rc}",
                              "stderr_tail": [ln for ln in tail.splitlines() if ln.strip()][-25:]}), flush=True)
        return rc


def rank_watchdog(timeout_s):
    """a rank that is still alive after `timeout_s` (a collective that never completes, a peer that died) leaves with code 3 instead of
    holding the driver: a daemon timer, no GPU call, no exec"""
    import threading

    def bail():
        sys.stderr.write(f"[bench] rank {os.environ.get('RANK', '0')}: --timeout {timeout_s:.0f} s exceeded, exiting\n")
        sys.stderr.flush()
        os._exit(3)
    t = threading.Timer(timeout_s, bail)
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU (C4/C5: 8)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-probe", action="store_true", help="skip the small-batch steps that time the unloaded launch loop (profiling runs)")
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default="bf16",
                    help="16-bit activation format of the trunk: bf16 (default) or the reference's AMP format fp16 with dynamic loss "
                         "scaling (BASELINE C5)")
    ap.add_argument("--exec", choices=["auto", "plan", "eager"], default="auto",
                    help="how the timed steps are issued: plan = launch-plan replay from C, eager = from Python; auto = plan on one rank (when it "
                         "records and its calibration is not slower), eager on multi-rank runs (the multi-rank replay has not run across GPUs yet)")
    ap.add_argument("--plan-verify-steps", type=int, default=12,
                    help="one rank: the two fastest of the three recordings are replayed this many steps each from the same state and must agree "
                         "bit for bit (0 = skip the check)")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="on: replay the captured step as a hipGraph (1 GPU, one stream); auto/off: issue every launch from Python "
                         "with the independent branches on side streams (the faster mode)")
    ap.add_argument("--no-eval-leg", action="store_true", help="skip the inference-throughput object (`eval`) measured after the training step (N = 1)")
    ap.add_argument("--no-fp16-leg", action="store_true", help="skip the 5 fp16 + loss-scaling steps run after the bf16 measurement (N = 1)")
    ap.add_argument("--timeout", type=float, default=1500.0,
                    help="seconds after which a rank (and, with --gpus N self-launch, the whole child job) is stopped and the bench exits non-zero")
    ap.add_argument("--fwd-only", action="store_true", help="diagnostic: loss only (no gradient); NOT the benchmark")
    ap.add_argument("--loss-only", action="store_true",
                    help="diagnostic: time only the reprojection loss fwd+bwd (round-1 v2 workload); NOT the benchmark")
    args = ap.parse_args()

    os.environ.setdefault("MGNET_P2P_TIMEOUT_S", "30")   # (a SyncBN mailbox wait that fails must fail within the bench's own time budget)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, args.timeout))   # nothing has touched the GPU in this process
    rank_watchdog(args.timeout)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # one rank per GPU.  (MGNET_DIST_BACKEND=gloo lets the multi-rank path be exercised on a box with fewer GPUs than ranks --
    # ranks then share devices, which RCCL refuses; a functional check only, never a measurement)
    backend = os.environ.get("MGNET_DIST_BACKEND", "nccl")
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)          # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=min(args.timeout, 600.0)))

    from mgnet_amd import _C
    from mgnet_amd.modeling.loss import _ReprojLossFn

    B, H, W = args.batch, args.height, args.width
    if not (args.loss_only or args.fwd_only):
        return full_step_bench(args, world, rank, dev)
    d = synth_batch(B, H, W, 1234 + rank, dev)
    if os.environ.get("MGN_REPROJ_RGBX"):   # diagnostic: context frames pixel-interleaved ([B,H,W,4] in memory, 4th channel unused)
        for k in ("prev", "nxt"):
            t = torch.zeros((B, 4, H, W), device=dev).contiguous(memory_format=torch.channels_last)
            t[:, :3] = d[k]
            d[k] = t
    u8_frames = bool(os.environ.get("MGN_REPROJ_U8"))
    if u8_frames:   # diagnostic: all three frames as uint8 RGBX ([B,H,W,4] in memory), the layout the training step uses
        for k in ("img", "prev", "nxt"):
            t = torch.zeros((B, 4, H, W), device=dev, dtype=torch.uint8).contiguous(memory_format=torch.channels_last)
            t[:, :3] = (d[k] * 255).round().to(torch.uint8)
            d[k] = t
    inv = [x.requires_grad_(not args.fwd_only) for x in d["inv"]]
    poses = d["poses"].requires_grad_(not args.fwd_only)
    nsteps = args.warmup + args.steps
    ev = HipEvents(args.steps)
    cfgs = [_C.make_reproj_cfg(B, H, W, 3) for _ in range(nsteps)]
    for k in range(args.steps):
        cfgs[args.warmup + k].prof_begin, cfgs[args.warmup + k].prof_end = ev.pairs[k]
    w = torch.ones(2, device=dev)

    def step(k):
        # data-parallel: each rank owns its B frames; the loss has no cross-rank term (per-rank means)
        losses = _ReprojLossFn.apply(cfgs[k], d["img"], d["prev"], d["nxt"], d["mask"], d["K"], poses, *inv)
        if args.fwd_only:
            return losses
        (losses * w).sum().backward()
        for x in inv:
            x.grad = None
        poses.grad = None
        return losses

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        last = step(args.warmup + k)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms = ev.elapsed_ms()
        kern_ms = float(np.mean(ms))
        npx = B * H * W
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("B") == B and tj.get("H") == H and tj.get("W") == W and bool(tj.get("u8_frames", False)) == u8_frames:
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": "training img/s at 1024x2048 Cityscapes, 1/2/4/8 MI355X; reprojection-loss HBM GB/s",
            "value": round(world * B * args.steps / dt, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"MGNet-Cityscapes-VideoSequence (C4/C5 shape): photometric reprojection loss "
                                   f"fwd+bwd only (SURVEY 8a group G; network rows N not in the step yet), "
                                   f"{B} frames/GPU of {H}x{W}, 3 scales, 2 context frames",
                       "frames_per_gpu": B, "height": H, "width": W, "parallelism": f"dp{world}",
                       "loss_photometric": float(last[0]), "loss_smoothness": float(last[1])},
            "roofline": reproj_roofline(kern_ms, npx, u8_frames, traffic),
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(H, W)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""HBM traffic of the reprojection kernel from the PMC counters, with a same-pass calibration of the counter on a dword
stream of KNOWN size.

  workload mode (run under rocprofv3 --pmc FETCH_SIZE, then again under --pmc WRITE_SIZE; tools/pmc_traffic2.sh):
      python3 tools/measure_traffic.py run
    launches, in one process: (a) the full MGNet training step of bench.py (B=8, 1024x2048; its reproj_march<true> launches are
    what is measured) and (b) the calibration kernel `reconstruct_kernel<false>` (csrc/geometry.hip) on a [8,1,1024,2048] fp32
    depth map: one coalesced dword load per lane = 67,108,864 B read, three dword stores per lane = 201,326,592 B written.
  summary mode:
      python3 tools/measure_traffic.py summarize <dir with fetch/ and write/ csv> -> profiles/traffic.json
    FETCH_SIZE / WRITE_SIZE are in KiB; factor_read = known / measured on the calibration kernel, same for writes; the
    reprojection kernel's counters are multiplied by those factors."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B, H, W = 8, 1024, 2048


def run():
    import torch
    from mgnet_amd import _C, add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model
    dev = torch.device("cuda:0")
    cfg = get_cfg(); add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
    torch.manual_seed(0)
    trainer = Trainer(cfg, build_model(cfg))
    batch = synthetic_batch(B, H, W, dev, seed=1234)
    depth = torch.rand(B, 1, H, W, device=dev) + 0.5
    A = torch.eye(3, device=dev).reshape(1, 9).repeat(B, 1).contiguous()
    t = torch.zeros(B, 3, device=dev)
    for _ in range(4):
        trainer.run_step(batch)
        _C.reconstruct_fwd(depth, A, t)
    torch.cuda.synchronize()


def summarize(root):
    acc = {}
    for f in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = "reproj" if "reproj_march" in r["Kernel_Name"] else ("calib" if "reconstruct_kernel" in r["Kernel_Name"] else None)
            if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]) * 1024.0)
            elif k == "reproj" and r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
                acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in acc.items()}
    px = B * H * W
    known_r, known_w = 4.0 * px, 12.0 * px
    fr, fw = known_r / avg[("calib", "FETCH_SIZE")], known_w / avg[("calib", "WRITE_SIZE")]
    rd, wr = avg[("reproj", "FETCH_SIZE")] * fr, avg[("reproj", "WRITE_SIZE")] * fw
    u8 = not os.environ.get("MGN_FRAMES_F32")   # (round 3: the training step hands the frames over as uint8 RGBX: 37 B/px algorithmic)
    out = {"B": B, "H": H, "W": W, "u8_frames": u8, "kernel": "reproj_march<true>", "hbm_bytes_per_launch": int(round(rd + wr)),
           "read_bytes_per_launch": int(round(rd)), "write_bytes_per_launch": int(round(wr)),
           "raw_counters_bytes": {"FETCH_SIZE": int(avg[("reproj", "FETCH_SIZE")]), "WRITE_SIZE": int(avg[("reproj", "WRITE_SIZE")])},
           "calibration": {"kernel": "reconstruct_kernel<false> on [8,1,1024,2048] fp32 (dword loads / stores, one per lane)",
                           "known_read_bytes": int(known_r), "known_write_bytes": int(known_w),
                           "FETCH_SIZE_bytes": int(avg[("calib", "FETCH_SIZE")]), "WRITE_SIZE_bytes": int(avg[("calib", "WRITE_SIZE")]),
                           "factor_read": round(fr, 4), "factor_write": round(fw, 4)},
           "algorithmic_bytes_per_launch": (37 if u8 else 61) * px,
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of tools/measure_traffic.py run (tools/pmc_traffic2.sh); "
                   "counters of the reprojection kernel multiplied by the factors that make the same counters of a dword stream "
                   "of known size (same process, same pass) equal to its byte count"}
    if ("reproj", "SQ_INSTS_VALU") in avg:
        # the limiter the counters name: vector-ALU issue.  SQ_INSTS_VALU counts wave instructions; a wave64 instruction takes
        # two cycles on a SIMD-32 (MI355X_MICROARCH.md, wave scheduling), SQ_ACTIVE_INST_VALU counts quad-cycles per SIMD,
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs.
        insts, act = avg[("reproj", "SQ_INSTS_VALU")], avg.get(("reproj", "SQ_ACTIVE_INST_VALU"))
        gui = avg.get(("reproj", "GRBM_GUI_ACTIVE"))
        out["valu"] = {"wave_instructions_per_launch": int(insts), "lane_instructions_per_px": round(insts * 64.0 / px, 1),
                       "SQ_ACTIVE_INST_VALU_quadcycles": None if act is None else int(act), "GRBM_GUI_ACTIVE_sum_over_xcds": None if gui is None else int(gui),
                       "active_frac": None if (act is None or not gui) else round(act * 4.0 / (1024.0 * gui / 8.0), 4),
                       "active_frac_is": "SQ_ACTIVE_INST_VALU x 4 cycles / (1024 SIMDs x kernel cycles): share of the kernel's cycles in which a SIMD's vector ALU is executing",
                       "note": "third pass of tools/pmc_traffic2.sh (--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE), same process"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else summarize(sys.argv[2])

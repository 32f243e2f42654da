"""Side streams of MGNet.forward: the training trajectory must be bit-identical with and without them, and run to run."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from mgnet_amd import add_mgnet_config, get_cfg
    from mgnet_amd.data import synthetic_batch
    from mgnet_amd.engine import Trainer
    from mgnet_amd.registry import build_model
    B, H, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    dev = torch.device("cuda:0")
    cfg = get_cfg(); add_mgnet_config(cfg)
    cfg.merge_from_file(os.path.join(ROOT, "configs", "bench-c4-cityscapes-videosequence.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(dev), "SOLVER.IMS_PER_BATCH", B, "MODEL.SEM_SEG_HEAD.OHEM_N_MIN", min(524287, B * H * W // 4 - 1)])
    torch.manual_seed(0)
    model = build_model(cfg); trainer = Trainer(cfg, model)
    batch = synthetic_batch(B, H, W, dev, seed=1234)
    out = []
    for _ in range(int(os.environ.get('DBG_STEPS', '10'))):
        l = trainer.run_step(batch)
        out.append({k: float(v.detach()) for k, v in l.items()})
    torch.cuda.synchronize()
    gn = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    print(json.dumps({"losses": out, "param_abs_sum": gn}))
    sys.exit(0)
for shape in [tuple(a.split("x")) for a in (sys.argv[1:] or ["2x256x512", "8x1024x2048"])]:
    res = {}
    for tag, env in (("off", "0"), ("off_b", "0"), ("off_c", "0"), ("on_a", "1"), ("on_b", "1"), ("on_c", "1"))[:int(os.environ.get("DBG_NRUNS", "6"))]:
        r = subprocess.run([sys.executable, __file__, "child", *shape], env=dict(os.environ, MGNET_STREAMS=env), capture_output=True, text=True)
        try:
            res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(tag, "FAILED", r.stderr[-2000:]); continue
    print(shape, {k: v["param_abs_sum"] for k, v in res.items()})
    for k in res:
        print("  ", k, [round(x["loss_center"], 6) for x in res[k]["losses"]], [round(x["loss_sem_seg"], 6) for x in res[k]["losses"]])
    if len(res) == 3:
        print("   off run-to-run identical:", res["off"] == res["off_b"] == res["off_c"])
    if len(res) == 6:
        print("   on run-to-run identical:", res["on_a"] == res["on_b"] == res["on_c"], " off run-to-run identical:", res["off"] == res["off_b"] == res["off_c"], " on == off:", res["on_a"] == res["off"])

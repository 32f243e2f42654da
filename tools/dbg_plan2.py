"""debug: the two-rank plan replay test body with per-rank logs and a hang dump"""
import faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    faulthandler.dump_traceback_later(45, exit=True)
    os.environ["MGNET_P2P_TIMEOUT_S"] = "20"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_dist_gpu import _plan_two_ranks
    import mgnet_amd.engine.trainer as T
    orig = T.Trainer.record_plan
    def rec(self, *a, **k):
        print(f"[{rank}] record_plan start", flush=True)
        r = orig(self, *a, **k)
        print(f"[{rank}] record_plan done {r.report}", flush=True)
        return r
    T.Trainer.record_plan = rec
    orig2 = T.Trainer.replay_plan
    def rep(self, *a, **k):
        print(f"[{rank}] replay start", flush=True)
        r = orig2(self, *a, **k)
        print(f"[{rank}] replay issued", flush=True)
        return r
    T.Trainer.replay_plan = rep
    print(f"[{rank}] result", _plan_two_ranks(rank, world), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

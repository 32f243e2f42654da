"""Thin training loop reproducing what tools/train_net.py delegates to detectron2 (SURVEY 3.1 hot loop): forward ->
sum of the loss dict -> backward (bucketed all-reduce overlapped) -> full-model clip -> Adam -> poly LR.
bf16 autocast replaces the reference's fp16 + GradScaler: no loss scaling / inf check needed."""
import torch

from ..events import EventStorage
from ..solver import build_lr_scheduler, build_optimizer
from .reducer import GradReducer


class Trainer:
    def __init__(self, cfg, model, bucket_bytes=32 << 20):
        self.cfg, self.model = cfg, model
        on_gpu = next(model.parameters()).is_cuda
        if on_gpu and cfg.SOLVER.OPTIMIZER == "ADAM":
            from .. import _C
            from ..solver import get_mgnet_optimizer_params
            groups = get_mgnet_optimizer_params(model, cfg.SOLVER.BASE_LR, head_lr_factor=cfg.SOLVER.HEAD_LR_FACTOR)
            self.reducer = GradReducer([p for g in groups for p in (g["params"] if isinstance(g["params"], list) else [g["params"]])],
                                       bucket_bytes, align=_C.optim_chunk(), flatten_params=True, average=False)
            self.optimizer = build_optimizer(cfg, model, reducer=self.reducer)
        else:
            self.optimizer = build_optimizer(cfg, model)
            self.reducer = GradReducer([p for g in self.optimizer.param_groups for p in g["params"]], bucket_bytes)
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        self.storage = EventStorage()
        self.iter = 0
        from ..checkpoint import Checkpointer
        self.checkpointer = Checkpointer(model, getattr(cfg, "OUTPUT_DIR", "") or "", optimizer=self.optimizer, scheduler=self.scheduler)

    def resume_or_load(self, resume=True):
        """detectron2 DefaultTrainer.resume_or_load (tools/train_net.py:234): MODEL.WEIGHTS, or the last checkpoint of
        OUTPUT_DIR (then training continues after its iteration)."""
        extra = self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        if resume and self.checkpointer.has_checkpoint():
            self.iter = int(extra.get("iteration", -1)) + 1
        return extra

    def save(self, name=None):
        return self.checkpointer.save(name or f"model_{self.iter - 1:07d}", iteration=self.iter - 1)

    def run_step(self, batched_inputs):
        self.model.train()
        self.reducer.zero_grad()
        with self.storage:
            loss_dict = self.model(batched_inputs)
            losses = sum(loss_dict.values())
            losses.backward()
        self.reducer.finish()
        self.optimizer.step()
        self.scheduler.step()
        self.iter += 1
        self.storage.step()
        return loss_dict

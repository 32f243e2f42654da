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
        self.optimizer = build_optimizer(cfg, model)
        self.scheduler = build_lr_scheduler(cfg, self.optimizer)
        self.reducer = GradReducer([p for g in self.optimizer.param_groups for p in g["params"]], bucket_bytes)
        self.storage = EventStorage()
        self.iter = 0

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

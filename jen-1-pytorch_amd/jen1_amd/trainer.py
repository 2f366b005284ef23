"""``UnifiedMultiTaskTrainer``: the training step of the reference around the HIP training path.

Host-side mirror of /root/reference/trainer.py:126-213 -- ``train(audio_emb, metadata)`` (three equal sub-batches, one
per task in the fixed order of config.py:93, a fresh mask / timestep draw per sub-batch, the sum of the three losses)
and the optimiser part of ``train_loop`` (trainer.py:139-150: zero_grad at the start of an accumulation window,
``(loss / grad_accum_every).backward()``, and every ``grad_accum_every`` micro-batches clip -> AdamW -> LinearLR).
Same names and argument meaning; what is NOT mirrored: the data loader, logging / tensorboard, evaluation and the
GradScaler (the HIP path trains in bf16 storage with float32 accumulation and float32 master weights, which needs no
loss scaling; ``skip_nonfinite`` of FusedAdamW keeps GradScaler's skip-on-overflow behaviour).

Data-parallel training (train.py:88-89): one process per GPU, every rank draws its own masks /
timesteps / noise, and the mean all-reduce of the flat gradient buffer (``optim.allreduce_gradients``, RCCL over xGMI)
runs once per optimiser step, right before the clip -- the gradient exchange is the only collective.
"""
from __future__ import annotations

import random as _random
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from .optim import FusedAdamW, GradExchange, LinearLR
from .tasks import get_conditioning, random_mask
from .train import GraphedLossStep

TASKS = ("text_guided", "music_inpaint", "music_cont")        # config.py:93


class UnifiedMultiTaskTrainer:
    """``conditioner(metadata, device)`` returns ``{"prompt": (embedding [b, 128, 1024], mask [b, 128])}`` like
    ``MultiConditioner.forward`` (conditioners.py:182-208); ``model`` is a ``jen1_amd.model.UNetCFG1d``."""

    def __init__(self, model, diffusion, conditioner: Callable, optimizer: FusedAdamW, lr_scheduler: Optional[LinearLR] = None,
                 grad_accum_every: int = 10, tasks: Sequence[str] = TASKS, device="cuda", process_group=None,
                 rng=_random, cross_attn_cond_ids: Sequence[str] = ("prompt",), global_cond_ids: Sequence[str] = (),
                 input_concat_ids: Sequence[str] = ("masked_input", "mask"), compute_dtype: Optional[str] = None,
                 use_graph: bool = True, allow_uneven_tasks: bool = False, bucket_bytes: int = 128 << 20, merge_tasks: bool = True):
        self.model, self.diffusion, self.conditioner, self.optimizer, self.lr_scheduler = model, diffusion, conditioner, optimizer, lr_scheduler
        self.grad_accum_every, self.tasks, self.device, self.group, self.rng = grad_accum_every, tuple(tasks), device, process_group, rng
        self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids = cross_attn_cond_ids, global_cond_ids, input_concat_ids
        self.graph = model.train_graph(compute_dtype)
        self.graph.attach_optimizer(optimizer)
        # forward + backward of one sub-batch replayed as a HIP graph (train.GraphedLossStep); the loss scaling of the
        # accumulation window (trainer.py:141) is folded into the captured backward
        self.graphed = GraphedLossStep(self.graph, diffusion, 1.0 / grad_accum_every) if use_graph else None
        self.grad_accum = 0
        self.global_step = 0
        self.allow_uneven_tasks = allow_uneven_tasks
        # task sub-batches that drew the same ``causal`` flag run as ONE pass through the network (same loss: the sum of the
        # per-task means, each sample weighted 1 / its sub-batch size); False = the reference's literal one pass per task
        self.merge_tasks = merge_tasks
        # DDP's gradient exchange (train.py:88-89): buckets in reverse execution order; in eager mode each bucket leaves as soon
        # as the backward pass has finished it, behind a replayed graph the buckets leave together right after the replay
        names = [n for n, _ in model.named_parameters()]
        self.exchange = GradExchange(optimizer, names, process_group, bucket_bytes)
        self.graph.exchange = self.exchange

    # trainer.py:215-247 / :249-278
    def random_mask(self, sequence, max_mask_length, task):
        return random_mask(sequence, max_mask_length, task, self.rng)

    def get_conditioning(self, cond):
        return get_conditioning(cond, self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids)

    def train(self, audio_emb: torch.Tensor, metadata) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """trainer.py:183-213.  Returns (sum of the task losses, {task: loss}); the per-task values stay on the device
        (the reference's ``.item()`` per sub-batch is a host sync the step does not need)."""
        loss_dict: Dict[str, torch.Tensor] = {}
        all_loss = torch.zeros((), device=self.device)
        batch_size = audio_emb.size(0)
        nt = len(self.tasks)
        if not self.allow_uneven_tasks:
            assert batch_size % nt == 0, "Batch size must be divisible by the number of tasks"       # trainer.py:187
        # BASELINE configs[3] puts 8 clips on a GPU (64 / 8), which three tasks do not divide: the first ``batch_size % 3``
        # tasks take one clip more (8 -> 3 / 3 / 2), task order as in config.py:93
        sizes = [batch_size // nt + (1 if i < batch_size % nt else 0) for i in range(nt)]
        start = 0
        parts = []                      # per task: what one pass of the reference's loop prepares (trainer.py:190-203)
        for i, task in enumerate(self.tasks):
            sub = sizes[i]
            if sub == 0:
                continue
            sub_audio_emb = audio_emb[start:start + sub]
            sub_metadata = metadata[start:start + sub]
            start += sub
            self.model.train()
            masked_input, mask, causal = self.random_mask(sub_audio_emb, sub_audio_emb.shape[2], task)
            conditioning = self.conditioner(sub_metadata, self.device)
            conditioning["masked_input"] = masked_input
            conditioning["mask"] = mask
            conditioning = self.get_conditioning(conditioning)
            t = torch.randint(0, self.diffusion.num_timesteps, (sub,), device=self.device).long()
            parts.append((task, sub_audio_emb, t, conditioning, bool(causal)))
        first = True
        if not self.merge_tasks:
            for task, x, t, conditioning, causal in parts:
                if self.graphed is not None:
                    if self.grad_accum == 0 and first:
                        self.optimizer.zero_grad()             # the captured step already contains the backward
                    loss = self.graphed(x, t, conditioning, causal)
                else:
                    loss = self.diffusion.training_loosses(self.graph, x, t, conditioning, causal=causal)
                first = False
                loss_dict[task] = loss.detach()
                all_loss = all_loss + loss
            return all_loss, loss_dict
        for flag in (False, True):
            group = [p for p in parts if p[4] == flag]
            if not group:
                continue
            cat = lambda xs: xs[0] if len(xs) == 1 else torch.cat(xs, dim=0)        # noqa: E731
            x = cat([p[1] for p in group])
            t = cat([p[2] for p in group])
            keys = group[0][3].keys()
            conditioning = {k: (None if group[0][3][k] is None else cat([p[3][k] for p in group])) for k in keys}
            w = torch.cat([torch.full((p[1].shape[0],), 1.0 / p[1].shape[0], device=self.device) for p in group])
            if self.graphed is not None:
                if self.grad_accum == 0 and first:
                    self.optimizer.zero_grad()
                per_sample = self.graphed(x, t, conditioning, flag, sample_weights=w)
                group_loss = (per_sample * w).sum()
            else:
                per_sample = self.diffusion.training_loosses(self.graph, x, t, conditioning, causal=flag, reduction="none")
                group_loss = (per_sample * w).sum()
            first = False
            all_loss = all_loss + group_loss
            o = 0
            for task, xs, *_ in group:
                loss_dict[task] = per_sample[o:o + xs.shape[0]].detach().mean()
                o += xs.shape[0]
        return all_loss, loss_dict

    def train_step(self, audio_emb: torch.Tensor, metadata) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], bool]:
        """one iteration of ``train_loop``'s body (trainer.py:134-150).  Returns (loss, per-task losses, whether an
        optimiser step was taken)."""
        last = self.grad_accum + 1 == self.grad_accum_every
        if self.graphed is None and last:
            self.exchange.begin()          # the backward pass of the window's last micro-batch releases the buckets
        all_task_loss, loss_dict = self.train(audio_emb, metadata)
        if self.graphed is None:
            if self.grad_accum == 0:
                self.optimizer.zero_grad()
            (all_task_loss / self.grad_accum_every).backward()
        self.grad_accum += 1
        stepped = False
        if self.grad_accum == self.grad_accum_every:
            if self.graphed is None:
                self.exchange.finish()                                         # DDP's exchange (train.py:88-89), overlapped
            else:
                self.exchange.blocking()                                       # behind the replayed graphs
            self.optimizer.step(None if self.lr_scheduler is None else self.lr_scheduler.get_last_lr())
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            self.grad_accum = 0
            stepped = True
        self.global_step += 1
        return all_task_loss.detach(), loss_dict, stepped

"""``UnifiedMultiTaskTrainer``: the training step of the reference around the HIP training path.

Host-side mirror of /root/reference/trainer.py:126-213 -- ``train(audio_emb, metadata)`` (three equal sub-batches, one
per task in the fixed order of config.py:93, a fresh mask / timestep draw per sub-batch, the sum of the three losses)
and the optimiser part of ``train_loop`` (trainer.py:139-150: zero_grad at the start of an accumulation window,
``(loss / grad_accum_every).backward()``, and every ``grad_accum_every`` micro-batches clip -> AdamW -> LinearLR).
Same names and argument meaning; what is NOT mirrored: the data loader, per-step logging / tensorboard and the
GradScaler (the HIP path trains in bf16 storage with float32 accumulation and float32 master weights, which needs no
loss scaling; ``skip_nonfinite`` of FusedAdamW keeps GradScaler's skip-on-overflow behaviour).

Data-parallel training (train.py:88-89): one process per GPU, every rank draws its own masks /
timesteps / noise, and the mean all-reduce of the flat gradient buffer (``optim.allreduce_gradients``, RCCL over xGMI)
runs once per optimiser step, right before the clip -- the gradient exchange is the only collective.
"""
from __future__ import annotations

import random as _random
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .optim import FusedAdamW, GradExchange, LinearLR, adopt_optimizer
from .tasks import get_conditioning, random_mask
from .train import GraphedLossStep

TASKS = ("text_guided", "music_inpaint", "music_cont")        # config.py:93


class UnifiedMultiTaskTrainer:
    """The reference's constructor (trainer.py:17-36), positional argument for positional argument, so the call site of
    train.py:110-125 works unchanged:

        UnifiedMultiTaskTrainer(config, rank, epoch_str, global_step, model, diffusion, conditioner, dls, optimizer,
                                lr_scheduler, scaler, logger, writers, grad_clip, grad_accum_every,
                                cross_attn_cond_ids=['prompt'], global_cond_ids=[], input_concat_ids=['masked_input', 'mask'])

    What each argument means here:
      config        read for ``tasks``, ``device``, ``num_epoch``, ``diffusion_type`` and
                    ``optimizer_config.lr`` (utils/config.py:84-100; ``jen1_amd.config.TrainConfig`` has the same fields)
      rank          kept (the reference's rank 0 logs: out of scope here)
      epoch_str / global_step   where ``train_loop`` resumes
      model         ``jen1_amd.model.UNetCFG1d`` (a ``.module`` wrapper, as DDP adds one, is unwrapped: the gradient exchange is
                    ``optim.GradExchange`` over ``process_group``, not a module wrapper)
      diffusion     ``GaussianDiffusion`` (``diffusion_type == 'gdm'``) or ``VDM``
      conditioner   ``conditioner(metadata, device) -> {"prompt": (embedding [b, 128, 1024], mask [b, 128])}``
                    (MultiConditioner.forward, conditioners.py:182-208)
      dls           (train_dl, valid_dl): iterables of ``(audio_emb, metadata)``
      optimizer     ``FusedAdamW``, or a ``torch.optim.AdamW`` whose hyper-parameters (one group: train.py:56-60) are taken over
                    into a ``FusedAdamW`` on the same parameters
      lr_scheduler  ``optim.LinearLR``, or torch's ``LinearLR`` (start / end factor and total_iters are taken over), or None
      scaler        accepted and NOT used: bf16 storage with float32 accumulation and master weights needs no loss scaling; its
                    skip-on-overflow behaviour is ``FusedAdamW(skip_nonfinite=True)``, switched on when an enabled scaler is passed
      logger / writers   ``logger.info`` lines and ``writer.add_scalar`` as in trainer.py:151-172 when given (None: silent)
      grad_clip     the clip norm of the fused optimiser step (nn.utils.clip_grad_norm_, trainer.py:145)
      grad_accum_every   micro-batches per optimiser step (trainer.py:139-149)
    Keyword-only extras (not in the reference): ``process_group``, ``rng``, ``compute_dtype``, ``use_graph``,
    ``allow_uneven_tasks``, ``bucket_bytes``, ``merge_tasks``, ``merge_causal``, ``grad_dtype``.  ``UnifiedMultiTaskTrainer.build(model, diffusion, conditioner,
    optimizer, ...)`` is the short form for code that has no config object."""

    def __init__(self, config, rank: int, epoch_str: int, global_step: int, model, diffusion, conditioner: Callable, dls, optimizer,
                 lr_scheduler, scaler, logger, writers, grad_clip, grad_accum_every: int,
                 cross_attn_cond_ids: Sequence[str] = ("prompt",), global_cond_ids: Sequence[str] = (),
                 input_concat_ids: Sequence[str] = ("masked_input", "mask"), *, process_group=None, rng=_random,
                 compute_dtype: Optional[str] = None, use_graph: bool = True, allow_uneven_tasks: bool = False,
                 bucket_bytes: int = 128 << 20, merge_tasks: bool = True, merge_causal: bool = True, grad_dtype: str = "f32"):
        self.config = config
        self.tasks = tuple(getattr(config, "tasks", TASKS))
        self.device = getattr(config, "device", "cuda")
        self.rank, self.epoch_str, self.global_step = rank, int(epoch_str), int(global_step)
        model = model.module if hasattr(model, "module") else model
        self.model, self.diffusion, self.conditioner = model, diffusion, conditioner
        self.train_dl, self.valid_dl = dls if dls is not None else (None, None)
        self.grad_clip, self.grad_accum_every = grad_clip, int(grad_accum_every)
        self.scaler, self.logger = scaler, logger
        self.writer, self.writer_val = writers if writers is not None else (None, None)
        self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids = cross_attn_cond_ids, global_cond_ids, input_concat_ids
        self.best_avg_total_loss = float("inf")
        self.group, self.rng = process_group, rng
        self.is_gdm = getattr(config, "diffusion_type", "gdm") == "gdm"
        scaling = bool(scaler is not None and getattr(scaler, "is_enabled", lambda: False)())
        optimizer, lr_scheduler = adopt_optimizer(list(model.parameters()), optimizer, lr_scheduler, grad_clip, scaling)
        self.optimizer, self.lr_scheduler = optimizer, lr_scheduler
        self.graph = model.train_graph(compute_dtype)
        self.graph.attach_optimizer(optimizer)
        # forward + backward of one sub-batch replayed as a HIP graph (train.GraphedLossStep); the loss scaling of the
        # accumulation window (trainer.py:141) is folded into the captured backward
        self.graphed = GraphedLossStep(self.graph, diffusion, 1.0 / self.grad_accum_every) if (use_graph and self.is_gdm) else None
        self.grad_accum = 0
        self.allow_uneven_tasks = allow_uneven_tasks
        # task sub-batches that drew the same ``causal`` flag run as ONE pass through the network (same loss: the sum of the
        # per-task means, each sample weighted 1 / its sub-batch size); False = the reference's literal one pass per task
        self.merge_tasks = merge_tasks and self.is_gdm
        # ... and sub-batches with DIFFERENT flags too: the flag travels per clip (train.CausalRows), one pass per micro-batch
        self.merge_causal = merge_causal and self.merge_tasks
        # DDP's gradient exchange (train.py:88-89): buckets in reverse execution order; in eager mode each bucket leaves as soon
        # as the backward pass has finished it, behind a replayed graph the regions leave between the segments of the replay
        names = [n for n, _ in model.named_parameters()]
        # grad_dtype="bf16": the buckets cross xGMI as bfloat16 (0.59 GB instead of 1.19 GB for the 296.5 M gradients) and are widened into
        # the float32 flat gradient again -- the optimiser, its moments and the master weights stay float32 (optim.GradExchange)
        self.exchange = GradExchange(optimizer, names, process_group, bucket_bytes, params=dict(model.named_parameters()), grad_dtype=grad_dtype)
        self.graph.exchange = self.exchange

    @classmethod
    def build(cls, model, diffusion, conditioner: Callable, optimizer: FusedAdamW, lr_scheduler: Optional[LinearLR] = None,
              grad_accum_every: int = 10, tasks: Sequence[str] = TASKS, device="cuda", cross_attn_cond_ids: Sequence[str] = ("prompt",),
              global_cond_ids: Sequence[str] = (), input_concat_ids: Sequence[str] = ("masked_input", "mask"), dls=None, save_dir: str = "",
              eval_interval: int = 30, num_epoch: int = 100, **extras):
        """the trainer without a config object / logger / writers (tests, bench.py): same object, short argument list"""
        from .config import TrainConfig
        cfg = TrainConfig(tasks=list(tasks), device=device, grad_accum_every=grad_accum_every, save_dir=save_dir, eval_interval=eval_interval,
                          num_epoch=num_epoch)
        return cls(cfg, 0, 0, 0, model, diffusion, conditioner, dls, optimizer, lr_scheduler, None, None, None, optimizer.max_norm,
                   grad_accum_every, cross_attn_cond_ids, global_cond_ids, input_concat_ids, **extras)

    # trainer.py:215-247 / :249-278
    def random_mask(self, sequence, max_mask_length, task):
        return random_mask(sequence, max_mask_length, task, self.rng)

    def get_conditioning(self, cond):
        return get_conditioning(cond, self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids)

    def prepare_parts(self, audio_emb: torch.Tensor, metadata) -> List[tuple]:
        """what one pass of the reference's task loop prepares (trainer.py:190-203): per task
        ``(task, sub_audio_emb, t, conditioning, causal)`` with a fresh mask / timestep draw per sub-batch"""
        batch_size = audio_emb.size(0)
        nt = len(self.tasks)
        if not self.allow_uneven_tasks:
            assert batch_size % nt == 0, "Batch size must be divisible by the number of tasks"       # trainer.py:187
        # BASELINE configs[3] puts 8 clips on a GPU (64 / 8), which three tasks do not divide: the first ``batch_size % 3``
        # tasks take one clip more (8 -> 3 / 3 / 2), task order as in config.py:93
        sizes = [batch_size // nt + (1 if i < batch_size % nt else 0) for i in range(nt)]
        start = 0
        parts = []
        for i, task in enumerate(self.tasks):
            sub = sizes[i]
            if sub == 0:
                continue
            sub_audio_emb = audio_emb[start:start + sub]
            sub_metadata = metadata[start:start + sub]
            start += sub
            masked_input, mask, causal = self.random_mask(sub_audio_emb, sub_audio_emb.shape[2], task)
            conditioning = self.conditioner(sub_metadata, self.device)
            conditioning["masked_input"] = masked_input
            conditioning["mask"] = mask
            conditioning = self.get_conditioning(conditioning)
            t = torch.randint(0, self.diffusion.num_timesteps, (sub,), device=self.device).long() if self.is_gdm else None
            parts.append((task, sub_audio_emb, t, conditioning, bool(causal)))
        return parts

    def train(self, audio_emb: torch.Tensor, metadata) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """trainer.py:183-213.  Returns (sum of the task losses, {task: loss}); the per-task values stay on the device
        (the reference's ``.item()`` per sub-batch is a host sync the step does not need)."""
        self.model.train()
        return self.run_parts(self.prepare_parts(audio_emb, metadata))

    def run_parts(self, parts: List[tuple], noises: Optional[Dict[str, torch.Tensor]] = None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """the loss of prepared task sub-batches (and, with a replayed graph, its backward).  ``noises`` {task: noise} injects the
        diffusion noise (parity tests against the reference's autograd; eager mode only -- the graph draws its own)."""
        loss_dict: Dict[str, torch.Tensor] = {}
        all_loss = torch.zeros((), device=self.device)
        assert noises is None or self.graphed is None, "injected noise needs use_graph=False"
        first = True
        armed = getattr(self, "_armed", False)                 # train_step: this micro-batch closes an accumulation window
        if not self.merge_tasks:
            for i, (task, x, t, conditioning, causal) in enumerate(parts):
                if self.graphed is not None:
                    if self.grad_accum == 0 and first:
                        self.optimizer.zero_grad()             # the captured step already contains the backward
                    # the replayed pass that completes the window's gradients carries the exchange (GraphedLossStep.exchange)
                    self.graphed.exchange = self.exchange if (armed and i == len(parts) - 1) else None
                    loss = self.graphed(x, t, conditioning, causal)
                elif self.is_gdm:
                    loss = self.diffusion.training_loosses(self.graph, x, t, conditioning, causal=causal,
                                                           noise=None if noises is None else noises[task])
                else:                                          # VDM draws its own continuous times (trainer.py:209-211)
                    loss = self.diffusion.training_loosses(self.graph, x, conditioning, causal=causal,
                                                           noise=None if noises is None else noises[task])
                first = False
                loss_dict[task] = loss.detach()
                all_loss = all_loss + loss
            return all_loss, loss_dict
        flags = [f for f in (False, True) if any(p[4] == f for p in parts)]
        if len(flags) == 2 and self.merge_causal and self.graph.per_clip_causal_ok(parts[0][1].shape[-1], parts[0][3]["cross_attn_cond"].shape[1]):
            flags = ["per clip"]               # ONE pass for all sub-batches: the causal flag travels per clip (train.CausalRows)
        for flag in flags:
            group = parts if flag == "per clip" else [p for p in parts if p[4] == flag]
            cat = lambda xs: xs[0] if len(xs) == 1 else torch.cat(xs, dim=0)        # noqa: E731
            x = cat([p[1] for p in group])
            t = cat([p[2] for p in group])
            keys = group[0][3].keys()
            conditioning = {k: (None if group[0][3][k] is None else cat([p[3][k] for p in group])) for k in keys}
            w = torch.cat([torch.full((p[1].shape[0],), 1.0 / p[1].shape[0], device=self.device) for p in group])
            if flag == "per clip":
                flag = torch.cat([torch.full((p[1].shape[0],), int(p[4]), dtype=torch.int32, device=self.device) for p in group])
            if self.graphed is not None:
                if self.grad_accum == 0 and first:
                    self.optimizer.zero_grad()
                self.graphed.exchange = self.exchange if (armed and (torch.is_tensor(flag) or flag == flags[-1])) else None
                per_sample = self.graphed(x, t, conditioning, flag, sample_weights=w)
                group_loss = (per_sample * w).sum()
            else:
                noise = None if noises is None else cat([noises[p[0]] for p in group])
                per_sample = self.diffusion.training_loosses(self.graph, x, t, conditioning, causal=flag, reduction="none", noise=noise)
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
        if last:
            self.exchange.begin()          # the backward pass of the window's last micro-batch releases the regions
        self._armed = last
        try:
            all_task_loss, loss_dict = self.train(audio_emb, metadata)
        finally:
            self._armed = False
            if self.graphed is not None:
                self.graphed.exchange = None
        if self.graphed is None:
            if self.grad_accum == 0:
                self.optimizer.zero_grad()
            (all_task_loss / self.grad_accum_every).backward()
        self.grad_accum += 1
        stepped = False
        if self.grad_accum == self.grad_accum_every:
            self.exchange.finish()                                             # DDP's exchange (train.py:88-89): what is still out
            self.optimizer.step(None if self.lr_scheduler is None else self.lr_scheduler.get_last_lr())
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            self.grad_accum = 0
            stepped = True
        self.global_step += 1
        return all_task_loss.detach(), loss_dict, stepped

    # ------------------------------------------------------------------ trainer.py:94-124
    @torch.no_grad()
    def eval(self) -> Tuple[Dict[str, float], int]:
        """trainer.py:94-124: the validation loss per task over ``self.valid_dl`` under no_grad (the denoiser runs on the sampling
        engine: ``training_loosses(model, ...)`` without gradients).  Returns ({task: summed loss}, batches)."""
        self.model.eval()
        loss_dict = {task: 0.0 for task in self.tasks}
        count = 0
        for audio_emb, metadata in (self.valid_dl if self.valid_dl is not None else ()):
            for task, x, t, conditioning, causal in self.prepare_parts(audio_emb.to(self.device), metadata):
                if self.is_gdm:
                    loss = self.diffusion.training_loosses(self.model, x, t, conditioning, causal=causal)
                else:
                    loss = self.diffusion.training_loosses(self.model, x, conditioning, causal=causal)
                loss_dict[task] += float(loss)
            count += 1
        return loss_dict, count

    def eval_all_tasks(self, epoch) -> float:
        """trainer.py:61-92: average validation loss per task; a new best total writes a checkpoint in the reference's wire format
        (``checkpoint.save_checkpoint`` = script_util.py:79-90, same file name pattern).  Logger / TensorBoard writer are optional;
        only rank 0 writes the file (the reference lets every rank write one under its own loss value)."""
        import os

        from .checkpoint import save_checkpoint
        loss_dict, count = self.eval()
        avg_total = 0.0
        for task in self.tasks:
            avg = loss_dict[task] / count if count > 0 else 0.0
            avg_total += avg
            if self.logger is not None:
                self.logger.info(f"Average validation loss for task {task}: {avg}")
            if self.rank == 0 and self.writer is not None:
                self.writer.add_scalar(f"loss/val_{task}", avg, self.global_step)
        if self.logger is not None:
            self.logger.info(f"Average total validation loss: {avg_total}")
        if avg_total < self.best_avg_total_loss:
            self.best_avg_total_loss = avg_total
            save_dir = getattr(self.config, "save_dir", None)
            if save_dir and self.rank == 0:
                os.makedirs(save_dir, exist_ok=True)
                oc = getattr(self.config, "optimizer_config", None)
                lr = getattr(oc, "lr", None) if oc is not None else None
                if lr is None:
                    lr = self.optimizer.lr if hasattr(self.optimizer, "lr") else None
                self.last_checkpoint = os.path.join(save_dir, f"Jen1_step_{self.global_step}_loss_{self.best_avg_total_loss}.pth")
                save_checkpoint(model=self.model, optimizer=self.optimizer, lr=lr, iteration=epoch, checkpoint_path=self.last_checkpoint,
                                logger=self.logger)
        if self.rank == 0 and self.writer is not None:
            self.writer.add_scalar("loss/val_total", avg_total, self.global_step)
        self.model.train()
        return avg_total

    # ------------------------------------------------------------------ trainer.py:126-181
    def train_loop(self, max_steps: Optional[int] = None) -> None:
        """``train_loop`` (trainer.py:126-181) over ``self.train_dl``, so that train.py:110-125's call site works unchanged: the
        micro-batch loop, ``eval_all_tasks`` every ``config.eval_interval`` micro-batches (:174-175) and once at the end (:181) --
        which is where the reference writes its best-so-far checkpoint.  The per-step logging / TensorBoard scalars (:151-172) are out
        of scope.  ``max_steps`` (not in the reference) bounds the number of micro-batches."""
        done = 0
        epoch = self.epoch_str
        interval = int(getattr(self.config, "eval_interval", 0) or 0)
        for epoch in range(self.epoch_str, int(self.epoch_str + int(getattr(self.config, "num_epoch", 1)) + 1)):
            for audio_emb, metadata in self.train_dl:
                step_before = self.global_step
                self.train_step(audio_emb.to(self.device), metadata)
                if interval > 0 and step_before % interval == 0 and step_before != 0:
                    self.eval_all_tasks(epoch=epoch)
                done += 1
                if max_steps is not None and done >= max_steps:
                    self.eval_all_tasks(epoch=epoch)
                    return
        self.eval_all_tasks(epoch=epoch)

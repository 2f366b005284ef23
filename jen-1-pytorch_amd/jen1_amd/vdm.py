"""``VDM``: the reference's second sampler / loss (``/root/reference/jen1/diffusion/vdm/vdm.py``), REPAIRED, on the HIP denoiser.

``Jen1.generate(use_gdm=False)`` -- the reference's default -- constructs this class (generation.py:54-58) and
``UnifiedMultiTaskTrainer`` uses it when ``config.diffusion_type != 'gdm'`` (trainer.py:209-211).  As shipped it cannot run
(both reproduced by running the reference):

  A-3  ``p_sample`` passes the 0-dim float ``time`` to the model, whose time embedding needs a batch vector
       (``rearrange('b -> b 1')`` fails), and indexes ``self.alphas[time]`` with that FLOAT tensor (vdm.py:44, :52-54);
  A-4  ``training_loosses`` multiplies ``alphas[B]`` / ``sigmas[B]`` against ``[B, C, T]`` tensors (vdm.py:86-87, :95): a shape
       error unless T == B.

The repair is the smallest one that makes the written formulas run, and it is a documented DEVIATION from the reference's text:
  * the model receives ``time.expand(B)`` (one continuous time per batch row, float32);
  * ``alphas`` / ``sigmas`` of a step are read by the step INDEX i (``self.alphas[i]``, ``[i + 1]``), which is what indexing by
    "the time of step i" can only mean;
  * in the loss, ``alphas`` / ``sigmas`` are reshaped to ``[B, 1, 1]`` before they multiply ``[B, C, T]`` tensors.
Everything else is as written: cosine ``alpha = cos(t pi / 2)``, ``sigma = sin(t pi / 2)`` over ``linspace(1, 0, step + 1)``
(vdm.py:39-41, :64-65), the v-parameterised update ``x_pred = alpha x - sigma v``, ``noise_pred = sigma x + alpha v``,
``x = alpha' x_pred + sigma' noise_pred`` with NO clamp (vdm.py:52-55), CFG dropout at sampling time (vdm.py:47), UNIFORM
training noise (vdm.py:80, :91) and the target ``noise alpha - x_t sigma`` (vdm.py:106) -- which uses x_t where the usual v target
has x_0; reproduced, not "fixed".  Parity is pinned against exactly this repair applied to the reference's own class
(``tests/golden/make_golden.py vdm`` subclasses the reference ``VDM`` and overrides the two broken methods with the lines above).

Sampling on ``jen1_amd.model.UNetCFG1d`` runs the fused stepper of ``diffusion.DDIMStepper`` (mode "vdm": one replayed graph per
step, row kind 3 of ``jen1_cfg_ddim_step``, continuous times through ``jen1_time_features_f32``); any other callable runs the
literal loop.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .diffusion import DDIMStepper, _check_model_errors
from .model import _STEPPER_CACHES, UNetCFG1d


class VDM(torch.nn.Module):
    def __init__(self, *, loss_type, device, cfg_dropout_proba=0.1, embedding_scale=0.8, batch_cfg=False, scale_cfg=False,
                 use_fp16=False):
        super().__init__()
        self.device = torch.device(device)
        self.cfg_dropout_proba, self.embedding_scale = cfg_dropout_proba, embedding_scale
        self.batch_cfg, self.scale_cfg, self.use_fp16 = batch_cfg, scale_cfg, use_fp16
        assert loss_type in {"l1", "l2"}
        self.loss_fn = F.l1_loss if loss_type == "l1" else F.mse_loss
        self.objective = "v"
        self._steps = 100

    # vdm.py:39-41
    def get_alpha_sigma(self, t):
        self.alphas = torch.cos(t * math.pi / 2)
        self.sigmas = torch.sin(t * math.pi / 2)

    def _call(self, model, x, t, conditioning, causal, dropout_rows=None):
        kw = dict(embedding=conditioning["cross_attn_cond"], embedding_mask=conditioning["cross_attn_masks"],
                  embedding_scale=self.embedding_scale, embedding_mask_proba=self.cfg_dropout_proba,
                  features=conditioning["global_cond"], channels_list=[conditioning["input_concat_cond"]],
                  batch_cfg=self.batch_cfg, scale_cfg=self.scale_cfg, causal=causal)
        if dropout_rows is not None:
            kw["dropout_rows"] = dropout_rows
        return model(x, t, **kw)

    # ------------------------------------------------------------------ sampling (vdm.py:42-79)
    def coeff_table(self, step: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """rows {alpha_i, sigma_i, alpha_{i+1}, sigma_{i+1}, 0, 3 (row kind), 0, 0} and the continuous times of the ``step`` steps,
        evaluated in float32 on the host like the reference's CPU path (vdm.py:64-65)"""
        step = self._steps if step is None else step
        steps = torch.linspace(1., 0., step + 1)
        al, sg = torch.cos(steps * math.pi / 2), torch.sin(steps * math.pi / 2)
        rows = [[al[i].item(), sg[i].item(), al[i + 1].item(), sg[i + 1].item(), 0.0, 3.0, 0.0, 0.0] for i in range(step)]
        return (torch.tensor(rows, dtype=torch.float32, device=self.device).reshape(step, 8), steps[:step].to(self.device, torch.float32))

    @torch.no_grad()
    def p_sample(self, x, i: int, model, conditioning, causal, dropout_rows=None):
        """one step of the repaired loop: ``i`` is the step index (see the module docstring)"""
        time = self.steps[i].expand(x.shape[0])
        v_pred = self._call(model, x, time, conditioning, causal, dropout_rows)
        x_pred = self.alphas[i] * x - self.sigmas[i] * v_pred
        noise_pred = self.sigmas[i] * x + self.alphas[i] * v_pred
        return self.alphas[i + 1] * x_pred + self.sigmas[i + 1] * noise_pred

    @torch.no_grad()
    def p_sample_loop(self, model, shape, conditioning, step=1000, return_all_timesteps=False, init_data=None, causal=False, *,
                      init_noise=None, dropout_rows: Optional[Sequence[torch.Tensor]] = None, use_graph: bool = True, fused: bool = True):
        audio = torch.randn(shape, device=self.device) if init_noise is None else init_noise.to(self.device, torch.float32).reshape(shape)
        if init_data is not None:
            audio = audio + init_data
        self.steps = torch.linspace(1., 0., step + 1, device=self.device)
        self.get_alpha_sigma(self.steps)
        fused = fused and isinstance(model, UNetCFG1d) and not (self.embedding_scale != 1.0 and not self.batch_cfg)
        audios = [audio]
        if fused:
            self._steps = step
            # one stepper (plan + schedule tables + captured graph) per (model, shape, causal, steps): later calls rebind the conditioning
            cache = self.__dict__.setdefault("_steppers", {})
            _STEPPER_CACHES.add(self)            # (an engine invalidation drops this model's steppers: model._drop_steppers_of)
            key = (id(model), id(model.engine()), tuple(shape), bool(causal), bool(use_graph), int(step), float(self.embedding_scale),
                   bool(self.batch_cfg), bool(self.scale_cfg), bool(model.deterministic))
            st = cache.get(key)
            if st is not None and st.model is model and st.eng is model.engine():
                st.rebind(conditioning)
            else:
                st = DDIMStepper(self, model, shape, conditioning, causal, use_graph, None, 0, "vdm")
                if len(cache) >= 8:
                    cache.pop(next(iter(cache)))
                cache[key] = st
            st.reset(audio)
        B = shape[0]
        for i in range(step):
            drop = None
            if self.cfg_dropout_proba > 0.0:
                if dropout_rows is not None:
                    drop = torch.as_tensor(dropout_rows[i])
                elif fused:          # (the literal loop lets the model draw them, like the reference)
                    drop = (torch.ones(B, dtype=torch.bool) if self.cfg_dropout_proba >= 1.0 else
                            torch.bernoulli(torch.full((B,), float(self.cfg_dropout_proba), device=self.device)).to(torch.bool))
            if fused:
                st.step(i, drop_rows=drop, set_rows=self.cfg_dropout_proba > 0.0)
                if return_all_timesteps:
                    audios.append(st.x.clone())
            else:
                audio = self.p_sample(audio, i, model, conditioning, causal, drop)
                audios.append(audio)
        if fused:
            audio = st.x.clone()
            st.check()
        else:
            _check_model_errors(model)           # the literal loop's last call (diffusion._check_model_errors)
        return audio if not return_all_timesteps else torch.stack(audios, dim=1)

    @torch.no_grad()
    def sample(self, model, shape, conditioning, step=100, return_all_timesteps=False, causal=False, init_data=None, **kw):
        """vdm.py:77-79"""
        return self.p_sample_loop(model, shape, conditioning, step, return_all_timesteps=return_all_timesteps, init_data=init_data,
                                  causal=causal, **kw)

    # ------------------------------------------------------------------ training (vdm.py:81-110)
    def q_sample(self, x_start, times, noise=None):
        """q(x_t | x_0) with the per-sample times broadcast over (C, T) (repair A-4); default noise is UNIFORM, as written"""
        if noise is None:
            noise = torch.rand_like(x_start)
        alphas, sigmas = torch.cos(times * math.pi / 2), torch.sin(times * math.pi / 2)
        shape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
        alphas, sigmas = alphas.reshape(shape), sigmas.reshape(shape)
        return x_start * alphas + noise * sigmas, alphas, sigmas

    def training_loosses(self, model, x_start, conditioning, noise=None, causal=False, *, times=None, dropout_rows=None):
        """vdm.py:89-110; ``times`` / ``dropout_rows`` inject the draws (parity tests)"""
        if noise is None:
            noise = torch.rand_like(x_start)
        if times is None:
            times = torch.rand(x_start.shape[0], device=self.device)
        times = times.to(self.device, torch.float32)
        x_t, alphas, sigmas = self.q_sample(x_start, times, noise=noise)
        if torch.is_grad_enabled() and getattr(model, "training", False) and hasattr(model, "train_graph"):
            model = model.train_graph()          # the differentiable HIP path (jen1_amd/train.py)
        model_out = self._call(model, x_t, times, conditioning, causal, dropout_rows)
        target = noise * alphas - x_t * sigmas
        loss = self.loss_fn(model_out, target, reduction="none")
        return loss.reshape(loss.shape[0], -1).mean(dim=1).mean()

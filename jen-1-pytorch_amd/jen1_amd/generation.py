"""``Jen1``: the top-level generate() surface of the reference around the HIP denoiser path.

Host-side mirror of /root/reference/generation.py:16-192 -- same constructor arguments, ``get_model_and_diffusion``,
``generate(prompt, seed, steps, batch_size, seconds, use_gdm, task, init_audio, init_audio_sr, inpainting_scope)``,
``get_mask``, ``get_emb``, ``get_conditioning`` -- with the two third-party models the reference constructs itself
passed in instead (they are outside this build):

  * ``audio_encoder``: the object the reference gets from ``EncodecModel.encodec_model_48khz()``; used exactly as
    generation.py uses it: ``.channels``, ``.encode(audio) -> [(codes, scale)]``, ``.quantizer.decode(codes)``,
    ``.decoder(emb)``  (generation.py:34, :95, :113, :130, :145-150);
  * ``conditioner``: the ``MultiConditioner`` of ``create_multi_conditioner`` (generation.py:29, :121-122),
    ``conditioner(batch_metadata, device) -> {"prompt": (emb [B,128,1024], mask [B,128])}``;
  * ``convert_audio``: ``encodec.utils.convert_audio`` (generation.py:95), identity by default.

Everything between them -- the masks, the conditioning dict, the 100-step DDIM loop over the UNet with the CFG pair,
captured as one HIP graph per shape -- runs on libjen1_hip.so through ``GaussianDiffusion.sample``.

Differences, all deliberate and visible:
  * ``use_gdm=False`` (the reference default) selects ``VDM``, which cannot run in the reference either (SURVEY.md
    Appendix A-3 / A-4); it raises ``NotImplementedError`` here instead of failing inside the sampler;
  * the reference reads ``flag`` before assignment when ``init_audio`` is given (generation.py:91-120); here
    ``flag`` is False in that case, i.e. the given audio is the ``init_data`` of the sampler, which is what the code
    evidently means;
  * ``init_audio.size() != 3`` (generation.py:85) compares a ``torch.Size`` with an int and is always true, so the
    reference repeats even batched audio ``batch_size`` times; here only audio without a batch axis is repeated;
  * ``music_cont`` indexes the mask with ``mask[:, cont_start:]`` on dim 1 (generation.py:106), which is a no-op slice of
    a size-1 axis that only works for cont_start == 0; here the time axis is sliced.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .checkpoint import load_checkpoint
from .config import GDMConfig, full_model_config
from .diffusion import GaussianDiffusion, get_beta_schedule
from .model import UNetCFG1d
from .tasks import get_conditioning, get_mask


class Jen1:
    def __init__(self, ckpt_path: Optional[str], device: str = "cuda", sample_rate: int = 48000,
                 cross_attn_cond_ids: Sequence[str] = ("prompt",), global_cond_ids: Sequence[str] = (),
                 input_concat_ids: Sequence[str] = ("masked_input", "mask"), *, audio_encoder, conditioner: Callable,
                 convert_audio: Optional[Callable] = None, model_config: Optional[dict] = None,
                 diffusion_config: Optional[GDMConfig] = None, compute_dtype: str = "bf16"):
        self.ckpt_path, self.device, self.sample_rate = ckpt_path, device, sample_rate
        self.conditioner, self.audio_encoder = conditioner, audio_encoder
        self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids = cross_attn_cond_ids, global_cond_ids, input_concat_ids
        self.convert_audio = convert_audio or (lambda wav, sr, target_sr, target_channels: wav)
        self.model_config = dict(model_config or full_model_config())
        self.diffusion_config = diffusion_config or GDMConfig()
        self.compute_dtype = compute_dtype
        self._model: Optional[UNetCFG1d] = None
        self.batch_size = 1

    # generation.py:36-74
    def get_model_and_diffusion(self, steps: int, use_gdm: bool):
        if not use_gdm:
            raise NotImplementedError("VDM sampling is broken in the reference (SURVEY.md Appendix A-3/A-4); pass use_gdm=True")
        dc = self.diffusion_config
        betas, alphas = get_beta_schedule(dc.noise_schedule, dc.steps)
        diffusion = GaussianDiffusion(steps=dc.steps, betas=betas.to(self.device, torch.float32), alphas=alphas, objective=dc.objective,
                                      loss_type=dc.loss_type, device=self.device, cfg_dropout_proba=dc.cfg_dropout_proba,
                                      embedding_scale=dc.embedding_scale, batch_cfg=dc.batch_cfg, scale_cfg=dc.scale_cfg,
                                      sampling_timesteps=steps, use_fp16=False)
        if self._model is None:          # the reference re-creates and re-loads the model on every call; once is enough
            cfg = dict(self.model_config)
            model = UNetCFG1d(context_embedding_features=cfg.pop("context_embedding_features", None),
                              context_embedding_max_length=cfg.pop("context_embedding_max_length", None),
                              compute_dtype=self.compute_dtype, device=self.device, **cfg)
            if self.ckpt_path is not None:
                model, _, _, _ = load_checkpoint(self.ckpt_path, model)
            self._model = model.eval()
        return diffusion, self._model

    # generation.py:76-132
    def generate(self, prompt, seed: int = -1, steps: int = 100, batch_size: int = 1, seconds: int = 30, use_gdm: bool = False,
                 task: str = "text_guided", init_audio: Optional[torch.Tensor] = None, init_audio_sr: Optional[int] = None,
                 inpainting_scope=None) -> torch.Tensor:
        seed = seed if seed != -1 else int(np.random.randint(0, 2 ** 32 - 1))
        torch.manual_seed(seed)
        self.batch_size = batch_size
        diffusion, model = self.get_model_and_diffusion(steps, use_gdm)
        channels = self.audio_encoder.channels
        sample_length = seconds * self.sample_rate
        flag = False
        if init_audio is not None and init_audio.dim() != 3:
            init_audio = init_audio.repeat(batch_size, 1, 1)
        if init_audio is None:
            flag = True
            init_audio = torch.zeros((batch_size, channels, sample_length))
            init_audio_sr = self.sample_rate
        init_audio = self.convert_audio(init_audio, init_audio_sr, self.sample_rate, channels)
        if task == "text_guided":
            mask = self.get_mask(sample_length, 0, seconds, batch_size)
            causal = False
        elif task == "music_inpaint":
            mask = self.get_mask(sample_length, inpainting_scope[0], inpainting_scope[1], batch_size)
            causal = False
        elif task == "music_cont":
            cont_length = sample_length - init_audio.size(2)
            cont_start = init_audio.size(2)
            mask = self.get_mask(sample_length, cont_start / self.sample_rate, seconds, batch_size)
            cont_audio = torch.randn(batch_size, channels, cont_length, device=init_audio.device)
            cont_audio = cont_audio * mask[:, :, cont_start:].to(cont_audio.device)
            init_audio = torch.cat([init_audio, cont_audio], dim=2)
            causal = True
        else:
            raise ValueError(f"unknown task {task!r}")
        with torch.no_grad():
            init_emb = self.get_emb(init_audio.to(self.device)).to(self.device)
            emb_shape = init_emb.shape
            mask = torch.nn.functional.interpolate(mask.to(self.device), size=(emb_shape[2]))
            masked_emb = init_emb * mask
            if flag:
                init_emb = None
            batch_metadata = [{"prompt": prompt} for _ in range(batch_size)]
            conditioning = self.conditioner(batch_metadata, self.device)
            conditioning["masked_input"] = masked_emb
            conditioning["mask"] = mask
            conditioning = self.get_conditioning(conditioning)
            sample_embs = diffusion.sample(model, tuple(emb_shape), conditioning, causal=causal, init_data=init_emb)
            samples = self.audio_encoder.decoder(sample_embs.to("cpu"))
        return samples

    # generation.py:134-150
    def get_mask(self, sample_size: int, start: float, end: float, batch_size: int) -> torch.Tensor:
        return get_mask(sample_size, start, end, batch_size, self.sample_rate)

    def get_emb(self, audio: torch.Tensor) -> torch.Tensor:
        encoded_frames = self.audio_encoder.encode(audio)
        codes = torch.cat([encoded[0] for encoded in encoded_frames], dim=-1)
        codes = codes.transpose(0, 1)
        return self.audio_encoder.quantizer.decode(codes)

    # generation.py:152-192
    def get_conditioning(self, cond):
        """as written in the reference: input-concat entries are read as ``cond[key][0]`` -- the FIRST batch element --
        and expanded over the batch (generation.py:173-180)"""
        return get_conditioning(cond, self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids, batch_size=self.batch_size)

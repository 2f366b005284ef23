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
  * ``use_gdm=False`` (the reference default) selects ``VDM``, which cannot run in the reference (SURVEY.md Appendix
    A-3 / A-4); here it runs the REPAIRED sampler of jen1_amd/vdm.py (same formulas, the three repairs listed there).  The
    reference also passes ``causal`` positionally into VDM.sample's ``step`` slot (generation.py:128 vs vdm.py:77), which would
    run zero steps; ``steps`` and ``causal`` go to their own parameters here;
  * the reference reads ``flag`` before assignment when ``init_audio`` is given (generation.py:91-120); here
    ``flag`` is False in that case, i.e. the given audio is the ``init_data`` of the sampler, which is what the code
    evidently means;
  * ``init_audio.size() != 3`` (generation.py:85) compares a ``torch.Size`` with an int and is always true, so the
    reference repeats even batched audio ``batch_size`` times; here only audio without a batch axis is repeated;
  * ``music_cont`` appends ``randn * mask[:, cont_start:]`` to the prefix (generation.py:105-107; the slice is on the
    size-1 channel axis); the mask is 0 over the whole extension and the extension is multiplied by the mask again before it
    reaches the network, so zeros are appended here (no noise draw: the sampler's own draws start at the same generator state
    only if the reference's draw is skipped too -- the test compares against the hand-built call, not a bit-stream);
  * the decoder gets the latents on its own device (``audio_encoder.decoder_device``, default "cpu" as in
    generation.py:129): with ``EncodecHIP`` they never leave HBM.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Sequence, Tuple

import numpy as np
import torch

from .checkpoint import load_checkpoint
from .config import GDMConfig, VDMConfig, full_model_config
from .diffusion import GaussianDiffusion, get_beta_schedule
from .model import UNetCFG1d
from .tasks import get_conditioning, get_mask


class Jen1:
    def __init__(self, ckpt_path: Optional[str], device: str = "cuda", sample_rate: int = 48000,
                 cross_attn_cond_ids: Sequence[str] = ("prompt",), global_cond_ids: Sequence[str] = (),
                 input_concat_ids: Sequence[str] = ("masked_input", "mask"), *, audio_encoder, conditioner: Callable,
                 convert_audio: Optional[Callable] = None, model_config: Optional[dict] = None,
                 diffusion_config: Optional[GDMConfig] = None, compute_dtype: str = "bf16", vdm_config: Optional[VDMConfig] = None):
        self.ckpt_path, self.device, self.sample_rate = ckpt_path, device, sample_rate
        self.conditioner, self.audio_encoder = conditioner, audio_encoder
        self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids = cross_attn_cond_ids, global_cond_ids, input_concat_ids
        self.convert_audio = convert_audio or (lambda wav, sr, target_sr, target_channels: wav)
        self.model_config = dict(model_config or full_model_config())
        self.diffusion_config = diffusion_config or GDMConfig()
        self.vdm_config = vdm_config or VDMConfig()
        self.compute_dtype = compute_dtype
        self._model: Optional[UNetCFG1d] = None
        self.batch_size = 1

    # generation.py:36-74
    def get_model_and_diffusion(self, steps: int, use_gdm: bool):
        if use_gdm:
            dc = self.diffusion_config
            betas, alphas = get_beta_schedule(dc.noise_schedule, dc.steps)
            diffusion = GaussianDiffusion(steps=dc.steps, betas=betas.to(self.device, torch.float32), alphas=alphas, objective=dc.objective,
                                          loss_type=dc.loss_type, device=self.device, cfg_dropout_proba=dc.cfg_dropout_proba,
                                          embedding_scale=dc.embedding_scale, batch_cfg=dc.batch_cfg, scale_cfg=dc.scale_cfg,
                                          sampling_timesteps=steps, use_fp16=False)
        else:                              # generation.py:54-58: the variational_diffusion block of the config
            from .vdm import VDM
            vc = self.vdm_config
            diffusion = VDM(loss_type=vc.loss_type, device=self.device, cfg_dropout_proba=vc.cfg_dropout_proba,
                            embedding_scale=vc.embedding_scale, batch_cfg=vc.batch_cfg, scale_cfg=vc.scale_cfg, use_fp16=False)
        if self._model is None:          # the reference re-creates and re-loads the model on every call; once is enough
            cfg = dict(self.model_config)
            model = UNetCFG1d(context_embedding_features=cfg.pop("context_embedding_features", None),
                              context_embedding_max_length=cfg.pop("context_embedding_max_length", None),
                              compute_dtype=self.compute_dtype, device=self.device, **cfg)
            if self.ckpt_path is not None:
                model, _, _, _ = load_checkpoint(self.ckpt_path, model)
            self._model = model.eval()
        return diffusion, self._model

    # ------------------------------------------------------------------ generate (generation.py:76-132)
    #
    # A request is planned in three independent pieces and only then touches the device:
    #   _task_window   which seconds are to be generated (the mask is 0 there) and whether the denoiser runs causally
    #   _known_audio   the waveform whose latents are "known": silence, the given audio, or the given prefix + placeholder
    #   _sample        latents of the known audio -> conditioning dict -> DDIM loop on the HIP path -> decoder
    def _task_window(self, task: str, seconds: float, inpainting_scope, prefix_samples: int) -> Tuple[float, float, bool]:
        """(start_s, end_s, causal): text_guided generates everything, music_inpaint the given scope, music_cont everything
        behind the given prefix with the causal network (generation.py:96-110)"""
        if task == "text_guided":
            return 0.0, float(seconds), False
        if task == "music_inpaint":
            if inpainting_scope is None or len(inpainting_scope) != 2:
                raise ValueError("music_inpaint needs inpainting_scope=(start_s, end_s)")
            return float(inpainting_scope[0]), float(inpainting_scope[1]), False
        if task == "music_cont":
            return prefix_samples / self.sample_rate, float(seconds), True
        raise ValueError(f"unknown task {task!r}")

    def _known_audio(self, task: str, init_audio: Optional[torch.Tensor], init_audio_sr: Optional[int], batch_size: int,
                     total_samples: int) -> Tuple[torch.Tensor, bool, int]:
        """([B, channels, n] waveform in the model's sample rate / channel count, whether it is only a placeholder, the number of
        samples of the given audio AFTER conversion to the model's sample rate -- generation.py:103 reads ``init_audio.size(2)``
        behind ``convert_audio``, so the continuation starts where the resampled prefix ends).
        Without ``init_audio`` the known audio is silence and the sampler starts from noise; audio without a batch axis is
        repeated over the batch; for music_cont the prefix is extended to the full length (the extension is masked out, its
        content never reaches the network)."""
        channels = self.audio_encoder.channels
        if init_audio is None:
            return torch.zeros((batch_size, channels, total_samples)), True, 0
        if init_audio.dim() == 2:
            init_audio = init_audio.unsqueeze(0).expand(batch_size, -1, -1)
        wav = self.convert_audio(init_audio, init_audio_sr, self.sample_rate, channels)
        prefix = int(wav.shape[2])
        if task == "music_cont":
            missing = total_samples - wav.shape[2]
            if missing < 0:
                raise ValueError("music_cont: init_audio is longer than the requested duration")
            # the reference appends noise * mask here (generation.py:105-107); the mask is 0 over the whole extension
            wav = torch.cat([wav, wav.new_zeros((wav.shape[0], wav.shape[1], missing))], dim=2)
        return wav, False, prefix

    def generate(self, prompt, seed: int = -1, steps: int = 100, batch_size: int = 1, seconds: int = 30, use_gdm: bool = False,
                 task: str = "text_guided", init_audio: Optional[torch.Tensor] = None, init_audio_sr: Optional[int] = None,
                 inpainting_scope=None) -> torch.Tensor:
        torch.manual_seed(seed if seed != -1 else int(np.random.randint(0, 2 ** 32 - 1)))
        self.batch_size = batch_size
        diffusion, model = self.get_model_and_diffusion(steps, use_gdm)
        total = int(seconds * self.sample_rate)
        wav, placeholder, prefix = self._known_audio(task, init_audio, init_audio_sr, batch_size, total)
        start_s, end_s, causal = self._task_window(task, seconds, inpainting_scope, prefix)
        keep = self.get_mask(total, start_s, end_s, batch_size)                 # 1 = keep the known audio, 0 = generate
        return self._sample(diffusion, model, prompt, wav, keep, causal, seed_with_audio=not placeholder, steps=steps)

    @torch.no_grad()
    def _sample(self, diffusion, model, prompt, wav: torch.Tensor, keep: torch.Tensor, causal: bool, seed_with_audio: bool,
                steps: int = 100) -> torch.Tensor:
        B = wav.shape[0]
        known = self.get_emb(wav.to(self.device)).to(self.device)               # [B, 128, T']
        keep = torch.nn.functional.interpolate(keep.to(self.device), size=known.shape[2])
        cond = self.conditioner([{"prompt": prompt}] * B, self.device)
        cond["masked_input"] = known * keep
        cond["mask"] = keep
        cond = self.get_conditioning(cond)
        extra = {} if isinstance(diffusion, GaussianDiffusion) else {"step": steps}       # VDM.sample takes the step count itself (vdm.py:77)
        z = diffusion.sample(model, tuple(known.shape), cond, causal=causal, init_data=known if seed_with_audio else None, **extra)
        # the reference hands the latents to its CPU decoder (generation.py:129-130); a decoder that lives on a device says so
        # (EncodecHIP.decoder_device) and gets them where they are
        return self.audio_encoder.decoder(z.to(getattr(self.audio_encoder, "decoder_device", "cpu")))

    # generation.py:134-150
    def get_mask(self, sample_size: int, start: float, end: float, batch_size: int) -> torch.Tensor:
        return get_mask(sample_size, start, end, batch_size, self.sample_rate)

    def get_emb(self, audio: torch.Tensor) -> torch.Tensor:
        """waveform -> continuous latents [B, 128, T']: the codes of every encoded segment side by side in time, summed
        codebook vectors (generation.py:145-150)"""
        per_segment = [codes for codes, _scale in self.audio_encoder.encode(audio)]          # each [B, n_q, T_seg]
        return self.audio_encoder.quantizer.decode(torch.cat(per_segment, dim=-1).permute(1, 0, 2))

    # generation.py:152-192
    def get_conditioning(self, cond):
        """as written in the reference: input-concat entries are read as ``cond[key][0]`` -- the FIRST batch element --
        and expanded over the batch (generation.py:173-180)"""
        return get_conditioning(cond, self.cross_attn_cond_ids, self.global_cond_ids, self.input_concat_ids, batch_size=self.batch_size)

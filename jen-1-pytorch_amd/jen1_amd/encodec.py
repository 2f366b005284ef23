"""Encodec 48 kHz pieces either side of the sampler, on the HIP kernels.

What the reference calls on ``EncodecModel.encodec_model_48khz()`` (third-party ``encodec==0.1.1``, requirements.txt:5,
not vendored, weights not available offline):
  * ``audio_encoder.quantizer.decode(codes)``  (generation.py:145-150): codes -> the 128-channel latents the denoiser
    works on  ->  ``ResidualVectorQuantizerHIP.decode`` (jen1_rvq_decode);
  * ``audio_encoder.decoder(sample_embs)``     (generation.py:130): latents ``[B, 128, T]`` -> stereo audio
    ``[B, 2, 320 T]`` through the SEANet decoder  ->  ``SEANetDecoderHIP``.
  * ``audio_encoder.encode(audio)``            (generation.py:146; dataloader.py:106-114): 1 s segments, RMS normalisation,
    SEANet encoder, nearest-codebook search  ->  ``EncodecHIP.encode`` (``SEANetEncoderHIP``,
    ``ResidualVectorQuantizerHIP.encode``).

The decoder restates encodec 0.1.1 ``modules/seanet.py::SEANetDecoder`` with the 48 kHz settings (dimension 128,
n_filters 32, ratios [8, 5, 4, 2], kernel 7, last kernel 7, residual kernel 3, 1 residual layer, compress 2, 2 LSTM
layers, ELU, non-causal, reflect padding, norm "time_group_norm", true skip off): every SConv1d / SConvTranspose1d is
convolution -> GroupNorm(1 group) [-> trim for the transposed ones, AFTER the norm], as in ``modules/conv.py``.
Convolutions run on jen1_train_gemm (the reflect padding is an index map, no padded copy), GroupNorm / ELU on the
train_ops kernels, the LSTM on jen1_lstm_layer.  Parameters are taken under the key names of the Hugging Face port
(``transformers.EncodecModel``: ``layers.N.conv.weight`` ...), which is the implementation of that architecture available offline:
PARITY IS PINNED AGAINST THAT PORT WITH SYNTHETIC WEIGHTS (tests/golden/encodec.npz); the real checkpoint and the
``encodec`` package itself are not available here, so parity against them is unpinned.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import lib as L
from .train import ConvGeom, TrainRuntime, _conv_forward, _operand, pad8

HOP_48K = 320


class ResidualVectorQuantizerHIP:
    """``quantizer.decode`` (encodec quantization/vq.py ResidualVectorQuantizer.decode -> core_vq decode): the sum over
    the n_q codebooks of the looked-up vectors.  ``tables``: float32 ``[n_q, bins, 128]`` (``layers.q.codebook.embed``)."""

    def __init__(self, tables: torch.Tensor, device="cuda"):
        self.lib = L.load()
        self.tables = tables.to(device, torch.float32).contiguous()
        self.device = torch.device(device)
        self._rt = None

    @classmethod
    def from_state_dict(cls, sd: Dict[str, torch.Tensor], device="cuda") -> "ResidualVectorQuantizerHIP":
        n = 0
        while f"layers.{n}.codebook.embed" in sd:
            n += 1
        assert n > 0, "no layers.N.codebook.embed entries"
        return cls(torch.stack([sd[f"layers.{q}.codebook.embed"] for q in range(n)]), device)

    @torch.no_grad()
    def encode(self, emb: torch.Tensor, n_q: Optional[int] = None) -> torch.Tensor:
        """ResidualVectorQuantization.encode (core_vq.py): per codebook the nearest entry of the running residual.
        emb float32 [B, 128, T] -> codes int64 [n_q, B, T].  The distance search -(|x|^2 - 2 x.E + |E|^2) is one float32
        GEMM per codebook (2 x.E with -|E|^2 as its bias; |x|^2 does not change the argmax)."""
        from .train import TrainRuntime, _operand
        if self._rt is None:
            self._rt = TrainRuntime("f32", self.device)
            self._neg_sq = (-(self.tables ** 2).sum(-1)).contiguous()          # [n_q][bins]
        rt = self._rt
        src = emb.device
        B, D, T = emb.shape
        nq = self.tables.shape[0] if n_q is None else n_q
        bins = self.tables.shape[1]
        res = emb.to(self.device, torch.float32).transpose(1, 2).reshape(B * T, D).contiguous()
        scores = torch.empty((B * T, bins), dtype=torch.float32, device=self.device)
        out = []
        for q in range(nq):
            rt.gemm(_operand(res.data_ptr(), D, 1), _operand(self.tables[q].data_ptr(), D, 1), scores.data_ptr(), B * T, bins, D,
                    dtype=L.F32, ldc_m=bins, bias=self._neg_sq[q], alpha=2.0)
            idx = scores.argmax(dim=-1)
            res = res - self.tables[q][idx]
            out.append(idx.view(B, T))
        return torch.stack(out).to(src)

    @torch.no_grad()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes int64 [n_q', B, T] (n_q' <= n_q) -> float32 [B, 128, T] on the codes' device"""
        src = codes.device
        c = codes.to(self.device, torch.int64).contiguous()
        nq, B, T = c.shape
        assert nq <= self.tables.shape[0]
        bins, D = self.tables.shape[1], self.tables.shape[2]
        out = torch.empty((B, D, T), dtype=torch.float32, device=self.device)
        s = torch.cuda.current_stream(self.device).cuda_stream
        L.check(self.lib.jen1_rvq_decode(c.data_ptr(), self.tables.data_ptr(), out.data_ptr(), nq, B, T, bins, D, s), "jen1_rvq_decode")
        return out.to(src)


class _SEANetOps:
    """the building blocks both halves of the SEANet share (encodec modules/conv.py, modules/lstm.py, modules/seanet.py)"""

    def __init__(self, params: Dict[str, torch.Tensor], lstm_name: str, compute_dtype: str, device):
        self.rt = TrainRuntime(compute_dtype, device)
        self.device = self.rt.device
        self.p = {k: v.detach().to(self.device, torch.float32).contiguous() for k, v in params.items()}
        assert "layers.0.norm.weight" in self.p, "only the 48 kHz model's norm='time_group_norm' is built (no weight_norm)"
        assert f"{lstm_name}.lstm.weight_ih_l0" in self.p, f"{lstm_name} must be the LSTM"
        self.lstm_name = lstm_name
        self.n_lstm = 0
        while f"{lstm_name}.lstm.weight_ih_l{self.n_lstm}" in self.p:
            self.n_lstm += 1
        dt = self.rt.tdtype
        # LSTM operands: W_ih as a linear weight (packed by the runtime), W_hh transposed [H][4H], the two biases summed
        self.whh_t = [self.p[f"{lstm_name}.lstm.weight_hh_l{l}"].t().contiguous().to(dt) for l in range(self.n_lstm)]
        self.whh = [self.p[f"{lstm_name}.lstm.weight_hh_l{l}"].to(dt).contiguous() for l in range(self.n_lstm)]
        # jen1_lstm_layer_multi (register-resident W_hh over H / 32 workgroups, one grid barrier per step) unless disabled
        self.lstm_multi = os.environ.get("JEN1_LSTM_MULTI", "1") == "1"
        self.last_lstm_counters: Optional[torch.Tensor] = None
        self._lstm_flags: List[torch.Tensor] = []
        self.lstm_bias = [(self.p[f"{lstm_name}.lstm.bias_ih_l{l}"] + self.p[f"{lstm_name}.lstm.bias_hh_l{l}"]).contiguous()
                          for l in range(self.n_lstm)]

    def _check_lstm(self) -> None:
        """raise if a grid barrier of the multi-workgroup LSTM timed out (its workgroups were not co-resident)"""
        flags, self._lstm_flags = self._lstm_flags, []
        if flags and int(torch.stack([f[:, 1].sum() for f in flags]).sum()) != 0:
            raise L.Jen1HipError("jen1_lstm_layer_multi: grid barrier time-out (workgroups not co-resident); set JEN1_LSTM_MULTI=0")

    def _to_rows(self, x_bct: torch.Tensor) -> torch.Tensor:
        B, C, T = x_bct.shape
        h = torch.zeros((B, T, pad8(C)), dtype=self.rt.tdtype, device=self.device)
        h[:, :, :C] = x_bct.transpose(1, 2)
        return h

    # ------------------------------------------------------------------ building blocks
    def _norm(self, x: torch.Tensor, name: str, C: int) -> torch.Tensor:
        rt, lib = self.rt, self.rt.lib
        B, Lx, ld = x.shape
        sums = torch.empty((B, 1, 2), dtype=torch.float32, device=x.device)
        y = (torch.zeros_like if ld != C else torch.empty_like)(x)
        s = rt.stream()
        dt = rt.dt_of(x)
        L.check(lib.jen1_gn_sums(x.data_ptr(), sums.data_ptr(), B, Lx, C, ld, 1, dt, s), "jen1_gn_sums")
        L.check(lib.jen1_gn_apply(x.data_ptr(), sums.data_ptr(), self.p[f"{name}.norm.weight"].data_ptr(), self.p[f"{name}.norm.bias"].data_ptr(),
                                  None, 0, y.data_ptr(), B, Lx, C, ld, 1, 1e-5, 0, dt, s), "jen1_gn_apply")
        return y

    def _conv(self, x: torch.Tensor, name: str, stride: int = 1) -> torch.Tensor:
        """SConv1d, non-causal: reflect padding of k - stride (split right-first) plus the extra right padding that
        makes the frame count whole (modules/conv.py get_extra_padding_for_conv1d), conv, GroupNorm(1)"""
        w, b = self.p[f"{name}.conv.weight"], self.p[f"{name}.conv.bias"]
        co, ci, k = w.shape
        Lx = x.shape[1]
        total = k - stride
        left = total - total // 2
        Lout = -(-(Lx - k + total) // stride) + 1                  # ceil((L - k + total) / stride) + 1
        right = (Lout - 1) * stride + k - left - Lx                # total // 2 + extra padding
        if Lx <= max(left, right):
            raise NotImplementedError(f"{name}: {Lx} frames are not more than the reflect padding ({left}, {right}); encodec's "
                                      "tiny-input case of pad1d is not built")
        g = ConvGeom("conv", k, stride, left, Lx, Lout, ci, co, reflect=True)
        y = _conv_forward(self.rt, x, self.rt.packed(w, "conv", x.dtype), b, g)
        return self._norm(y, name, co)

    def _conv_transpose(self, x: torch.Tensor, name: str, stride: int) -> torch.Tensor:
        """SConvTranspose1d, non-causal: full transposed conv, GroupNorm(1) over the UNtrimmed length, then trim"""
        w, b = self.p[f"{name}.conv.weight"], self.p[f"{name}.conv.bias"]
        ci, co, k = w.shape
        Lx = x.shape[1]
        full = (Lx - 1) * stride + k
        g = ConvGeom("convT", k, stride, 0, Lx, full, ci, co)
        y = self._norm(_conv_forward(self.rt, x, self.rt.packed(w, "convT", x.dtype), b, g), name, co)
        total = k - stride
        right = total // 2
        return y[:, total - right: full - right].contiguous()

    def _elu(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.empty_like(x)
        L.check(self.rt.lib.jen1_act_forward(x.data_ptr(), y.data_ptr(), x.numel(), 2, self.rt.dt_of(x), self.rt.stream()), "jen1_act_forward")
        return y

    def _resblock(self, x: torch.Tensor, name: str) -> torch.Tensor:
        h = self._conv(self._elu(x), f"{name}.block.1")
        h = self._conv(self._elu(h), f"{name}.block.3")
        return self._conv(x, f"{name}.shortcut") + h

    def _lstm(self, x: torch.Tensor) -> torch.Tensor:
        """SLSTM: y = LSTM(x) + x over the time axis (modules/lstm.py)"""
        rt = self.rt
        B, T, H = x.shape
        dt = rt.dt_of(x)
        h = x
        for l in range(self.n_lstm):
            wih = rt.packed(self.p[f"{self.lstm_name}.lstm.weight_ih_l{l}"], "linear", x.dtype)
            gin = torch.empty((B, T, 4 * H), dtype=torch.float32, device=x.device)
            rt.gemm(_operand(h.data_ptr(), h.shape[-1], 1), _operand(wih.data_ptr(), wih.shape[-1], 1), gin.data_ptr(), B * T, 4 * H, H,
                    dtype=dt, ldc_m=4 * H, bias=self.lstm_bias[l], c_f32=True)
            y = torch.empty_like(x)
            last = l == self.n_lstm - 1
            groups = (B + 7) // 8
            if self.lstm_multi and groups * (H // 32) <= 256:
                hbuf = torch.empty((groups, 2, 16, H), dtype=torch.float32, device=x.device)
                cnt = torch.zeros((groups, 32), dtype=torch.int32, device=x.device)
                L.check(rt.lib.jen1_lstm_layer_multi(gin.data_ptr(), self.whh[l].data_ptr(), x.data_ptr() if last else None, y.data_ptr(),
                                                     hbuf.data_ptr(), cnt.data_ptr(), B, T, H, y.shape[-1], dt, rt.stream()),
                        "jen1_lstm_layer_multi")
                self.last_lstm_counters = cnt        # cnt[:, 1] != 0 reports a barrier time-out (checked in _check_lstm)
                self._lstm_flags.append(cnt)
            else:
                L.check(rt.lib.jen1_lstm_layer(gin.data_ptr(), self.whh_t[l].data_ptr(), x.data_ptr() if last else None, y.data_ptr(), B, T, H,
                                               y.shape[-1], dt, rt.stream()), "jen1_lstm_layer")
            h = y
        return h



class SEANetDecoderHIP(_SEANetOps):
    def __init__(self, params: Dict[str, torch.Tensor], ratios: Sequence[int] = (8, 5, 4, 2), n_residual_layers: int = 1,
                 compute_dtype: str = "bf16", device="cuda"):
        super().__init__(params, "layers.1", compute_dtype, device)
        self.ratios, self.n_res = list(ratios), n_residual_layers
        idx = 2
        self.stages: List[tuple] = []
        for r in self.ratios:
            self.stages.append((idx + 1, r, [idx + 2 + j for j in range(self.n_res)]))
            idx += 2 + self.n_res
        self.last = idx + 1
        assert f"layers.{self.last}.conv.weight" in self.p, f"expected the output convolution at layers.{self.last}"

    @classmethod
    def from_module(cls, decoder: torch.nn.Module, ratios: Sequence[int] = (8, 5, 4, 2), **kw) -> "SEANetDecoderHIP":
        """from a ``transformers`` EncodecDecoder (its state_dict already uses the key names this class reads)"""
        return cls({k: v for k, v in decoder.state_dict().items()}, ratios, **kw)

    # ------------------------------------------------------------------ SEANetDecoder.forward
    @torch.no_grad()
    def __call__(self, emb: torch.Tensor) -> torch.Tensor:
        """latents [B, 128, T] (any device) -> audio float32 [B, channels, hop * T] on the same device"""
        src = emb.device
        x = emb.to(self.device, torch.float32)
        h = self._conv(self._to_rows(x), "layers.0")
        h = self._lstm(h)
        for conv_idx, ratio, res in self.stages:
            h = self._conv_transpose(self._elu(h), f"layers.{conv_idx}", ratio)
            for r in res:
                h = self._resblock(h, f"layers.{r}")
        h = self._conv(self._elu(h), f"layers.{self.last}")
        ch = self.p[f"layers.{self.last}.conv.weight"].shape[0]
        out = h[:, :, :ch].to(torch.float32).transpose(1, 2).contiguous().to(src)
        self._check_lstm()
        return out


class SEANetEncoderHIP(_SEANetOps):
    """encodec modules/seanet.py::SEANetEncoder, 48 kHz settings: conv k7, then per ratio (reversed: 2, 4, 5, 8) a residual
    block, ELU and a strided conv (k = 2 r); LSTM; ELU; conv k7 to the 128 latent channels"""

    def __init__(self, params: Dict[str, torch.Tensor], ratios: Sequence[int] = (8, 5, 4, 2), n_residual_layers: int = 1,
                 compute_dtype: str = "bf16", device="cuda"):
        n_stage = len(ratios) * (n_residual_layers + 2)
        super().__init__(params, f"layers.{1 + n_stage}", compute_dtype, device)
        self.ratios, self.n_res = list(reversed(list(ratios))), n_residual_layers
        self.last = 1 + n_stage + 2
        assert f"layers.{self.last}.conv.weight" in self.p, f"expected the output convolution at layers.{self.last}"

    @classmethod
    def from_module(cls, encoder: torch.nn.Module, ratios: Sequence[int] = (8, 5, 4, 2), **kw) -> "SEANetEncoderHIP":
        return cls({k: v for k, v in encoder.state_dict().items()}, ratios, **kw)

    @torch.no_grad()
    def __call__(self, audio: torch.Tensor) -> torch.Tensor:
        """audio [B, channels, L] -> latents float32 [B, 128, ceil(L / 320)] on the same device"""
        src = audio.device
        h = self._conv(self._to_rows(audio.to(self.device, torch.float32)), "layers.0")
        idx = 1
        for r in self.ratios:
            for j in range(self.n_res):
                h = self._resblock(h, f"layers.{idx + j}")
            idx += self.n_res
            h = self._conv(self._elu(h), f"layers.{idx + 1}", stride=r)
            idx += 2
        h = self._lstm(h)
        h = self._conv(self._elu(h), f"layers.{self.last}")
        ch = self.p[f"layers.{self.last}.conv.weight"].shape[0]
        out = h[:, :, :ch].to(torch.float32).transpose(1, 2).contiguous().to(src)
        self._check_lstm()
        return out


class EncodecHIP:
    """the slice of ``EncodecModel`` generation.py touches: ``.channels``, ``.sample_rate``, ``.quantizer.decode``,
    ``.decoder``, ``.encode``"""

    def __init__(self, decoder: SEANetDecoderHIP, quantizer: ResidualVectorQuantizerHIP, channels: int = 2, sample_rate: int = 48000,
                 encode: Optional[Callable] = None, encoder: Optional["SEANetEncoderHIP"] = None, segment: float = 1.0,
                 overlap: float = 0.01, normalize: bool = True, n_q: Optional[int] = None):
        self.decoder, self.quantizer, self.channels, self.sample_rate, self._encode = decoder, quantizer, channels, sample_rate, encode
        self.encoder, self.normalize, self.n_q = encoder, normalize, n_q
        self.decoder_device = decoder.device       # Jen1.generate hands the sampled latents over where they are (no host round trip)
        self.segment_length = int(segment * sample_rate)                                   # encodec model.py segment_length
        self.segment_stride = max(1, int((1 - overlap) * self.segment_length))             # encodec model.py segment_stride

    @torch.no_grad()
    def encode(self, audio: torch.Tensor):
        """``EncodecModel.encode`` of the package (model.py): the audio is cut into 1 s segments with 1 % overlap, every
        segment is normalised by the RMS of its mono mix, encoded and quantised with ALL codebooks (the reference never
        sets a target bandwidth) -> ``[(codes [B, n_q, T_seg], scale [B, 1]), ...]``, which is what ``get_emb``
        concatenates (generation.py:145-150; dataloader.py:106-114)."""
        if self._encode is not None:
            return self._encode(audio)
        if self.encoder is None:
            raise NotImplementedError("no encoder: construct EncodecHIP with encoder=SEANetEncoderHIP(...) or encode=<callable>")
        assert audio.dim() == 3 and 0 < audio.shape[1] <= 2
        frames = []
        for offset in range(0, audio.shape[-1], self.segment_stride):
            x = audio[:, :, offset: offset + self.segment_length]
            scale = None
            if self.normalize:
                mono = x.mean(dim=1, keepdim=True)
                scale = 1e-8 + mono.pow(2).mean(dim=2, keepdim=True).sqrt()
                x = x / scale
                scale = scale.view(-1, 1)
            codes = self.quantizer.encode(self.encoder(x), self.n_q).transpose(0, 1)       # [B, n_q, T]
            frames.append((codes, scale))
        return frames

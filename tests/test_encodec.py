"""Encodec pieces either side of the sampler (generation.py:130, :145-150).

CPU: the numpy restatement (oracle/encodec_oracle.py) against tests/golden/encodec.npz, which holds what the Hugging
Face port of Encodec 48 kHz produced with synthetic weights (tests/golden/make_golden.py encodec) -- the pin of the
oracle.  GPU (-m gpu): the HIP decoder / RVQ decode against the same fixture, float32 mode <= 1e-3 of the largest
reference entry (BASELINE's gate), bf16 <= 5e-2.  Parity against the ``encodec`` package and its checkpoint is unpinned
(neither is available here); see jen1_amd/encodec.py.
"""
import json

import numpy as np
import pytest
import torch

from helpers import SEED, golden, rel_err
from jen1_amd.init_fill import fill, fill_normal


def _params():
    g = golden("encodec")
    return {k: fill("encodec.decoder." + k, tuple(s), SEED) for k, s in json.loads(str(g["schema"]))}


def _tables(n_q):
    return np.stack([fill_normal(f"encodec.quantizer.layers.{i}.codebook.embed", (1024, 128), SEED) for i in range(n_q)])


def test_oracle_seanet_decoder_matches_reference_port():
    from oracle import encodec_oracle as EO
    g = golden("encodec")
    taps = {}
    y = EO.seanet_decoder(_params(), fill_normal("encodec.emb", (2, 128, 37), 5), taps=taps)
    assert y.shape == g["decoder.y"].shape == (2, 2, 37 * 320)
    assert rel_err(taps["conv0"][:, ::8, :], g["decoder.tap.conv0"]) < 1e-5
    assert rel_err(taps["lstm"][:, ::8, :], g["decoder.tap.lstm"]) < 1e-5
    assert rel_err(y, g["decoder.y"]) < 1e-4


def test_oracle_rvq_decode_matches_reference_port():
    from oracle import encodec_oracle as EO
    g = golden("encodec")
    y = EO.rvq_decode(g["rvq.codes"], _tables(int(g["rvq.n_q"])))
    assert y.shape == g["rvq.y"].shape and rel_err(y, g["rvq.y"]) < 1e-6


def _enc_params():
    g = golden("encodec")
    return {k: fill("encodec.encoder." + k, tuple(s), SEED) for k, s in json.loads(str(g["enc_schema"]))}


def _audio():
    return fill_normal("encodec.audio", (2, 2, 9600 + 123), 5) * 0.3


def test_oracle_seanet_encoder_and_rvq_encode_match_reference_port():
    from oracle import encodec_oracle as EO
    g = golden("encodec")
    e = EO.seanet_encoder(_enc_params(), _audio())
    assert e.shape == g["encoder.y"].shape == (2, 128, 31)
    assert rel_err(e, g["encoder.y"]) < 1e-4
    codes = EO.rvq_encode(g["encoder.y"], _tables(16))
    assert codes.shape == g["encoder.codes"].shape == (16, 2, 31)
    assert (codes == g["encoder.codes"]).mean() > 0.999          # nearest-neighbour ties can flip with the summation order


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("f32", 1e-3), ("bf16", 5e-2)])
def test_hip_seanet_encoder_vs_reference_port(mode, tol):
    from jen1_amd.encodec import SEANetEncoderHIP
    g = golden("encodec")
    enc = SEANetEncoderHIP({k: torch.from_numpy(v) for k, v in _enc_params().items()}, compute_dtype=mode)
    e = enc(torch.from_numpy(_audio()))
    assert e.device.type == "cpu" and tuple(e.shape) == (2, 128, 31)
    assert rel_err(e.numpy(), g["encoder.y"]) < tol


@pytest.mark.gpu
def test_hip_rvq_encode_and_segmented_encode():
    """codes of the nearest-neighbour search against the port (same latents in), then EncodecModel.encode's segment
    loop (1 s segments, 1 % overlap, RMS normalisation, all codebooks) against the numpy restatement"""
    from jen1_amd.encodec import EncodecHIP, ResidualVectorQuantizerHIP, SEANetDecoderHIP, SEANetEncoderHIP
    from oracle import encodec_oracle as EO
    g = golden("encodec")
    quant = ResidualVectorQuantizerHIP(torch.from_numpy(_tables(16)))
    codes = quant.encode(torch.from_numpy(g["encoder.y"]))
    assert tuple(codes.shape) == (16, 2, 31) and codes.dtype == torch.int64
    assert (codes.numpy() == g["encoder.codes"]).mean() > 0.999
    # the first codebook's choice really is the nearest entry (float64 check of every frame)
    x = g["encoder.y"].transpose(0, 2, 1).reshape(-1, 128).astype(np.float64)
    d = ((x[:, None, :] - _tables(1)[0][None].astype(np.float64)) ** 2).sum(-1)
    chosen = d[np.arange(x.shape[0]), codes[0].numpy().reshape(-1)]
    assert np.all(chosen <= d.min(axis=1) * (1 + 1e-6))
    enc = SEANetEncoderHIP({k: torch.from_numpy(v) for k, v in _enc_params().items()}, compute_dtype="f32")
    dec = SEANetDecoderHIP({k: torch.from_numpy(v) for k, v in _params().items()}, compute_dtype="f32")
    model = EncodecHIP(dec, quant, encoder=enc)
    audio = fill_normal("encodec.audio.long", (1, 2, 48000 + 47520 // 2), 7) * 0.2       # 2 segments, the second one short
    frames = model.encode(torch.from_numpy(audio))
    want = EO.encode_frames(_enc_params(), _tables(16), audio)
    assert len(frames) == len(want) == 2
    for (c, s), (cw, sw) in zip(frames, want):
        assert tuple(c.shape) == cw.shape and rel_err(s.numpy(), sw) < 1e-6
        assert (c.numpy() == cw).mean() > 0.995
    # get_emb (generation.py:145-150) on top of it
    emb = quant.decode(torch.cat([f[0] for f in frames], dim=-1).transpose(0, 1))
    assert tuple(emb.shape) == (1, 128, sum(f[0].shape[-1] for f in frames))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,tol", [("f32", 1e-3), ("bf16", 5e-2)])
def test_hip_seanet_decoder_vs_reference_port(mode, tol):
    from jen1_amd.encodec import SEANetDecoderHIP
    g = golden("encodec")
    dec = SEANetDecoderHIP({k: torch.from_numpy(v) for k, v in _params().items()}, compute_dtype=mode)
    emb = torch.from_numpy(fill_normal("encodec.emb", (2, 128, 37), 5))
    y = dec(emb)                                   # CPU latents in, CPU audio out (generation.py:129-130)
    assert y.device.type == "cpu" and tuple(y.shape) == (2, 2, 37 * 320)
    assert rel_err(y.numpy(), g["decoder.y"]) < tol
    y2 = dec(emb.cuda())
    assert y2.device.type == "cuda" and rel_err(y2.cpu().numpy(), g["decoder.y"]) < tol
    assert int(dec.last_lstm_counters[:, 1].sum()) == 0           # no grid-barrier time-out in the multi-workgroup LSTM
    # the single-workgroup LSTM kernel gives the same audio
    dec.lstm_multi = False
    assert rel_err(dec(emb).numpy(), g["decoder.y"]) < tol


@pytest.mark.gpu
def test_hip_lstm_multi_many_sequences_and_long():
    """more sequences than one group of 8 and a long sequence: multi-workgroup LSTM against the single-workgroup kernel"""
    from jen1_amd.encodec import SEANetDecoderHIP
    dec = SEANetDecoderHIP({k: torch.from_numpy(v) for k, v in _params().items()}, compute_dtype="f32")
    for B, T in ((11, 40), (2, 700)):
        x = torch.randn((B, T, 512), device="cuda") * 0.5
        dec.lstm_multi = True
        a = dec._lstm(x)
        assert int(dec.last_lstm_counters[:, 1].sum()) == 0
        dec.lstm_multi = False
        b = dec._lstm(x)
        torch.cuda.synchronize()
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5, (B, T)
    # bf16 mode with more than one sequence: the recurrent product runs on the matrix cores (16 sequences per group,
    # h split into bf16 high + low); against the float32 single-workgroup kernel on the same (bf16-rounded) input
    dec16 = SEANetDecoderHIP({k: torch.from_numpy(v) for k, v in _params().items()}, compute_dtype="bf16")
    for B, T in ((2, 64), (11, 40), (20, 33), (3, 700)):
        x = (torch.randn((B, T, 512), device="cuda") * 0.5).to(torch.bfloat16)
        dec16.lstm_multi = True
        a = dec16._lstm(x)
        assert int(dec16.last_lstm_counters[:, 1].sum()) == 0
        dec.lstm_multi = False
        b = dec._lstm(x.float())
        torch.cuda.synchronize()
        assert rel_err(a.float().cpu().numpy(), b.cpu().numpy()) < 2e-2, (B, T)


@pytest.mark.gpu
def test_hip_decoder_vs_oracle_other_lengths():
    """lengths the fixture does not hold (odd, long enough for several GEMM tiles), against the numpy restatement"""
    from jen1_amd.encodec import SEANetDecoderHIP
    from oracle import encodec_oracle as EO
    p = _params()
    dec = SEANetDecoderHIP({k: torch.from_numpy(v) for k, v in p.items()}, compute_dtype="f32")
    for B, T in ((1, 9), (3, 75)):
        emb = fill_normal(f"encodec.emb.{T}", (B, 128, T), 6)
        want = EO.seanet_decoder(p, emb)
        got = dec(torch.from_numpy(emb)).numpy()
        assert got.shape == want.shape == (B, 2, 320 * T)
        assert rel_err(got, want) < 1e-3, (B, T)


@pytest.mark.gpu
def test_hip_rvq_decode_vs_reference_port():
    from jen1_amd.encodec import ResidualVectorQuantizerHIP
    g = golden("encodec")
    n_q = int(g["rvq.n_q"])
    q = ResidualVectorQuantizerHIP(torch.from_numpy(_tables(n_q)))
    codes = torch.from_numpy(g["rvq.codes"])
    y = q.decode(codes)
    assert y.device.type == "cpu" and rel_err(y.numpy(), g["rvq.y"]) < 1e-6
    y8 = q.decode(codes[:8].cuda())               # a lower bandwidth uses the first codebooks only
    want = sum(torch.from_numpy(_tables(n_q))[i][codes[i]] for i in range(8)).transpose(1, 2)
    assert y8.device.type == "cuda" and rel_err(y8.cpu().numpy(), want.numpy()) < 1e-6


@pytest.mark.gpu
def test_jen1_generate_with_hip_encodec_halves():
    """generation.py end to end with the HIP quantizer decode + SEANet decoder around the HIP sampler"""
    from jen1_amd import synth
    from jen1_amd.config import GDMConfig, tiny_model_config
    from jen1_amd.encodec import EncodecHIP, ResidualVectorQuantizerHIP, SEANetDecoderHIP
    from jen1_amd.generation import Jen1
    dec = SEANetDecoderHIP({k: torch.from_numpy(v) for k, v in _params().items()}, compute_dtype="bf16")
    quant = ResidualVectorQuantizerHIP(torch.from_numpy(_tables(16)))

    def encode(audio):                              # stand-in for the encoder half: deterministic codes per 320 samples
        B, _, n = audio.shape
        base = (audio[:, :, : n // 320 * 320].reshape(B, 2, n // 320, 320).mean(dim=(1, 3)) * 1e3).round().long().abs() % 1024
        return [(torch.stack([(base + 37 * q) % 1024 for q in range(16)], dim=1), None)]

    enc = EncodecHIP(dec, quant, encode=encode)
    cond = synth.conditioning(2, 300, "text_guided")
    emb, msk = torch.from_numpy(cond["cross_attn_cond"]).cuda(), torch.from_numpy(cond["cross_attn_masks"]).cuda()
    j = Jen1(None, device="cuda", audio_encoder=enc, conditioner=lambda md, device: {"prompt": (emb[:len(md)], msk[:len(md)])},
             model_config=tiny_model_config(), diffusion_config=GDMConfig(), compute_dtype="bf16")
    wav = j.generate("strings", seed=2, steps=3, batch_size=2, seconds=1, use_gdm=True)
    assert tuple(wav.shape) == (2, 2, 48000) and torch.isfinite(wav).all() and float(wav.abs().max()) > 0
    # the sampled latents go to the HIP decoder where they are: no host round trip (EncodecHIP.decoder_device)
    assert enc.decoder_device.type == "cuda" and wav.device.type == "cuda"
    seen = []
    inner = enc.decoder
    enc.decoder = lambda z: (seen.append(z.device.type), inner(z))[1]
    j.generate("strings", seed=2, steps=2, batch_size=2, seconds=1, use_gdm=True)
    assert seen == ["cuda"]

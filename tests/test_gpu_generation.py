"""-m gpu: the ``Jen1.generate`` surface (generation.py:16-192) around the HIP sampler, with stand-ins for the two
third-party models the reference constructs itself (Encodec, the T5 conditioner: outside this build)."""
import numpy as np
import pytest
import torch

from jen1_amd import synth
from jen1_amd.config import GDMConfig, tiny_model_config

pytestmark = pytest.mark.gpu

HOP = 320          # Encodec 48 kHz: one latent frame per 320 samples


class _Quantizer:
    def __init__(self, n_q=4, bins=64, dim=128):
        g = torch.Generator().manual_seed(11)
        self.tables = torch.randn((n_q, bins, dim), generator=g) * 0.3

    def decode(self, codes):                      # [n_q, B, T] -> [B, dim, T]: the sum of the codebook vectors
        out = 0
        for q in range(codes.shape[0]):
            out = out + self.tables.to(codes.device)[q][codes[q]]
        return out.transpose(1, 2)


class StubAudioEncoder:
    """the slice of ``encodec.EncodecModel`` generation.py touches"""
    channels = 2
    sample_rate = 48000

    def __init__(self):
        self.quantizer = _Quantizer()
        self.decoder_calls = 0

    def encode(self, audio):                      # -> [(codes [B, n_q, T], scale)]
        B, _, n = audio.shape
        frames = audio[:, :, : n // HOP * HOP].reshape(B, 2, n // HOP, HOP).mean(dim=(1, 3))
        base = (frames * 1000).round().long().abs() % 64
        codes = torch.stack([(base + 7 * q) % 64 for q in range(4)], dim=1)
        return [(codes, None)]

    def decoder(self, emb):                       # [B, 128, T] -> [B, 2, HOP * T]
        self.decoder_calls += 1
        assert emb.device.type == "cpu"           # generation.py:129 moves the latents to the CPU first
        y = emb[:, :2].repeat_interleave(HOP, dim=2)
        return torch.tanh(y)


@pytest.fixture(scope="module")
def jen1():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.generation import Jen1
    cond = synth.conditioning(8, 300, "text_guided")
    emb = torch.from_numpy(cond["cross_attn_cond"]).cuda()
    msk = torch.from_numpy(cond["cross_attn_masks"]).cuda()

    def conditioner(batch_metadata, device):
        n = len(batch_metadata)
        assert all(set(m) == {"prompt"} for m in batch_metadata)
        return {"prompt": (emb[:n].to(device), msk[:n].to(device))}

    return Jen1(None, device="cuda", audio_encoder=StubAudioEncoder(), conditioner=conditioner, model_config=tiny_model_config(),
                diffusion_config=GDMConfig(), compute_dtype="f32")


def test_generate_text_guided_is_seeded_and_shaped(jen1):
    a = jen1.generate("a calm piano piece", seed=3, steps=4, batch_size=2, seconds=2, use_gdm=True)
    b = jen1.generate("a calm piano piece", seed=3, steps=4, batch_size=2, seconds=2, use_gdm=True)
    c = jen1.generate("a calm piano piece", seed=4, steps=4, batch_size=2, seconds=2, use_gdm=True)
    assert a.shape == (2, 2, 2 * 48000) and a.device.type == "cpu" and torch.isfinite(a).all()
    assert torch.allclose(a, b, atol=1e-4)        # same seed: same trajectory (float atomics reorder sums at the 1e-7 level)
    assert float((a - c).abs().max()) > 1e-3


def test_generate_matches_hand_built_sampler_call(jen1):
    """generate() == get_mask + get_emb + conditioner + get_conditioning + GaussianDiffusion.sample + decoder by hand"""
    torch.manual_seed(9)
    out = jen1.generate("x", seed=9, steps=3, batch_size=2, seconds=2, use_gdm=True, task="music_inpaint", inpainting_scope=(0.5, 1.5))
    diffusion, model = jen1.get_model_and_diffusion(3, True)
    torch.manual_seed(9)
    jen1.batch_size = 2
    n = 2 * 48000
    audio = torch.zeros((2, 2, n))
    mask = jen1.get_mask(n, 0.5, 1.5, 2)
    assert float(mask[0, 0, 24000 - 1]) == 1 and float(mask[0, 0, 24000]) == 0 and float(mask[0, 0, 72000]) == 1
    emb = jen1.get_emb(audio.cuda())
    m = torch.nn.functional.interpolate(mask.cuda(), size=(emb.shape[2]))
    cond = jen1.conditioner([{"prompt": "x"}] * 2, "cuda")
    cond["masked_input"], cond["mask"] = emb * m, m
    cond = jen1.get_conditioning(cond)
    assert cond["input_concat_cond"].shape == (2, 129, emb.shape[2])
    z = diffusion.sample(model, tuple(emb.shape), cond, causal=False, init_data=None)
    want = jen1.audio_encoder.decoder(z.cpu())
    assert torch.allclose(out, want, atol=1e-4)


def test_generate_continuation_and_errors(jen1):
    prefix = torch.randn((2, 48000)) * 0.1                                   # 1 s of stereo audio, no batch axis
    out = jen1.generate("y", seed=1, steps=3, batch_size=2, seconds=2, use_gdm=True, task="music_cont", init_audio=prefix,
                        init_audio_sr=48000)
    assert out.shape == (2, 2, 2 * 48000) and torch.isfinite(out).all()
    # use_gdm defaults to False -> VDM: broken in the reference, the repaired sampler here (jen1_amd/vdm.py)
    a = jen1.generate("y", seed=5, steps=3, batch_size=2, seconds=2)
    b = jen1.generate("y", seed=5, steps=3, batch_size=2, seconds=2)
    assert a.shape == (2, 2, 2 * 48000) and torch.isfinite(a).all() and torch.allclose(a, b, atol=1e-4)
    with pytest.raises(ValueError):
        jen1.generate("y", steps=3, seconds=2, use_gdm=True, task="nope")

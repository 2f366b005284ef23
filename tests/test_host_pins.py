"""The host functions either side of the denoiser pinned against the reference's OWN trainer.py / generation.py /
utils/script_util.py (tests/golden/host_pins.npz, written by tests/golden/make_golden.py host, which imports those files
unmodified and also performs the checkpoint cross-load in both directions): SURVEY.md section 8 rows a14, f2, f3."""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from helpers import golden
from jen1_amd import synth
from jen1_amd import tasks as T
from jen1_amd.checkpoint import load_checkpoint, save_checkpoint
from jen1_amd.config import UNetSpec, tiny_model_config
from jen1_amd.init_fill import fill, fill_normal
from jen1_amd.model import UNetCFG1d
from jen1_amd.optim import FusedAdamW


def test_random_mask_matches_reference_trainer():
    """trainer.py:215-247 with ``random.seed(k)``: same mask, same causal flag, for every case the reference can run (its
    ``random.randint`` gets float bounds and refuses non-integral ones: exactly the lengths that are not multiples of 5)"""
    g = golden("host_pins")
    cases = json.loads(str(g["mask.cases"]))
    assert len(cases) == 63
    n_ok = 0
    for L_, task, seed, status in cases:
        integral = float(L_ * 0.2).is_integer() and float(L_ * 0.8).is_integer()
        if status != "ok":
            assert task != "text_guided" and not integral, (L_, task, seed, status)
            continue
        assert task == "text_guided" or integral
        seq = torch.from_numpy(synth.latents(3, L_, key="clip"))
        masked, mask, causal = T.random_mask(seq, L_, task, rng=random.Random(seed))
        k = f"mask.{L_}.{task}.{seed}"
        assert mask.shape == (3, 1, L_) and torch.equal(masked, seq * mask)
        for b in range(3):
            assert np.array_equal(mask[b, 0].numpy().astype(np.uint8), g[k]), k
        assert bool(causal) == bool(g[k + ".causal"]), k
        n_ok += 1
    assert n_ok >= 40
    # both values of the coin text_guided flips for ``causal`` occur in the fixture
    assert {bool(g[f"mask.1500.text_guided.{s}.causal"]) for s in range(7)} == {True, False}


def test_get_conditioning_matches_reference_trainer_and_generation():
    """trainer.py:249-278 and generation.py:152-192 (input-concat entries read as cond[key][0] and expanded over the batch)"""
    g = golden("host_pins")
    t = lambda k: torch.from_numpy(g["cond.in." + k])
    cond = {"prompt": (t("emb"), t("msk")), "style": (t("emb2"), t("msk2")), "g": (t("glob"), None), "masked_input": t("masked_in"),
            "mask": t("keep")}
    ids = dict(cross_attn_cond_ids=["prompt", "style"], global_cond_ids=["g"], input_concat_ids=["masked_input", "mask"])
    for form, kw in (("trainer", {}), ("generation", {"batch_size": 3})):
        r = T.get_conditioning(cond, **ids, **kw)
        assert set(r) == {"cross_attn_cond", "cross_attn_masks", "global_cond", "input_concat_cond"}
        for k, v in r.items():
            want = g[f"cond.{form}.{k}"]
            assert tuple(v.shape) == want.shape and np.array_equal(v.numpy(), want), (form, k)
    # the generation form really is different: every batch row carries the FIRST element's concat channels
    assert not np.array_equal(g["cond.trainer.input_concat_cond"], g["cond.generation.input_concat_cond"])


def test_get_mask_matches_reference_generation():
    """generation.py:134-143: ones with [floor(start sr), ceil(end sr)) zeroed"""
    g = golden("host_pins")
    for n, a, b, bs, z0, z1, nz in json.loads(str(g["get_mask.cases"])):
        m = T.get_mask(n, a, b, bs, 48000)
        assert m.shape == (bs, 1, n)
        z = np.flatnonzero(m[0, 0].numpy() == 0)
        assert (int(z[0]), int(z[-1]), len(z)) == (z0, z1, nz)
        assert all(torch.equal(m[i], m[0]) for i in range(bs))


def _digest(t: torch.Tensor) -> str:
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()[:16]


def test_checkpoint_format_matches_reference_writer(tmp_path):
    """script_util.py:79-124.  The fixture was written while the reference's save_checkpoint / load_checkpoint and the build's
    exchanged files in both directions (asserted equal there, outcome recorded); here the same training state is rebuilt
    without the reference -- torch AdamW on the CPU, two steps on seeded gradients -- and must give the recorded tensors bit for
    bit after a trip through FusedAdamW and the build's checkpoint file."""
    g = golden("host_pins")
    assert json.loads(str(g["ckpt.cross_load"])) == {"reference_file_into_build": True, "build_file_into_reference": True}
    cfg = tiny_model_config()
    keys = json.loads(str(g["ckpt.keys"]))
    assert keys == [k for k, _ in UNetSpec(**cfg).param_shapes()]
    assert json.loads(str(g["ckpt.file_keys"])) == ["epoch", "learning_rate", "model", "optimizer"]
    # the state the reference's model + optimiser were in when the file was written
    model = UNetCFG1d(**cfg, init_seed=1234, compute_dtype="f32", device="cpu")              # filled like make_golden's _build
    params = list(model.parameters())
    ref_opt = torch.optim.AdamW(params, lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1)
    for it in range(2):
        for i, p in enumerate(params):
            p.grad = torch.from_numpy(fill_normal(f"ckpt.grad.{it}.{i}", tuple(p.shape), 3)) * 1e-2
        ref_opt.step()
    want = json.loads(str(g["ckpt.param_digest"]))
    assert {k: _digest(v) for k, v in model.state_dict().items()} == want
    assert sorted(ref_opt.state_dict()["param_groups"][0]) == json.loads(str(g["ckpt.opt_group_keys"]))
    # through the build's optimiser and file format and back
    for p in params:
        p.grad = None
    opt = FusedAdamW(model.parameters())
    opt.load_state_dict(ref_opt.state_dict())
    path = str(tmp_path / "G_2.pth")
    save_checkpoint(model, opt, 3e-5, 2, path)
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(blob) == ["epoch", "learning_rate", "model", "optimizer"] and list(blob["model"]) == keys
    assert sorted(blob["optimizer"]["param_groups"][0]) == json.loads(str(g["ckpt.opt_group_keys"]))
    m2 = UNetCFG1d(**cfg, compute_dtype="f32", device="cpu", init_seed=None)
    o2 = FusedAdamW(m2.parameters())
    _, _, lr, epoch = load_checkpoint(path, m2, optimizer=o2)
    assert (lr, epoch, o2.step_count) == (3e-5, 2, 2)
    assert {k: _digest(v) for k, v in m2.state_dict().items()} == want
    ea, eq = json.loads(str(g["ckpt.exp_avg_digest"])), json.loads(str(g["ckpt.exp_avg_sq_digest"]))
    for i, (p, o) in enumerate(zip(o2.params, o2.offsets)):
        assert _digest(o2.exp_avg[o:o + p.numel()].view_as(p)) == ea[i]
        assert _digest(o2.exp_avg_sq[o:o + p.numel()].view_as(p)) == eq[i]

"""Checkpoint wire format of the reference.

Same file layout and loader tolerances as /root/reference/utils/script_util.py:79-148:
``torch.save({'model': state_dict, 'epoch', 'optimizer', 'learning_rate'})``; on load, keys missing from the file
keep the model's current value and a ``_orig_mod.`` prefix (torch.compile) is accepted.  The model's parameters use
the reference ``state_dict`` key schema, so files are interchangeable in both directions;
after loading, the HIP engine repacks its weights on next use.
"""
from __future__ import annotations

import os
from typing import Optional

import torch


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def save_checkpoint(model, optimizer, lr, iteration, checkpoint_path, logger=None) -> None:
    """script_util.py:79-90 (without the old-checkpoint cleanup, which is run management, not format)."""
    if logger is not None:
        logger.info(f"Saving model and optimizer state at iteration {iteration} to {checkpoint_path}")
    torch.save({"model": _unwrap(model).state_dict(), "epoch": iteration,
                "optimizer": optimizer.state_dict() if optimizer is not None else None, "learning_rate": lr}, checkpoint_path)


def load_checkpoint(checkpoint_path, model, logger=None, optimizer=None):
    """script_util.py:93-124: returns (model, optimizer, learning_rate, epoch)."""
    assert os.path.isfile(checkpoint_path)
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    epoch, learning_rate = ck["epoch"], ck["learning_rate"]
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer"])
    saved = ck["model"]
    m = _unwrap(model)
    new_state = {}
    for k, v in m.state_dict().items():
        if k in saved:
            new_state[k] = saved[k]
        elif f"_orig_mod.{k}" in saved:
            new_state[k] = saved[f"_orig_mod.{k}"]
        else:
            if logger is not None:
                logger.info("%s is not in the checkpoint" % k)
            new_state[k] = v
    m.load_state_dict(new_state)
    _repack(m)
    if logger is not None:
        logger.info(f"Loaded checkpoint '{checkpoint_path}' (epoch {epoch})")
    return model, optimizer, learning_rate, epoch


def load_model_diffsize(checkpoint_path, model):
    """script_util.py:127-148: copy every tensor whose key (with or without ``_orig_mod.``) and size match."""
    assert os.path.isfile(checkpoint_path)
    saved = torch.load(checkpoint_path, map_location="cpu", weights_only=False)["model"]
    m = _unwrap(model)
    state = m.state_dict()
    for k, v in saved.items():
        k2 = k.replace("_orig_mod.", "")
        if k in state and state[k].size() == v.size():
            state[k] = v
        elif k2 in state and state[k2].size() == v.size():
            state[k2] = v
        else:
            print("[WARNING] Parameter mismatch :", k)
    m.load_state_dict(state, strict=False)
    _repack(m)
    return model


def _repack(m) -> None:
    inv = getattr(m, "_invalidate", None)
    if callable(inv):
        inv()

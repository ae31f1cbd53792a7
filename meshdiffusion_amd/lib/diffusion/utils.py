"""Checkpoint I/O -- mirror of the reference's lib/diffusion/utils.py:6-30.
Same file format: torch.save({optimizer, model, ema, step}); non-strict model load."""
import logging
import os

import torch


def restore_checkpoint(ckpt_dir, state, device, strict=False):
    if not os.path.exists(ckpt_dir):
        os.makedirs(os.path.dirname(ckpt_dir), exist_ok=True)
        logging.warning(f"No checkpoint found at {ckpt_dir}. Returned the same state as input")
        if strict:
            raise FileNotFoundError(ckpt_dir)
        return state
    loaded = torch.load(ckpt_dir, map_location=device, weights_only=False)
    state["optimizer"].load_state_dict(loaded["optimizer"])
    state["model"].load_state_dict(loaded["model"], strict=False)
    state["ema"].load_state_dict(loaded["ema"])
    state["step"] = loaded["step"]
    return state


def save_checkpoint(ckpt_dir, state):
    torch.save({"optimizer": state["optimizer"].state_dict(), "model": state["model"].state_dict(),
                "ema": state["ema"].state_dict(), "step": state["step"]}, ckpt_dir)

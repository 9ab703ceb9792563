"""Checkpoint reader: the reference saves ``{"model": model.module.state_dict(), "optimizer": ...}``
every ``save_step`` to ``{ckpt_path}/{step}.pth.tar`` (train.py:149-159) and restores it in ``get_model``
(utils/model.py:11-35).  Only the inference subset is uploaded: ``mel_encoder.*`` (the training-only aligner, 12.4 M of
the 41.3 M parameters) and the optimizer state are read from disk and dropped."""
from __future__ import annotations

import os

import torch

from .model import FastSpeech2Align


def inference_state_dict(ckpt):
    """Pick the model state out of whatever ``torch.load`` returned and drop what the inference path never reads."""
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):  # a DataParallel wrapper saved without .module (train.py:42 wraps the model)
            k = k[len("module."):]
        if k.startswith("mel_encoder.") or k.endswith("num_batches_tracked"):
            continue
        out[k] = v
    return out


def load_checkpoint(model: FastSpeech2Align, path: str) -> FastSpeech2Align:
    # weights_only: a checkpoint is tensors + an optimizer state dict; do not unpickle arbitrary objects
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(inference_state_dict(ckpt))
    return model


def get_model(args, configs, device, train: bool = False) -> FastSpeech2Align:
    """utils/model.py:11-35 for the inference case (``train=False``)."""
    if train:
        raise NotImplementedError("training is out of scope for this path (SURVEY.md §2)")
    preprocess_config, model_config, train_config = configs
    model = FastSpeech2Align(preprocess_config, model_config).to(device)
    if getattr(args, "restore_step", None):
        load_checkpoint(model, os.path.join(train_config["path"]["ckpt_path"], "{}.pth.tar".format(args.restore_step)))
    model.eval()
    model.requires_grad_ = False  # the reference assigns (not calls) this attribute, utils/model.py:34
    return model

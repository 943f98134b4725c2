"""`utils` of the reference (reference utils.py:1-54): log dir, config, str2bool, checkpoint loading."""
import argparse
import os

import torch


def get_log_dir(model_name, cfg):
    import yaml

    log_dir = os.path.join("logs", model_name)
    os.makedirs(log_dir, exist_ok=True)
    with open(os.path.join(log_dir, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f, default_flow_style=False)
    return log_dir


def get_config():
    # reference utils.py:16-27 (only max_iteration is ever read, by the training loop)
    return {1: dict(max_iteration=700000, lr=1e-4, momentum=0.99, betas=(0.9, 0.999), weight_decay=0, interval_validate=1000)}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def load_checkpoint(model, optimizer, filename="model_best.pth.tar"):
    """reference utils.py:39-54: dict(epoch, model_state_dict, optim_state_dict, loss, ...) written by train.py:119-128.
    A missing file prints a note and returns the model untouched, as the reference does.  Checkpoints saved from
    nn.DataParallel (keys prefixed with `module.`, reference README.md:13,52) are accepted."""
    start_epoch, loss = 0, []
    if os.path.isfile(filename):
        checkpoint = torch.load(filename, map_location="cpu", weights_only=False)
        start_epoch = checkpoint["epoch"]
        sd = checkpoint["model_state_dict"]
        if sd and all(k.startswith("module.") for k in sd):
            sd = {k[7:]: v for k, v in sd.items()}
        model.load_state_dict(sd)
        if optimizer is not None and "optim_state_dict" in checkpoint:
            try:
                optimizer.load_state_dict(checkpoint["optim_state_dict"])
            except ValueError:
                pass  # optimiser built for a different parameter set: irrelevant for inference
        loss = checkpoint["loss"]
    else:
        print("=> no checkpoint found at '{}'".format(filename))
    return model, start_epoch, optimizer, loss

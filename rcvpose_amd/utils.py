"""Host-side helpers the reference keeps in utils.py (:1-54) and that `--mode=test` touches: the run configuration,
the command line's boolean flag type and the checkpoint reader.  Same names and return values; own code."""
import argparse
import pathlib

_TRUE = frozenset(("yes", "true", "t", "y", "1"))
_FALSE = frozenset(("no", "false", "f", "n", "0"))


def str2bool(value):
    """argparse `type=` for flags written as text (reference utils.py:29-37): yes/true/t/y/1 and no/false/f/n/0."""
    if isinstance(value, bool):
        return value
    word = value.lower()
    if word in _TRUE or word in _FALSE:
        return word in _TRUE
    raise argparse.ArgumentTypeError("Boolean value expected.")


def get_config():
    """The reference's single run configuration, keyed 1 (utils.py:16-27); only `max_iteration` is ever read, by training."""
    cfg = {"max_iteration": 700000, "lr": 1e-4, "momentum": 0.99, "betas": (0.9, 0.999), "weight_decay": 0, "interval_validate": 1000}
    return {1: cfg}


def get_log_dir(model_name, cfg):
    """logs/<model_name>/ with the configuration dumped next to the logs (reference utils.py:6-13)."""
    import yaml

    folder = pathlib.Path("logs") / model_name
    folder.mkdir(parents=True, exist_ok=True)
    (folder / "config.yaml").write_text(yaml.safe_dump(cfg, default_flow_style=False))
    return str(folder)


def load_checkpoint(model, optimizer, filename="model_best.pth.tar"):
    """-> (model, epoch, optimizer, loss), reference utils.py:39-54.  The file is the dict train.py:119-128 saves (epoch,
    model_state_dict, optim_state_dict, loss).  A missing file prints the reference's note and hands the model back
    untouched (epoch 0, loss []).  Checkpoints written from nn.DataParallel (`module.` key prefix, reference README.md:13,52)
    load as well; an optimiser that does not fit the stored state is left alone (irrelevant for inference)."""
    import torch

    path = pathlib.Path(filename)
    if not path.is_file():
        print("=> no checkpoint found at '{}'".format(filename))
        return model, 0, optimizer, []
    stored = torch.load(str(path), map_location="cpu", weights_only=False)
    weights = stored["model_state_dict"]
    if weights and all(name.startswith("module.") for name in weights):
        weights = {name[len("module."):]: tensor for name, tensor in weights.items()}
    model.load_state_dict(weights)
    if optimizer is not None and "optim_state_dict" in stored:
        try:
            optimizer.load_state_dict(stored["optim_state_dict"])
        except ValueError:
            pass
    return model, stored["epoch"], optimizer, stored["loss"]

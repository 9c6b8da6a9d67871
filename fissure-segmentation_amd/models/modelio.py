"""Checkpoint contract of the reference (models/modelio.py:20-89): constructors record their
arguments in `self.config`; `save` writes {'config', 'model_state'}; `load` rebuilds `cls(**config)`
and loads the weights non-strictly.  Written against inspect.signature (getargspec is gone in 3.11)."""
import functools
import inspect

import torch
from torch import nn


def store_config_args(init):
    params = [p for p in inspect.signature(init).parameters.values()][1:]  # drop self
    positional = [p for p in params if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]

    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        cfg = {p.name: p.default for p in positional if p.default is not inspect.Parameter.empty}
        cfg.update({p.name: a for p, a in zip(positional, args)})
        cfg.update(kwargs)
        self.config = cfg
        return init(self, *args, **kwargs)
    return wrapped


class LoadableModel(nn.Module):
    def __init__(self, *args, **kwargs):
        if not hasattr(self, "config"):
            raise RuntimeError("decorate the constructor of a LoadableModel with @store_config_args")
        super().__init__(*args, **kwargs)

    def save(self, path):
        state = {k: v for k, v in self.state_dict().items() if not k.endswith(".grid")}
        torch.save({"config": self.config, "model_state": state}, path)

    @classmethod
    def load(cls, path, device):
        ckpt = torch.load(path, map_location=torch.device(device))
        model = cls(**ckpt["config"])
        model.load_state_dict(ckpt["model_state"], strict=False)
        return model

"""Checkpoint I/O for both models.

Reference behaviour being replaced:
  * waveglow/train.py:41-60 pickles the whole model OBJECT (`{'model': model_for_saving, 'iteration', 'optimizer',
    'learning_rate'}`); waveglow/inference.py:37 and inference.py:66 read it back with `torch.load(path)['model']`, which
    only works while a class `WaveGlow` is importable from a top-level module called `glow`.
  * train.py:66-76,125-135 stores Tacotron as `{'iteration', 'state_dict', 'optimizer', 'learning_rate'}`.

Here both are written as plain state_dicts (+ the constructor config, so a file is self-describing), and BOTH legacy
layouts still load: a pickled reference object is unpickled with this package's classes standing in for module `glow`
(they keep the reference's class names and `state_dict` keys), and only its `state_dict()` is used - the weights end up in a
freshly constructed MI355X-native model.
"""
import contextlib
import os
import sys

import torch

FORMAT = "t2s-state-dict-v1"


def save_checkpoint(model, optimizer, learning_rate, iteration, filepath, config=None):
    """Same call as the reference's save_checkpoint (waveglow/train.py:52, train.py:66), state_dict layout."""
    if config is None:
        config = getattr(model, "config", None)
    payload = {"format": FORMAT, "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "config": config,
               "iteration": int(iteration), "optimizer": optimizer.state_dict() if optimizer is not None else None,
               "learning_rate": learning_rate}
    tmp = filepath + ".tmp"
    torch.save(payload, tmp)
    os.replace(tmp, filepath)                        # never leave a half-written checkpoint behind


@contextlib.contextmanager
def _legacy_glow_module():
    """While unpickling, `glow.WaveGlow` / `glow.WN` / `glow.Invertible1x1Conv` resolve to this package's classes."""
    from . import glow as our_glow
    had = sys.modules.get("glow")
    sys.modules["glow"] = our_glow
    try:
        yield
    finally:
        if had is None:
            sys.modules.pop("glow", None)
        else:
            sys.modules["glow"] = had


def _load_file(path):
    assert os.path.isfile(path), path
    with _legacy_glow_module():
        return torch.load(path, map_location="cpu", weights_only=False)


def state_dict_of(checkpoint):
    """The model weights of any supported layout."""
    if "state_dict" in checkpoint:                   # this package, or the reference's Tacotron files
        return checkpoint["state_dict"]
    if "model" in checkpoint:                        # reference WaveGlow files: a pickled module object
        m = checkpoint["model"]
        return m if isinstance(m, dict) else m.state_dict()
    raise KeyError("no 'state_dict' or 'model' entry in the checkpoint")


def waveglow_config_from_state_dict(sd, hint=None):
    """Constructor arguments of glow.WaveGlow recovered from tensor shapes (and, for the early-output schedule, from the
    attributes the reference object carries when there is one)."""
    n_mel = sd["upsample.weight"].shape[0]
    flows = sorted({int(k.split(".")[1]) for k in sd if k.startswith("convinv.")})
    chans = [sd["convinv.%d.conv.weight" % k].shape[0] for k in flows]
    n_group = chans[0]
    n_early_every = getattr(hint, "n_early_every", None)
    n_early_size = getattr(hint, "n_early_size", None)
    if n_early_every is None or n_early_size is None:
        drops = [k for k in range(1, len(chans)) if chans[k] < chans[k - 1]]
        if drops:
            n_early_every, n_early_size = drops[0], chans[drops[0] - 1] - chans[drops[0]]
        else:
            n_early_every, n_early_size = len(chans) + 1, 2
    n_layers = len({int(k.split(".")[3]) for k in sd if k.startswith("WN.0.in_layers.")})
    v = sd.get("WN.0.in_layers.0.weight_v", sd.get("WN.0.in_layers.0.weight"))
    return dict(n_mel_channels=n_mel, n_flows=len(flows), n_group=n_group, n_early_every=int(n_early_every),
                n_early_size=int(n_early_size), WN_config=dict(n_layers=n_layers, n_channels=v.shape[1], kernel_size=v.shape[2]))


def load_checkpoint(checkpoint_path, model, optimizer=None):
    """Reference signature (waveglow/train.py:41, train.py:125): returns (model, optimizer, iteration)."""
    ck = _load_file(checkpoint_path)
    model.load_state_dict(state_dict_of(ck))
    if optimizer is not None and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    iteration = int(ck.get("iteration", 0))
    print("Loaded checkpoint '{}' (iteration {})".format(checkpoint_path, iteration))
    return model, optimizer, iteration


def load_waveglow(path, device="cuda"):
    """Drop-in for `torch.load(path)['model']` (waveglow/inference.py:37, inference.py:66): a ready WaveGlow on `device`,
    whatever layout the file has."""
    from .glow import WaveGlow
    ck = _load_file(path)
    sd = state_dict_of(ck)
    cfg = ck.get("config") or waveglow_config_from_state_dict(sd, ck.get("model") if not isinstance(ck.get("model"), dict) else None)
    if not any(k.endswith("weight_g") for k in sd):
        raise ValueError("checkpoint has weight-norm already removed; load it into a model after WaveGlow.remove_weightnorm")
    model = WaveGlow(**cfg)
    model.load_state_dict(sd, strict=True)
    return model.to(device)

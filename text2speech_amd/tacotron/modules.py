"""Parameter containers with the reference's names and ``state_dict`` keys
(reference tacotron/modules.py:11-22,94-137,248-284).  They hold weights; the math runs in
libt2s_hip.so (see tacotron.py).  Only the live classes are provided — the Tacotron-1 leftovers
(CBHG, Highway, ...) are never instantiated by the reference model."""
import torch
from torch import nn


def owner_engine(mod):
    """The engine of the Tacotron this sub-module belongs to.  The sub-modules are parameter containers; their ``forward`` (kept
    for callers that use the reference's module surface directly, reference modules.py:19-22,131-137, tacotron.py:192-220,
    395-466) runs the same kernels as the whole-model paths.  Forward only: training goes through ``Tacotron.forward``."""
    ref = mod.__dict__.get("_owner")
    owner = ref() if ref is not None else None
    if owner is None:
        raise RuntimeError("%s.forward runs on the engine of the Tacotron it belongs to; build it through Tacotron(...)"
                           % type(mod).__name__)
    return owner._eng()


class OwnedModule(nn.Module):
    """nn.Module whose back-reference to its owner (a weak reference in __dict__) stays out of pickles and deep copies; the owner
    re-establishes it (Tacotron._adopt)."""

    def __getstate__(self):
        d = self.__dict__.copy()
        d.pop("_owner", None)
        d.pop("_step", None)        # Decoder: the device-side step state (raw pointers) is not part of the model
        return d


class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain="linear"):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))


class ConvNorm(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1, bias=True,
                 w_init_gain="linear"):
        super().__init__()
        if padding is None:
            assert kernel_size % 2 == 1
            padding = int(dilation * (kernel_size - 1) / 2)
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                              dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))


class Prenet(OwnedModule):
    def __init__(self, in_dim, sizes):
        super().__init__()
        in_sizes = [in_dim] + sizes[:-1]
        self.layers = nn.ModuleList([LinearNorm(i, o, bias=False) for i, o in zip(in_sizes, sizes)])

    def forward(self, x, masks=None):
        """Reference modules.py:19-22 (dropout 0.5 with training=True ALWAYS).  masks [items, 1, 2, prenet_dim] of {0,1} injects
        the draws."""
        with torch.no_grad():
            return owner_engine(self).prenet_forward(x, masks)


class Postnet(OwnedModule):
    """Five 1-d convolutions + BatchNorm (reference modules.py:94-129)."""

    def __init__(self, hparams):
        super().__init__()
        n_mel, emb, ks, n = (hparams["n_mel_channels"], hparams["postnet_embedding_dim"],
                             hparams["postnet_kernel_size"], hparams["postnet_n_convolutions"])
        dims = [n_mel] + [emb] * (n - 1) + [n_mel]
        self.convolutions = nn.ModuleList()
        for i in range(n):
            gain = "tanh" if i < n - 1 else "linear"
            self.convolutions.append(nn.Sequential(
                ConvNorm(dims[i], dims[i + 1], kernel_size=ks, stride=1, padding=(ks - 1) // 2, dilation=1,
                         w_init_gain=gain),
                nn.BatchNorm1d(dims[i + 1])))

    def forward(self, x, train_masks=None):
        """Reference modules.py:131-137: x [B, n_mel, T] -> the postnet's residual (the caller adds it to x)."""
        with torch.no_grad():
            eng = owner_engine(self)
            eng.prepare(x.device)
            return eng.postnet(x.detach().to(torch.float32), train_masks, eng.fresh_seed())


def get_mask_from_lengths(lengths):
    """True on valid positions (reference modules.py:280-284 returns the same mask as uint8 and is
    CUDA-only; callers negate it)."""
    max_len = int(torch.max(lengths).item())
    ids = torch.arange(0, max_len, device=lengths.device)
    return ids < lengths.unsqueeze(1)

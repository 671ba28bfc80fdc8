"""Seeded synthetic weights and inputs for the WaveGlow / Tacotron-2 hot path.

Everything here is generated from CPU ``torch.Generator`` streams so that the
same tensors can be rebuilt bit-for-bit in the survey container (where the
reference is imported to make golden vectors) and on the GPU box (where the
reference does not exist).  Shapes and key names follow the reference
``state_dict`` layout (reference waveglow/glow.py:111-152,
179-205 and tacotron/tacotron.py:15-34,167-260).

Nothing in this file is on the product path: it only feeds tests, bench.py and
the golden-vector generator.
"""
import math
from collections import OrderedDict

import torch

WAVEGLOW_DEFAULT = dict(
    n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4, n_early_size=2,
    WN_config=dict(n_layers=8, n_channels=512, kernel_size=3))

WAVEGLOW_SMALL = dict(
    n_mel_channels=80, n_flows=12, n_group=8, n_early_every=4, n_early_size=2,
    WN_config=dict(n_layers=8, n_channels=64, kernel_size=3))


def _randn(gen, *shape, std=1.0):
    return torch.randn(*shape, generator=gen, dtype=torch.float32) * std


def waveglow_flow_sizes(cfg):
    """(n_remaining_channels, n_half) per flow — reference glow.py:193-205."""
    n_half = cfg["n_group"] // 2
    n_rem = cfg["n_group"]
    out = []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def waveglow_state(cfg=WAVEGLOW_DEFAULT, seed=1234, end_seed=7, end_std=0.02, wn_gain=1.0):
    """Deterministic WaveGlow ``state_dict`` with weight-norm (g, v) pairs.

    ``WN.end`` is zero-initialised in the reference (glow.py:128-130), which
    makes every coupling an identity; it is overwritten with N(0, end_std^2)
    from its own stream so the whole WN stack is exercised.
    ``end_std`` / ``wn_gain`` (a factor on every weight-norm gain of the in / cond / res_skip layers) make the stress weights
    of the parity tests: 0.03 / 1.25 gives max |log_s| 3-4 instead of 1.3 (tools/numerics_study.py, profiles/r03_numerics.md).
    """
    gen = torch.Generator().manual_seed(seed)
    gen_end = torch.Generator().manual_seed(end_seed)
    n_mel = cfg["n_mel_channels"]
    wn = cfg["WN_config"]
    C, nl, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
    n_cond = n_mel * cfg["n_group"]
    sd = OrderedDict()
    sd["upsample.weight"] = _randn(gen, n_mel, n_mel, 1024, std=1.0 / math.sqrt(4 * n_mel))
    sd["upsample.bias"] = _randn(gen, n_mel, std=0.05)

    def wn_pair(prefix, out_c, in_c, k, gain=1.0):
        v = _randn(gen, out_c, in_c, k, std=1.0 / math.sqrt(in_c * k))
        g = v.flatten(1).norm(dim=1).view(out_c, 1, 1) * (gain * (0.75 + 0.5 * torch.rand(out_c, 1, 1, generator=gen)))
        sd[prefix + ".bias"] = _randn(gen, out_c, std=0.05)
        sd[prefix + ".weight_g"] = g
        sd[prefix + ".weight_v"] = v

    for k, (n_rem, n_half) in enumerate(waveglow_flow_sizes(cfg)):
        for i in range(nl):
            wn_pair(f"WN.{k}.in_layers.{i}", 2 * C, C, ks, gain=wn_gain)
        for i in range(nl):
            rs = 2 * C if i < nl - 1 else C
            wn_pair(f"WN.{k}.res_skip_layers.{i}", rs, C, 1, gain=0.5 * wn_gain)
        for i in range(nl):
            wn_pair(f"WN.{k}.cond_layers.{i}", 2 * C, n_cond, 1, gain=wn_gain)
        wn_pair(f"WN.{k}.start", C, n_half, 1)
        sd[f"WN.{k}.end.weight"] = _randn(gen_end, 2 * n_half, C, 1, std=end_std)
        sd[f"WN.{k}.end.bias"] = _randn(gen_end, 2 * n_half, std=end_std)
    for k, (n_rem, n_half) in enumerate(waveglow_flow_sizes(cfg)):
        q, _ = torch.linalg.qr(_randn(gen, n_rem, n_rem))
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        # a mild, well-conditioned departure from orthonormal so logdet != 0
        q = q @ torch.diag(0.8 + 0.4 * torch.rand(n_rem, generator=gen))
        sd[f"convinv.{k}.conv.weight"] = q.contiguous().view(n_rem, n_rem, 1)
    return sd


def waveglow_inputs(batch, n_samples, n_mel=80, hop=256, seed=31):
    """mel ~ N(0,1) [B,n_mel,F], audio ~ U(-0.5,0.5) [B,T]."""
    gen = torch.Generator().manual_seed(seed)
    frames = n_samples // hop + 1
    mel = _randn(gen, batch, n_mel, frames)
    audio = torch.rand(batch, n_samples, generator=gen, dtype=torch.float32) - 0.5
    return mel, audio


# --------------------------------------------------------------------------- Tacotron-2

TACOTRON_HPARAMS = {
    # the keys the reference model reads (hparams.py:98-172)
    "embedding_size": 512, "speaker_embedding_size": 16,
    "enc_conv_num_layers": 3, "enc_conv_channels": 512, "enc_conv_kernel_size": 5,
    "n_mel_channels": 80, "n_frames_per_step": 1,
    "attention_rnn_dim": 1024, "decoder_rnn_dim": 1024, "prenet_dim": 256,
    "max_decoder_steps": 1000, "gate_threshold": 0.5,
    "p_attention_dropout": 0.1, "p_decoder_dropout": 0.1,
    "attention_dim": 128, "attention_location_n_filters": 32,
    "attention_location_kernel_size": 31,
    "postnet_embedding_dim": 512, "postnet_kernel_size": 5, "postnet_n_convolutions": 5,
    "mask_padding": True,
}


def tacotron_state(hp=TACOTRON_HPARAMS, n_vocab=80, num_speakers=2, seed=4321):
    """Deterministic Tacotron-2 ``state_dict`` (key names: SURVEY.md 8b)."""
    gen = torch.Generator().manual_seed(seed)
    E = hp["embedding_size"]
    Cc = hp["enc_conv_channels"]
    ks = hp["enc_conv_kernel_size"]
    n_mel = hp["n_mel_channels"] * hp["n_frames_per_step"]
    A, D, P = hp["attention_rnn_dim"], hp["decoder_rnn_dim"], hp["prenet_dim"]
    ad, nf, lk = hp["attention_dim"], hp["attention_location_n_filters"], hp["attention_location_kernel_size"]
    sd = OrderedDict()

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        sd[name + ".weight"] = _randn(gen, out_f, in_f, std=gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = _randn(gen, out_f, std=0.05)

    def conv(name, out_c, in_c, k, bias=True, gain=1.0):
        sd[name + ".weight"] = _randn(gen, out_c, in_c, k, std=gain / math.sqrt(in_c * k))
        if bias:
            sd[name + ".bias"] = _randn(gen, out_c, std=0.05)

    def bn(name, c):
        sd[name + ".weight"] = 0.75 + 0.5 * torch.rand(c, generator=gen)
        sd[name + ".bias"] = _randn(gen, c, std=0.1)
        sd[name + ".running_mean"] = _randn(gen, c, std=0.1)
        sd[name + ".running_var"] = 0.5 + torch.rand(c, generator=gen)
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def lstm_cell(name, in_f, hid, suffix=""):
        s = 1.0 / math.sqrt(hid)
        sd[f"{name}.weight_ih{suffix}"] = (torch.rand(4 * hid, in_f, generator=gen) * 2 - 1) * s
        sd[f"{name}.weight_hh{suffix}"] = (torch.rand(4 * hid, hid, generator=gen) * 2 - 1) * s
        sd[f"{name}.bias_ih{suffix}"] = (torch.rand(4 * hid, generator=gen) * 2 - 1) * s
        sd[f"{name}.bias_hh{suffix}"] = (torch.rand(4 * hid, generator=gen) * 2 - 1) * s

    sd["embedding.weight"] = _randn(gen, n_vocab, E, std=0.3)
    sd["speaker_embed_table.weight"] = _randn(gen, num_speakers, hp["speaker_embedding_size"])
    lin("deep_linear", 512, hp["speaker_embedding_size"])
    for i in range(hp["enc_conv_num_layers"]):
        conv(f"encoder.convolutions.{i}.0.conv", Cc, Cc, ks, gain=1.4)
        bn(f"encoder.convolutions.{i}.1", Cc)
    lstm_cell("encoder.lstm", Cc, Cc // 2, "_l0")
    lstm_cell("encoder.lstm", Cc, Cc // 2, "_l0_reverse")
    lin("decoder.prenet.layers.0.linear_layer", P, n_mel, bias=False, gain=1.4)
    lin("decoder.prenet.layers.1.linear_layer", P, P, bias=False, gain=1.4)
    lstm_cell("decoder.attention_rnn", P + Cc, A)
    lin("decoder.attention_layer.query_layer.linear_layer", ad, A, bias=False)
    lin("decoder.attention_layer.memory_layer.linear_layer", ad, Cc, bias=False)
    lin("decoder.attention_layer.v.linear_layer", 1, ad, bias=False, gain=4.0)
    conv("decoder.attention_layer.location_layer.location_conv.conv", nf, 2, lk, bias=False)
    lin("decoder.attention_layer.location_layer.location_dense.linear_layer", ad, nf, bias=False)
    lstm_cell("decoder.decoder_rnn", A + Cc, D)
    lin("decoder.linear_projection.linear_layer", n_mel, D + Cc)
    lin("decoder.gate_layer.linear_layer", 1, D + Cc)
    pe, pk, pn = hp["postnet_embedding_dim"], hp["postnet_kernel_size"], hp["postnet_n_convolutions"]
    dims = [hp["n_mel_channels"]] + [pe] * (pn - 1) + [hp["n_mel_channels"]]
    for i in range(pn):
        conv(f"postnet.convolutions.{i}.0.conv", dims[i + 1], dims[i], pk)
        bn(f"postnet.convolutions.{i}.1", dims[i + 1])
    return sd

#!/usr/bin/env python3
"""Generate Tacotron-2 golden vectors by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference); writes data only
(tests/golden/tacotron_*.npz).  Inputs/weights are regenerated from seeds by
text2speech_amd.synth.  Harness shims, per SURVEY.md 8c (none touches arithmetic):
  * empty stub modules for absent third-party imports (librosa, jamo, unidecode, inflect, nltk)
    that the model code never calls;
  * tacotron.tacotron.get_mask_from_lengths rebound to a CPU bool version with the intended meaning
    (the reference's is CUDA-only, modules.py:280-284);
  * dropout draws are captured by replaying the global RNG stream with the same seed
    (verified: F.dropout(x, p, True) == x * empty_like(x).bernoulli_(1-p) / (1-p) draw for draw).

usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_tacotron.py
"""
import os
import sys
import types
import warnings

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for name in ["librosa", "librosa.filters", "librosa.core", "librosa.util", "jamo", "unidecode", "inflect", "nltk",
             "nltk.tokenize"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["librosa.filters"].mel = None
sys.modules["librosa.core"].load = None
sys.modules["librosa.util"].pad_center = None
sys.modules["librosa.util"].tiny = None
for fn in ("hangul_to_jamo", "h2j", "j2h"):
    setattr(sys.modules["jamo"], fn, None)
sys.modules["unidecode"].unidecode = None
sys.modules["inflect"].engine = lambda *a, **k: None
sys.path.insert(0, "/root/reference")

import tacotron.tacotron as ref_taco  # noqa: E402  (the reference)
from tacotron.loss_function import Tacotron2Loss  # noqa: E402
from hparams import hparams as ref_hparams  # noqa: E402
from text2speech_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def cpu_mask_from_lengths(lengths):
    ids = torch.arange(0, int(torch.max(lengths).item()))
    return ids < lengths.unsqueeze(1)


ref_taco.get_mask_from_lengths = cpu_mask_from_lengths


def build():
    torch.manual_seed(0)
    m = ref_taco.Tacotron(ref_hparams, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state(), strict=True)
    return m


def bern(shape, keep):
    return torch.empty(*shape).bernoulli_(keep)


def pack(m):
    return np.packbits(m.numpy().astype(np.uint8).reshape(-1))


def gen_inference(n_steps=200, seed=11):
    m = build().eval()
    text = (torch.arange(64) % 78 + 2)[None]
    m.decoder.gate_threshold = 2.0
    m.decoder.max_decoder_steps = n_steps
    torch.manual_seed(seed)
    masks = torch.stack([torch.stack([bern((1, 256), 0.5), bern((1, 256), 0.5)], 1) for _ in range(n_steps)])
    torch.manual_seed(seed)
    with torch.no_grad():
        mel, mel_post, gate, align = m.inference(text, None)
        enc = m.encoder.inference(m.embedding(text).transpose(1, 2))
    np.savez_compressed(os.path.join(OUT, "tacotron_infer.npz"), mel=mel.numpy(), mel_post=mel_post.numpy(),
                        gate=gate.numpy(), align=align.numpy(), enc=enc.numpy(), prenet_masks=pack(masks),
                        prenet_masks_shape=np.array(masks.shape))
    print("infer", tuple(mel.shape), tuple(gate.shape), float(mel.std()), float(align.max()))


def ragged_batch(seed=21, B=4, T_in=40, T_out=50):
    gen = torch.Generator().manual_seed(seed)
    in_len = torch.tensor([T_in, T_in - 4, T_in - 9, T_in - 20])[:B]
    out_len = torch.tensor([T_out, T_out - 6, T_out - 13, T_out - 25])[:B]
    text = torch.randint(2, 80, (B, T_in), generator=gen)
    mel = torch.randn(B, 80, T_out, generator=gen)
    for b in range(B):
        text[b, in_len[b]:] = 0
        mel[b, :, out_len[b]:] = 0
    gate = torch.zeros(B, T_out)
    for b in range(B):
        gate[b, out_len[b] - 1:] = 1
    return text, in_len, mel, gate, out_len


def gen_forward_eval(seed=12):
    m = build().eval()
    text, in_len, mel, gate_t, out_len = ragged_batch()
    B, _, T_out = mel.shape
    torch.manual_seed(seed)
    m1, m2 = bern((T_out + 1, B, 256), 0.5), bern((T_out + 1, B, 256), 0.5)
    masks = torch.stack([m1, m2], 2)
    torch.manual_seed(seed)
    with torch.no_grad():
        out = m((text, in_len, mel, int(in_len.max()), torch.zeros(B), out_len))
        enc = m.encoder(m.embedding(text).transpose(1, 2), in_len)
        loss = Tacotron2Loss()(out, (mel, gate_t))
    np.savez_compressed(os.path.join(OUT, "tacotron_fwd_eval.npz"), mel=out[0].numpy(), mel_post=out[1].numpy(),
                        gate=out[2].numpy(), align=out[3].numpy(), enc=enc.numpy(), loss=np.float64(loss.item()),
                        prenet_masks=pack(masks), prenet_masks_shape=np.array(masks.shape))
    print("fwd_eval", [tuple(o.shape) for o in out], float(loss))


def gen_forward_train(seed=13):
    m = build().train()
    text, in_len, mel, gate_t, out_len = ragged_batch()
    B, _, T_out = mel.shape
    T_in = text.size(1)
    torch.manual_seed(seed)
    enc_m = [bern((B, 512, T_in), 0.5) for _ in range(3)]
    m1, m2 = bern((T_out + 1, B, 256), 0.5), bern((T_out + 1, B, 256), 0.5)
    att_m, dec_m = [], []
    for _ in range(T_out):
        att_m.append(bern((B, 1024), 0.9))
        dec_m.append(bern((B, 1024), 0.9))
    post_m = [bern((B, 512 if i < 4 else 80, T_out), 0.5) for i in range(5)]
    torch.manual_seed(seed)
    out = m((text, in_len, mel, int(in_len.max()), torch.zeros(B), out_len))
    loss = Tacotron2Loss()(out, (mel, gate_t))
    loss.backward()
    named = dict(m.named_parameters())
    grads = {}
    for key in ["decoder.attention_rnn.weight_hh", "decoder.linear_projection.linear_layer.weight",
                "decoder.attention_layer.location_layer.location_conv.conv.weight",
                "encoder.lstm.weight_hh_l0_reverse", "encoder.convolutions.0.0.conv.weight",
                "postnet.convolutions.4.0.conv.weight", "embedding.weight", "decoder.prenet.layers.0.linear_layer.weight"]:
        g = named[key].grad.detach().flatten()
        step = max(1, g.numel() // 16384)
        grads["grad::" + key] = g[::step].contiguous().numpy()
        grads["gradsq::" + key] = np.float64((g.double() ** 2).sum().item())
    np.savez_compressed(os.path.join(OUT, "tacotron_fwd_train.npz"), mel=out[0].detach().numpy(),
                        mel_post=out[1].detach().numpy(), gate=out[2].detach().numpy(), align=out[3].detach().numpy(),
                        loss=np.float64(loss.item()),
                        enc_masks=pack(torch.stack(enc_m)), prenet_masks=pack(torch.stack([m1, m2], 2)),
                        att_masks=pack(torch.stack(att_m)), dec_masks=pack(torch.stack(dec_m)),
                        post_masks_512=pack(torch.stack(post_m[:4])), post_masks_80=pack(post_m[4]), **grads)
    print("fwd_train loss", float(loss))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    gen_inference()
    gen_forward_eval()
    gen_forward_train()

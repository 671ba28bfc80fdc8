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
                "postnet.convolutions.4.0.conv.weight", "embedding.weight", "decoder.prenet.layers.0.linear_layer.weight",
                # its saved input is zeroed on padded frames by parse_output's .data.masked_fill_ before backward runs
                "postnet.convolutions.0.0.conv.weight"]:
        g = named[key].grad.detach().flatten()
        step = max(1, g.numel() // 16384)
        grads["grad::" + key] = g[::step].contiguous().numpy()
        grads["gradsq::" + key] = np.float64((g.double() ** 2).sum().item())
    names = sorted(n for n, p_ in named.items() if p_.grad is not None)
    grads["all_names"] = np.array(names)
    grads["all_gradsq"] = np.array([(named[n].grad.double() ** 2).sum().item() for n in names], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "tacotron_fwd_train.npz"), mel=out[0].detach().numpy(),
                        mel_post=out[1].detach().numpy(), gate=out[2].detach().numpy(), align=out[3].detach().numpy(),
                        loss=np.float64(loss.item()),
                        enc_masks=pack(torch.stack(enc_m)), prenet_masks=pack(torch.stack([m1, m2], 2)),
                        att_masks=pack(torch.stack(att_m)), dec_masks=pack(torch.stack(dec_m)),
                        post_masks_512=pack(torch.stack(post_m[:4])), post_masks_80=pack(post_m[4]), **grads)
    print("fwd_train loss", float(loss))


def full_size_batch(seed=21, B=32, T_in=256, T_out=800):
    """BASELINE configs[1] / SURVEY.md 8d row 2: ragged input_lengths 256-4i, output_lengths 800-12i (the same draws as
    tests/test_tacotron_train_gpu.py::_ragged)."""
    gen = torch.Generator().manual_seed(seed)
    in_len = torch.tensor([T_in - 4 * i for i in range(B)])
    out_len = torch.tensor([T_out - 12 * i for i in range(B)])
    text = torch.randint(2, 80, (B, T_in), generator=gen)
    mel = torch.randn(B, 80, T_out, generator=gen)
    gate = torch.zeros(B, T_out)
    for b in range(B):
        text[b, in_len[b]:] = 0
        mel[b, :, out_len[b]:] = 0
        gate[b, out_len[b] - 1:] = 1
    return text, in_len, mel, gate, out_len


def full_size_masks(seed, B, T_in, T_out):
    """The dropout draws of ONE reference training forward at this shape, in the order the reference makes them on the global
    CPU RNG (encoder convs, the hoisted prenet over T_out+1 frames, per decoder step attention-LSTM then decoder-LSTM, postnet).
    The test regenerates them from the seed with this same sequence of calls; `mask_sums` pins that it got the same bits."""
    torch.manual_seed(seed)
    enc_m = [bern((B, 512, T_in), 0.5) for _ in range(3)]
    m1, m2 = bern((T_out + 1, B, 256), 0.5), bern((T_out + 1, B, 256), 0.5)
    att_m, dec_m = [], []
    for _ in range(T_out):
        att_m.append(bern((B, 1024), 0.9))
        dec_m.append(bern((B, 1024), 0.9))
    post_m = [bern((B, 512 if i < 4 else 80, T_out), 0.5) for i in range(5)]
    sums = [float(t.double().sum()) for t in enc_m] + [float(m1.double().sum()), float(m2.double().sum()),
            float(torch.stack(att_m).double().sum()), float(torch.stack(dec_m).double().sum())] + \
        [float(t.double().sum()) for t in post_m]
    # a position-weighted checksum as well: equal counts with permuted bits must not pass
    w = lambda t: float((t.flatten().double() * (torch.arange(t.numel(), dtype=torch.float64) % 9973 + 1)).sum())
    wsums = [w(enc_m[0]), w(m1), w(torch.stack(att_m)), w(torch.stack(dec_m)), w(post_m[0]), w(post_m[4])]
    return np.array(sums + wsums, dtype=np.float64)


def gen_forward_train_full(seed=23):
    """Loss, outputs' checksums and EVERY parameter gradient's sum / squared norm of the reference at the benchmarked training
    shape (B=32, T_in 256, T_out 800, ragged).  ~3 min and ~10 GB on 8 cores."""
    B, T_in, T_out = 32, 256, 800
    m = build().train()
    text, in_len, mel, gate_t, out_len = full_size_batch(B=B, T_in=T_in, T_out=T_out)
    mask_sums = full_size_masks(seed, B, T_in, T_out)
    torch.manual_seed(seed)
    out = m((text, in_len, mel, int(in_len.max()), torch.zeros(B), out_len))
    loss = Tacotron2Loss()(out, (mel, gate_t))
    print("fwd_train_full forward done, loss", float(loss), flush=True)
    loss.backward()
    named = {n: p for n, p in m.named_parameters() if p.grad is not None}
    names = sorted(named)
    res = {"loss": np.float64(loss.item()), "seed": np.int64(seed), "mask_sums": mask_sums, "all_names": np.array(names),
           "all_gradsum": np.array([named[n].grad.double().sum().item() for n in names], dtype=np.float64),
           "all_gradsq": np.array([(named[n].grad.double() ** 2).sum().item() for n in names], dtype=np.float64)}
    for key in ["decoder.attention_rnn.weight_hh", "decoder.decoder_rnn.weight_ih", "decoder.linear_projection.linear_layer.weight",
                "decoder.attention_layer.location_layer.location_conv.conv.weight", "decoder.attention_layer.query_layer.linear_layer.weight",
                "encoder.lstm.weight_hh_l0_reverse", "encoder.convolutions.0.0.conv.weight", "postnet.convolutions.4.0.conv.weight",
                "embedding.weight", "decoder.prenet.layers.0.linear_layer.weight"]:
        g = named[key].grad.detach().flatten()
        step = max(1, g.numel() // 4096)
        res["grad::" + key] = g[::step].contiguous().numpy()
    for i, nm in enumerate(["mel", "mel_post", "gate", "align"]):
        o = out[i].detach().double()
        if nm == "gate":
            o = torch.where(o == 1e3, torch.zeros_like(o), o)       # parse_output's fill on padded frames
        res[nm + "_sum"] = np.float64(o.sum().item())
        res[nm + "_sq"] = np.float64((o ** 2).sum().item())
    # rows of the outputs at a few frames pin them individually (mel of entry 0 and entry 31 at frames 0, 399, last valid)
    res["mel_rows"] = torch.stack([out[1][0, :, 0], out[1][0, :, 399], out[1][0, :, 799], out[1][31, :, 0], out[1][31, :, 427]]).detach().numpy()
    res["align_rows"] = torch.stack([out[3][0, 10], out[3][0, 700], out[3][31, 400]]).detach().numpy()
    np.savez_compressed(os.path.join(OUT, "tacotron_train_full.npz"), **res)
    print("fwd_train_full loss", float(loss))


def gen_inference_long(n_steps=1000, seed=31):
    """BASELINE configs[4]: 1000 forced decoder frames (B = 1, 64 symbols).  The prenet's always-on dropout draws are regenerated
    by the test from the seed (2 x 256 bits per frame in the order below); mel / gate / alignment rows at frames 0, 199, 499, 999
    and whole-tensor checksums are stored."""
    m = build().eval()
    text = (torch.arange(64) % 78 + 2)[None]
    m.decoder.gate_threshold = 2.0
    m.decoder.max_decoder_steps = n_steps
    torch.manual_seed(seed)
    masks = torch.stack([torch.stack([bern((1, 256), 0.5), bern((1, 256), 0.5)], 1) for _ in range(n_steps)])
    torch.manual_seed(seed)
    with torch.no_grad():
        mel, mel_post, gate, align = m.inference(text, None)
    frames = [0, 199, 499, 999]
    np.savez_compressed(os.path.join(OUT, "tacotron_infer_1000.npz"), seed=np.int64(seed), frames=np.array(frames),
                        mel=mel[0, :, frames].numpy(), mel_post=mel_post[0, :, frames].numpy(), gate=gate[0, frames].numpy(),
                        align=align[0, frames].numpy(), mask_sum=np.float64(masks.double().sum().item()),
                        mel_sq_by_100=np.array([(mel[0, :, i:i + 100].double() ** 2).sum().item() for i in range(0, n_steps, 100)]),
                        mel_post_full=mel_post[0].numpy().astype(np.float32))
    print("infer_1000", tuple(mel.shape), float(mel.std()), float(align.max()))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--train-full" in sys.argv:
        gen_forward_train_full()
        sys.exit(0)
    if "--infer-long" in sys.argv:
        gen_inference_long()
        sys.exit(0)
    gen_inference()
    gen_forward_eval()
    gen_forward_train()

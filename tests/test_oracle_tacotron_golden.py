"""The Tacotron-2 CPU oracle against vectors produced by the reference itself
(tools/gen_golden_tacotron.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import tacotron_oracle as O
from text2speech_amd import synth

HP = synth.TACOTRON_HPARAMS


def _rel(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def unpack(g, key, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(g[key])[:n].reshape(shape).astype(np.float32))


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


def test_inference(golden_dir):
    g = np.load(os.path.join(golden_dir, "tacotron_infer.npz"))
    sd = synth.tacotron_state()
    text = (torch.arange(64) % 78 + 2)[None]
    masks = unpack(g, "prenet_masks", tuple(g["prenet_masks_shape"]))
    with torch.no_grad():
        enc = O.encoder(sd, HP, text)
        mel, mel_post, gate, align = O.tacotron_inference(sd, HP, text, 200, masks)
    assert _rel(enc, g["enc"]) < 1e-5
    assert _rel(mel, g["mel"]) < 1e-4
    assert _rel(mel_post, g["mel_post"]) < 1e-4
    assert _rel(gate, g["gate"]) < 1e-4
    assert _rel(align, g["align"]) < 1e-4


def test_forward_eval_ragged(golden_dir):
    g = np.load(os.path.join(golden_dir, "tacotron_fwd_eval.npz"))
    sd = synth.tacotron_state()
    text, in_len, mel_t, gate_t, out_len = ragged_batch()
    masks = {"prenet": unpack(g, "prenet_masks", tuple(g["prenet_masks_shape"]))}
    with torch.no_grad():
        enc = O.encoder(sd, HP, text, in_len)
        out = O.tacotron_forward(sd, HP, text, in_len, mel_t, out_len, masks)
        loss = O.tacotron_loss(out, mel_t, gate_t)
    assert _rel(enc, g["enc"]) < 1e-5
    assert _rel(out[0], g["mel"]) < 1e-4
    assert _rel(out[1], g["mel_post"]) < 1e-4
    assert _rel(out[2], g["gate"]) < 1e-4
    assert _rel(out[3], g["align"]) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-4
    # parse_output semantics (reference tacotron.py:67-76): zeros / 1e3 beyond output_lengths
    assert float(out[0][3, :, 25:].abs().max()) == 0.0 and float(out[2][3, 25:].min()) == 1e3


def test_forward_train_with_captured_dropout(golden_dir):
    g = np.load(os.path.join(golden_dir, "tacotron_fwd_train.npz"))
    sd = synth.tacotron_state()
    text, in_len, mel_t, gate_t, out_len = ragged_batch()
    B, T_in, T_out = 4, 40, 50
    masks = {
        "enc": unpack(g, "enc_masks", (3, B, 512, T_in)),
        "prenet": unpack(g, "prenet_masks", (T_out + 1, B, 2, 256)),
        "att": unpack(g, "att_masks", (T_out, B, 1024)),
        "dec": unpack(g, "dec_masks", (T_out, B, 1024)),
        "post": list(unpack(g, "post_masks_512", (4, B, 512, T_out))) + [unpack(g, "post_masks_80", (B, 80, T_out))],
    }
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    out = O.tacotron_forward(sd, HP, text, in_len, mel_t, out_len, masks, training=True)
    loss = O.tacotron_loss(out, mel_t, gate_t)
    loss.backward()
    assert _rel(out[0].detach(), g["mel"]) < 2e-4
    assert _rel(out[1].detach(), g["mel_post"]) < 2e-4
    assert _rel(out[3].detach(), g["align"]) < 2e-4
    assert abs(float(loss) - float(g["loss"])) < 2e-4
    # every parameter gradient of the reference (squared norms) and the sampled tensors.  Includes the first postnet
    # convolution's weight, whose saved input the reference's parse_output zeroes on padded frames before backward runs
    # (tacotron.py:73 `.data.masked_fill_`): plain autograd of the same forward differs there by tens of percent.
    for n, gq in zip((str(x) for x in g["all_names"]), g["all_gradsq"]):
        sq = float((sd[n].grad.double() ** 2).sum())
        assert abs(sq - gq) <= 2e-3 * gq + 1e-10, (n, sq, gq)
    for key in g.files:
        if key.startswith("grad::"):
            flat = sd[key[len("grad::"):]].grad.flatten()
            assert _rel(flat[::max(1, flat.numel() // 16384)], g[key]) < 1e-3, key


def test_oracle_inference_1000_frames_vs_reference():
    """The oracle against the reference's own 1000-frame decode (BASELINE configs[4]) with the reference's prenet-dropout draws
    regenerated from the seed: no drift beyond f32 rounding (1e-5 at frame 999)."""
    import os
    import numpy as np
    import torch
    from text2speech_amd import synth
    from oracle import tacotron_oracle as O
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tacotron_infer_1000.npz"))
    n = 1000
    torch.manual_seed(int(g["seed"]))
    bern = lambda: torch.empty(1, 256).bernoulli_(0.5)
    masks = torch.stack([torch.stack([bern(), bern()], 1) for _ in range(n)])
    assert float(masks.double().sum()) == float(g["mask_sum"])
    ids = (torch.arange(64) % 78 + 2)[None]
    with torch.no_grad():
        mel, post, gate, align = O.tacotron_inference(synth.tacotron_state(), synth.TACOTRON_HPARAMS, ids, n, masks)
    rel = lambda a, b: float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).norm() / torch.as_tensor(b).double().norm())
    for i, f in enumerate(int(x) for x in g["frames"]):
        assert rel(mel[0, :, f], g["mel"][:, i]) < 2e-5 and rel(align[0, f], g["align"][i]) < 5e-5, f
    assert rel(post[0], g["mel_post_full"]) < 1e-5

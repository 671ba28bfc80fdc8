"""GPU parity of the Tacotron-2 training step: training-mode teacher-forced forward -> Tacotron2Loss -> hand-written
backward, every parameter gradient against CPU autograd through the oracle with the same dropout masks, and against
the gradients the reference itself produced (tests/golden/tacotron_fwd_train.npz)."""
import os

import numpy as np
import pytest
import torch

from text2speech_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HP = synth.TACOTRON_HPARAMS


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu().flatten()
    b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def unpack(g, key, shape):
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(g[key])[:n].reshape(shape).astype(np.uint8))


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


@pytest.fixture(scope="module")
def trained(golden_dir):
    from oracle import tacotron_oracle as O
    from text2speech_amd.tacotron import Tacotron
    _lib.load()
    g = np.load(os.path.join(golden_dir, "tacotron_fwd_train.npz"))
    sd = synth.tacotron_state()
    m = Tacotron(HP, 80, num_speakers=2)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    text, in_len, mel_t, gate_t, out_len = ragged_batch()
    B, T_in, T_out = 4, 40, 50
    tm = {"enc": list(unpack(g, "enc_masks", (3, B, 512, T_in))), "att": unpack(g, "att_masks", (T_out, B, 1024)),
          "dec": unpack(g, "dec_masks", (T_out, B, 1024)),
          "post": list(unpack(g, "post_masks_512", (4, B, 512, T_out))) + [unpack(g, "post_masks_80", (B, 80, T_out))]}
    pm = unpack(g, "prenet_masks", (T_out + 1, B, 2, 256))
    out = m((text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV)),
            prenet_masks=pm, train_masks=tm)
    mt, gt = mel_t.to(DEV), gate_t.to(DEV)
    loss = torch.nn.functional.mse_loss(out[0], mt) + torch.nn.functional.mse_loss(out[1], mt) + \
        torch.nn.functional.binary_cross_entropy_with_logits(out[2].reshape(-1, 1), gt.reshape(-1, 1))
    loss.backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    # CPU autograd through the oracle with the same masks
    sd_cpu = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    masks = {"enc": [t.float() for t in tm["enc"]], "prenet": pm.float(), "att": tm["att"].float(), "dec": tm["dec"].float(),
             "post": [t.float() for t in tm["post"]]}
    oo = O.tacotron_forward(sd_cpu, HP, text, in_len, mel_t, out_len, masks, training=True)
    lo = O.tacotron_loss(oo, mel_t, gate_t)
    lo.backward()
    want = {k: v.grad for k, v in sd_cpu.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    return dict(got=got, want=want, loss=float(loss), loss_o=float(lo), g=g)


def test_loss_matches(trained):
    assert abs(trained["loss"] - trained["loss_o"]) < 2e-4 * max(1.0, abs(trained["loss_o"]))
    assert abs(trained["loss"] - float(trained["g"]["loss"])) < 2e-4 * max(1.0, abs(trained["loss_o"]))


@pytest.mark.parametrize("prefix", ["postnet.", "decoder.linear_projection", "decoder.gate_layer", "decoder.decoder_rnn",
                                    "decoder.attention_rnn", "decoder.attention_layer", "decoder.prenet", "encoder.lstm",
                                    "encoder.convolutions", "embedding"])
def test_param_grads_vs_oracle(trained, prefix):
    got, want = trained["got"], trained["want"]
    names = [n for n in want if n.startswith(prefix)]
    assert names, prefix
    worst = []
    for n in names:
        assert n in got, "no gradient for " + n
        assert got[n].shape == want[n].shape, n
        # a conv bias in front of a batch-statistics BatchNorm has an exactly-zero gradient: both sides are rounding
        # noise there, so differences below an absolute floor count as equal
        diff = float((got[n].double() - want[n].double()).norm())
        worst.append((0.0 if diff < 2e-5 else _rel(got[n], want[n]), n))
    worst.sort(reverse=True)
    assert worst[0][0] < 5e-3, [(r, n, float(got[n].norm()), float(want[n].norm())) for r, n in worst[:6]]


def test_grads_vs_reference_golden(trained):
    g = trained["g"]
    for key in g.files:
        if not key.startswith("grad::"):
            continue
        name = key[len("grad::"):]
        flat = trained["got"][name].flatten()
        step = max(1, flat.numel() // 16384)
        assert _rel(flat[::step], g[key]) < 5e-3, name


def test_batch12_step_vs_oracle():
    """Batch 12 (> 8 items: the LSTM cells and the BPTT data gradients take the f32 matrix-core path, csrc/sbgemm.hip):
    loss and every parameter gradient against CPU autograd through the oracle, seeded masks, ragged lengths."""
    from oracle import tacotron_oracle as O
    from text2speech_amd.tacotron import Tacotron
    _lib.load()
    B, T_in, T_out = 12, 22, 20       # T_in % 4 != 0: the in-loop d_memory accumulation (the deferred form needs % 4)
    gen = torch.Generator().manual_seed(5)
    in_len = torch.tensor([T_in - (i * 3) // 2 for i in range(B)])
    out_len = torch.tensor([T_out - i for i in range(B)])
    text = torch.randint(2, 80, (B, T_in), generator=gen)
    mel_t = torch.randn(B, 80, T_out, generator=gen)
    gate_t = torch.zeros(B, T_out)
    for b in range(B):
        text[b, in_len[b]:] = 0
        mel_t[b, :, out_len[b]:] = 0
        gate_t[b, out_len[b] - 1:] = 1
    bern = lambda *s: (torch.rand(*s, generator=gen) < 0.5).to(torch.uint8)
    bern9 = lambda *s: (torch.rand(*s, generator=gen) < 0.9).to(torch.uint8)
    tm = {"enc": [bern(B, 512, T_in) for _ in range(3)], "att": bern9(T_out, B, 1024), "dec": bern9(T_out, B, 1024),
          "post": [bern(B, 512, T_out) for _ in range(4)] + [bern(B, 80, T_out)]}
    pm = bern(T_out + 1, B, 2, 256)
    sd = synth.tacotron_state()
    m = Tacotron(HP, 80, num_speakers=2)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    out = m((text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV)),
            prenet_masks=pm, train_masks=tm)
    mt, gt = mel_t.to(DEV), gate_t.to(DEV)
    loss = torch.nn.functional.mse_loss(out[0], mt) + torch.nn.functional.mse_loss(out[1], mt) + \
        torch.nn.functional.binary_cross_entropy_with_logits(out[2].reshape(-1, 1), gt.reshape(-1, 1))
    loss.backward()
    torch.cuda.synchronize()
    got = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    sd_cpu = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    masks = {"enc": [t.float() for t in tm["enc"]], "prenet": pm.float(), "att": tm["att"].float(), "dec": tm["dec"].float(),
             "post": [t.float() for t in tm["post"]]}
    oo = O.tacotron_forward(sd_cpu, HP, text, in_len, mel_t, out_len, masks, training=True)
    lo = O.tacotron_loss(oo, mel_t, gate_t)
    lo.backward()
    assert abs(float(loss) - float(lo)) < 2e-4 * max(1.0, abs(float(lo)))
    for k in range(4):
        assert _rel(out[k], oo[k].detach()) < 1e-3, k
    worst = []
    for n, v in sd_cpu.items():
        if not (torch.is_tensor(v) and v.is_floating_point() and v.grad is not None):
            continue
        assert n in got, n
        diff = float((got[n].double() - v.grad.double()).norm())
        worst.append((0.0 if diff < 2e-5 else _rel(got[n], v.grad), n))
    worst.sort(reverse=True)
    assert worst[0][0] < 5e-3, worst[:6]


def test_tacotron2loss_matches_torch():
    """Tacotron2Loss (csrc/loss_ops.hip) against the reference's formula in stock torch ops: value and the three gradients."""
    from text2speech_amd.tacotron.loss_function import Tacotron2Loss
    _lib.load()
    gen = torch.Generator().manual_seed(3)
    B, T = 5, 37
    mel = torch.randn(B, 80, T, generator=gen)
    post = torch.randn(B, 80, T, generator=gen)
    gate = torch.randn(B, T, generator=gen) * 4
    tgt = torch.randn(B, 80, T, generator=gen)
    gt = (torch.rand(B, T, generator=gen) < 0.2).float()
    a = [t.clone().to(DEV).requires_grad_(True) for t in (mel, post, gate)]
    loss = Tacotron2Loss()([a[0], a[1], a[2], None], (tgt.to(DEV), gt.to(DEV)))
    (loss * 1.7).backward()
    b = [t.clone().double().requires_grad_(True) for t in (mel, post, gate)]
    ref = torch.nn.functional.mse_loss(b[0], tgt.double()) + torch.nn.functional.mse_loss(b[1], tgt.double()) + \
        torch.nn.functional.binary_cross_entropy_with_logits(b[2].view(-1, 1), gt.double().view(-1, 1))
    (ref * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 1e-6 * max(1.0, abs(float(ref)))
    for x, y in zip(a, b):
        assert _rel(x.grad, y.grad) < 1e-6
    with pytest.raises(Exception):
        Tacotron2Loss()([mel, post, gate, None], (tgt, gt))          # host tensors: no CPU path


def test_backward_is_bitwise_reproducible():
    """The BPTT runs on three HIP streams (decoder-cell chain, attention chain, location-conv backward) and uses no atomics:
    the same step three times must give bit-identical gradients."""
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    _lib.load()
    B, T_in, T_out = 10, 36, 30
    gen = torch.Generator().manual_seed(8)
    in_len = torch.tensor([T_in - i for i in range(B)])
    out_len = torch.tensor([T_out - i for i in range(B)])
    text = torch.randint(2, 80, (B, T_in), generator=gen)
    mel_t = torch.randn(B, 80, T_out, generator=gen)
    gate_t = torch.zeros(B, T_out)
    for b in range(B):
        text[b, in_len[b]:] = 0
        mel_t[b, :, out_len[b]:] = 0
        gate_t[b, out_len[b] - 1:] = 1
    bern = lambda *s: (torch.rand(*s, generator=gen) < 0.5).to(torch.uint8)
    tm = {"enc": [bern(B, 512, T_in) for _ in range(3)], "att": bern(T_out, B, 1024), "dec": bern(T_out, B, 1024),
          "post": [bern(B, 512, T_out) for _ in range(4)] + [bern(B, 80, T_out)]}
    pm = bern(T_out + 1, B, 2, 256)
    m = Tacotron(HP, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state(), strict=True)
    m = m.to(DEV).train()
    crit = Tacotron2Loss()
    runs = []
    for _ in range(3):
        m.load_state_dict(synth.tacotron_state(), strict=True)       # BatchNorm running statistics back to the start
        m.zero_grad(set_to_none=True)
        out = m((text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV)),
                prenet_masks=pm, train_masks=tm)
        crit(out, (mel_t.to(DEV), gate_t.to(DEV))).backward()
        torch.cuda.synchronize()
        runs.append({n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    for n in runs[0]:
        assert torch.equal(runs[0][n], runs[1][n]) and torch.equal(runs[0][n], runs[2][n]), n

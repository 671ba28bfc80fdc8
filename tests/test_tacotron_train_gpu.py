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


def _ragged(B, T_in, T_out, gen, din=None, dout=None):
    in_len = torch.tensor([T_in - (din(i) if din else i) for i in range(B)])
    out_len = torch.tensor([T_out - (dout(i) if dout else i) for i in range(B)])
    text = torch.randint(2, 80, (B, T_in), generator=gen)
    mel_t = torch.randn(B, 80, T_out, generator=gen)
    gate_t = torch.zeros(B, T_out)
    for b in range(B):
        text[b, in_len[b]:] = 0
        mel_t[b, :, out_len[b]:] = 0
        gate_t[b, out_len[b] - 1:] = 1
    return text, in_len, mel_t, gate_t, out_len


def _seeded_masks(B, T_in, T_out, gen):
    bern = lambda *s: (torch.rand(*s, generator=gen) < 0.5).to(torch.uint8)
    bern9 = lambda *s: (torch.rand(*s, generator=gen) < 0.9).to(torch.uint8)
    tm = {"enc": [bern(B, 512, T_in) for _ in range(3)], "att": bern9(T_out, B, 1024), "dec": bern9(T_out, B, 1024),
          "post": [bern(B, 512, T_out) for _ in range(4)] + [bern(B, 80, T_out)]}
    pm = bern(T_out + 1, B, 2, 256)
    masks = {"enc": [t.float() for t in tm["enc"]], "prenet": pm.float(), "att": tm["att"].float(), "dec": tm["dec"].float(),
             "post": [t.float() for t in tm["post"]]}
    return tm, pm, masks


@pytest.mark.parametrize("B,T_in,T_out", [
    (12, 22, 20),     # > 8 items: f32 matrix-core cells (csrc/sbgemm.hip); T_in % 4 != 0: in-loop d_memory accumulation
    (32, 64, 64),     # BASELINE configs[1] batch (train.py:216-225): two 16-item groups per sbgemm workgroup, deferred d_memory
])
def test_step_vs_oracle(B, T_in, T_out):
    """Loss, the four outputs and every parameter gradient against CPU autograd through the oracle, seeded masks, ragged
    lengths."""
    from oracle import tacotron_oracle as O
    from text2speech_amd.tacotron import Tacotron
    _lib.load()
    gen = torch.Generator().manual_seed(5)
    text, in_len, mel_t, gate_t, out_len = _ragged(B, T_in, T_out, gen, din=lambda i: (i * 3) // 2 if B == 12 else i)
    tm, pm, masks = _seeded_masks(B, T_in, T_out, gen)
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


def test_three_adam_steps_vs_oracle_batch32():
    """zero_grad -> forward -> Tacotron2Loss -> backward -> Adam (reference train.py:216-225; Adam lr / weight decay as
    train.py:187-189) three times at B=32 on the GPU (FusedAdam) and through the oracle (torch.optim.Adam on CPU) with the
    same injected masks: the loss of every step within 1e-3 rel, the weights after the third step within 5e-3 rel.  (Adam
    divides by sqrt(v): where the true gradient is zero - the conv biases in front of a batch-statistics BatchNorm - the update
    is the sign of rounding noise on both sides, so those tensors are left out; elsewhere near-zero elements make the weights a
    looser check than the losses.)"""
    from oracle import tacotron_oracle as O
    from text2speech_amd.optim import FusedAdam
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    _lib.load()
    B, T_in, T_out = 32, 48, 40
    gen = torch.Generator().manual_seed(17)
    text, in_len, mel_t, gate_t, out_len = _ragged(B, T_in, T_out, gen)
    tm, pm, masks = _seeded_masks(B, T_in, T_out, gen)
    sd = synth.tacotron_state()
    m = Tacotron(HP, 80, num_speakers=2)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    crit = Tacotron2Loss()
    x = (text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV))
    y = (mel_t.to(DEV), gate_t.to(DEV))
    losses = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        loss = crit(m(x, prenet_masks=pm, train_masks=tm), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    # oracle side: same steps with stock torch on the host
    sd_cpu = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
              for k, v in sd.items()}
    params = [v for v in sd_cpu.values() if torch.is_tensor(v) and v.requires_grad]
    opt_o = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-6)
    losses_o = []
    for _ in range(3):
        opt_o.zero_grad(set_to_none=True)
        lo = O.tacotron_loss(O.tacotron_forward(sd_cpu, HP, text, in_len, mel_t, out_len, masks, training=True), mel_t, gate_t)
        lo.backward()
        opt_o.step()
        losses_o.append(float(lo))
    for a, b in zip(losses, losses_o):
        assert abs(a - b) < 1e-3 * abs(b), (losses, losses_o)
    assert losses[2] < losses[0]
    worst = []
    for n, p in m.named_parameters():
        v = sd_cpu[n]
        if v.requires_grad and v.grad is not None and not (".conv.bias" in n and "convolutions" in n):
            worst.append((_rel(p.detach(), v.detach()), n))
    worst.sort(reverse=True)
    assert len(worst) > 40 and worst[0][0] < 5e-3, worst[:6]


def test_full_size_config_step_properties():
    """BASELINE configs[1] at full size (B=32, T_in=256, T_out=800, ragged as SURVEY.md 8d: input_lengths 256-4i,
    output_lengths 800-12i; ~20 GB of saves, three-stream BPTT): every output and gradient finite, the same seeded step
    twice gives bit-identical gradients (device-drawn masks reproduce under torch.manual_seed), and three FusedAdam steps
    on the fixed batch lower the loss."""
    from text2speech_amd.optim import FusedAdam
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    _lib.load()
    B, T_in, T_out = 32, 256, 800
    gen = torch.Generator().manual_seed(21)
    text, in_len, mel_t, gate_t, out_len = _ragged(B, T_in, T_out, gen, din=lambda i: 4 * i, dout=lambda i: 12 * i)
    x = (text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV))
    y = (mel_t.to(DEV), gate_t.to(DEV))
    crit = Tacotron2Loss()

    def one_step(seed):
        m = Tacotron(HP, 80, num_speakers=2)
        m.load_state_dict(synth.tacotron_state(), strict=True)
        m = m.to(DEV).train()
        torch.manual_seed(seed)
        out = m(x)
        loss = crit(out, y)
        loss.backward()
        torch.cuda.synchronize()
        return m, out, float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    m, out, loss_a, ga = one_step(3)
    assert len(ga) == len([p for p in m.parameters()]) - 3      # speaker table + deep_linear are dead weights (tacotron.py:27-29)
    for t in out:
        assert bool(torch.isfinite(t).all())
    for n, g_ in ga.items():
        assert bool(torch.isfinite(g_).all()), n
        assert float(g_.abs().max()) > 0 or "convolutions" in n, n
    # padded positions are masked exactly as the reference's parse_output does (tacotron.py:67-76)
    assert float(out[0][5, :, int(out_len[5]):].abs().max()) == 0.0 and float(out[2][5, int(out_len[5]):].min()) == 1e3
    _, _, loss_b, gb = one_step(3)
    assert loss_a == loss_b
    for n in ga:
        assert torch.equal(ga[n], gb[n]), n
    _, _, loss_c, _ = one_step(4)
    assert loss_c != loss_a          # a different seed draws different dropout masks
    del ga, gb
    opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)
    tm = None
    losses = []
    torch.manual_seed(11)
    for _ in range(4):
        m.zero_grad(set_to_none=True)
        torch.manual_seed(11)          # fixed masks: the loss sequence is then a pure function of the weights
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert losses[-1] < losses[0], losses


def _reference_draw_order_masks(seed, B, T_in, T_out):
    """The dropout draws of one reference training forward, regenerated from the seed in the order the reference makes them on
    the global CPU RNG (tools/gen_golden_tacotron.py::full_size_masks)."""
    bern = lambda shape, keep: torch.empty(*shape).bernoulli_(keep)
    torch.manual_seed(seed)
    enc_m = [bern((B, 512, T_in), 0.5) for _ in range(3)]
    m1, m2 = bern((T_out + 1, B, 256), 0.5), bern((T_out + 1, B, 256), 0.5)
    att_m, dec_m = [], []
    for _ in range(T_out):
        att_m.append(bern((B, 1024), 0.9))
        dec_m.append(bern((B, 1024), 0.9))
    post_m = [bern((B, 512 if i < 4 else 80, T_out), 0.5) for i in range(5)]
    att_m, dec_m = torch.stack(att_m), torch.stack(dec_m)
    w = lambda t: float((t.flatten().double() * (torch.arange(t.numel(), dtype=torch.float64) % 9973 + 1)).sum())
    sums = [float(t.double().sum()) for t in enc_m] + [float(m1.double().sum()), float(m2.double().sum()),
                                                      float(att_m.double().sum()), float(dec_m.double().sum())] + \
        [float(t.double().sum()) for t in post_m] + [w(enc_m[0]), w(m1), w(att_m), w(dec_m), w(post_m[0]), w(post_m[4])]
    u8 = lambda t: t.to(torch.uint8)
    tm = {"enc": [u8(t) for t in enc_m], "att": u8(att_m), "dec": u8(dec_m), "post": [u8(t) for t in post_m]}
    return tm, u8(torch.stack([m1, m2], 2)), np.array(sums, dtype=np.float64)


def test_benchmarked_shape_b32_t800_vs_reference_golden(golden_dir):
    """BASELINE configs[1] at the shape bench.py times (B=32, T_in 256, T_out 800, ragged as SURVEY.md 8d) against the
    REFERENCE's own training step (tests/golden/tacotron_train_full.npz, tools/gen_golden_tacotron.py --train-full): the loss,
    checksums and rows of the four outputs, the squared norm of every parameter gradient and strided samples of ten tensors.
    The dropout masks are the reference's own draws, regenerated from the shipped seed (checked bit-count and weighted sums)."""
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    _lib.load()
    g = np.load(os.path.join(golden_dir, "tacotron_train_full.npz"))
    B, T_in, T_out = 32, 256, 800
    tm, pm, sums = _reference_draw_order_masks(int(g["seed"]), B, T_in, T_out)
    assert np.array_equal(sums, g["mask_sums"]), "the CPU RNG on this box does not reproduce the reference's dropout draws"
    gen = torch.Generator().manual_seed(21)
    text, in_len, mel_t, gate_t, out_len = _ragged(B, T_in, T_out, gen, din=lambda i: 4 * i, dout=lambda i: 12 * i)
    m = Tacotron(HP, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state(), strict=True)
    m = m.to(DEV).train()
    x = (text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV))
    out = m(x, prenet_masks=pm, train_masks=tm)
    loss = Tacotron2Loss()(out, (mel_t.to(DEV), gate_t.to(DEV)))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(g["loss"])) < 2e-4 * float(g["loss"]), (float(loss), float(g["loss"]))
    for i, nm in enumerate(["mel", "mel_post", "gate", "align"]):
        o = out[i].detach().double()
        if nm == "gate":
            o = torch.where(o == 1e3, torch.zeros_like(o), o)
        sq = float((o ** 2).sum())
        assert abs(sq - float(g[nm + "_sq"])) < 2e-3 * float(g[nm + "_sq"]), nm
    rows = torch.stack([out[1][0, :, 0], out[1][0, :, 399], out[1][0, :, 799], out[1][31, :, 0], out[1][31, :, 427]])
    assert _rel(rows, g["mel_rows"]) < 1e-3
    arows = torch.stack([out[3][0, 10], out[3][0, 700], out[3][31, 400]])
    assert _rel(arows, g["align_rows"]) < 1e-3
    got = {n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None}
    names = [str(n) for n in g["all_names"]]
    assert sorted(names) == sorted(got)
    worst_sq = []
    for n, gq in zip(names, g["all_gradsq"]):
        sq = float((got[n].double() ** 2).sum())
        # conv biases in front of a batch-statistics BatchNorm: the true gradient is zero, both sides are rounding noise
        if gq < 1e-9:
            assert sq < 1e-8, (n, sq, gq)
            continue
        worst_sq.append((abs(sq - gq) / gq, n))
    worst_sq.sort(reverse=True)
    assert worst_sq[0][0] < 1e-2, worst_sq[:6]
    worst = []
    for key in g.files:
        if key.startswith("grad::"):
            name = key[len("grad::"):]
            flat = got[name].flatten()
            worst.append((_rel(flat[::max(1, flat.numel() // 4096)], g[key]), name))
    worst.sort(reverse=True)
    assert len(worst) == 10 and worst[0][0] < 5e-3, worst[:4]
    m._eng().check_lstm_xbuf()          # no hand-off wait of the split BiLSTM kernels (forward and BPTT) expired
    print("B=32 T_out=800 step vs reference: loss %.6f / %.6f, worst |dsq|/sq %.2e (%s), worst sampled rel %.2e (%s)"
          % (float(loss), float(g["loss"]), worst_sq[0][0], worst_sq[0][1], worst[0][0], worst[0][1]))


def test_pooled_buffers_carry_nothing_between_steps():
    """The large save / scratch buffers of the training step come from a per-shape pool and are NOT cleared between steps
    (tacotron.pool_take): whatever a step leaves in them must not reach the next one.  Two batches of ONE shape but different
    ragged lengths (the second one's valid regions are shorter, so stale data of the first would sit right behind them), same
    model object: the second step's outputs and gradients must be bit-identical to the same step on a fresh model."""
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    _lib.load()
    B, T_in, T_out = 12, 40, 36
    gen = torch.Generator().manual_seed(9)
    batches = [_ragged(B, T_in, T_out, gen, din=lambda i: 0, dout=lambda i: 0),
               _ragged(B, T_in, T_out, gen, din=lambda i: 2 * i, dout=lambda i: 2 * i if i else 0)]
    masks = [_seeded_masks(B, T_in, T_out, gen) for _ in batches]
    crit = Tacotron2Loss()

    def step(m, k):
        text, in_len, mel_t, gate_t, out_len = batches[k]
        tm, pm, _ = masks[k]
        m.load_state_dict(synth.tacotron_state(), strict=True)          # BatchNorm running statistics back to the start
        m.zero_grad(set_to_none=True)
        out = m((text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV)),
                prenet_masks=pm, train_masks=tm)
        crit(out, (mel_t.to(DEV), gate_t.to(DEV))).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in out], {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    def fresh():
        m = Tacotron(HP, 80, num_speakers=2)
        m.load_state_dict(synth.tacotron_state(), strict=True)
        return m.to(DEV).train()

    m = fresh()
    step(m, 0)                       # full-length batch first: fills every pooled buffer up to the shape's extent
    out_b, grads_b = step(m, 1)      # shorter valid regions on the same buffers
    out_f, grads_f = step(fresh(), 1)
    for a, b in zip(out_b, out_f):
        assert torch.equal(a, b)
    assert sorted(grads_b) == sorted(grads_f)
    for n in grads_f:
        assert torch.equal(grads_b[n], grads_f[n]), n


def test_pooled_planes_cleared_when_padded_length_shrinks():
    """ADVICE r3 (high x2): operand planes are pooled by SHAPE, and the padded row count Lp = ceil(T/256)*256 + 2*halo is the
    same for 256 different T.  The reference's collate pads every batch to its own maximum (utils/data_utils.py:117), so a batch
    padded to T = 37/33 follows one padded to 40/36 on the same buffers: rows 33..35 must read as the k=5 convolutions' ZERO
    padding (forward: encoder / postnet layers and, through the batch statistics, every output; backward: the weight-gradient
    contraction over the whole plane and the data gradient's last two frames).  Second step bit-identical to a fresh model; the
    pool is also run with a tiny cap so that eviction happens between the two steps of a third model."""
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    from text2speech_amd.tacotron.tacotron import BufferPool
    _lib.load()
    B = 6
    gen = torch.Generator().manual_seed(19)
    shapes = [(40, 36), (37, 33)]
    batches = [_ragged(B, ti, to, gen, din=lambda i: 0, dout=lambda i: 0) for ti, to in shapes]
    masks = [_seeded_masks(B, ti, to, gen) for ti, to in shapes]
    crit = Tacotron2Loss()

    def step(m, k):
        text, in_len, mel_t, gate_t, out_len = batches[k]
        tm, pm, _ = masks[k]
        m.load_state_dict(synth.tacotron_state(), strict=True)
        m.zero_grad(set_to_none=True)
        out = m((text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV)),
                prenet_masks=pm, train_masks=tm)
        crit(out, (mel_t.to(DEV), gate_t.to(DEV))).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in out], {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    def fresh(cap=None):
        m = Tacotron(HP, 80, num_speakers=2)
        m.load_state_dict(synth.tacotron_state(), strict=True)
        m = m.to(DEV).train()
        if cap is not None:
            m._eng().pool = BufferPool(cap_bytes=cap)
        return m

    out_f, grads_f = step(fresh(), 1)
    m = fresh()
    step(m, 0)
    out_b, grads_b = step(m, 1)
    for a, b in zip(out_b, out_f):
        assert torch.equal(a, b)
    assert sorted(grads_b) == sorted(grads_f)
    for n in grads_f:
        assert torch.equal(grads_b[n], grads_f[n]), n
    # bounded pool: a 1 MB cap evicts nearly everything a step returns; results unchanged, pool stays under the cap
    m2 = fresh(cap=1 << 20)
    step(m2, 0)
    out_c, grads_c = step(m2, 1)
    pool = m2._eng().pool
    m2.__dict__.pop("_last_bwd", None)
    assert pool.evicted > 0 and pool.free_bytes <= pool.cap_bytes, (pool.evicted, pool.free_bytes)
    for a, b in zip(out_c, out_f):
        assert torch.equal(a, b)
    for n in grads_f:
        assert torch.equal(grads_c[n], grads_f[n]), n


def test_dropout_masks_fresh_per_call_and_reproducible():
    """Every training forward / inference draws NEW dropout masks (reference: F.dropout on the global RNG,
    modules.py:21, tacotron.py:193,368,383) and torch.manual_seed reproduces them."""
    from text2speech_amd.tacotron import Tacotron
    _lib.load()
    B, T_in, T_out = 3, 24, 16
    gen = torch.Generator().manual_seed(2)
    text, in_len, mel_t, gate_t, out_len = _ragged(B, T_in, T_out, gen)
    m = Tacotron(HP, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state(), strict=True)
    m = m.to(DEV).train()
    x = (text.to(DEV), in_len.to(DEV), mel_t.to(DEV), int(in_len.max()), torch.zeros(B, device=DEV), out_len.to(DEV))
    with torch.no_grad():
        torch.manual_seed(77)
        a = [t.clone() for t in m(x)]
        b = [t.clone() for t in m(x)]
        torch.manual_seed(77)
        c = [t.clone() for t in m(x)]
    assert not torch.equal(a[1], b[1]) and not torch.equal(a[0], b[0])
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2])
    m.eval()
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, 12
    ids = text[:1, :20].to(DEV)
    torch.manual_seed(5)
    i1 = m.inference(ids, None)[0].clone()
    i2 = m.inference(ids, None)[0].clone()
    torch.manual_seed(5)
    i3 = m.inference(ids, None)[0].clone()
    assert not torch.equal(i1, i2) and torch.equal(i1, i3)      # the always-on prenet dropout (modules.py:21)


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

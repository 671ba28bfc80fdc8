"""ORACLE — test infrastructure, not product code.

CPU restatement (stock PyTorch ops) of the reference Tacotron-2 path, as pure
functions over a ``state_dict`` with the reference's key names.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED by ``tools/gen_golden_tacotron.py`` (runs the reference with
import stubs for absent third-party packages) ->
``tests/golden/tacotron_*.npz`` -> ``tests/test_oracle_tacotron_golden.py``.

Dropout is an explicit input everywhere: ``masks`` are {0,1} tensors that the
reference drew from the global RNG (the prenet's dropout is always on,
reference modules.py:21).
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _conv_bn(sd, prefix, x, training, x_saved=None):
    """ConvNorm + BatchNorm1d (reference tacotron.py:177-186 / modules.py:105-129).  x_saved: what the reference's autograd
    holds as this convolution's input when its backward runs, if that differs from x (see postnet)."""
    w, b = sd[prefix + ".0.conv.weight"], sd[prefix + ".0.conv.bias"]
    pad = (w.size(2) - 1) // 2
    if x_saved is None:
        y = F.conv1d(x, w, b, padding=pad)
    else:
        # value and d/dx, d/dbias from conv(x, w); d/dw from x_saved: the second term is zero in value and carries only that
        ghost = F.conv1d(x_saved.detach(), w, None, padding=pad)
        y = F.conv1d(x, w.detach(), b, padding=pad) + (ghost - ghost.detach())
    if training:
        mean = y.mean(dim=(0, 2))
        var = y.var(dim=(0, 2), unbiased=False)
    else:
        mean, var = sd[prefix + ".1.running_mean"], sd[prefix + ".1.running_var"]
    y = (y - mean.view(1, -1, 1)) / torch.sqrt(var.view(1, -1, 1) + BN_EPS)
    return y * sd[prefix + ".1.weight"].view(1, -1, 1) + sd[prefix + ".1.bias"].view(1, -1, 1)


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch LSTMCell semantics, gate order i, f, g, o."""
    gates = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    H = h.size(1)
    i, f, g, o = gates[:, :H], gates[:, H:2 * H], gates[:, 2 * H:3 * H], gates[:, 3 * H:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def encoder(sd, hp, ids, lengths=None, training=False, masks=None):
    """Embedding + Encoder.forward / .inference (reference tacotron.py:40,192-220).
    ``lengths`` (sorted descending) gives packed-sequence semantics: each direction of the
    BiLSTM runs over the valid steps only and padded outputs are zero."""
    x = F.embedding(ids, sd["embedding.weight"]).transpose(1, 2)
    for i in range(hp["enc_conv_num_layers"]):
        x = F.relu(_conv_bn(sd, f"encoder.convolutions.{i}", x, training))
        if training:
            x = x * masks["enc"][i] * 2.0
    x = x.transpose(1, 2)                                    # [B, T, C]
    B, T, C = x.shape
    H = C // 2
    if lengths is None:
        lengths = torch.full((B,), T, dtype=torch.long)
    T_out = int(lengths.max())
    out = torch.zeros(B, T_out, 2 * H, dtype=x.dtype)
    for d, suf in enumerate(["_l0", "_l0_reverse"]):
        w_ih, w_hh = sd["encoder.lstm.weight_ih" + suf], sd["encoder.lstm.weight_hh" + suf]
        b_ih, b_hh = sd["encoder.lstm.bias_ih" + suf], sd["encoder.lstm.bias_hh" + suf]
        for b in range(B):
            n = int(lengths[b])
            h = torch.zeros(1, H, dtype=x.dtype)
            c = torch.zeros(1, H, dtype=x.dtype)
            steps = range(n) if d == 0 else range(n - 1, -1, -1)
            for t in steps:
                h, c = lstm_cell(x[b:b + 1, t], h, c, w_ih, w_hh, b_ih, b_hh)
                out[b, t, d * H:(d + 1) * H] = h[0]
    return out


def prenet(sd, x, mask):
    """Prenet.forward (reference modules.py:19-22); mask [..., 2, prenet_dim] in {0,1}, p = 0.5."""
    for i in range(2):
        x = F.relu(x @ sd[f"decoder.prenet.layers.{i}.linear_layer.weight"].t()) * mask[..., i, :] * 2.0
    return x


class DecoderState:
    def __init__(self, sd, hp, memory, memory_lengths=None):
        """Decoder.initialize_decoder_states (reference tacotron.py:276-307)."""
        B, T, _ = memory.shape
        z = lambda n: torch.zeros(B, n, dtype=memory.dtype)
        self.ah, self.ac = z(hp["attention_rnn_dim"]), z(hp["attention_rnn_dim"])
        self.dh, self.dc = z(hp["decoder_rnn_dim"]), z(hp["decoder_rnn_dim"])
        self.w, self.wc = z(T), z(T)
        self.ctx = z(memory.size(2))
        self.memory = memory
        self.pmem = memory @ sd["decoder.attention_layer.memory_layer.linear_layer.weight"].t()
        if memory_lengths is not None:
            ids = torch.arange(T)
            self.pad = ~(ids[None, :] < memory_lengths[:, None])     # True on padding (tacotron.py:415)
        else:
            self.pad = None


def decode_step(sd, hp, st, x, drop_a=None, drop_d=None):
    """Decoder.decode (reference tacotron.py:355-393).  x = prenet output [B, prenet_dim].
    drop_a / drop_d: {0,1} masks for the two LSTM-output dropouts in training (p = 0.1)."""
    p = "decoder."
    st.ah, st.ac = lstm_cell(torch.cat((x, st.ctx), -1), st.ah, st.ac,
                             sd[p + "attention_rnn.weight_ih"], sd[p + "attention_rnn.weight_hh"],
                             sd[p + "attention_rnn.bias_ih"], sd[p + "attention_rnn.bias_hh"])
    if drop_a is not None:
        st.ah = st.ah * drop_a / (1.0 - hp["p_attention_dropout"])
    # location-sensitive attention (tacotron.py:124-166)
    a = p + "attention_layer."
    q = st.ah @ sd[a + "query_layer.linear_layer.weight"].t()                       # [B, 128]
    cat = torch.stack((st.w, st.wc), 1)                                             # [B, 2, T]
    lk = sd[a + "location_layer.location_conv.conv.weight"]
    loc = F.conv1d(cat, lk, padding=(lk.size(2) - 1) // 2).transpose(1, 2)          # [B, T, 32]
    loc = loc @ sd[a + "location_layer.location_dense.linear_layer.weight"].t()     # [B, T, 128]
    e = torch.tanh(q.unsqueeze(1) + loc + st.pmem) @ sd[a + "v.linear_layer.weight"].t()
    e = e.squeeze(-1)
    if st.pad is not None:
        e = e.masked_fill(st.pad, -float("inf"))
    st.w = F.softmax(e, dim=1)
    st.ctx = torch.bmm(st.w.unsqueeze(1), st.memory).squeeze(1)
    st.wc = st.wc + st.w
    st.dh, st.dc = lstm_cell(torch.cat((st.ah, st.ctx), -1), st.dh, st.dc,
                             sd[p + "decoder_rnn.weight_ih"], sd[p + "decoder_rnn.weight_hh"],
                             sd[p + "decoder_rnn.bias_ih"], sd[p + "decoder_rnn.bias_hh"])
    if drop_d is not None:
        st.dh = st.dh * drop_d / (1.0 - hp["p_decoder_dropout"])
    hc = torch.cat((st.dh, st.ctx), 1)
    mel = hc @ sd[p + "linear_projection.linear_layer.weight"].t() + sd[p + "linear_projection.linear_layer.bias"]
    gate = hc @ sd[p + "gate_layer.linear_layer.weight"].t() + sd[p + "gate_layer.linear_layer.bias"]
    return mel, gate, st.w


def postnet(sd, hp, x, training=False, masks=None, x_saved=None):
    """Postnet.forward (reference modules.py:131-137).  x_saved: the first convolution's input as the reference's backward
    sees it - Tacotron.parse_output zeroes the padded frames of mel_outputs through ``.data.masked_fill_`` (tacotron.py:73)
    AFTER the postnet ran, on the tensor that convolution saved, so its weight gradient is taken against the masked mel."""
    n = hp["postnet_n_convolutions"]
    for i in range(n):
        x = _conv_bn(sd, f"postnet.convolutions.{i}", x, training, x_saved if i == 0 else None)
        if i < n - 1:
            x = torch.tanh(x)
        if training:
            x = x * masks["post"][i] * 2.0
    return x


def tacotron_forward(sd, hp, text, text_lengths, mels, output_lengths, masks, training=False):
    """Tacotron.forward, teacher forced (reference tacotron.py:36-49,395-429,67-76).
    masks["prenet"]: [T_out+1, B, 2, prenet_dim]."""
    memory = encoder(sd, hp, text, text_lengths, training, masks)
    B, n_mel, T_out = mels.shape
    frames = torch.cat((torch.zeros(1, B, n_mel, dtype=mels.dtype), mels.permute(2, 0, 1)), 0)   # go frame first
    pre = prenet(sd, frames, masks["prenet"])
    st = DecoderState(sd, hp, memory, text_lengths)
    mel_out, gate_out, aligns = [], [], []
    for t in range(T_out):
        da = masks["att"][t] if training else None
        dd = masks["dec"][t] if training else None
        m, g, w = decode_step(sd, hp, st, pre[t], da, dd)
        mel_out.append(m)
        gate_out.append(g.squeeze(1))
        aligns.append(w)
    mel_out = torch.stack(mel_out, 2)                         # [B, n_mel, T_out]
    gate_out = torch.stack(gate_out, 1)                       # [B, T_out]
    aligns = torch.stack(aligns, 1)                           # [B, T_out, T_in]
    pad = None
    if hp["mask_padding"] and output_lengths is not None:
        ids = torch.arange(T_out)
        pad = ~(ids[None, :] < output_lengths[:, None])
    mel_post = mel_out + postnet(sd, hp, mel_out, training, masks,
                                 None if pad is None else mel_out.masked_fill(pad.unsqueeze(1), 0.0))
    if pad is not None:
        mel_out = mel_out.masked_fill(pad.unsqueeze(1), 0.0)
        mel_post = mel_post.masked_fill(pad.unsqueeze(1), 0.0)
        gate_out = gate_out.masked_fill(pad, 1e3)
    return mel_out, mel_post, gate_out, aligns


def tacotron_inference(sd, hp, text, n_steps, prenet_masks, gate_threshold=None, training=False, masks=None):
    """Tacotron.inference (reference tacotron.py:51-65,431-466) for a fixed number of steps
    (or until sigmoid(gate) > gate_threshold).  prenet_masks: [n_steps, B, 2, prenet_dim].  training: the modules are in
    .train() mode as the reference would have them if inference() were called on a training model - batch-statistics BatchNorm
    and dropout in encoder / postnet (masks['enc'], masks['post']), dropout on both LSTM outputs (masks['att'], masks['dec']:
    [n_steps, B, H])."""
    memory = encoder(sd, hp, text, None, training, masks)
    B = text.size(0)
    st = DecoderState(sd, hp, memory, None)
    x = torch.zeros(B, hp["n_mel_channels"], dtype=memory.dtype)
    mel_out, gate_out, aligns = [], [], []
    for t in range(n_steps):
        m, g, w = decode_step(sd, hp, st, prenet(sd, x, prenet_masks[t]), masks["att"][t] if training else None,
                              masks["dec"][t] if training else None)
        mel_out.append(m)
        gate_out.append(g)
        aligns.append(w)
        if gate_threshold is not None and bool((torch.sigmoid(g) > gate_threshold).all()):
            break
        x = m
    mel_out = torch.stack(mel_out, 2)
    gate_out = torch.stack(gate_out, 1)                       # [B, T, 1] (reference returns [1, T, 1])
    aligns = torch.stack(aligns, 1)
    mel_post = mel_out + postnet(sd, hp, mel_out, training, masks)
    return mel_out, mel_post, gate_out, aligns


def tacotron_loss(outputs, mel_target, gate_target):
    """Tacotron2Loss (reference tacotron/loss_function.py:7-18)."""
    mel_out, mel_post, gate_out, _ = outputs
    return F.mse_loss(mel_out, mel_target) + F.mse_loss(mel_post, mel_target) + \
        F.binary_cross_entropy_with_logits(gate_out.reshape(-1, 1), gate_target.reshape(-1, 1))

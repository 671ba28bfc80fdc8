"""MI355X-native Tacotron-2 with the reference module API.

Drop-in for reference ``tacotron/tacotron.py``: ``Tacotron(hparams, n_vocab, ..., num_speakers)``
with ``forward(inputs)``, ``inference(inputs, speaker_id)``, ``parse_batch``, ``parse_output`` and
the same ``state_dict`` keys.  ``Encoder`` / ``Decoder`` / ``Attention`` /
``LocationLayer`` are parameter containers; the math is issued by ``_TacoEngine`` through the C ABI
of ``libt2s_hip.so`` (include/t2s_hip.h).  No CPU / eager fallback.

Dropout: the prenet's dropout is always on in the reference (modules.py:21).  Masks can be injected
(``prenet_masks=`` {0,1} bytes) for parity tests; otherwise they are drawn on the device from a seed
that is fresh for every call (``_TacoEngine.fresh_seed``, reproducible under ``torch.manual_seed``).
Training mode (BatchNorm batch statistics, encoder / LSTM / postnet dropout) runs here; its backward
is ``tacotron/autograd.py``.
"""
import ctypes
import sys
from math import sqrt

import torch
from torch import nn

from .. import _lib
from .modules import ConvNorm, LinearNorm, Postnet, Prenet, get_mask_from_lengths, OwnedModule

BN_EPS = 1e-5


class LocationLayer(nn.Module):
    def __init__(self, attention_n_filters, attention_kernel_size, attention_dim):
        super().__init__()
        padding = int((attention_kernel_size - 1) / 2)
        self.location_conv = ConvNorm(2, attention_n_filters, kernel_size=attention_kernel_size, padding=padding,
                                      bias=False, stride=1, dilation=1)
        self.location_dense = LinearNorm(attention_n_filters, attention_dim, bias=False, w_init_gain="tanh")


def _lengths_from_mask(mask, T):
    """int32 lengths from the reference's padding mask (True on padding, a suffix of every row: ~get_mask_from_lengths)."""
    if mask is None:
        return None
    return (T - mask.to(torch.int32).sum(1)).to(torch.int32).contiguous()


class Attention(OwnedModule):
    def __init__(self, attention_rnn_dim, embedding_dim, attention_dim, attention_location_n_filters,
                 attention_location_kernel_size):
        super().__init__()
        self.query_layer = LinearNorm(attention_rnn_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.memory_layer = LinearNorm(embedding_dim, attention_dim, bias=False, w_init_gain="tanh")
        self.v = LinearNorm(attention_dim, 1, bias=False)
        self.location_layer = LocationLayer(attention_location_n_filters, attention_location_kernel_size,
                                            attention_dim)
        self.score_mask_value = -float("inf")

    def forward(self, attention_hidden_state, memory, processed_memory, attention_weights_cat, mask):
        """Reference tacotron.py:145-166: (attention_context [B, E], attention_weights [B, T]) from the attention LSTM's output,
        the encoder outputs, their memory_layer projection, the previous and cumulative weights stacked as [B, 2, T], and the
        padding mask (True on padding; None: no masking).  One call of the kernels the decode loop uses (t2s_taco_attention);
        forward only - training goes through Tacotron.forward."""
        from .modules import owner_engine
        with torch.no_grad():
            eng = owner_engine(self)
            dev = memory.device
            P = eng.prepare(dev)
            B, T, E = memory.shape
            ad = self.query_layer.linear_layer.out_features
            w = attention_weights_cat[:, 0].detach().to(torch.float32).contiguous().clone()
            wc = attention_weights_cat[:, 1].detach().to(torch.float32).contiguous().clone()
            h = _f32(attention_hidden_state)
            mem, pmem = _f32(memory), _f32(processed_memory)
            ctx = torch.empty(B, E, dtype=torch.float32, device=dev)
            q = torch.empty(B, ad, dtype=torch.float32, device=dev)
            e = torch.empty(B, T, dtype=torch.float32, device=dev)
            len32 = _lengths_from_mask(mask, T)
            conv = self.location_layer.location_conv.conv
            _lib.call("t2s_taco_attention", _lib.ptr(h), _lib.ptr(mem), _lib.ptr(pmem), _lib.ptr(len32), _lib.ptr(w), _lib.ptr(wc),
                      _lib.ptr(ctx), _lib.ptr(q), _lib.ptr(e), _lib.ptr(P["w_query"]), _lib.ptr(P["w_loc_conv"]),
                      _lib.ptr(P["w_loc_dense"]), _lib.ptr(P["w_loc_denseT"]), _lib.ptr(P["w_v"]), B, T, h.size(1), ad, E,
                      conv.out_channels, conv.kernel_size[0], _lib.current_stream())
            return ctx, w


class Encoder(OwnedModule):
    """3 x (Conv1d k5 + BatchNorm1d) + BiLSTM (reference tacotron.py:167-190)."""

    def __init__(self, hparams):
        super().__init__()
        C, ks = hparams["enc_conv_channels"], hparams["enc_conv_kernel_size"]
        self.convolutions = nn.ModuleList([
            nn.Sequential(ConvNorm(C, C, kernel_size=ks, stride=1, padding=(ks - 1) // 2, dilation=1,
                                   w_init_gain="relu"), nn.BatchNorm1d(C))
            for _ in range(hparams["enc_conv_num_layers"])])
        self.lstm = nn.LSTM(C, C // 2, 1, batch_first=True, bidirectional=True)

    def forward(self, x, input_lengths, train_masks=None):
        """Reference tacotron.py:192-207: embedded inputs x [B, C, T] (batch sorted by decreasing length, as
        pack_padded_sequence wants) -> [B, max(input_lengths), 2H], zero beyond each length."""
        from .modules import owner_engine
        with torch.no_grad():
            eng = owner_engine(self)
            eng.prepare(x.device)
            return eng.encode(None, input_lengths, train_masks, eng.fresh_seed(), embedded=x)[0]

    def inference(self, x, train_masks=None):
        """Reference tacotron.py:209-220: the same without lengths."""
        from .modules import owner_engine
        with torch.no_grad():
            eng = owner_engine(self)
            eng.prepare(x.device)
            return eng.encode(None, None, train_masks, eng.fresh_seed(), embedded=x)[0]


class Decoder(OwnedModule):
    """Prenet, attention LSTMCell, attention, decoder LSTMCell, projection, gate (reference tacotron.py:223-260)."""

    def __init__(self, hparams):
        super().__init__()
        hp = hparams
        self.n_mel_channels = hp["n_mel_channels"]
        self.n_frames_per_step = hp["n_frames_per_step"]
        self.encoder_embedding_dim = hp["enc_conv_channels"]
        self.attention_rnn_dim = hp["attention_rnn_dim"]
        self.decoder_rnn_dim = hp["decoder_rnn_dim"]
        self.prenet_dim = hp["prenet_dim"]
        self.max_decoder_steps = hp["max_decoder_steps"]
        self.gate_threshold = hp["gate_threshold"]
        self.p_attention_dropout = hp["p_attention_dropout"]
        self.p_decoder_dropout = hp["p_decoder_dropout"]
        n_out = hp["n_mel_channels"] * hp["n_frames_per_step"]
        self.prenet = Prenet(n_out, [hp["prenet_dim"], hp["prenet_dim"]])
        self.attention_rnn = nn.LSTMCell(hp["prenet_dim"] + hp["enc_conv_channels"], hp["attention_rnn_dim"])
        self.attention_layer = Attention(hp["attention_rnn_dim"], hp["enc_conv_channels"], hp["attention_dim"],
                                         hp["attention_location_n_filters"], hp["attention_location_kernel_size"])
        self.decoder_rnn = nn.LSTMCell(hp["attention_rnn_dim"] + hp["enc_conv_channels"], hp["decoder_rnn_dim"], 1)
        self.linear_projection = LinearNorm(hp["decoder_rnn_dim"] + hp["enc_conv_channels"], n_out)
        self.gate_layer = LinearNorm(hp["decoder_rnn_dim"] + hp["enc_conv_channels"], 1, bias=True,
                                     w_init_gain="sigmoid")

    def forward(self, memory, decoder_inputs, memory_lengths, prenet_masks=None, train_masks=None):
        """Reference tacotron.py:395-429: teacher-forced decode.  memory [B, T_in, E], decoder_inputs [B, n_mel, T_out] ->
        (mel_outputs [B, n_mel, T_out], gate_outputs [B, T_out], alignments [B, T_out, T_in])."""
        from .modules import owner_engine
        with torch.no_grad():
            eng = owner_engine(self)
            dev = memory.device
            len32 = None if memory_lengths is None else memory_lengths.to(device=dev, dtype=torch.int32).contiguous()
            mem = memory.detach().to(torch.float32).contiguous()
            return eng.decode_teacher(mem, len32, decoder_inputs, prenet_masks, None, train_masks)

    def inference(self, memory, prenet_masks=None, train_masks=None):
        """Reference tacotron.py:431-466: autoregressive decode until the gate fires (or max_decoder_steps).  Returns
        (mel_outputs [B, n_mel, T], gate_outputs [B, T, 1], alignments [B, T, T_in])."""
        from .modules import owner_engine
        with torch.no_grad():
            return owner_engine(self).decode_free(memory, prenet_masks, None, 64, train_masks)

    # ---- single decoder steps on module-held state (reference tacotron.py:262-307, 355-393) ----
    def get_go_frame(self, memory):
        """Reference tacotron.py:262-274: the all-zero first decoder input [B, n_mel * n_frames_per_step]."""
        return memory.new_zeros(memory.size(0), self.n_mel_channels * self.n_frames_per_step)

    def initialize_decoder_states(self, memory, mask):
        """Reference tacotron.py:276-307: zero LSTM states / attention weights / context, store memory and processed memory.
        mask: True on padding (~get_mask_from_lengths(memory_lengths)), or None.  The state lives in one device block the step
        kernels update in place; the reference's attribute names are views of it."""
        from .modules import owner_engine
        with torch.no_grad():
            eng = owner_engine(self)
            dev = memory.device
            eng.prepare(dev)
            mem = memory.detach().to(torch.float32).contiguous()
            B, T_in, E = mem.shape
            len32 = _lengths_from_mask(mask, T_in)
            Pd, D = self.prenet_dim, self.decoder_rnn_dim
            # two step slots: the step kernels ping-pong h by step parity and index their per-step buffers by the step number
            pre = torch.zeros(2, B, Pd, dtype=torch.float32, device=dev)
            hc = torch.empty(2, B, D + E, dtype=torch.float32, device=dev)
            extra = dict(pre_all=pre, hc_all=hc)
            d, S = eng._decoder_struct(mem, len32, 2, True, extra)
            d.att_xbuf = None       # (its tags are step numbers, used once per sequence: the two step slots here repeat them)
            d.pace_flag = None
            self.__dict__["_step"] = dict(d=d, S=S, n=0, len32=len32, B=B, T_in=T_in, E=E)
            self.memory, self.processed_memory, self.mask = memory, S["pmem"], mask
            self._publish_state()

    def _publish_state(self):
        st = self.__dict__["_step"]
        S, n = st["S"], st["n"]
        cur = (lambda a, b: S[b] if n & 1 else S[a])       # after n steps the newest h sits in slot n & 1
        self.attention_hidden, self.attention_cell = cur("att_h0", "att_h1"), S["att_c"]
        self.decoder_hidden, self.decoder_cell = cur("dec_h0", "dec_h1"), S["dec_c"]
        self.attention_weights, self.attention_weights_cum, self.attention_context = S["att_w"], S["att_wcum"], S["ctx"]

    def decode(self, decoder_input, attention_dropout_mask=None, decoder_dropout_mask=None):
        """Reference tacotron.py:355-393: ONE decoder step on the stored state.  decoder_input [B, prenet_dim] is the prenet's
        output, as in the reference.  Returns (decoder_output [B, n_mel], gate_prediction [B, 1], attention_weights [B, T_in]).
        In .train() mode the two LSTM outputs get dropout (tacotron.py:368,383): {0,1} masks [B, H] can be injected, else they are
        drawn on the device.  Forward only."""
        from .modules import owner_engine
        st = self.__dict__.get("_step")
        if st is None:
            raise _lib.T2SError("Decoder.decode before Decoder.initialize_decoder_states")
        with torch.no_grad():
            eng = owner_engine(self)
            d, S, n, B = st["d"], st["S"], st["n"], st["B"]
            dev = S["ctx"].device
            P = eng.prepare(dev)
            slot = n & 1
            S["pre_all"][slot].copy_(decoder_input.detach().to(torch.float32).reshape(B, self.prenet_dim))
            keep = []
            if self.training:
                seed = eng.fresh_seed()
                am = eng._drop(attention_dropout_mask, (B, self.attention_rnn_dim), 1 - self.p_attention_dropout, dev, seed + 301)
                dm = eng._drop(decoder_dropout_mask, (B, self.decoder_rnn_dim), 1 - self.p_decoder_dropout, dev, seed + 302)
                # the kernels index the masks by step number: hand them a base such that base + slot * B * H is this step's mask
                d.att_drop = am.data_ptr() - slot * B * self.attention_rnn_dim
                d.dec_drop = dm.data_ptr() - slot * B * self.decoder_rnn_dim
                d.att_drop_scale = 1.0 / (1.0 - self.p_attention_dropout)
                d.dec_drop_scale = 1.0 / (1.0 - self.p_decoder_dropout)
                keep = [am, dm]
            else:
                d.att_drop = d.dec_drop = None
                d.att_drop_scale = d.dec_drop_scale = 1.0
            _lib.call("t2s_taco_decode_steps", ctypes.byref(d), slot, 1, _lib.current_stream())
            n_mel = self.n_mel_channels * self.n_frames_per_step
            proj = torch.empty(B, n_mel + 1, dtype=torch.float32, device=dev)
            D, E = self.decoder_rnn_dim, st["E"]
            eng._gemv(P["w_proj"], S["hc_all"][slot], n_mel + 1, B, D + E, proj, bias=P["b_proj"])
            st["n"] = n + 1
            st["keep"] = keep
            self._publish_state()
            return proj[:, :n_mel].contiguous(), proj[:, n_mel:].contiguous(), S["align_out"][:, slot].clone()


class _DecoderStruct(ctypes.Structure):
    """Mirror of ``t2s_taco_decoder`` (include/t2s_hip.h)."""
    _I = ["B", "T_in", "n_mel", "prenet_dim", "enc_dim", "att_rnn_dim", "dec_rnn_dim", "att_dim", "loc_filters",
          "loc_kernel", "T_cap", "teacher_forced", "mask_steps"]
    _P1 = ["att_w_ih", "att_w_hh", "att_b_ih", "att_b_hh", "dec_w_ih", "dec_w_hh", "dec_b_ih", "dec_b_hh",
           "w_query", "w_loc_conv", "w_loc_dense", "w_v", "w_proj", "b_proj", "w_projpre", "b_projpre", "w_loc_denseT",
           "w_pre2",
           "memory", "pmem", "mem_lengths", "pre_all", "prenet_masks", "att_drop", "dec_drop"]
    _F = ["att_drop_scale", "dec_drop_scale"]
    _P2 = ["att_h0", "att_h1", "att_c", "dec_h0", "dec_h1", "dec_c", "att_w", "att_wcum", "ctx", "q", "energies",
           "pre1", "pre2", "q_part", "mel_gate_out", "align_out", "hc_all", "att_gates_all", "att_c_all",
           "dec_gates_all", "dec_c_all", "att_h_all", "q_all", "wcum_all", "gate_part", "w_pre2T", "ploc", "dec_in_part", "att_xbuf", "pace_flag"]
    _fields_ = ([(n, ctypes.c_int) for n in _I] + [(n, ctypes.c_void_p) for n in _P1] +
                [(n, ctypes.c_float) for n in _F] + [(n, ctypes.c_void_p) for n in _P2])


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


class BufferPool:
    """Per-engine pool of step-sized device buffers, keyed by (call site, shape, dtype, device).

    Bounded: the bytes sitting in free lists never exceed ``cap_bytes`` (default 16 GiB, ``T2S_POOL_CAP_GB``); when a returned
    buffer takes the pool over the cap, the free lists of the least recently used keys are dropped (their memory goes back to
    torch's caching allocator).  Batches padded to many different lengths (the reference's collate pads to the per-batch
    maximum, utils/data_utils.py:117) therefore cost re-allocations, not unbounded growth.  Leased buffers are in no list."""

    def __init__(self, cap_bytes=None):
        import os
        if cap_bytes is None:
            cap_bytes = int(float(os.environ.get("T2S_POOL_CAP_GB", "16")) * (1 << 30))
        self.cap_bytes = int(cap_bytes)
        self.free = {}          # key -> [tensors]
        self.stamp = {}         # key -> tick of the last take / give_back
        self.free_bytes = 0
        self.tick = 0
        self.evicted = 0        # buffers dropped so far (diagnostics / tests)

    def take(self, key):
        self.tick += 1
        self.stamp[key] = self.tick
        lst = self.free.get(key)
        if lst:
            t = lst.pop()
            self.free_bytes -= t.numel() * t.element_size()
            return t
        return None

    def give_back(self, key, t):
        self.tick += 1
        self.stamp[key] = self.tick
        self.free.setdefault(key, []).append(t)
        self.free_bytes += t.numel() * t.element_size()
        if self.free_bytes > self.cap_bytes:
            self._evict(keep=key)

    def _evict(self, keep=None):
        for key in sorted(self.free, key=lambda k: self.stamp.get(k, 0)):
            if self.free_bytes <= self.cap_bytes:
                break
            if key == keep and len(self.free) > 1:
                continue
            lst = self.free.pop(key)
            self.stamp.pop(key, None)
            for t in lst:
                self.free_bytes -= t.numel() * t.element_size()
                self.evicted += 1
        if self.free_bytes > self.cap_bytes and keep in self.free:     # one key alone over the cap: keep its newest buffers
            lst = self.free[keep]
            while lst and self.free_bytes > self.cap_bytes:
                t = lst.pop(0)
                self.free_bytes -= t.numel() * t.element_size()
                self.evicted += 1

    def clear(self):
        self.free.clear()
        self.stamp.clear()
        self.free_bytes = 0


class _Lease:
    """A buffer taken from the engine's pool; goes back when its holder (the autograd context of one step, or that
    step's backward scratch) is released.  While leased it is in no free list, so two live steps never share a buffer; reuse is
    stream-ordered like the caching allocator's."""

    def __init__(self, pool, key, tensor):
        self.pool, self.key, self.tensor = pool, key, tensor

    def __del__(self):
        try:
            self.pool.give_back(self.key, self.tensor)
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass


def pool_take(pool, holder, tag, shape, dtype, device, zero_once=False, extent=None):
    """Buffer of `shape` for call site `tag`: reused from the pool when one is free, else allocated.  `holder` (a list) keeps
    the lease.  Three kinds of buffer:
      * plain (zero_once=False): every element is rewritten by the kernels that own it - never cleared;
      * zero_once with extent=None: the elements the kernels never write depend on the shape and the call site only (weight packs,
        the zero bias) - zeroed when created;
      * zero_once with an `extent` (the valid row counts the producers write, e.g. (T,) of a [.., Lp, 32] plane whose
        Lp = ceil(T/256)*256 + 2*halo is shared by 256 different T): zeroed when created AND cleared again whenever the buffer
        comes back out of the pool for a different extent - a batch padded to T = 780 must not read the previous batch's rows
        780..799 as its convolutions' zero padding."""
    key = (tag, tuple(int(x) for x in shape), dtype, str(device))
    t = pool.take(key)
    if t is None:
        t = (torch.zeros if zero_once else torch.empty)(*shape, dtype=dtype, device=device)
    elif zero_once and extent is not None and getattr(t, "_t2s_extent", None) != tuple(extent):
        if t.is_cuda:
            _lib.call("t2s_zero_fill", _lib.ptr(t), t.numel() * t.element_size(), _lib.current_stream())
        else:
            t.zero_()
    if zero_once and extent is not None:
        t._t2s_extent = tuple(extent)
    holder.append(_Lease(pool, key, t))
    return t


class _TacoEngine:
    pool = None     # per-engine BufferPool (created on first use)

    def __init__(self, model):
        self.m = model
        self.prep = None
        self.prep_key = None

    # ------------------------------------------------------------------ weight preparation
    def _pack_conv_bn(self, seq, dev, halo):
        """ConvNorm + eval BatchNorm1d -> packed GEMM planes with the BN affine folded in."""
        conv, bn = seq[0].conv, seq[1]
        st = _lib.current_stream()
        O, Cin, Kt = conv.weight.shape
        Cpad = -(-Cin // 32) * 32
        Mpad = _lib.padded_rows(O)
        A_hi = torch.zeros(Kt * Cpad // 32, Mpad, 32, dtype=torch.bfloat16, device=dev)
        A_lo = torch.zeros_like(A_hi)
        bias = torch.zeros(Mpad, dtype=torch.float32, device=dev)
        scale = torch.empty(O, dtype=torch.float32, device=dev)
        bfold = torch.empty(O, dtype=torch.float32, device=dev)
        w, cb = _f32(conv.weight), (None if conv.bias is None else _f32(conv.bias))
        g, be, mu, var = _f32(bn.weight), _f32(bn.bias), _f32(bn.running_mean), _f32(bn.running_var)
        _lib.call("t2s_bn_fold", _lib.ptr(g), _lib.ptr(be), _lib.ptr(mu), _lib.ptr(var), _lib.ptr(cb), float(bn.eps), O,
                  _lib.ptr(scale), _lib.ptr(bfold), st)
        _lib.call("t2s_pack_conv_weight", _lib.ptr(w), _lib.ptr(scale), 1, _lib.ptr(bfold), O, Cin, Kt, 0, 0, 0, Mpad, 0,
                  Cpad, _lib.ptr(A_hi), _lib.ptr(A_lo), _lib.ptr(bias), 0, st)
        return dict(A_hi=A_hi, A_lo=A_lo, bias=bias, Mpad=Mpad, Cin=Cin, Cout=O, taps=Kt, keep=(w, cb, g, be, mu, var, scale, bfold))

    def prepare(self, dev):
        m = self.m
        key = tuple(p._version for p in m.parameters()) + (str(dev), m.training)
        if not m.training:       # BatchNorm running statistics are folded into the eval-mode weights
            key += tuple(b._version for b in m.buffers())
        if self.prep is not None and self.prep_key == key:
            return self.prep
        st = _lib.current_stream()
        enc, dec = m.encoder, m.decoder
        P = {}
        P["emb"] = _f32(m.embedding.weight)
        # (eval-mode packs fold the BatchNorm running statistics; training mode uses batch statistics and the plain packs)
        P["enc_convs"] = None if m.training else [self._pack_conv_bn(seq, dev, 2) for seq in enc.convolutions]
        if m.training:
            P["enc_convs_plain"] = [self._pack_conv_plain(seq, dev) for seq in enc.convolutions]
            P["post_convs_plain"] = [self._pack_conv_plain(seq, dev) for seq in m.postnet.convolutions]
        # BiLSTM: one input-projection GEMM for both directions, bias = b_ih + b_hh
        H = enc.lstm.hidden_size
        w_ih = torch.cat([_f32(enc.lstm.weight_ih_l0), _f32(enc.lstm.weight_ih_l0_reverse)], 0).contiguous()
        b_ih = torch.cat([_f32(enc.lstm.bias_ih_l0), _f32(enc.lstm.bias_ih_l0_reverse)], 0).contiguous()
        b_hh = torch.cat([_f32(enc.lstm.bias_hh_l0), _f32(enc.lstm.bias_hh_l0_reverse)], 0).contiguous()
        Cin = w_ih.size(1)
        Cpad, Mpad = -(-Cin // 32) * 32, _lib.padded_rows(8 * H)
        A_hi = torch.zeros(Cpad // 32, Mpad, 32, dtype=torch.bfloat16, device=dev)
        A_lo = torch.zeros_like(A_hi)
        bias = torch.zeros(Mpad, dtype=torch.float32, device=dev)
        for bvec, accum in ((b_ih, 0), (b_hh, 1)):
            _lib.call("t2s_pack_conv_weight", _lib.ptr(w_ih), None, 0, _lib.ptr(bvec), 8 * H, Cin, 1, 0, 0, 0, Mpad, 0, Cpad,
                      _lib.ptr(A_hi), _lib.ptr(A_lo), _lib.ptr(bias), accum, st)
        P["lstm_in"] = dict(A_hi=A_hi, A_lo=A_lo, bias=bias, Mpad=Mpad, Cin=Cin, Cout=8 * H, keep=(w_ih, b_ih, b_hh))
        P["H"] = H
        P["whhT"] = []
        for w in (enc.lstm.weight_hh_l0, enc.lstm.weight_hh_l0_reverse):
            w = _f32(w)
            wt = torch.empty(H, 4 * H, dtype=torch.float32, device=dev)
            _lib.call("t2s_transpose", _lib.ptr(w), _lib.ptr(wt), 4 * H, H, st)
            P["whhT"].append((wt, w))
        P["post_convs"] = None if m.training else [self._pack_conv_bn(seq, dev, 2) for seq in m.postnet.convolutions]
        # decoder
        al = dec.attention_layer
        P["w_mem"] = _f32(al.memory_layer.linear_layer.weight)
        for name, t in [("att_w_ih", dec.attention_rnn.weight_ih), ("att_w_hh", dec.attention_rnn.weight_hh),
                        ("att_b_ih", dec.attention_rnn.bias_ih), ("att_b_hh", dec.attention_rnn.bias_hh),
                        ("dec_w_ih", dec.decoder_rnn.weight_ih), ("dec_w_hh", dec.decoder_rnn.weight_hh),
                        ("dec_b_ih", dec.decoder_rnn.bias_ih), ("dec_b_hh", dec.decoder_rnn.bias_hh),
                        ("w_query", al.query_layer.linear_layer.weight),
                        ("w_loc_conv", al.location_layer.location_conv.conv.weight),
                        ("w_loc_dense", al.location_layer.location_dense.linear_layer.weight),
                        ("w_v", al.v.linear_layer.weight),
                        ("w_pre1", dec.prenet.layers[0].linear_layer.weight),
                        ("w_pre2", dec.prenet.layers[1].linear_layer.weight)]:
            P[name] = _f32(t)
        n_mel = dec.n_mel_channels * dec.n_frames_per_step
        Pd = dec.prenet_dim
        DE = dec.linear_projection.linear_layer.in_features
        # one row block [n_mel mel rows | gate row | prenet_dim precomposed prenet rows] so that the projection and
        # the next step's prenet layer 0 are ONE launch (t2s_taco_decode_steps)
        P["w_proj_all"] = torch.zeros(n_mel + 1 + Pd, DE, dtype=torch.float32, device=dev)
        P["b_proj_all"] = torch.zeros(n_mel + 1 + Pd, dtype=torch.float32, device=dev)
        P["w_proj_all"][:n_mel + 1] = torch.cat([_f32(dec.linear_projection.linear_layer.weight),
                                               _f32(dec.gate_layer.linear_layer.weight)], 0)
        P["b_proj_all"][:n_mel + 1] = torch.cat([_f32(dec.linear_projection.linear_layer.bias),
                                               _f32(dec.gate_layer.linear_layer.bias)], 0)
        P["w_proj"], P["b_proj"] = P["w_proj_all"][:n_mel + 1], P["b_proj_all"][:n_mel + 1]
        # w_projpre = W_pre1 . W_proj[:n_mel]  (prenet layer 0 applied straight to [h_dec | ctx])
        projT = torch.empty(DE, n_mel, dtype=torch.float32, device=dev)
        _lib.call("t2s_transpose", _lib.ptr(P["w_proj"]), _lib.ptr(projT), n_mel, DE, st)
        P["w_projpre"] = P["w_proj_all"][n_mel + 1:]
        _lib.call("t2s_gemv", _lib.ptr(P["w_pre1"]), n_mel, n_mel, None, 0, 0, _lib.ptr(projT), n_mel, n_mel, None, 0, 0,
                  None, 0, 0, None, None, _lib.ptr(P["w_projpre"]), 1, DE, Pd, DE, 0, None, 0, 1.0, st)
        P["b_projpre"] = P["b_proj_all"][n_mel + 1:]
        _lib.call("t2s_gemv", _lib.ptr(P["w_pre1"]), n_mel, n_mel, None, 0, 0, _lib.ptr(P["b_proj"]), n_mel, n_mel, None, 0,
                  0, None, 0, 0, None, None, _lib.ptr(P["b_projpre"]), Pd, 1, Pd, 1, 0, None, 0, 1.0, st)
        P["_projT"] = projT
        F_, AD_ = P["w_loc_dense"].size(1), P["w_loc_dense"].size(0)
        P["w_loc_denseT"] = torch.empty(F_, AD_, dtype=torch.float32, device=dev)
        _lib.call("t2s_transpose", _lib.ptr(P["w_loc_dense"]), _lib.ptr(P["w_loc_denseT"]), AD_, F_, st)
        P["w_pre2T"] = torch.empty(Pd, Pd, dtype=torch.float32, device=dev)     # ABI v4: the folded, sparse prenet layer 1
        _lib.call("t2s_transpose", _lib.ptr(P["w_pre2"]), _lib.ptr(P["w_pre2T"]), Pd, Pd, st)
        self.prep, self.prep_key = P, key
        return P

    # ------------------------------------------------------------------ building blocks
    def _planes(self, shape, dev, leases=None, tag=None, extent=None):
        """A (hi, lo) pair of bf16 operand planes whose padding (halo rows, rows past the valid length, channels past the operand's
        extent) must read as zero.  The kernels that fill them write rows halo .. halo + T only, so inside a training step
        (leases = the step's lease list) they come from the engine's pool, zeroed when created and again whenever the valid row
        count `extent` = (T,) differs from the one the buffer last held (the padded row count Lp is shared by 256 different T);
        otherwise fresh zero-filled tensors."""
        if leases is None or tag is None:
            hi = torch.zeros(*shape, dtype=torch.bfloat16, device=dev)
            return hi, torch.zeros_like(hi)
        if self.pool is None:
            self.pool = BufferPool()
        assert extent is not None, "pooled planes need their valid extent"
        return (pool_take(self.pool, leases, tag + "_h", shape, torch.bfloat16, dev, zero_once=True, extent=extent),
                pool_take(self.pool, leases, tag + "_l", shape, torch.bfloat16, dev, zero_once=True, extent=extent))

    def _conv(self, layer, Xh, Xl, B, L, Lp, halo, act, out_planes=True, out_f32=None, f32_cl=0):
        dev = Xh.device
        oc = -(-layer["Cout"] // 32)
        Oh = Ol = None
        if out_planes:
            Oh = torch.zeros(B, oc, Lp, 32, dtype=torch.bfloat16, device=dev)
            Ol = torch.zeros_like(Oh)
        _lib.call("t2s_conv_bias_act", _lib.ptr(layer["A_hi"]), _lib.ptr(layer["A_lo"]), _lib.ptr(layer["bias"]),
                  _lib.ptr(Xh), _lib.ptr(Xl), _lib.ptr(Oh), _lib.ptr(Ol), _lib.ptr(out_f32), f32_cl, B, layer["Cin"],
                  layer["Cout"], layer.get("taps", 1), 1, act, L, Lp, halo, layer["Mpad"], _lib.current_stream())
        return Oh, Ol

    def encode(self, ids, lengths, train_masks=None, seed=0, save=None, embedded=None, max_len=None):
        """Embedding + Encoder.forward / .inference (reference tacotron.py:40,192-220).  In training mode the
        convolutions use batch statistics and dropout(0.5) (masks from ``train_masks['enc']`` or drawn here).
        ``embedded`` [B, E, T] f32: start from embedded inputs instead of ids (Encoder.forward's own argument)."""
        m, P = self.m, self.prep
        E = m.embedding.embedding_dim
        halo = 2
        st = _lib.current_stream()
        leases = None if save is None else save.setdefault("_leases", [])
        if embedded is not None:
            dev = embedded.device
            B, _, T = embedded.shape
            Lp = _lib.plane_rows(T, halo)
            ids64 = None
            emb32 = embedded.detach().to(torch.float32).contiguous()
            Xh, Xl = self._planes((B, -(-E // 32), Lp, 32), dev, leases, "enc_x", (T,))
            _lib.call("t2s_f32_to_planes", _lib.ptr(emb32), B, E, T, Lp, halo, _lib.ptr(Xh), _lib.ptr(Xl), st)
        else:
            dev = ids.device
            B, T = ids.shape
            Lp = _lib.plane_rows(T, halo)
            ids64 = ids.to(torch.int64).contiguous()
            Xh, Xl = self._planes((B, -(-E // 32), Lp, 32), dev, leases, "enc_x", (T,))
            _lib.call("t2s_embed_planes", _lib.ptr(ids64), _lib.ptr(P["emb"]), B, T, E, m.embedding.num_embeddings, Lp, halo,
                      _lib.ptr(Xh), _lib.ptr(Xl), st)
        if m.training:
            given = None if train_masks is None else train_masks.get("enc")
            for i, (seq, layer) in enumerate(zip(m.encoder.convolutions, P["enc_convs_plain"])):
                mk = self._drop(None if given is None else given[i], (B, layer["Cout"], T), 0.5, dev, seed + 101 + i)
                Xh, Xl = self._conv_bn_train(seq, layer, Xh, Xl, B, T, Lp, halo, 1, mk,
                                             save=None if save is None else save.setdefault("enc_convs", []),
                                             leases=leases, tag="enc_conv%d" % i)
        else:
            for layer in P["enc_convs"]:
                Xh, Xl = self._conv(layer, Xh, Xl, B, T, Lp, halo, 1)
        H = P["H"]
        gx = torch.empty(B, T, 8 * H, dtype=torch.float32, device=dev)
        self._conv(P["lstm_in"], Xh, Xl, B, T, Lp, halo, 0, out_planes=False, out_f32=gx, f32_cl=1)
        if lengths is not None:
            len32 = lengths.to(device=dev, dtype=torch.int32).contiguous()
            # the longest entry: given by the caller (the batch tuple's max_len, reference tacotron.py:81-83 / parse_batch) or read
            # back from the device - a host synchronisation that keeps the host from enqueueing this step while the previous one
            # still runs
            T_out = int(max_len) if max_len is not None else int(lengths.max().item())
        else:
            len32, T_out = None, T
        memory = torch.empty(B, T_out, 2 * H, dtype=torch.float32, device=dev)
        gsave = csave = None
        if save is not None:
            # (the recurrence writes, and its backward reads, the steps below each entry's length only: no clearing)
            if self.pool is None:
                self.pool = BufferPool()
            gsave = pool_take(self.pool, leases, "enc_gsave", (B, T, 2, 4 * H), torch.float32, dev)
            csave = pool_take(self.pool, leases, "enc_csave", (B, T, 2, H), torch.float32, dev)
            save.update(enc_gates=gsave, enc_c=csave)
        xb = self._lstm_xbuf("fwd", B, dev) if H == 256 and T < 4095 else None
        if xb is not None:      # W_hh resident, four workgroups per (element, direction) exchanging h per step (t2s_taco_encoder_lstm_split)
            _lib.call("t2s_taco_encoder_lstm_split", _lib.ptr(gx), _lib.ptr(P["whhT"][0][0]), _lib.ptr(P["whhT"][1][0]),
                      _lib.ptr(len32), _lib.ptr(memory), B, T, H, T_out, _lib.ptr(gsave), _lib.ptr(csave), _lib.ptr(xb[0]), xb[1], st)
        else:
            _lib.call("t2s_taco_encoder_lstm", _lib.ptr(gx), _lib.ptr(P["whhT"][0][0]), _lib.ptr(P["whhT"][1][0]),
                      _lib.ptr(len32), _lib.ptr(memory), B, T, H, T_out, _lib.ptr(gsave), _lib.ptr(csave), st)
        if save is not None:
            save.update(enc_ids=ids64, enc_gx=gx, enc_Xh=Xh, enc_Xl=Xl, enc_T=T, enc_Lp=Lp, enc_len32=len32, memory=memory)
        return memory, len32

    def _lstm_xbuf(self, which, B, dev):
        """Exchange buffer of the split BiLSTM recurrence (t2s_taco_encoder_lstm_split / _bwd_split) and this launch's epoch, or None
        (T2S_LSTM_SEQ_SPLIT=0: the one-workgroup kernels).  One buffer per direction of use (forward / backward: the two may be in
        flight on different streams), engine-owned, zeroed once; `check_lstm_xbuf()` reads the error words."""
        import os
        if os.environ.get("T2S_LSTM_SEQ_SPLIT", "1") == "0":
            return None
        bufs = self.__dict__.setdefault("_xbufs", {})
        key = (which, int(B), str(dev))
        ent = bufs.get(key)
        if ent is None:
            n = int(_lib.load().t2s_taco_lstm_xbuf_bytes(int(B)))
            ent = bufs[key] = [torch.zeros(n // 8, dtype=torch.int64, device=dev), 0]
        ent[1] = (ent[1] + 1) & 0xFFFFF
        if ent[1] == 0:         # (wrapped: tags of 2^20 launches ago could match again - start over from a clean buffer)
            ent[0].zero_()
            ent[1] = 1
        return ent[0], ent[1]

    def check_lstm_xbuf(self):
        """Synchronises; raises if a bounded wait of the split BiLSTM kernels ever expired on one of this engine's buffers."""
        for key, (buf, _) in self.__dict__.get("_xbufs", {}).items():
            if int(buf[-1].item()) != 0:
                raise _lib.T2SError("split BiLSTM recurrence %s: a hand-off wait expired (results of that launch are invalid)" % (key,))
        # ... and of the last teacher-forced decode's exchange buffers (t2s_taco_decoder::att_xbuf / pace_flag: their last / second
        # 8-byte word is raised by a bounded wait that expired)
        S = self.__dict__.get("_last_decoder_S")
        if S is not None:
            for name, words in (("att_xbuf", (-2, -1)), ("pace_flag", (2, 3))):
                t = S.get(name)
                if t is not None and any(float(t.view(-1)[w].item()) != 0.0 for w in words):
                    raise _lib.T2SError("teacher-forced decode: a bounded wait on %s expired (results of that call are invalid)" % name)

    def _gemv(self, W, x, rows, items, K, y, act=0, mask=None, smask=0, mask_scale=1.0, bias=None, sy_item=None,
              sx=None):
        _lib.call("t2s_gemv", _lib.ptr(W), K, K, None, 0, 0, _lib.ptr(x), K, K if sx is None else sx, None, 0, 0, None, 0,
                  0, _lib.ptr(bias), None, _lib.ptr(y), rows if sy_item is None else sy_item, 1, rows, items, act,
                  _lib.ptr(mask), smask, mask_scale, _lib.current_stream())

    def _decoder_struct(self, memory, len32, T_cap, teacher, extra):
        m, P = self.m, self.prep
        dec = m.decoder
        dev = memory.device
        B, T_in, E = memory.shape
        A, D, Pd = dec.attention_rnn_dim, dec.decoder_rnn_dim, dec.prenet_dim
        al = dec.attention_layer
        ad = al.query_layer.linear_layer.out_features
        n_mel = dec.n_mel_channels * dec.n_frames_per_step
        # recurrent state and scratch that must start at zero: views of one buffer cleared by one launch
        shapes = dict(att_h0=(B, A), att_h1=(B, A), att_c=(B, A), dec_h0=(B, D), dec_h1=(B, D), dec_c=(B, D),
                      att_w=(B, T_in), att_wcum=(B, T_in), ctx=(B, E), q=(B, ad), energies=(B, T_in),
                      pre1=(B, Pd), pre2=(B, Pd), q_part=(A // 2, B, ad))
        if B > 8 and T_in <= 512:
            shapes["att_xbuf"] = (2 * (B * T_in + 1),)      # (8-byte granules + error word, as f32 pairs; zero = no tag matches)
        if B > 8 and teacher:
            shapes["pace_flag"] = (4,)                      # step counter + error word of the paced decoder cells
        if not teacher:
            shapes["align_out"] = (B, T_cap, T_in)          # rows past the stop step stay zero
            if B <= 8 and A == 1024 and D == 1024 and getattr(self, "decode_stream", True):
                shapes["gate_part"] = (3, B, 4 * A)         # streamed gate partials (ABI v4): zero = h_att(-1) . W_hh_att
                shapes["ploc"] = (B, T_in, ad)              # location term of the next step's attention (zero: w = w_cum = 0)
        offs, tot = {}, 0
        for name, sh in shapes.items():
            n = 1
            for x in sh:
                n *= int(x)
            offs[name] = (tot, n)
            tot += -(-n // 64) * 64                         # 256-byte granules
        flat = torch.empty(tot, dtype=torch.float32, device=dev)
        _lib.call("t2s_zero_fill", _lib.ptr(flat), tot * 4, _lib.current_stream())
        S = {name: flat[o:o + n].view(*shapes[name]) for name, (o, n) in offs.items()}
        if teacher:     # every (item, step) row is written by the attention kernel of its step
            S["align_out"] = torch.empty(B, T_cap, T_in, dtype=torch.float32, device=dev)
        pmem = torch.empty(B, T_in, ad, dtype=torch.float32, device=dev)
        self._gemv(P["w_mem"], memory, ad, B * T_in, E, pmem)
        S["pmem"], S["memory"] = pmem, memory
        d = _DecoderStruct()
        for k, v in dict(B=B, T_in=T_in, n_mel=n_mel, prenet_dim=Pd, enc_dim=E, att_rnn_dim=A, dec_rnn_dim=D, att_dim=ad,
                         loc_filters=al.location_layer.location_conv.conv.out_channels,
                         loc_kernel=al.location_layer.location_conv.conv.kernel_size[0], T_cap=T_cap,
                         teacher_forced=1 if teacher else 0, mask_steps=0).items():
            setattr(d, k, v)
        for name in ["att_w_ih", "att_w_hh", "att_b_ih", "att_b_hh", "dec_w_ih", "dec_w_hh", "dec_b_ih", "dec_b_hh",
                     "w_query", "w_loc_conv", "w_loc_dense", "w_v", "w_proj", "b_proj", "w_projpre", "b_projpre",
                     "w_loc_denseT", "w_pre2"]:
            setattr(d, name, P[name].data_ptr())
        for name, t in S.items():
            setattr(d, name, t.data_ptr())
        d.w_pre2T = P["w_pre2T"].data_ptr() if "gate_part" in S else None
        d.mem_lengths = None if len32 is None else len32.data_ptr()
        d.att_drop_scale = d.dec_drop_scale = 1.0
        for name, t in extra.items():
            setattr(d, name, None if t is None else t.data_ptr())
            S[name] = t
        return d, S

    def postnet(self, mel, train_masks=None, seed=0, save=None):
        """Postnet.forward (reference modules.py:131-137): 5 x conv+BN, tanh on the first four; in training mode
        batch statistics and dropout(0.5) after every layer."""
        P = self.prep
        m = self.m
        B, C, T = mel.shape
        halo = 2
        Lp = _lib.plane_rows(T, halo)
        dev = mel.device
        mel = mel.contiguous()
        leases = None if save is None else save.setdefault("_leases", [])
        Xh, Xl = self._planes((B, -(-C // 32), Lp, 32), dev, leases, "post_x", (T,))
        _lib.call("t2s_f32_to_planes", _lib.ptr(mel), B, C, T, Lp, halo, _lib.ptr(Xh), _lib.ptr(Xl), _lib.current_stream())
        n = len(m.postnet.convolutions)
        out = torch.empty(B, C, T, dtype=torch.float32, device=dev)
        if m.training:
            given = None if train_masks is None else train_masks.get("post")
            for i, (seq, layer) in enumerate(zip(m.postnet.convolutions, P["post_convs_plain"])):
                mk = self._drop(None if given is None else given[i], (B, layer["Cout"], T), 0.5, dev, seed + 201 + i)
                sv = None if save is None else save.setdefault("post_convs", [])
                if i < n - 1:
                    Xh, Xl = self._conv_bn_train(seq, layer, Xh, Xl, B, T, Lp, halo, 2, mk, save=sv, leases=leases,
                                                 tag="post_conv%d" % i)
                else:
                    self._conv_bn_train(seq, layer, Xh, Xl, B, T, Lp, halo, 0, mk, want_planes=False, out_f32=out, save=sv)
            return out
        for i, layer in enumerate(P["post_convs"]):
            if i < n - 1:
                Xh, Xl = self._conv(layer, Xh, Xl, B, T, Lp, halo, 2)
            else:
                self._conv(layer, Xh, Xl, B, T, Lp, halo, 0, out_planes=False, out_f32=out)
        return out

    def _masks(self, prenet_masks, n, B, Pd, dev, seed):
        if prenet_masks is not None:
            mk = prenet_masks.to(device=dev, dtype=torch.uint8).contiguous()
            assert mk.numel() >= n * B * 2 * Pd, "prenet_masks too short"
            return mk
        mk = torch.empty(n, B, 2, Pd, dtype=torch.uint8, device=dev)
        _lib.call("t2s_bernoulli_mask", _lib.ptr(mk), mk.numel(), int(seed), 0, 0.5, _lib.current_stream())
        return mk

    def _drop(self, given, shape, keep, dev, seed):
        """{0,1} bytes: injected (parity tests) or drawn on the device with keep probability ``keep``."""
        if given is not None:
            return given.to(device=dev, dtype=torch.uint8).contiguous()
        mk = torch.empty(*shape, dtype=torch.uint8, device=dev)
        _lib.call("t2s_bernoulli_mask", _lib.ptr(mk), mk.numel(), int(seed), 0, float(keep), _lib.current_stream())
        return mk

    def _pack_conv_plain(self, seq, dev):
        """ConvNorm alone (training mode: BatchNorm uses batch statistics, so nothing is folded)."""
        conv = seq[0].conv
        O, Cin, Kt = conv.weight.shape
        Cpad, Mpad = -(-Cin // 32) * 32, _lib.padded_rows(O)
        # repacked every training step (the weights move): pooled, zeroed once - the pack writes every entry inside (O, Cin, Kt),
        # the padding around it depends on that triple only (it is part of the tag).  The leases live in the layer dict, which
        # the step's saves reference too, so a buffer is not recycled while a backward still reads it.
        if self.pool is None:
            self.pool = BufferPool()
        leases, tag = [], "pack_plain_%d_%d_%d" % (O, Cin, Kt)
        A_hi = pool_take(self.pool, leases, tag + "_h", (Kt * Cpad // 32, Mpad, 32), torch.bfloat16, dev, zero_once=True)
        A_lo = pool_take(self.pool, leases, tag + "_l", (Kt * Cpad // 32, Mpad, 32), torch.bfloat16, dev, zero_once=True)
        bias = pool_take(self.pool, leases, tag + "_b", (Mpad,), torch.float32, dev, zero_once=True)
        w, cb = _f32(conv.weight), (None if conv.bias is None else _f32(conv.bias))
        _lib.call("t2s_pack_conv_weight", _lib.ptr(w), None, 0, _lib.ptr(cb), O, Cin, Kt, 0, 0, 0, Mpad, 0, Cpad,
                  _lib.ptr(A_hi), _lib.ptr(A_lo), _lib.ptr(bias), 0, _lib.current_stream())
        return dict(A_hi=A_hi, A_lo=A_lo, bias=bias, Mpad=Mpad, Cin=Cin, Cout=O, taps=Kt, keep=(w, cb), _leases=leases)

    def _conv_bn_train(self, seq, layer, Xh, Xl, B, T, Lp, halo, act, mask, want_planes=True, out_f32=None, save=None,
                       leases=None, tag=None):
        """conv -> BatchNorm1d with batch statistics (+ running-stat update, as nn.BatchNorm1d.train() does) ->
        activation -> dropout mask (reference tacotron.py:193-194; modules.py:131-137)."""
        bn = seq[1]
        dev = Xh.device
        C = layer["Cout"]
        y = torch.empty(B, C, T, dtype=torch.float32, device=dev)
        self._conv(layer, Xh, Xl, B, T, Lp, halo, 0, out_planes=False, out_f32=y)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        var = torch.empty(C, dtype=torch.float32, device=dev)
        Oh = Ol = None
        if want_planes:
            Oh, Ol = self._planes((B, -(-C // 32), Lp, 32), dev, leases, tag, (T,))
        g, be = _f32(bn.weight), _f32(bn.bias)
        _lib.call("t2s_bn_train", _lib.ptr(y), _lib.ptr(g), _lib.ptr(be), float(bn.eps), act, _lib.ptr(mask), 2.0, B, C, T, Lp,
                  halo, _lib.ptr(mean), _lib.ptr(var), _lib.ptr(Oh), _lib.ptr(Ol), _lib.ptr(out_f32), _lib.current_stream())
        # bookkeeping of nn.BatchNorm1d in training mode: one launch
        mom = 0.1 if bn.momentum is None else bn.momentum
        _lib.call("t2s_bn_running_update", _lib.ptr(mean), _lib.ptr(var), _lib.ptr(bn.running_mean), _lib.ptr(bn.running_var),
                  _lib.ptr(bn.num_batches_tracked), float(mom), B * T, C, _lib.current_stream())
        if save is not None:
            save.append(dict(seq=seq, layer=layer, Xh=Xh, Xl=Xl, y=y, mean=mean, var=var, mask=mask, act=act, B=B, T=T, Lp=Lp,
                             halo=halo))
        return Oh, Ol

    # ------------------------------------------------------------------ whole-model paths
    @staticmethod
    def fresh_seed():
        """A new dropout seed per call, drawn from torch's default generator: every forward / inference draws different
        masks (reference: F.dropout draws from the global RNG each call, modules.py:21, tacotron.py:193,368,383) and
        ``torch.manual_seed`` reproduces them.  Host-side draw: no device synchronisation."""
        return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())

    def inference(self, ids, prenet_masks=None, seed=None, chunk=64, train_masks=None):
        """Tacotron.inference (reference tacotron.py:51-65).  In ``.train()`` mode the reference simply runs with its modules in
        training mode: the encoder / postnet convolutions use batch statistics and dropout(0.5) (tacotron.py:193, modules.py:131-137)
        and both LSTM outputs get dropout (tacotron.py:368,383); the same here (``train_masks`` as in ``forward``, 'att' / 'dec' of
        shape [max_decoder_steps, B, H])."""
        if seed is None:
            seed = self.fresh_seed()
        m = self.m
        dec = m.decoder
        dev = ids.device
        self.prepare(dev)
        memory, _ = self.encode(ids, None, train_masks, seed)
        mel, gate, align = self.decode_free(memory, prenet_masks, seed, chunk, train_masks)
        mel_post = mel + self.postnet(mel, train_masks, seed)
        return [mel, mel_post, gate, align]

    def decode_free(self, memory, prenet_masks=None, seed=None, chunk=64, train_masks=None):
        """Decoder.inference (reference tacotron.py:431-466): autoregressive decode of encoder outputs ``memory`` [B, T_in, E]
        until every entry's gate passes the threshold or max_decoder_steps.  Returns (mel [B, n_mel, T], gate [B, T, 1],
        alignments [B, T, T_in])."""
        if seed is None:
            seed = self.fresh_seed()
        m = self.m
        dec = m.decoder
        dev = memory.device
        self.prepare(dev)
        memory = memory.detach().to(torch.float32).contiguous()
        B = memory.size(0)
        T_cap = int(dec.max_decoder_steps)
        n_mel = dec.n_mel_channels * dec.n_frames_per_step
        mk = self._masks(prenet_masks, T_cap, B, dec.prenet_dim, dev, seed)
        mel_gate = torch.zeros(B, n_mel + 1, T_cap, dtype=torch.float32, device=dev)
        extra = dict(prenet_masks=mk, mel_gate_out=mel_gate)
        if m.training:
            tm = train_masks or {}
            extra["att_drop"] = self._drop(tm.get("att"), (T_cap, B, dec.attention_rnn_dim), 1 - dec.p_attention_dropout, dev, seed + 301)
            extra["dec_drop"] = self._drop(tm.get("dec"), (T_cap, B, dec.decoder_rnn_dim), 1 - dec.p_decoder_dropout, dev, seed + 302)
        d, S = self._decoder_struct(memory, None, T_cap, False, extra)
        if m.training:
            d.att_drop_scale = 1.0 / (1.0 - dec.p_attention_dropout)
            d.dec_drop_scale = 1.0 / (1.0 - dec.p_decoder_dropout)
        d.mask_steps = mk.numel() // (B * 2 * dec.prenet_dim)
        stop = torch.full((B,), -1, dtype=torch.int32, device=dev)
        st = _lib.current_stream()
        s0 = 0
        n_done = T_cap
        while s0 < T_cap:
            n = min(chunk, T_cap - s0)
            _lib.call("t2s_taco_decode_steps", ctypes.byref(d), s0, n, st)
            _lib.call("t2s_taco_stop_check", _lib.ptr(mel_gate), B, n_mel, T_cap, s0, n, float(dec.gate_threshold),
                      _lib.ptr(stop), st)
            s0 += n
            sv = stop.cpu()                     # sparse poll: one host sync per chunk, not per frame
            if bool((sv >= 0).all()):
                n_done = int(sv.max().item()) + 1
                break
        else:
            print("Warning! Reached max decoder steps", file=sys.stderr)     # reference tacotron.py:458 (stdout there)
        mel = mel_gate[:, :n_mel, :n_done].contiguous()
        gate = mel_gate[:, n_mel, :n_done].unsqueeze(-1).contiguous()          # [B, T, 1] as the reference returns
        align = S["align_out"][:, :n_done].contiguous()
        return mel, gate, align

    def forward(self, text, text_lengths, mels, output_lengths, prenet_masks=None, seed=None, train_masks=None, save=None,
                max_len=None):
        if seed is None:
            seed = self.fresh_seed()
        m = self.m
        dec = m.decoder
        dev = text.device
        self.prepare(dev)
        if max_len is not None:
            max_len = int(max_len)
            if not 0 < max_len <= text.size(1):
                raise _lib.T2SError("max_len %d outside (0, %d]" % (max_len, text.size(1)))
        memory, len32 = self.encode(text, text_lengths, train_masks, seed, save=save, max_len=max_len)
        mel, gate, align = self.decode_teacher(memory, len32, mels, prenet_masks, seed, train_masks, save)
        B, n_mel, T_out = mels.shape
        mel_post = mel + self.postnet(mel, train_masks, seed, save=save)
        return self._finish_forward(mel, mel_post, gate, align, output_lengths, save)

    def decode_teacher(self, memory, len32, mels, prenet_masks=None, seed=None, train_masks=None, save=None):
        """Decoder.forward (reference tacotron.py:395-429): teacher-forced decode of ``memory`` [B, T_in, E] against the target
        frames ``mels`` [B, n_mel, T_out] (``len32``: int32 memory lengths on the device, or None).  Returns (mel [B, n_mel, T_out],
        gate [B, T_out], alignments [B, T_out, T_in])."""
        if seed is None:
            seed = self.fresh_seed()
        m = self.m
        dec = m.decoder
        dev = memory.device
        self.prepare(dev)
        B, n_mel, T_out = mels.shape
        Pd, D, E = dec.prenet_dim, dec.decoder_rnn_dim, memory.size(2)
        P = self.prep
        # teacher forcing: prenet hoisted over [go frame ; all target frames] (reference tacotron.py:409-412)
        frames = torch.cat((torch.zeros(1, B, n_mel, dtype=torch.float32, device=dev),
                            mels.to(torch.float32).permute(2, 0, 1)), 0).contiguous()
        mk = self._masks(prenet_masks, T_out + 1, B, Pd, dev, seed)
        items = (T_out + 1) * B
        p1 = torch.empty(items, Pd, dtype=torch.float32, device=dev)
        pre_all = torch.empty(items, Pd, dtype=torch.float32, device=dev)
        self._gemv(P["w_pre1"], frames, Pd, items, n_mel, p1, act=1, mask=mk, smask=2 * Pd, mask_scale=2.0)
        mk1 = mk.view(-1)[Pd:]
        self._gemv(P["w_pre2"], p1, Pd, items, Pd, pre_all, act=1, mask=mk1, smask=2 * Pd, mask_scale=2.0)
        hc_all = torch.empty(T_out, B, D + E, dtype=torch.float32, device=dev)
        extra = dict(pre_all=pre_all, hc_all=hc_all)
        if m.training:      # dropout on both LSTM outputs (reference tacotron.py:368-369,383-384)
            tm = train_masks or {}
            extra["att_drop"] = self._drop(tm.get("att"), (T_out, B, dec.attention_rnn_dim), 1 - dec.p_attention_dropout,
                                           dev, seed + 301)
            extra["dec_drop"] = self._drop(tm.get("dec"), (T_out, B, D), 1 - dec.p_decoder_dropout, dev, seed + 302)
        if save is not None:        # per-step state the decoder backward needs (teacher-forced, training)
            A_, T_in_ = dec.attention_rnn_dim, memory.size(1)
            ad_ = dec.attention_layer.query_layer.linear_layer.out_features
            # 1.2 GB at B=32, T_out=800.  Every element is written by the decode steps (each step stores the gates / cell state /
            # query / cumulative weights of ALL batch entries and ALL encoder positions), so nothing is cleared: the buffers
            # come from the engine's pool and go back when this step's autograd context is released.
            if self.pool is None:
                self.pool = BufferPool()
            leases = save.setdefault("_leases", [])
            zf = lambda name, *sh: pool_take(self.pool, leases, name, sh, torch.float32, dev)
            extra.update(att_gates_all=zf("att_gates_all", T_out, B, 4 * A_), att_c_all=zf("att_c_all", T_out, B, A_),
                         dec_gates_all=zf("dec_gates_all", T_out, B, 4 * D), dec_c_all=zf("dec_c_all", T_out, B, D),
                         att_h_all=zf("att_h_all", T_out, B, A_), q_all=zf("q_all", T_out, B, ad_),
                         wcum_all=zf("wcum_all", T_out, B, T_in_))
        elif B > 8:
            # no-grad teacher-forced forward at 9+ items: with a copy of every step's h_att at hand the library takes the decoder cells
            # off the serial chain (helper stream, a chunk of steps behind the attention chain: t2s_taco_decode_steps) exactly as in
            # training - the only save that path needs besides hc_all
            extra["att_h_all"] = torch.empty(T_out, B, dec.attention_rnn_dim, dtype=torch.float32, device=dev)
        if B > 8 and extra.get("att_h_all") is not None:
            # scratch of the decoder cells' per-chunk input product (t2s_taco_decoder::dec_in_part: 16 steps x B items x 4 D)
            extra["dec_in_part"] = torch.empty(16, B, 4 * D, dtype=torch.float32, device=dev)
        d, S = self._decoder_struct(memory, len32, T_out, True, extra)
        if m.training:
            d.att_drop_scale = 1.0 / (1.0 - dec.p_attention_dropout)
            d.dec_drop_scale = 1.0 / (1.0 - dec.p_decoder_dropout)
        _lib.call("t2s_taco_decode_steps", ctypes.byref(d), 0, T_out, _lib.current_stream())
        self.__dict__["_last_decoder_S"] = {k: S.get(k) for k in ("att_xbuf", "pace_flag")}     # (check_lstm_xbuf reads their error words)
        # hoisted projection + gate over all steps (reference tacotron.py:387-392)
        proj = torch.empty(T_out * B, n_mel + 1, dtype=torch.float32, device=dev)
        self._gemv(P["w_proj"], hc_all, n_mel + 1, T_out * B, D + E, proj, bias=P["b_proj"])
        proj = proj.view(T_out, B, n_mel + 1)
        mel = proj[:, :, :n_mel].permute(1, 2, 0).contiguous()
        gate = proj[:, :, n_mel].permute(1, 0).contiguous()
        if save is not None:
            save.update(S=S, frames=frames, p1=p1, pre_all=pre_all, hc_all=hc_all, prenet_masks=mk, len32=len32, B=B, T_out=T_out,
                        n_mel=n_mel)
        return mel, gate, S["align_out"]

    def _finish_forward(self, mel, mel_post, gate, align, output_lengths, save):
        m = self.m
        dev = mel.device
        B, n_mel, T_out = mel.shape
        # Tacotron.parse_output (reference tacotron.py:67-76) here, in one launch.  The reference fills through `.data` AFTER the
        # postnet has run, on the tensor the postnet's first convolution saved for backward: that convolution's weight gradient
        # is taken against the MASKED mel.  Same here: its saved input planes are re-derived from the masked tensor.
        if m.hparams["mask_padding"] and output_lengths is not None:
            olen32 = output_lengths.to(device=dev, dtype=torch.int32).contiguous()
            _lib.call("t2s_taco_parse_output", _lib.ptr(mel), _lib.ptr(mel_post), _lib.ptr(gate), _lib.ptr(olen32), B, n_mel, T_out,
                      _lib.current_stream())
            if save is not None and save.get("post_convs"):
                s0 = save["post_convs"][0]
                _lib.call("t2s_f32_to_planes", _lib.ptr(mel), B, n_mel, T_out, s0["Lp"], s0["halo"], _lib.ptr(s0["Xh"]),
                          _lib.ptr(s0["Xl"]), _lib.current_stream())
            self._keep_olen = olen32
        return [mel, mel_post, gate, align]

    def prenet_forward(self, x, masks=None, seed=None):
        """Prenet.forward (reference modules.py:19-22): two Linear + ReLU + dropout(0.5), the dropout ALWAYS on.  x [..., n_in]."""
        if seed is None:
            seed = self.fresh_seed()
        P = self.prep
        dev = x.device
        self.prepare(dev)
        P = self.prep
        Pd = self.m.decoder.prenet_dim
        n_in = x.shape[-1]
        x2 = x.detach().to(torch.float32).reshape(-1, n_in).contiguous()
        items = x2.size(0)
        mk = self._masks(masks, items, 1, Pd, dev, seed)
        p1 = torch.empty(items, Pd, dtype=torch.float32, device=dev)
        out = torch.empty(items, Pd, dtype=torch.float32, device=dev)
        self._gemv(P["w_pre1"], x2, Pd, items, n_in, p1, act=1, mask=mk, smask=2 * Pd, mask_scale=2.0)
        self._gemv(P["w_pre2"], p1, Pd, items, Pd, out, act=1, mask=mk.view(-1)[Pd:], smask=2 * Pd, mask_scale=2.0)
        return out.view(*x.shape[:-1], Pd)


class Tacotron(nn.Module):
    def __init__(self, hparams, n_vocab, mel_dim=80, linear_dim=1025, r=5, padding_idx=None, num_speakers=1):
        super().__init__()
        self.hparams = hparams
        self.mel_dim = mel_dim
        self.linear_dim = linear_dim
        embedding_dim = hparams["embedding_size"]
        self.embedding = nn.Embedding(n_vocab, embedding_dim)
        std = sqrt(2.0 / (n_vocab + embedding_dim))
        val = sqrt(3.0) * std
        self.embedding.weight.data.uniform_(-val, val)
        # dead weights the reference keeps in its state_dict (tacotron.py:27-29; speaker_id is unused by the math)
        self.speaker_embed_table = nn.Embedding(num_speakers, hparams["speaker_embedding_size"])
        self.deep_linear = nn.Linear(hparams["speaker_embedding_size"], 512)
        self.encoder = Encoder(hparams)
        self.decoder = Decoder(hparams)
        self.postnet = Postnet(hparams)
        self.__dict__["_engine"] = None
        self._adopt()

    def _adopt(self):
        """The sub-modules reach the engine through a weak reference to their owner (kept out of the module tree)."""
        import weakref
        for sub in (self.encoder, self.decoder, self.decoder.prenet, self.decoder.attention_layer, self.postnet):
            sub.__dict__["_owner"] = weakref.ref(self)

    def _eng(self):
        if self.__dict__.get("_engine") is None:
            self.__dict__["_engine"] = _TacoEngine(self)
        return self.__dict__["_engine"]

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_engine"] = None
        return d

    def __setstate__(self, state):
        super().__setstate__(state)
        self._adopt()           # unpickled / deep-copied sub-modules belong to THIS object

    def _check(self, t):
        if not t.is_cuda:
            raise _lib.T2SError("Tacotron (MI355X build) needs CUDA/HIP tensors; got %s - there is no CPU fallback" % t.device)

    def forward(self, inputs, prenet_masks=None, train_masks=None):
        """Teacher-forced forward (reference tacotron.py:36-49).  In ``.train()`` mode BatchNorm uses batch statistics
        and every dropout is live (``train_masks`` = {'enc': [3 x [B,C,T_in]], 'att': [T,B,H], 'dec': [T,B,H],
        'post': [5 x [B,C,T]]} of {0,1} injects the draws; otherwise they are drawn on the device from a fresh seed per
        call, reproducible under ``torch.manual_seed``).  With autograd enabled on trainable parameters the call goes
        through ``tacotron/autograd.py``: forward with saves + hand-written HIP backward."""
        text_inputs, text_lengths, mels, max_len, speaker_id, output_lengths = inputs
        self._check(text_inputs)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import tacotron_forward_with_grad
            # parse_output (tacotron.py:49) has been applied by the engine: one HIP launch inside the forward, see there
            return list(tacotron_forward_with_grad(self, text_inputs, text_lengths.data, mels, output_lengths.data, prenet_masks,
                                                   train_masks, self._host_max_len(max_len)))
        with torch.no_grad():
            out = self._eng().forward(text_inputs, text_lengths.data, mels, output_lengths.data, prenet_masks,
                                      train_masks=train_masks, max_len=self._host_max_len(max_len))
        return self._as_module_dtype(out)

    @staticmethod
    def _host_max_len(max_len):
        """The batch tuple's `max_len` (reference parse_batch: torch.max(input_lengths).item(), a host number) when it is one; a
        device tensor or None makes the encoder read the maximum back itself."""
        if isinstance(max_len, (int, float)) and not isinstance(max_len, bool):
            return int(max_len)
        return None

    def inference(self, inputs, speaker_id=None, prenet_masks=None, train_masks=None):
        """Autoregressive decode (reference tacotron.py:51-65).  Works in ``.train()`` mode as the reference's does (batch-statistics
        BatchNorm and live dropout everywhere; ``train_masks`` injects the draws), although inference.py:61 calls it in eval mode."""
        self._check(inputs)
        with torch.no_grad():
            out = self._eng().inference(inputs, prenet_masks, train_masks=train_masks)
        return self._as_module_dtype(self.parse_output(out))

    def _as_module_dtype(self, outputs):
        """After ``model.half()`` (reference inference.py:61) the reference's outputs are half tensors; the kernels here
        compute in f32 from the (half-rounded) weights and the results are cast on the way out."""
        dt = self.embedding.weight.dtype
        if dt in (torch.float16, torch.bfloat16):
            return [o.to(dt) if torch.is_floating_point(o) else o for o in outputs]
        return outputs

    def parse_output(self, outputs, output_lengths=None):
        """Reference tacotron.py:67-76: zero mel / 1e3 gate beyond each output length."""
        if self.hparams["mask_padding"] and output_lengths is not None:
            o0, o1, o2 = outputs[0].data, outputs[1].data, outputs[2].data
            B, n_mel, T = o0.shape
            if all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (o0, o1, o2)) and o2.numel() == B * T \
                    and int(output_lengths.numel()) == B:
                olen32 = output_lengths.to(device=o0.device, dtype=torch.int32).contiguous()
                _lib.call("t2s_taco_parse_output", _lib.ptr(o0), _lib.ptr(o1), _lib.ptr(o2), _lib.ptr(olen32), B, n_mel, T,
                          _lib.current_stream())
                return outputs
            mask = ~get_mask_from_lengths(output_lengths.to(outputs[0].device))
            T = outputs[0].size(2)
            if mask.size(1) < T:
                mask = torch.cat([mask, mask.new_ones(mask.size(0), T - mask.size(1))], 1)
            # .data, as the reference does (tacotron.py:73-75): the fill bypasses autograd
            outputs[0].data.masked_fill_(mask.unsqueeze(1), 0.0)
            outputs[1].data.masked_fill_(mask.unsqueeze(1), 0.0)
            outputs[2].data.masked_fill_(mask, 1e3)
        return outputs

    def parse_batch(self, batch):
        """Reference tacotron.py:78-89."""
        text_padded, input_lengths, mel_padded, gate_padded, speaker_id, output_lengths = batch
        dev = self.embedding.weight.device
        text_padded = text_padded.to(dev).long()
        input_lengths = input_lengths.to(dev).long()
        max_len = torch.max(input_lengths.data).item()
        mel_padded = mel_padded.to(dev).float()
        gate_padded = gate_padded.to(dev).float()
        speaker_id = speaker_id.to(dev).float()
        output_lengths = output_lengths.to(dev).long()
        return ((text_padded, input_lengths, mel_padded, max_len, speaker_id, output_lengths),
                (mel_padded, gate_padded))

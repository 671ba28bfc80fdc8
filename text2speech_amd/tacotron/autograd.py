"""Training path of the MI355X Tacotron-2: teacher-forced forward with saved per-step state and a hand-scheduled
backward that issues HIP kernels only (reference: autograd over tacotron/tacotron.py:36-49,355-429 and
tacotron/modules.py:19-22,94-137, driven by train.py:219-225 `y_pred = model(x); loss = criterion(y_pred, y);
loss.backward()`).

Structure of the backward (all through the C ABI of libt2s_hip.so):
  postnet  : BatchNorm(batch statistics)+tanh+dropout backward (t2s_bn_bwd), conv data gradients as convolutions
             with transposed / mirrored weights (t2s_conv_accumulate), conv weight gradients as time-contracting GEMMs
             (t2s_plane_transpose + t2s_wgrad_gemm)
  decoder  : projection, then T_out steps in reverse: LSTM-cell pointwise backward (t2s_lstm_cell_bwd), W^T dgates
             (t2s_gemv on transposed weights), attention backward (t2s_taco_att_bwd); weight gradients of both LSTM cells,
             the query / memory / prenet / projection Linears as ONE split-K GEMM each over all (step, batch) items
             (t2s_rows_to_tm + t2s_wgrad_gemm)
  encoder  : BiLSTM BPTT, convolutions, embedding
"""
import ctypes
import os

import torch

from .. import _lib
from .tacotron import _f32, pool_take, BufferPool

vp, i32, f32c, lng = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long


class _AttBwd(ctypes.Structure):
    """Mirror of t2s_att_bwd."""
    _fields_ = ([("dctx1", vp), ("sc1", lng), ("dctx2", vp), ("sc2", lng), ("dctx3", vp), ("sc3", lng),
                 ("w_cur", vp), ("s_wcur", lng), ("w_prev", vp), ("wc_prev", vp), ("s_wprev", lng), ("s_wcprev", lng),
                 ("q", vp), ("pmem", vp), ("memory", vp), ("lengths", vp), ("w_loc_conv", vp), ("w_loc_dense", vp),
                 ("w_v", vp), ("dw_carry", vp), ("dwc_carry", vp), ("d_q", vp), ("d_pmem", vp), ("d_memory", vp),
                 ("dD_part", vp), ("dK_part", vp), ("dv_part", vp), ("dw_buf", vp), ("df_buf", vp), ("dq_part", vp),
                 ("dctx_out", vp)] +
                [(n, i32) for n in ("B", "T", "att_dim", "enc_dim", "loc_f", "loc_ks")] +
                [("ctx", vp), ("s_ctx", lng), ("dw_carry_out", vp), ("dwc_carry_out", vp)])


class _Bptt(ctypes.Structure):
    """Mirror of t2s_taco_bptt."""
    _fields_ = ([(n, i32) for n in ("B", "T_in", "T_out", "T_cap", "prenet_dim", "enc_dim", "att_rnn_dim", "dec_rnn_dim",
                                    "att_dim", "loc_filters", "loc_kernel")] +
                [(n, vp) for n in ("W_dT", "W_aT", "w_query", "w_loc_conv", "w_loc_dense", "w_v", "dec_gates_all", "dec_c_all",
                                   "att_gates_all", "att_c_all", "q_all", "wcum_all", "align", "pmem", "memory", "lengths",
                                   "att_drop", "dec_drop")] +
                [("att_drop_scale", f32c), ("dec_drop_scale", f32c)] +
                [(n, vp) for n in ("d_hc", "out_d", "out_a", "dg_d", "dg_a", "dq_all", "dc_d", "dc_a", "dw_c",
                                   "dwc_c", "d_pmem", "d_memory", "dD_part", "dK_part", "dv_part", "dw_buf", "df_buf",
                                   "dq_part", "dctx_all", "ctx_all")] +
                [("s_ctx_step", lng), ("s_ctx_item", lng), ("dw_c2", vp), ("dwc_c2", vp), ("att_xbuf", vp)])


class _BnBwd(ctypes.Structure):
    """Mirror of t2s_bn_bwd_args."""
    _fields_ = [("x", vp), ("mean", vp), ("var", vp), ("gamma", vp), ("beta", vp), ("eps", f32c), ("dout_f32", vp),
                ("dout_hi", vp), ("dout_lo", vp), ("mask", vp), ("mask_scale", f32c), ("act", i32), ("dgamma", vp),
                ("dbeta", vp), ("dx_hi", vp), ("dx_lo", vp), ("B", i32), ("C", i32), ("T", i32), ("Lp", i32), ("halo", i32)]


def _p(t, off_elems=0):
    """raw address of a tensor (+ element offset), None -> NULL"""
    if t is None:
        return None
    return vp(t.data_ptr() + off_elems * t.element_size())


def _ru(a, b):
    return -(-a // b) * b


import os as _os
_IW_WGS = int(_os.environ.get("T2S_IW_WGS", "4096"))


class _Bwd:
    def __init__(self, eng, sv):
        self.eng, self.sv = eng, sv
        self.m = eng.m
        self.dev = sv["memory"].device
        self.st = _lib.current_stream()
        self.grads = {}
        self.keep = []
        if eng.pool is None:
            eng.pool = BufferPool()
        self.leases = []            # buffers of this backward pass, back in the engine's pool when it is released
        self.zero_bias = pool_take(eng.pool, self.leases, "zero_bias", (8192,), torch.float32, self.dev, zero_once=True)
        self._seq = 0
        # accumulators / carries of this pass come out of one arena cleared by ONE launch; its size is what the previous pass of
        # this engine asked for (the first pass, and any request that does not fit, falls back to a fill per buffer)
        self._za, self._za_off, self._za_need = None, 0, 0
        cap = getattr(eng, "zero_arena_elems", 0)
        if cap:
            self._za = pool_take(eng.pool, self.leases, "zero_arena", (cap,), torch.float32, self.dev)
            _lib.call("t2s_zero_fill", _p(self._za), cap * 4, self.st)

    def new(self, *shape, tag=None):
        """f32 scratch every element of which its producer writes.  tag: take it from the engine's pool (large buffers)."""
        if tag is not None:
            return pool_take(self.eng.pool, self.leases, tag, shape, torch.float32, self.dev)
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def zeros(self, *shape):
        """f32 scratch that must START at zero (accumulators, carries): cleared every step.  Never handed out as a gradient."""
        n = 1
        for d in shape:
            n *= int(d)
        n_al = _ru(max(n, 1), 64)                   # 256-byte granules
        self._za_need += n_al
        if self._za is not None and self._za_off + n_al <= self._za.numel():
            t = self._za[self._za_off:self._za_off + n].view(*shape)
            self._za_off += n_al
            return t
        return torch.zeros(*shape, dtype=torch.float32, device=self.dev)

    def close(self):
        """Join the helper stream; remember the arena size this pass needed (the next pass of this engine allocates it)."""
        if self.__dict__.get("side_used"):
            torch.cuda.current_stream(self.dev).wait_stream(self.side_stream())
        self.eng.zero_arena_elems = max(getattr(self.eng, "zero_arena_elems", 0), self._za_need)

    def bf(self, *shape, tag=None, extent=()):
        """bf16 operand planes.  Their producers write the valid region and leave padding (halo rows, rows / columns / K-blocks
        past the operand's extent) untouched, and that padding must read as zero.  Which elements are padding depends on the
        shape, the call site and the valid row counts `extent` (the batch's padded T, the item count: several of them share one
        padded shape) - never on the data or on per-entry lengths (ragged entries are zeroed BY VALUE inside the valid region).
        A pooled buffer (tag = the call site, + loop index where several are alive at once) is zeroed at creation and again
        whenever it is handed out for a different extent.  No tag: a fresh zero-filled tensor."""
        if tag is not None:
            return pool_take(self.eng.pool, self.leases, tag, shape, torch.bfloat16, self.dev, zero_once=True, extent=tuple(extent))
        return torch.zeros(*shape, dtype=torch.bfloat16, device=self.dev)

    def seq(self):
        self._seq += 1
        return self._seq

    # ------------------------------------------------------------------ generic pieces
    def gemv(self, W, ld, K, x_ptr, sx, y_ptr, sy_item, rows, items):
        _lib.call("t2s_gemv", _p(W), ld, K, None, 0, 0, x_ptr, K, sx, None, 0, 0, None, 0, 0, None, None, y_ptr, sy_item, 1,
                  rows, items, 0, None, 0, 1.0, self.st)

    def transpose(self, w2d):
        R, C = w2d.shape
        out = self.new(C, R)
        _lib.call("t2s_transpose", _p(w2d), _p(out), R, C, self.st)
        self.keep.append(w2d)
        return out

    def items_wgrad(self, items, a_srcs, x_srcs, M, N_cols):
        """sum over items of A[item][m] * [X | 1][item][n] via the split-K GEMM.  a_srcs / x_srcs: lists of
        (ptr, ld, C, row_off, shift).  Returns (P [nsplit][M4][N], nsplit, M4, N); column N-1 is the ones column."""
        items_pad = _ru(items, 32)
        nch = items_pad // 32
        M4 = _ru(M, 4)
        Mpad = _lib.padded_rows(M4)
        N = N_cols + 1
        Npad = _ru(N, 256)
        n_ = self.seq()             # the n-th call of a backward pass is the same call site in every step
        ex = (items,)
        A = (self.bf(nch, Mpad, 32, tag=("iw_Ah", n_), extent=ex), self.bf(nch, Mpad, 32, tag=("iw_Al", n_), extent=ex))
        X = (self.bf(nch, Npad, 32, tag=("iw_Xh", n_), extent=ex), self.bf(nch, Npad, 32, tag=("iw_Xl", n_), extent=ex))
        for (ptr, ld, C, off, shift) in a_srcs:
            _lib.call("t2s_rows_to_tm", ptr, ld, items, items_pad, shift, C, _p(A[0]), _p(A[1]), Mpad, off, self.st)
        for (ptr, ld, C, off, shift) in x_srcs:
            _lib.call("t2s_rows_to_tm", ptr, ld, items, items_pad, shift, C, _p(X[0]), _p(X[1]), Npad, off, self.st)
        _lib.call("t2s_tm_ones_row", _p(X[0]), _p(X[1]), 1, items_pad, 0, items, Npad, N_cols, self.st)
        # split-K slabs: 16 where the item count allows.  Fewer for the products that are many output tiles already (the LSTM cells':
        # 176 / 128 tiles, 16 slabs of 4096 x 2561 floats = 671 MB written and read back) measured EQUAL - T2S_IW_WGS = target number
        # of workgroups, 256 / 768 / 1536 / 4096: 82.2-83.5 ms per train step for all of them (profiles/r04_iw_split_ab.txt)
        tiles = -(-M4 // 256) * -(-N // 256)
        ks = max(1, min(16, nch, -(-_IW_WGS // tiles)))
        P = self.new(ks, M4, N)
        _lib.call("t2s_wgrad_gemm", _p(A[0]), _p(A[1]), _p(X[0]), _p(X[1]), _p(self.zero_bias), _p(P), 1, M4, N, Mpad, Npad,
                  nch, 0, nch, ks, self.st)
        self.keep += [A, X]
        return P, ks, M4, N

    def slab_to_grad(self, P, ks, M4, N, param, O, Cin, col_off, bias_param=None, Kt=1, tap_stride=0, row_off=0):
        w = _f32(param)
        dW = self.new(*param.shape)
        db = None if bias_param is None else self.new(O)
        _lib.call("t2s_wn_backward", _p(P), ks, M4, N, row_off, col_off, tap_stride, N - 1, 1, _p(w), None, O, Cin, Kt, _p(dW), None,
                  _p(db), 0, self.st)
        self.grads[id(param)] = dW
        if bias_param is not None:
            self.grads[id(bias_param)] = db
        self.keep.append(w)

    # ------------------------------------------------------------------ conv + BatchNorm stack (postnet / encoder)
    def side_stream(self):
        """The engine's helper stream for work nothing on the backward's dependent chain waits for (joined in close())."""
        side = getattr(self.eng, "enc_side_stream", None)
        if side is None:
            side = self.eng.enc_side_stream = torch.cuda.Stream(device=self.dev)
        return side

    def conv_bn_stack_backward(self, saves, dout_f32=None, dout_planes=None, wgrad_side=False):
        """Backward of [conv -> BN(batch stats) -> act -> dropout] x n.  The gradient of the stack's output comes as f32
        [B][C_last][T] or as planes.  Returns the gradient w.r.t. the stack's input as planes (hi, lo).
        wgrad_side: the weight gradients (nothing downstream needs them) go to the helper stream, layer by layer behind an
        event - for the postnet they then run beside the decoder's BPTT loop, which leaves most of the chip idle."""
        d_planes = dout_planes
        main_t = torch.cuda.current_stream(self.dev)
        side_t = self.side_stream() if wgrad_side else None
        for i in reversed(range(len(saves))):
            s = saves[i]
            conv, bn, layer = s["seq"][0].conv, s["seq"][1], s["layer"]
            B, T, Lp, halo = s["B"], s["T"], s["Lp"], s["halo"]
            Cout, Cin, Kt = layer["Cout"], layer["Cin"], layer["taps"]
            occ = -(-Cout // 32)
            cs_ = self.seq()
            ex = (T,)
            dconv = (self.bf(B, occ, Lp, 32, tag=("cs_dch", cs_), extent=ex), self.bf(B, occ, Lp, 32, tag=("cs_dcl", cs_), extent=ex))
            dgamma, dbeta = self.new(Cout), self.new(Cout)
            g32, b32 = _f32(bn.weight), _f32(bn.bias)
            a = _BnBwd(x=s["y"].data_ptr(), mean=s["mean"].data_ptr(), var=s["var"].data_ptr(), gamma=g32.data_ptr(),
                       beta=b32.data_ptr(), eps=float(bn.eps),
                       dout_f32=None if d_planes is not None else dout_f32.data_ptr(),
                       dout_hi=None if d_planes is None else d_planes[0].data_ptr(),
                       dout_lo=None if d_planes is None else d_planes[1].data_ptr(),
                       mask=None if s["mask"] is None else s["mask"].data_ptr(), mask_scale=2.0, act=s["act"],
                       dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), dx_hi=dconv[0].data_ptr(), dx_lo=dconv[1].data_ptr(),
                       B=B, C=Cout, T=T, Lp=Lp, halo=halo)
            part = torch.empty(B * Cout * 2, dtype=torch.float64, device=self.dev)      # per-element sums, added in a fixed order
            _lib.call("t2s_bn_bwd", ctypes.byref(a), _p(part), self.st)
            self.keep.append(part)
            self.grads[id(bn.weight)], self.grads[id(bn.bias)] = dgamma, dbeta
            self.keep += [g32, b32, dconv]
            # conv weight gradient: contraction over time per batch element (split-K slabs), bias from the ones row
            nt = -(-Lp // 32)
            Cin_pad = _ru(Cin, 32)
            Mpad = _lib.padded_rows(Cout)
            Ncols = Kt * Cin_pad
            N = Ncols + 1
            Npad = _ru(N, 256)
            A = (self.bf(B, nt, Mpad, 32, tag=("cs_Ah", cs_), extent=ex), self.bf(B, nt, Mpad, 32, tag=("cs_Al", cs_), extent=ex))
            X = (self.bf(B, nt, Npad, 32, tag=("cs_Xh", cs_), extent=ex), self.bf(B, nt, Npad, 32, tag=("cs_Xl", cs_), extent=ex))
            st_main = self.st
            if side_t is not None:          # dconv is final: the weight-gradient sequence of this layer moves to the helper stream
                ev = torch.cuda.Event()
                ev.record(main_t)
                side_t.wait_event(ev)
                self.st = _lib.c_vp(side_t.cuda_stream)
            _lib.call("t2s_plane_transpose", _p(dconv[0]), _p(dconv[1]), B, occ, occ, Lp, 0, _p(A[0]), _p(A[1]), Mpad, 0, self.st)
            icc = Cin_pad // 32
            for tap in range(Kt):
                _lib.call("t2s_plane_transpose", _p(s["Xh"]), _p(s["Xl"]), B, icc, icc, Lp, tap - Kt // 2, _p(X[0]), _p(X[1]),
                          Npad, tap * Cin_pad, self.st)
            _lib.call("t2s_tm_ones_row", _p(X[0]), _p(X[1]), B, Lp, halo, T, Npad, Ncols, self.st)
            M4 = _ru(Cout, 4)
            P = self.new(B, M4, N)
            _lib.call("t2s_wgrad_gemm", _p(A[0]), _p(A[1]), _p(X[0]), _p(X[1]), _p(self.zero_bias), _p(P), B, M4, N, Mpad,
                      Npad, nt, 0, nt, 1, self.st)
            self.slab_to_grad(P, B, M4, N, conv.weight, Cout, Cin, 0, conv.bias, Kt=Kt, tap_stride=Cin_pad)
            self.keep += [A, X, P]
            if side_t is not None:
                self.st = st_main
                self.side_used = True
            # data gradient: convolution of dconv with the transposed, tap-mirrored weight
            w32 = _f32(conv.weight)
            Opad = _ru(Cout, 32)
            Mi = _lib.padded_rows(Cin)
            At = (self.bf(Kt * Opad // 32, Mi, 32, tag=("cs_Ath", cs_)), self.bf(Kt * Opad // 32, Mi, 32, tag=("cs_Atl", cs_)))
            _lib.call("t2s_pack_transposed", _p(w32), None, Cout, Cin, Kt, 1, Opad, Mi, 0, _p(At[0]), _p(At[1]), 0, self.st)
            d_in = (self.bf(B, icc, Lp, 32, tag=("cs_dih", cs_), extent=ex), self.bf(B, icc, Lp, 32, tag=("cs_dil", cs_), extent=ex))
            _lib.call("t2s_conv_accumulate", _p(At[0]), _p(At[1]), _p(self.zero_bias), _p(dconv[0]), _p(dconv[1]), 0, _p(d_in[0]),
                      _p(d_in[1]), B, Cout, _ru(Cin, 4), Kt, 1, 1, T, Lp, halo, Mi, 0, self.st)
            self.keep += [w32, At, d_in]
            d_planes = d_in
        return d_planes

    # ------------------------------------------------------------------ the whole thing
    def run(self, g_mel, g_mel_post, g_gate):
        sv, m, dev, st = self.sv, self.m, self.dev, self.st
        dec = m.decoder
        P = self.eng.prep
        B, T, n_mel = sv["B"], sv["T_out"], sv["n_mel"]
        A, D, Pd = dec.attention_rnn_dim, dec.decoder_rnn_dim, dec.prenet_dim
        S = sv["S"]
        memory = sv["memory"]
        T_in, E = memory.size(1), memory.size(2)
        al = dec.attention_layer
        ad = al.query_layer.linear_layer.out_features
        F_, KS = al.location_layer.location_conv.conv.out_channels, al.location_layer.location_conv.conv.kernel_size[0]
        z = lambda t_: self.zeros(*t_.shape) if t_ is None else t_.to(torch.float32).contiguous()
        g_mel = self.zeros(B, n_mel, T) if g_mel is None else g_mel.to(torch.float32).contiguous()
        g_mel_post = self.zeros(B, n_mel, T) if g_mel_post is None else g_mel_post.to(torch.float32).contiguous()
        g_gate = self.zeros(B, T) if g_gate is None else g_gate.to(torch.float32).contiguous()
        # ---- postnet: mel_post = mel + postnet(mel) ----
        d_in = self.conv_bn_stack_backward(sv["post_convs"], g_mel_post, wgrad_side=os.environ.get("T2S_POSTNET_WGRAD_SIDE", "0") == "1")
        d_mel = self.new(B, n_mel, T)
        _lib.call("t2s_add3", _p(g_mel), _p(g_mel_post), None, d_mel.numel(), _p(d_mel), st)
        s0 = sv["post_convs"][0]
        _lib.call("t2s_planes_to_f32", _p(d_in[0]), _p(d_in[1]), B, n_mel, T, s0["Lp"], s0["halo"], _p(d_mel), 1, st)
        # ---- projection + gate (hoisted over all steps): proj[t,b,:] = W_proj [h_dec | ctx] + b ----
        items = T * B
        NP = _ru(n_mel + 1, 4)
        d_proj = self.zeros(T, B, NP)
        d_proj[:, :, :n_mel] = d_mel.permute(2, 0, 1)
        d_proj[:, :, n_mel] = g_gate.permute(1, 0)
        d_proj = d_proj.view(items, NP)
        DE = D + E
        w_proj = self.zeros(NP, DE)
        w_proj[:n_mel + 1] = P["w_proj"]
        w_projT = self.transpose(w_proj)                                     # [DE][NP]
        d_hc = self.new(items, DE)
        # d[h_dec | ctx] = d_proj . W_proj over all T*B items: a 1x1 "convolution" along the item axis on the split-bf16 GEMM
        # (K = 81 is too short and 25600 items too many for the GEMV kernels: 4.5 ms there, ~0.1 ms here)
        Cpad_p, Mpad_p = _ru(NP, 32), _lib.padded_rows(DE)
        Ap = (self.bf(Cpad_p // 32, Mpad_p, 32), self.bf(Cpad_p // 32, Mpad_p, 32))
        bias_p = self.zeros(Mpad_p)
        _lib.call("t2s_pack_conv_weight", _p(w_projT), None, 0, None, DE, NP, 1, 0, 0, 0, Mpad_p, 0, Cpad_p, _p(Ap[0]), _p(Ap[1]),
                  _p(bias_p), 0, st)
        Lp_p = _lib.plane_rows(items, 0)
        Xp = (self.bf(1, Cpad_p // 32, Lp_p, 32), self.bf(1, Cpad_p // 32, Lp_p, 32))
        _lib.call("t2s_rows_to_planes", _p(d_proj), 1, items, NP, Lp_p, 0, _p(Xp[0]), _p(Xp[1]), st)
        _lib.call("t2s_conv_bias_act", _p(Ap[0]), _p(Ap[1]), _p(bias_p), _p(Xp[0]), _p(Xp[1]), None, None, _p(d_hc), 1, 1, NP, DE,
                  1, 1, 0, items, Lp_p, 0, Mpad_p, st)
        self.keep += [Ap, Xp, bias_p]
        hc_all = sv["hc_all"]
        Pp, ks, M4, N = self.items_wgrad(items, [(_p(d_proj), NP, n_mel + 1, 0, 0)], [(_p(hc_all), DE, DE, 0, 0)], n_mel + 1, DE)
        lp, gl = dec.linear_projection.linear_layer, dec.gate_layer.linear_layer
        # rows [0, n_mel) -> linear_projection, row n_mel -> gate_layer
        dWp, dbp = self.new(n_mel + 1, DE), self.new(n_mel + 1)
        _lib.call("t2s_wn_backward", _p(Pp), ks, M4, N, 0, 0, 0, N - 1, 1, _p(w_proj), None, n_mel + 1, DE, 1, _p(dWp), None, _p(dbp),
                  0, st)
        self.grads[id(lp.weight)], self.grads[id(lp.bias)] = dWp[:n_mel].contiguous(), dbp[:n_mel].contiguous()
        self.grads[id(gl.weight)], self.grads[id(gl.bias)] = dWp[n_mel:].contiguous(), dbp[n_mel:].contiguous()
        # ---- decoder BPTT ----
        W_dT = self.transpose(torch.cat([P["dec_w_ih"], P["dec_w_hh"]], 1).contiguous())      # [A+E+D][4D]
        W_aT = self.transpose(torch.cat([P["att_w_ih"], P["att_w_hh"]], 1).contiguous())      # [Pd+E+A][4A]
        KD, KA = A + E + D, Pd + E + A
        # written whole, step by step, by the BPTT kernels (every batch entry, every column): pooled, never cleared
        out_d = self.new(T, B, KD, tag="bptt_out_d")
        out_a = self.new(T, B, KA, tag="bptt_out_a")
        dg_d, dg_a = self.new(T, B, 4 * D, tag="bptt_dg_d"), self.new(T, B, 4 * A, tag="bptt_dg_a")
        dq_all = self.new(T, B, ad, tag="bptt_dq_all")
        dc_d, dc_a = self.zeros(B, D), self.zeros(B, A)
        # carries of the attention weights: [3][B][T_in] (own part + the parts reaching in from the neighbouring 32-position
        # chunks) in two sets - the one-launch attention backward reads one and writes the other; the three-launch form works
        # in place on the first [B][T_in] of dw_c / dwc_c
        dw_c, dwc_c = self.zeros(3, B, T_in), self.zeros(3, B, T_in)
        dw_c2, dwc_c2 = self.zeros(3, B, T_in), self.zeros(3, B, T_in)
        d_pmem, d_memory = self.zeros(B, T_in, ad), self.zeros(B, T_in, E)
        nch = (T_in + 31) // 32                     # partial parameter gradients: one slot per (batch element, 32-position chunk)
        dD_p, dK_p, dv_p = self.zeros(B * nch, ad * F_), self.zeros(B * nch, F_ * 2 * KS), self.zeros(B * nch, ad)
        dw_buf, df_buf, dq_part = self.new(B, T_in), self.new(B, T_in, 32), self.new(B, nch, ad)
        # deferred d_memory (needs T_in % 4 == 0 for the slab layout); otherwise the loop accumulates it step by step
        dctx_all = self.new(T, B, E) if T_in % 4 == 0 else None
        align = S["align_out"]                     # [B][T_cap][T_in]
        T_cap = align.size(1)
        att_drop, dec_drop = S.get("att_drop"), S.get("dec_drop")
        w_v = P["w_v"].reshape(-1).contiguous()
        bp = _Bptt(B=B, T_in=T_in, T_out=T, T_cap=T_cap, prenet_dim=Pd, enc_dim=E, att_rnn_dim=A, dec_rnn_dim=D, att_dim=ad,
                   loc_filters=F_, loc_kernel=KS, W_dT=_p(W_dT), W_aT=_p(W_aT), w_query=_p(P["w_query"]), w_loc_conv=_p(P["w_loc_conv"]),
                   w_loc_dense=_p(P["w_loc_dense"]), w_v=_p(w_v), dec_gates_all=_p(S["dec_gates_all"]),
                   dec_c_all=_p(S["dec_c_all"]), att_gates_all=_p(S["att_gates_all"]), att_c_all=_p(S["att_c_all"]),
                   q_all=_p(S["q_all"]), wcum_all=_p(S["wcum_all"]), align=_p(align), pmem=_p(S["pmem"]), memory=_p(memory),
                   lengths=_p(sv["len32"]), att_drop=_p(att_drop), dec_drop=_p(dec_drop),
                   att_drop_scale=1.0 / (1.0 - dec.p_attention_dropout), dec_drop_scale=1.0 / (1.0 - dec.p_decoder_dropout),
                   d_hc=_p(d_hc), out_d=_p(out_d), out_a=_p(out_a), dg_d=_p(dg_d), dg_a=_p(dg_a), dq_all=_p(dq_all),
                   dc_d=_p(dc_d), dc_a=_p(dc_a), dw_c=_p(dw_c), dwc_c=_p(dwc_c), d_pmem=_p(d_pmem),
                   d_memory=_p(d_memory), dD_part=_p(dD_p), dK_part=_p(dK_p), dv_part=_p(dv_p), dw_buf=_p(dw_buf),
                   df_buf=_p(df_buf), dq_part=_p(dq_part), dctx_all=_p(dctx_all))
        if dctx_all is not None:                   # the forward's contexts: hc_all[t][b] = [h_dec | ctx]
            bp.ctx_all, bp.s_ctx_step, bp.s_ctx_item = _p(sv["hc_all"], D), B * (D + E), D + E
            bp.dw_c2, bp.dwc_c2 = _p(dw_c2), _p(dwc_c2)
            if T_in <= 512 and ad == 128:
                # exchange buffer of the attention cell's backward folded into the attention backward's launch (t2s_taco_bptt::att_xbuf):
                # 8-byte granules as f32 pairs, zero = no tag matches
                att_xbuf = self.zeros(2 * (B * nch * ad + 3))       # (+ error word, pace word, pace error word)
                bp.att_xbuf = _p(att_xbuf)
                self.keep.append(att_xbuf)
        _lib.call("t2s_taco_bptt_steps", ctypes.byref(bp), T, 0, st)        # the whole reversed loop, enqueued from C++
        if dctx_all is not None:
            # d_memory[b] = sum_t w[t][b][:] (x) d_ctx[t][b][:]: one contraction over the decoder steps per batch element
            # (all batch elements at once: one set of time-major planes per element, the GEMM's per-element slabs ARE the result -
            # five launches instead of 5 x B)
            items_pad = _ru(T, 32)
            nch_ = items_pad // 32
            Mpad_, N_ = _lib.padded_rows(T_in), E + 1
            Npad_ = _ru(N_, 256)
            ex = (T, T_in)
            Am = (self.bf(B, nch_, Mpad_, 32, tag="dmem_Ah", extent=ex), self.bf(B, nch_, Mpad_, 32, tag="dmem_Al", extent=ex))
            Xm = (self.bf(B, nch_, Npad_, 32, tag="dmem_Xh", extent=ex), self.bf(B, nch_, Npad_, 32, tag="dmem_Xl", extent=ex))
            _lib.call("t2s_rows_to_tm_batched", _p(align), T_in, T_cap * T_in, T, items_pad, 0, T_in, _p(Am[0]), _p(Am[1]),
                      nch_ * Mpad_ * 32, Mpad_, 0, B, st)
            _lib.call("t2s_rows_to_tm_batched", _p(dctx_all), B * E, E, T, items_pad, 0, E, _p(Xm[0]), _p(Xm[1]),
                      nch_ * Npad_ * 32, Npad_, 0, B, st)
            _lib.call("t2s_tm_ones_row", _p(Xm[0]), _p(Xm[1]), B, items_pad, 0, T, Npad_, E, st)
            Pm_ = self.new(B, T_in, N_, tag="dmem_P")
            _lib.call("t2s_wgrad_gemm", _p(Am[0]), _p(Am[1]), _p(Xm[0]), _p(Xm[1]), _p(self.zero_bias), _p(Pm_), B, T_in, N_, Mpad_,
                      Npad_, nch_, 0, nch_, 1, st)
            _lib.call("t2s_wn_backward", _p(Pm_), 1, B * T_in, N_, 0, 0, 0, N_ - 1, 1, _p(memory), None, B * T_in, E, 1,
                      _p(d_memory), None, None, 0, st)
            self.keep += [dctx_all, Am, Xm, Pm_]
        self.keep += [dw_buf, df_buf, dq_part, dw_c2, dwc_c2]
        # ---- d_memory is final once the memory layer's share is in: the encoder BiLSTM's BPTT starts now, on the side stream ----
        items_m = B * T_in
        W_mT = self.transpose(P["w_mem"])                                    # [E][ad]
        d_mem2 = self.new(B, T_in, E)
        self.gemv(W_mT, ad, ad, _p(d_pmem), ad, _p(d_mem2), E, E, items_m)
        d_mem_tot = self.new(B, T_in, E)
        _lib.call("t2s_add3", _p(d_memory), _p(d_mem2), None, d_mem_tot.numel(), _p(d_mem_tot), st)
        self.d_memory = d_mem_tot
        if os.environ.get("T2S_ENC_BWD_SIDE", "1") != "0":
            from .autograd_encoder import encoder_lstm_backward
            self.enc_pending = encoder_lstm_backward(self, d_mem_tot)
        # ---- weight gradients over all (step, batch) items ----
        ar = dec.attention_rnn
        Pa, ks, M4, N = self.items_wgrad(items, [(_p(dg_a), 4 * A, 4 * A, 0, 0)],
                                         [(_p(sv["pre_all"]), Pd, Pd, 0, 0), (_p(hc_all, D), DE, E, Pd, B),
                                          (_p(S["att_h_all"]), A, A, Pd + E, B)], 4 * A, KA)
        self.slab_to_grad(Pa, ks, M4, N, ar.weight_ih, 4 * A, Pd + E, 0, ar.bias_ih)
        self.slab_to_grad(Pa, ks, M4, N, ar.weight_hh, 4 * A, A, Pd + E, ar.bias_hh)
        dr = dec.decoder_rnn
        Pdd, ks, M4, N = self.items_wgrad(items, [(_p(dg_d), 4 * D, 4 * D, 0, 0)],
                                          [(_p(S["att_h_all"]), A, A, 0, 0), (_p(hc_all, D), DE, E, A, 0),
                                           (_p(hc_all), DE, D, A + E, B)], 4 * D, KD)
        self.slab_to_grad(Pdd, ks, M4, N, dr.weight_ih, 4 * D, A + E, 0, dr.bias_ih)
        self.slab_to_grad(Pdd, ks, M4, N, dr.weight_hh, 4 * D, D, A + E, dr.bias_hh)
        Pq, ks, M4, N = self.items_wgrad(items, [(_p(dq_all), ad, ad, 0, 0)], [(_p(S["att_h_all"]), A, A, 0, 0)], ad, A)
        self.slab_to_grad(Pq, ks, M4, N, al.query_layer.linear_layer.weight, ad, A, 0)
        # memory layer: pmem = W_mem memory
        Pm, ks, M4, N = self.items_wgrad(items_m, [(_p(d_pmem), ad, ad, 0, 0)], [(_p(memory), E, E, 0, 0)], ad, E)
        self.slab_to_grad(Pm, ks, M4, N, al.memory_layer.linear_layer.weight, ad, E, 0)
        # attention parameters accumulated per batch element
        loc = al.location_layer
        for part, param, n in ((dD_p, loc.location_dense.linear_layer.weight, ad * F_),
                               (dK_p, loc.location_conv.conv.weight, F_ * 2 * KS), (dv_p, al.v.linear_layer.weight, ad)):
            g = self.new(*param.shape)
            _lib.call("t2s_sum_axis0", _p(part), B * nch, n, _p(g), st)
            if part is dD_p:                        # the slots hold dD^T [F][att_dim]
                gt = self.new(ad, F_)
                _lib.call("t2s_transpose", _p(g), _p(gt), F_, ad, st)
                g = gt
            self.grads[id(param)] = g
        # prenet (hoisted): pre_all = drop(relu(W2 drop(relu(W1 frames))))
        d_pre = out_a[:, :, :Pd].contiguous().view(items, Pd)
        dz2 = self.new(items, Pd)
        _lib.call("t2s_relu_drop_bwd", _p(d_pre), _p(sv["pre_all"]), 2.0, dz2.numel(), _p(dz2), st)
        l0, l1 = dec.prenet.layers[0].linear_layer, dec.prenet.layers[1].linear_layer
        P2, ks, M4, N = self.items_wgrad(items, [(_p(dz2), Pd, Pd, 0, 0)], [(_p(sv["p1"]), Pd, Pd, 0, 0)], Pd, Pd)
        self.slab_to_grad(P2, ks, M4, N, l1.weight, Pd, Pd, 0)
        W2T = self.transpose(P["w_pre2"])
        d_p1 = self.new(items, Pd)
        self.gemv(W2T, Pd, Pd, _p(dz2), Pd, _p(d_p1), Pd, Pd, items)
        dz1 = self.new(items, Pd)
        _lib.call("t2s_relu_drop_bwd", _p(d_p1), _p(sv["p1"]), 2.0, dz1.numel(), _p(dz1), st)
        P1, ks, M4, N = self.items_wgrad(items, [(_p(dz1), Pd, Pd, 0, 0)], [(_p(sv["frames"]), n_mel, n_mel, 0, 0)], Pd, n_mel)
        self.slab_to_grad(P1, ks, M4, N, l0.weight, Pd, n_mel, 0)
        self.d_memory = d_mem_tot
        self.keep += [d_proj, w_proj, d_hc, out_d, out_a, dg_d, dg_a, dq_all, d_pmem, d_memory, dz2, dz1, d_p1, d_pre, w_v]
        return self.grads


class _TacotronFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, text, text_lengths, mels, output_lengths, prenet_masks, train_masks, max_len, *params):
        eng = model._eng()
        sv = {}
        with torch.no_grad():
            out = eng.forward(text, text_lengths, mels, output_lengths, prenet_masks, train_masks=train_masks, save=sv, max_len=max_len)
        ctx.model, ctx.sv, ctx.params = model, sv, params
        ctx.mark_non_differentiable(out[3])
        return tuple(out)

    @staticmethod
    def backward(ctx, g_mel, g_mel_post, g_gate, g_align):
        eng = ctx.model._eng()
        with torch.no_grad():
            bw = _Bwd(eng, ctx.sv)
            grads = bw.run(g_mel, g_mel_post, g_gate)
            from .autograd_encoder import encoder_backward
            encoder_backward(bw, bw.d_memory)
            bw.close()
        outs = []
        for p in ctx.params:
            g = grads.get(id(p))
            outs.append(None if g is None else g.reshape(p.shape).to(p.dtype))
        ctx.model.__dict__["_last_bwd"] = bw          # keeps scratch alive until the next step's backward
        return (None, None, None, None, None, None, None, None, *outs)


def tacotron_forward_with_grad(model, text, text_lengths, mels, output_lengths, prenet_masks, train_masks, max_len=None):
    params = list(model.parameters())
    return list(_TacotronFn.apply(model, text, text_lengths, mels, output_lengths, prenet_masks, train_masks, max_len, *params))

"""Encoder part of the Tacotron-2 backward (reference tacotron.py:192-209 under autograd): BiLSTM BPTT, the input
projection's weight / data gradients through the GEMM paths, the conv + BatchNorm stack, and the embedding."""
import os

import torch

from .. import _lib
from .tacotron import _f32


def encoder_lstm_backward(bw, d_memory):
    """The BiLSTM's BPTT alone, on the engine's side stream: 2 x B/2 workgroups walking T steps (3.3 ms at B = 32, T = 256) leave
    seven eighths of the chip idle, and nothing but the rest of the encoder backward needs its result - so it is launched as soon
    as d_memory is final and runs beside the decoder's weight-gradient GEMMs.  encoder_backward() joins it."""
    from .autograd import _p
    sv, P = bw.sv, bw.eng.prep
    lstm = bw.m.encoder.lstm
    B, T = sv["enc_ids"].shape
    H = P["H"]
    memory = sv["memory"]
    dgx = bw.zeros(B, T, 8 * H)
    hprev = bw.zeros(B, T, 2 * H)
    whh = [_f32(lstm.weight_hh_l0), _f32(lstm.weight_hh_l0_reverse)]
    main = torch.cuda.current_stream(memory.device)
    side = getattr(bw.eng, "enc_side_stream", None)
    if side is None:
        side = bw.eng.enc_side_stream = torch.cuda.Stream(device=memory.device)
    ready = torch.cuda.Event()
    ready.record(main)              # d_memory, the cleared dgx / hprev and the forward's saves precede this point on the main stream
    side.wait_event(ready)
    xb = bw.eng._lstm_xbuf("bwd", B, memory.device) if H == 256 and T < 4095 else None
    if xb is not None:      # W_hh resident over four workgroups per (element, direction): t2s_taco_encoder_lstm_bwd_split
        _lib.call("t2s_taco_encoder_lstm_bwd_split", _p(d_memory), _p(memory), _p(sv["enc_gates"]), _p(sv["enc_c"]), _p(whh[0]),
                  _p(whh[1]), _p(sv["enc_len32"]), _p(dgx), _p(hprev), B, T, H, memory.size(1), _p(xb[0]), xb[1],
                  _lib.c_vp(side.cuda_stream))
    else:
        _lib.call("t2s_taco_encoder_lstm_bwd", _p(d_memory), _p(memory), _p(sv["enc_gates"]), _p(sv["enc_c"]), _p(whh[0]),
                  _p(whh[1]), _p(sv["enc_len32"]), _p(dgx), _p(hprev), B, T, H, memory.size(1), _lib.c_vp(side.cuda_stream))
    done = torch.cuda.Event()
    done.record(side)
    bw.keep += [d_memory, whh]
    return dict(dgx=dgx, hprev=hprev, done=done, main=main)


def encoder_backward(bw, d_memory):
    """d_memory: [B][T_enc][2H] gradient w.r.t. the encoder output (already summed over its consumers)."""
    from .autograd import _p, _ru
    sv, m, P, st = bw.sv, bw.m, bw.eng.prep, bw.st
    enc = m.encoder
    lstm = enc.lstm
    ids = sv["enc_ids"]
    B, T = ids.shape
    H = P["H"]
    memory = sv["memory"]
    T_out = memory.size(1)
    halo, Lp = 2, sv["enc_Lp"]
    # ---- BiLSTM BPTT (launched earlier on the side stream, or now) ----
    pend = bw.__dict__.pop("enc_pending", None) or encoder_lstm_backward(bw, d_memory)
    dgx, hprev = pend["dgx"], pend["hprev"]
    pend["main"].wait_event(pend["done"])
    items = B * T
    for d, (w_hh, b_hh) in enumerate([(lstm.weight_hh_l0, lstm.bias_hh_l0), (lstm.weight_hh_l0_reverse, lstm.bias_hh_l0_reverse)]):
        Pw, ks, M4, N = bw.items_wgrad(items, [(_p(dgx, d * 4 * H), 8 * H, 4 * H, 0, 0)], [(_p(hprev, d * H), 2 * H, H, 0, 0)],
                                       4 * H, H)
        bw.slab_to_grad(Pw, ks, M4, N, w_hh, 4 * H, H, 0, b_hh)
    # ---- input projection gx = W_ih x + b: weight gradient contracts over time per batch element ----
    nt = -(-Lp // 32)
    items_pad = nt * 32
    M = 8 * H
    Mpad = _lib.padded_rows(M)
    A = (bw.bf(B, nt, Mpad, 32, tag="enc_Ah", extent=(T,)), bw.bf(B, nt, Mpad, 32, tag="enc_Al", extent=(T,)))
    _lib.call("t2s_rows_to_tm_batched", _p(dgx), M, T * M, T, items_pad, halo, M, _p(A[0]), _p(A[1]), nt * Mpad * 32, Mpad, 0, B, st)
    Cin = lstm.input_size
    icc = _ru(Cin, 32) // 32
    N = Cin + 1
    Npad = _ru(N, 256)
    X = (bw.bf(B, nt, Npad, 32, tag="enc_Xh", extent=(T,)), bw.bf(B, nt, Npad, 32, tag="enc_Xl", extent=(T,)))
    _lib.call("t2s_plane_transpose", _p(sv["enc_Xh"]), _p(sv["enc_Xl"]), B, icc, icc, Lp, 0, _p(X[0]), _p(X[1]), Npad, 0, st)
    _lib.call("t2s_tm_ones_row", _p(X[0]), _p(X[1]), B, Lp, halo, T, Npad, Cin, st)
    Pi = bw.new(B, M, N)
    _lib.call("t2s_wgrad_gemm", _p(A[0]), _p(A[1]), _p(X[0]), _p(X[1]), _p(bw.zero_bias), _p(Pi), B, M, N, Mpad, Npad, nt, 0, nt,
              1, st)
    bw.slab_to_grad(Pi, B, M, N, lstm.weight_ih_l0, 4 * H, Cin, 0, lstm.bias_ih_l0, row_off=0)
    bw.slab_to_grad(Pi, B, M, N, lstm.weight_ih_l0_reverse, 4 * H, Cin, 0, lstm.bias_ih_l0_reverse, row_off=4 * H)
    # ---- data gradient of the projection: d_x = W_ih^T dgx, as a 1-tap convolution over planes ----
    w_cat = P["lstm_in"]["keep"][0]                                  # [8H][Cin] (both directions)
    Mi = _lib.padded_rows(Cin)
    At = (bw.bf(M // 32, Mi, 32, tag="enc_Ath"), bw.bf(M // 32, Mi, 32, tag="enc_Atl"))
    _lib.call("t2s_pack_transposed", _p(w_cat), None, M, Cin, 1, 0, M, Mi, 0, _p(At[0]), _p(At[1]), 0, st)
    dgp = (bw.bf(B, M // 32, Lp, 32, tag="enc_dgph", extent=(T,)), bw.bf(B, M // 32, Lp, 32, tag="enc_dgpl", extent=(T,)))
    _lib.call("t2s_rows_to_planes", _p(dgx), B, T, M, Lp, halo, _p(dgp[0]), _p(dgp[1]), st)
    dx = (bw.bf(B, icc, Lp, 32, tag="enc_dxh", extent=(T,)), bw.bf(B, icc, Lp, 32, tag="enc_dxl", extent=(T,)))
    _lib.call("t2s_conv_accumulate", _p(At[0]), _p(At[1]), _p(bw.zero_bias), _p(dgp[0]), _p(dgp[1]), 0, _p(dx[0]), _p(dx[1]), B, M,
              Cin, 1, 1, 1, T, Lp, halo, Mi, 0, st)
    # ---- conv + BatchNorm stack, then the embedding ----
    d_emb_in = bw.conv_bn_stack_backward(sv["enc_convs"], dout_planes=dx, wgrad_side=os.environ.get("T2S_ENC_WGRAD_SIDE", "0") == "1")
    E = m.embedding.embedding_dim
    V = m.embedding.num_embeddings
    d_emb = bw.new(V, E)
    _lib.call("t2s_embedding_grad", _p(ids), _p(d_emb_in[0]), _p(d_emb_in[1]), B, T, E, V, Lp, halo, _p(d_emb), st)
    bw.grads[id(m.embedding.weight)] = d_emb
    bw.keep += [dgx, hprev, A, X, Pi, At, dgp, dx, d_emb_in]

"""Training path of the MI355X WaveGlow: forward with saved activations and a hand-scheduled backward
that issues the gfx950 kernels (reference: autograd over waveglow/glow.py:207-249, driven by
waveglow/train.py:116-123 `outputs = model((mel, audio)); loss = criterion(outputs); loss.backward()`).

``waveglow_forward_with_grad`` returns tensors hooked into autograd through one custom Function, so the
reference's training loop (criterion -> loss.backward() -> optimizer.step()) runs unchanged; every
parameter's ``.grad`` is produced by libt2s_hip.so, not by eager PyTorch.

Memory model (sized for 288 GB HBM3E): every WN layer's input, gate output and sigmoid stay
resident as split-bf16 planes (~105 MB per layer, 10.1 GB per step at 8 x 16000; tanh is rebuilt as gate output / sigmoid) instead of being
recomputed.
"""
import ctypes
import os

import torch

from . import _lib
from .glow import _f32c, _vg

L_ = _lib


def _ptr(t):
    return _lib.ptr(t)


class _TrainState:
    pass


def _bf(*shape, dev):
    return torch.zeros(*shape, dtype=torch.bfloat16, device=dev)


def _alloc_train(eng, B, L, dev):
    g = eng.geom()
    key = ("train", B, L, str(dev))
    st = eng.ws.get(key)
    if st is not None:
        return st
    m = eng.m
    Lp = _lib.plane_rows(L, g["halo"])
    xc, sc = g["Cpad"] // 32, g["Spad"] // 32
    C, nl = g["C"], g["nl"]
    st = _TrainState()
    st.Lp = Lp
    pl = lambda ch: (_bf(B, ch, Lp, 32, dev=dev), _bf(B, ch, Lp, 32, dev=dev))
    st.S_planes = pl(sc)
    # Training forward = the no-grad forward's kernels (WN.end folded into the gate GEMM's epilogue, residual-only GEMM on
    # 128-row tiles) + what the backward needs: every layer's input x, gate output and sigmoid.  The skip sum is never formed in
    # the forward; the backward rebuilds it once per flow from the saved gate outputs (t2s_wg_skip_sum), which is why the gate
    # outputs of a flow's layers sit side by side in ONE plane set (layer i = chunks [xc i, xc (i + 1)) of every batch entry).
    st.fold_train = C % 16 == 0 and eng.use_fold and not os.environ.get("T2S_TRAIN_NO_FOLD")
    st.act_bchunks = nl * xc if st.fold_train else 0
    st.layers, st.AF, st.GF = [], [], []
    for k in range(m.n_flows):
        if st.fold_train:
            st.AF.append(pl(nl * xc))
            st.GF.append(pl(nl * xc))
            st.layers.append([dict(X=pl(xc)) for _ in range(nl)])
        else:
            st.layers.append([dict(X=pl(xc), A=pl(xc), G=pl(xc)) for _ in range(nl)])   # tanh is rebuilt as A / G
    if st.fold_train:
        st.fold_acc = torch.zeros(_lib.load().t2s_wg_gate_fold_slots(B, C, L), B, 8, L, dtype=torch.float32, device=dev)
        st.skip = [torch.zeros(B, xc, Lp, 32, dtype=torch.float32, device=dev)] * m.n_flows      # one buffer, rebuilt per flow
        st.Mskip = _lib.padded_rows(C)
        st.A_skip = (_bf(nl * g["Cpad"] // 32, st.Mskip, 32, dev=dev), _bf(nl * g["Cpad"] // 32, st.Mskip, 32, dev=dev))
        st.bias_skip = torch.zeros(st.Mskip, dtype=torch.float32, device=dev)
        st.sw_scratch_side = torch.empty(_lib.load().t2s_small_wgrad_scratch(B, xc), dtype=torch.float32, device=dev)
    else:
        st.skip = [torch.zeros(B, xc, Lp, 32, dtype=torch.float32, device=dev) for _ in range(m.n_flows)]
    st.wn_out = [None] * m.n_flows
    # backward scratch
    nt = Lp // 32
    st.nt = nt
    # DS (d_skip of a flow) and the flow-wide d_pre planes exist twice: flow k works in parity k & 1, so the data-gradient chain
    # of the next flow never waits for the weight-gradient stream to finish reading this flow's (it may lag by one flow)
    st.DX = pl(xc)
    st.DS2 = [pl(xc), pl(xc)]
    # d_pre of ALL layers of a flow side by side (layer i = chunks [2 xc i, 2 xc (i + 1)) of every batch entry; 0.6 GB at
    # 8 x 16000): the conditioning gradient d_spect += sum_i W_cond,i^T d_pre_i is then ONE GEMM per flow with K = n_layers * 2C
    # instead of one 192-workgroup GEMM and one read-modify-write of d_spect per layer, and a layer's d_pre stays valid until the
    # flow ends (the weight-gradient stream may lag behind the data-gradient chain)
    st.DP2 = [pl(nl * 2 * xc), pl(nl * 2 * xc)]
    st.dp_chunks = nl * 2 * xc
    st.DSp = pl(sc)
    st.N2 = 3 * C + g["n_cond"] + 1
    st.N2pad = -(-st.N2 // 256) * 256
    st.N1 = C + 1
    st.N1pad = -(-st.N1 // 256) * 256
    st.M2pad = _lib.padded_rows(2 * C)
    tm = lambda rows: (_bf(B, nt, rows, 32, dev=dev), _bf(B, nt, rows, 32, dev=dev))
    st.TM_dp = tm(st.M2pad)
    st.TM_drs = tm(st.M2pad)
    st.TM_x = tm(st.N2pad)
    st.TM_act = tm(st.N1pad)
    # split-K of the weight-gradient GEMMs over (batch element, non-halo time chunk) flattened: as many slabs as fill the
    # chip's 256 CUs in one round of 256x256 tiles (a split tied to the batch gave 288-576 workgroups = 2-3 rounds)
    st.k0, st.k1 = g["halo"] // 32, -(-(g["halo"] + L) // 32)
    ksteps = B * (st.k1 - st.k0)

    def nsplit(M, N, fill=256):
        tiles = -(-M // 256) * -(-N // 256)
        ns = max(1, min(fill // tiles, ksteps // 8))
        while ns > 1 and -(-ksteps // ns) * (ns - 1) >= ksteps:      # every slab owns at least one K-step
            ns -= 1
        return ns
    # (its K-blocks are whole 32-row blocks shifted by up to +-halo rows: in bounds only when halo % 32 == 0, which geom() ensures)
    st.cl_ok = C % 32 == 0 and g["n_cond"] % 32 == 0 and g["halo"] % 32 == 0 and not os.environ.get("T2S_WGRAD_TM")
    assert not st.cl_ok or (st.k0 * 32 >= g["halo"] and st.k1 * 32 + g["halo"] <= Lp), "wgrad_cl K-blocks leave the plane"
    # res/skip weight gradient: N = C + 1 with the bias as an all-ones column would open another column of 256-wide tiles for
    # that one column whenever C % 256 == 0; the ping-pong kernel computes the bias gradient as row sums instead (bias_cols:
    # four partial-sum columns C .. C+3) and the GEMM is exactly C wide
    st.bias_cols = 1 if (st.cl_ok and C % 256 == 0 and os.environ.get("T2S_WGRAD_PP", "1") != "0"
                         and os.environ.get("T2S_WGRAD_BIAS_COL") != "ones") else 0
    st.N1g = C if st.bias_cols else st.N1                  # columns of the GEMM proper
    # workgroups a weight-gradient launch aims for (T2S_WGRAD_FILL2 / _FILL1: the gate / the res-skip convolution's): next to the
    # data-gradient stream what counts is CU-time per unit of work, and half the slabs are half the partial-sum traffic
    fill2, fill1 = int(os.environ.get("T2S_WGRAD_FILL2", "256")), int(os.environ.get("T2S_WGRAD_FILL1", "128"))
    st.ks2, st.ks1 = nsplit(2 * C, st.N2, fill2), nsplit(2 * C, st.N1g, fill1)
    # the last layer of a flow has no residual rows (M = C): twice the slabs of half the height, the same slab buffer
    st.ks1_last = nsplit(C, st.N1g, fill1) if st.cl_ok and nsplit(C, st.N1g, fill1) * C <= st.ks1 * 2 * C else st.ks1
    # floats per slab row: a multiple of 4 so that the channel-last kernel's epilogue stores whole 16-byte pieces
    st.ld2, st.ld1 = (-(-st.N2 // 4) * 4, -(-(st.N1g + 4 * st.bias_cols) // 4) * 4) if st.cl_ok else (st.N2, st.N1)
    st.P2 = torch.empty(st.ks2, 2 * C, st.ld2, dtype=torch.float32, device=dev)
    st.P1 = torch.empty(st.ks1, 2 * C, st.ld1, dtype=torch.float32, device=dev)
    st.P1b = torch.empty_like(st.P1)        # second slab set: the res/skip weight gradient on the data-gradient stream (backward_train)
    st.Mc = _lib.padded_rows(C)
    st.Ms = _lib.padded_rows(g["n_cond"])
    # PERM_PAIR8 row order for the backward's transposed operands (16-byte epilogue pieces) where the consuming GEMM runs on 256-row
    # ping-pong tiles: rows = C for W_rs^T / W_in^T, n_cond for W_cond^T
    st.p8_x = int(_lib.load().t2s_wg_bwd_pair8_ok(B, C, L))
    st.p8_c = int(_lib.load().t2s_wg_bwd_pair8_ok(B, g["n_cond"], L))
    # Transposed weight operands of the backward's data-gradient GEMMs, one set per (flow, layer) (0.8 GB at config.json
    # defaults): they depend on the weights only, so the forward produces them on its pack stream, under its own GEMMs, and the
    # backward's dependent chain has three small launches per layer less.
    pair = lambda kch, M: (_bf(kch, M, 32, dev=dev), _bf(kch, M, 32, dev=dev))
    st.A_rsT = [[pair(2 * C // 32, st.Mc) for _ in range(nl)] for _ in range(m.n_flows)]
    st.A_inT = [[pair(g["ks"] * 2 * C // 32, st.Mc) for _ in range(nl)] for _ in range(m.n_flows)]
    st.A_cT = [pair(nl * 2 * C // 32, st.Ms) for _ in range(m.n_flows)]                                      # K = (layer, channel)
    st.Winv = [torch.empty(eng._flow_geom(k)[1], eng._flow_geom(k)[1], dtype=torch.float32, device=dev) for k in range(m.n_flows)]
    st.zero_bias = torch.zeros(max(st.M2pad, st.Ms, 1024), dtype=torch.float32, device=dev)
    st.sw_scratch = torch.empty(_lib.load().t2s_small_wgrad_scratch(B, xc), dtype=torch.float32, device=dev)
    # channel-last weight-gradient GEMM (t2s_wgrad_cl): constant chunks and the per-layer operand tables (built on first use)
    st.zero_plane = _bf(Lp, 32, dev=dev)
    st.ones_plane = _bf(Lp, 32, dev=dev)
    st.ones_plane[g["halo"]:g["halo"] + L, 0] = 1.0
    st.cl_tables = {}
    eng.ws[key] = st
    return st


def _act_ptrs(ts, k, i, which):
    """(hi, lo) device pointers of layer i's saved gate output (which = "A") or sigmoid ("G") in flow k."""
    if not ts.fold_train:
        t = ts.layers[k][i][which]
        return _ptr(t[0]), _ptr(t[1])
    pair = ts.AF[k] if which == "A" else ts.GF[k]
    off = 2 * i * (pair[0].size(1) // len(ts.layers[k])) * pair[0].size(2) * 32          # bytes: xc chunks x Lp rows x 32 ch
    return _lib.c_vp(pair[0].data_ptr() + off), _lib.c_vp(pair[1].data_ptr() + off)


def _chunk_rows(pair, n_chunks, shift=0, first=0):
    """t2s_wgrad_chunk rows of chunks [first, first + n_chunks) of a (hi, lo) plane pair [B, chunks, Lp, 32], rows shifted by
    `shift` (a dilated tap)."""
    hi, lo = pair
    Bc, nch, Lp, _ = hi.shape
    return [[hi.data_ptr() + 2 * ((first + c) * Lp + shift) * 32, lo.data_ptr() + 2 * ((first + c) * Lp + shift) * 32, nch * Lp * 32]
            for c in range(n_chunks)]


def _chunk_table(ts, rows, dev):
    """Pad to 8 chunks per 256-row tile with the zero chunk; int64 [n][3] in device memory."""
    z = [ts.zero_plane.data_ptr(), ts.zero_plane.data_ptr(), 0]
    rows = rows + [z] * (-len(rows) % 8)
    return torch.tensor(rows, dtype=torch.int64).to(dev)


def _cl_tables(ts, key, lay_sv, last, xc, sc, ks, d, dev, layer):
    flow = key[0]
    """Operand tables of the two weight-gradient GEMMs of one WN layer (the buffers they point at are allocated once per shape,
    so the tables are built once)."""
    t = ts.cl_tables.get(key)
    if t is not None:
        return t
    ones = [[ts.ones_plane.data_ptr(), ts.zero_plane.data_ptr(), 0]]
    DS, DP = ts.DS2[flow & 1], ts.DP2[flow & 1]
    a1 = ([] if last else _chunk_rows(ts.DX, xc)) + _chunk_rows(DS, xc)               # [d_x ; d_skip]
    acts = _chunk_rows(ts.AF[flow], xc, first=layer * xc) if ts.fold_train else _chunk_rows(lay_sv["A"], xc)
    b1 = acts + ([] if ts.bias_cols else ones)                                          # [acts | 1] (or the kernel's row sums)
    a2 = _chunk_rows(DP, 2 * xc, first=layer * 2 * xc)                                  # d_pre (tanh half, sigmoid half)
    b2 = []
    for tap in range(ks):
        b2 += _chunk_rows(lay_sv["X"], xc, (tap - ks // 2) * d)                         # x shifted by the dilated tap
    b2 += _chunk_rows(ts.S_planes, sc) + ones                                           # [ .. | spect | 1]
    t = tuple(_chunk_table(ts, r, dev) for r in (a1, b1, a2, b2))
    ts.cl_tables[key] = t
    return t


def forward_train(eng, mel, audio):
    """WaveGlow.forward (glow.py:207-249) keeping what the backward needs."""
    m = eng.m
    eng._check_inputs(mel, audio)
    dev = audio.device
    B, T = audio.shape
    G = m.n_group
    L = T // G
    g = eng.geom()
    C, nl, ks = g["C"], g["nl"], g["ks"]
    ts = _alloc_train(eng, B, L, dev)
    # The per-step weight work (weight-norm + pack of every convolution, the folded WN.end matrices, the 12 log-determinants and
    # inverses of the 1x1 convolutions, the transposed operands of the backward) depends on the weights only: it runs on the
    # engine's pack stream, flow by flow, and the main stream waits for flow k's event only - as in the no-grad forward.
    main = torch.cuda.current_stream(dev)
    pack_s = eng.pack_stream if getattr(eng, "pack_stream", None) is not None else torch.cuda.Stream(device=dev)
    eng.pack_stream = pack_s
    if os.environ.get("T2S_NO_SIDE_STREAM") or os.environ.get("T2S_PACK_OVERLAP") == "0":
        pack_s = main
    pack_events = []
    log_det = torch.empty(m.n_flows, dtype=torch.float32, device=dev)
    Ws = [_f32c(m.convinv[k].conv.weight) for k in range(m.n_flows)]
    pack_s.wait_stream(main)
    with torch.cuda.stream(pack_s):
        eng.pack_weights(dev, force=True, flow_events=pack_events, res_pair8=ts.fold_train)
        stp = _lib.current_stream()
        if m.n_flows <= 16:         # B*L*logdet(W_k) and W_k^-1 of every flow in one launch (the table travels as a kernel argument)
            jobs = torch.tensor([[Ws[k].data_ptr(), log_det.data_ptr() + 4 * k, ts.Winv[k].data_ptr(), eng._flow_geom(k)[1]]
                                 for k in range(m.n_flows)], dtype=torch.int64)
            _lib.call("t2s_small_logdet_inv_batch_host", ctypes.c_void_p(jobs.data_ptr()), m.n_flows, float(B * L), stp)
            ts.keep_jobs = jobs
        else:
            for k in range(m.n_flows):
                _lib.call("t2s_small_logdet_inv", _ptr(Ws[k]), eng._flow_geom(k)[1], float(B * L),
                          _lib.c_vp(log_det.data_ptr() + 4 * k), _ptr(ts.Winv[k]), stp)
    pair8 = 1 if (ts.fold_train and eng.packed.get("res_pair8")) else 0
    st = _lib.current_stream()
    w = dict(Lp=ts.Lp, Sh=ts.S_planes[0], Sl=ts.S_planes[1])
    eng._upsample(mel, B, L, w)
    audio32 = _f32c(audio)
    z = torch.empty(B, G, L, dtype=torch.float32, device=dev)
    _lib.call("t2s_wg_audio_squeeze", _ptr(audio32), _ptr(z), B, T, G, L, 0, st)
    log_s_list = []
    keep = [audio32] + Ws
    for k in range(m.n_flows):
        c_off, n_rem, n_half = eng._flow_geom(k)
        fl = eng.packed["flows"][k]
        Wk = Ws[k]
        main.wait_event(pack_events[k])
        _lib.call("t2s_wg_convinv", _ptr(z), _ptr(Wk), B, G, c_off, n_rem, L, st)
        wn = m.WN[k]
        b_start = _f32c(wn.start.bias)
        keep.append(b_start)
        lay = ts.layers[k]
        _lib.call("t2s_wg_start", _ptr(z), _ptr(fl["w_start"]), _ptr(b_start), B, G, c_off, n_half, C, L, ts.Lp, g["halo"],
                  _ptr(lay[0]["X"][0]), _ptr(lay[0]["X"][1]), st)
        for i in range(nl):
            ly, sv = fl["layers"][i], lay[i]
            last = i == nl - 1
            a_h, a_l = _act_ptrs(ts, k, i, "A")
            g_h, g_l = _act_ptrs(ts, k, i, "G")
            if ts.fold_train:
                _lib.call("t2s_wg_in_cond_gate_fold_train", _ptr(ly["A1h"]), _ptr(ly["A1l"]), _ptr(ly["b1"]), _ptr(sv["X"][0]),
                          _ptr(sv["X"][1]), _ptr(ts.S_planes[0]), _ptr(ts.S_planes[1]), a_h, a_l, g_h, g_l, ts.act_bchunks,
                          _ptr(ly["fold_A"]), _ptr(ts.fold_acc), 1 if i == 0 else 0, B, C, g["n_cond"], ks, 2 ** i, L, ts.Lp,
                          g["halo"], g["Mpad1"], st)
                if not last:        # the last layer has no residual half; its skip half lives in the fold
                    nxt = lay[i + 1]["X"]
                    _lib.call("t2s_wg_res_only_train", _ptr(ly["A2h"]), _ptr(ly["A2l"]), _ptr(ly["b2"]), a_h, a_l, ts.act_bchunks,
                              _ptr(sv["X"][0]), _ptr(sv["X"][1]), _ptr(nxt[0]), _ptr(nxt[1]), B, C, L, ts.Lp, g["halo"],
                              ly["Mpad2"], pair8, st)
                continue
            _lib.call("t2s_wg_in_cond_gate_train", _ptr(ly["A1h"]), _ptr(ly["A1l"]), _ptr(ly["b1"]), _ptr(sv["X"][0]),
                      _ptr(sv["X"][1]), _ptr(ts.S_planes[0]), _ptr(ts.S_planes[1]), a_h, a_l,
                      None, None, g_h, g_l, B, C, g["n_cond"], ks, 2 ** i,
                      L, ts.Lp, g["halo"], g["Mpad1"], st)
            nxt = None if last else lay[i + 1]["X"]
            _lib.call("t2s_wg_res_skip_train", _ptr(ly["A2h"]), _ptr(ly["A2l"]), _ptr(ly["b2"]), a_h, a_l,
                      None if last else _ptr(sv["X"][0]), None if last else _ptr(sv["X"][1]),
                      None if last else _ptr(nxt[0]), None if last else _ptr(nxt[1]), _ptr(ts.skip[k]), B, C,
                      0 if last else C, 1 if i == 0 else 0, L, ts.Lp, g["halo"], ly["Mpad2"], st)
        # the backward's transposed operands of this flow (weights only; per-row scales from this flow's pack): pack stream
        with torch.cuda.stream(pack_s):
            stp = _lib.current_stream()
            wn_k = m.WN[k]
            for i in range(nl):
                last_i = i == nl - 1
                rows2 = C if last_i else 2 * C
                pk = fl["layers"][i]
                v_rs = _f32c(_vg(wn_k.res_skip_layers[i])[0])
                v_in = _f32c(_vg(wn_k.in_layers[i])[0])
                v_c = _f32c(_vg(wn_k.cond_layers[i])[0])
                keep.extend([v_rs, v_in, v_c])
                # (PERM_PAIR8 row order where the backward GEMM that consumes the operand runs on 256-row ping-pong tiles: 16-byte
                # epilogue pieces, t2s_wg_bwd_pair8_ok; the flags are kept in ts for the calls of the backward)
                _lib.call("t2s_pack_transposed", _ptr(v_rs), _ptr(pk["s_rs"]), rows2, C, 1, 0, rows2, ts.Mc, 0,
                          _ptr(ts.A_rsT[k][i][0]), _ptr(ts.A_rsT[k][i][1]), ts.p8_x, stp)
                _lib.call("t2s_pack_transposed", _ptr(v_in), _ptr(pk["s_in"]), 2 * C, C, ks, 1, 2 * C, ts.Mc, 0,
                          _ptr(ts.A_inT[k][i][0]), _ptr(ts.A_inT[k][i][1]), ts.p8_x, stp)
                _lib.call("t2s_pack_transposed", _ptr(v_c), _ptr(pk["s_cond"]), 2 * C, g["n_cond"], 1, 0, 2 * C, ts.Ms, i * 2 * C,
                          _ptr(ts.A_cT[k][0]), _ptr(ts.A_cT[k][1]), ts.p8_c, stp)
        log_s = torch.empty(B, n_half, L, dtype=torch.float32, device=dev)
        ts.wn_out[k] = torch.empty(B, 2 * n_half, L, dtype=torch.float32, device=dev)
        if ts.fold_train:
            eng._end(k, z, log_s, B, L, dict(Lp=ts.Lp, fold_acc=ts.fold_acc, wn_out=ts.wn_out[k]), c_off, n_half, reverse=False)
        else:
            eng._end(k, z, log_s, B, L, dict(Lp=ts.Lp), c_off, n_half, reverse=False, wn_out=ts.wn_out[k], skip=ts.skip[k])
        log_s_list.append(log_s)
    main.wait_stream(pack_s)            # log-determinants, inverses, transposed operands: all in before the outputs are used
    ts.z_final = z
    ts.mel = _f32c(mel)
    ts.B, ts.L = B, L
    ts.keep = keep
    return z, log_s_list, log_det, ts


def backward_train(eng, ts, gz, g_log_s, g_log_det):
    """Gradients of every parameter, flows and layers walked in reverse.  Returns {id(param): grad}."""
    m = eng.m
    g = eng.geom()
    C, nl, ks, n_cond, halo = g["C"], g["nl"], g["ks"], g["n_cond"], g["halo"]
    B, L, Lp, nt = ts.B, ts.L, ts.Lp, ts.nt
    G = m.n_group
    dev = ts.z_final.device
    st = _lib.current_stream()
    grads = {}
    zb = ts.zero_bias
    # working copies (both are updated in place flow by flow), made by t2s_add3: no eager operator inside the step
    dz = torch.empty(B, G, L, dtype=torch.float32, device=dev)
    if gz is None:
        _lib.call("t2s_scale_by_scalar", _ptr(ts.z_final), dz.numel(), _ptr(ts.zero_bias), 0.0, _ptr(dz), st)
    else:
        gz = gz.to(torch.float32).contiguous()
        _lib.call("t2s_add3", _ptr(gz), None, None, dz.numel(), _ptr(dz), st)
    zw = torch.empty_like(ts.z_final)
    _lib.call("t2s_add3", _ptr(ts.z_final), None, None, zw.numel(), _ptr(zw), st)
    # upstream gradients of the n_flows log_det_W outputs: 0-dim device tensors (or None)
    g_log_det = [None if t is None else t.detach().to(torch.float32).contiguous() for t in g_log_det]
    xc, sc = g["Cpad"] // 32, g["Spad"] // 32
    # Two streams.  Per layer the weight-gradient work (seven HBM-bound plane transposes, two weight-gradient GEMMs, three
    # weight-norm reductions) never feeds the data-gradient chain (gate backward -> W_in^T / W_cond^T accumulate), so it runs
    # on a side stream; events guard the d-plane buffers the two share (DX / DS: read by the weight-gradient GEMM before the
    # chain updates DX in place; every layer has its own slice of the flow-wide d_pre planes, which the next flow reuses only
    # after the two streams have joined).  The split-K slabs belong to the side stream alone.
    main_s = torch.cuda.current_stream(dev)
    two = not os.environ.get("T2S_WG_BWD_ONE_STREAM")
    side_s = getattr(eng, "bwd_side_stream", None)
    if two and side_s is None:
        # T2S_WG_SIDE_PRIO=-1: the weight-gradient stream at high priority (it is the longer of the two chains)
        side_s = eng.bwd_side_stream = torch.cuda.Stream(device=dev, priority=int(os.environ.get("T2S_WG_SIDE_PRIO", "0")))
    if not two:
        side_s = main_s
    st2 = _lib.c_vp(side_s.cuda_stream)
    side_s.wait_stream(main_s)
    cl = ts.cl_ok       # weight-gradient GEMMs straight from the channel-last planes (no time-major copies)
    cl_third_ok = cl
    if not cl:
        # conditioning rows + the ones row of the in/cond weight-gradient GEMM are the same for every layer
        _lib.call("t2s_plane_transpose", _ptr(ts.S_planes[0]), _ptr(ts.S_planes[1]), B, sc, sc, Lp, 0, _ptr(ts.TM_x[0]),
                  _ptr(ts.TM_x[1]), ts.N2pad, ks * C, st2)
        _lib.call("t2s_tm_ones_row", _ptr(ts.TM_x[0]), _ptr(ts.TM_x[1]), B, Lp, halo, L, ts.N2pad, ts.N2 - 1, st2)
        _lib.call("t2s_tm_ones_row", _ptr(ts.TM_act[0]), _ptr(ts.TM_act[1]), B, Lp, halo, L, ts.N1pad, C, st2)
    dsp_init = 1
    side_done = {}                          # flow -> event: the weight-gradient stream has finished that flow
    per_layer_cond = bool(os.environ.get("T2S_WCOND_PER_LAYER"))
    keep = []

    def new(*shape):
        return torch.empty(*shape, dtype=torch.float32, device=dev)

    # Parameter gradients of one flow live in one flat bucket, handed to RCCL as soon as the flow is done
    # (text2speech_amd/distributed.py) while the next flow's backward kernels run.
    sync = getattr(eng, "grad_sync", None)

    class _Bucket:
        def __init__(self, params):
            params = list(params)
            self.flat = torch.empty(sum(p.numel() for p in params) + 4 * len(params), dtype=torch.float32, device=dev)
            self.off = 0

        def take(self, *shape):
            n = 1
            for d_ in shape:
                n *= int(d_)
            v = self.flat[self.off:self.off + n].view(*shape)
            self.off += -(-n // 4) * 4
            return v

        def ship(self, events=()):
            """events: what must have completed before the bucket is whole (one per stream that wrote gradients into it)."""
            if sync is not None:
                sync.reduce_async(self.flat[:self.off], ready_events=events)

    bucket = None

    def wn_grads(conv, P, nsplit, Prows, Pcols, col_off, tap_stride, col_bias, O, Cin, Kt, with_bias=True, stream=None, nb=1):
        v, gg = _vg(conv)
        v32 = _f32c(v)
        g32 = None if gg is None else _f32c(gg)
        dv = bucket.take(*v.shape)
        dg = None if gg is None else bucket.take(*gg.shape)
        db = bucket.take(O) if with_bias else None
        keep.extend([v32, g32])
        _lib.call("t2s_wn_backward", _ptr(P), nsplit, Prows, Pcols, 0, col_off, tap_stride, col_bias, nb, _ptr(v32), _ptr(g32), O,
                  Cin, Kt, _ptr(dv), _ptr(dg), _ptr(db), 0, st if stream is None else stream)
        if gg is None:
            grads[id(conv.weight)] = dv
        else:
            grads[id(conv.weight_v)] = dv
            grads[id(conv.weight_g)] = dg
        if with_bias:
            grads[id(conv.bias)] = db

    def scale_of(conv, saved):
        """(v, per-row scale g/|v| that this step's forward pack applied and kept)."""
        v32 = _f32c(_vg(conv)[0])
        keep.append(v32)
        return v32, saved

    # T2S_WG_P1_MAIN=0: the res/skip weight gradient on the weight-gradient stream (the round-3 start)
    p1_main = two and os.environ.get("T2S_WG_P1_MAIN", "1") != "0"
    p1_count, p1_free = [0], [None, None]
    # T2S_WG_P1_THIRD=1 (experimental): the res/skip weight gradient on a THIRD stream, started beside the gate backward - both
    # are 128-workgroup launches that only READ DX / DS, so together they fill the chip; the chain waits for it only in front of
    # W_in^T's in-place update of DX
    p1_third = p1_main and cl_third_ok and os.environ.get("T2S_WG_P1_THIRD", "0") == "1"
    p1_s = None
    if p1_third:
        p1_s = getattr(eng, "bwd_p1_stream", None)
        if p1_s is None:
            p1_s = eng.bwd_p1_stream = torch.cuda.Stream(device=dev)
        p1_s.wait_stream(main_s)
    st3 = _lib.c_vp(p1_s.cuda_stream) if p1_third else None
    ev_p1_done = None
    wcond_side = two and cl and not per_layer_cond and os.environ.get("T2S_WG_WCOND_SIDE", "1") == "1"
    for k in reversed(range(m.n_flows)):
        c_off, n_rem, n_half = eng._flow_geom(k)
        wn = m.WN[k]
        fl = eng.packed["flows"][k]
        lay = ts.layers[k]
        nj = 2 * n_half
        bucket = _Bucket(list(wn.parameters()) + list(m.convinv[k].parameters()))
        DS, DP = ts.DS2[k & 1], ts.DP2[k & 1]
        if side_done.get(k + 2) is not None:
            main_s.wait_event(side_done[k + 2])     # the weight-gradient stream has finished with this parity's DS / d_pre planes
        # ---- affine coupling backward, un-apply (a1 restored in zw) ----
        d_out = new(B, nj, L)
        gls, gls_scalar = g_log_s[k], 0
        if gls is not None:
            if all(s_ == 0 for s_ in gls.stride()) and gls.dtype == torch.float32:
                gls_scalar = 1          # a broadcast scalar (WaveGlowLoss): ONE float stands for every element, nothing to expand
            else:
                gls = gls.to(torch.float32).contiguous()
        keep.append(gls)
        _lib.call("t2s_wg_affine_backward", _ptr(zw), _ptr(dz), _ptr(ts.wn_out[k]), _ptr(gls), gls_scalar, _ptr(d_out), B, G,
                  c_off, n_half, L, st)
        # ---- WN.end: weight / bias gradient, and d_skip = W_end^T d_out as planes ----
        w_end = _f32c(wn.end.weight)
        dW_end = bucket.take(*wn.end.weight.shape)
        db_end = bucket.take(nj)
        if ts.fold_train:
            # The forward never formed the skip sum (WN.end is folded into the gate GEMM there).  WN.end's weight gradient needs
            # it once: skip = sum_i (W_skip,i acts_i + b_skip,i) as ONE GEMM over the flow-wide gate-output planes, K = nl * C,
            # with the skip rows of the nl res_skip weights packed one after the other along K.  On the weight-gradient stream.
            ev_dout = torch.cuda.Event()
            ev_dout.record(main_s)
            side_s.wait_event(ev_dout)
            for i in range(nl):
                conv = wn.res_skip_layers[i]
                v, gg = _vg(conv)
                v32, g32, b32 = _f32c(v), (None if gg is None else _f32c(gg)), _f32c(conv.bias)
                keep.extend([v32, g32, b32])
                r0 = C if i < nl - 1 else 0                      # skip rows: the second half, or all rows of the last layer
                _lib.call("t2s_pack_conv_weight", _lib.c_vp(v32.data_ptr() + 4 * r0 * C),
                          None if g32 is None else _lib.c_vp(g32.data_ptr() + 4 * r0), 0, _lib.c_vp(b32.data_ptr() + 4 * r0), C, C, 1,
                          0, 0, 0, ts.Mskip, i * g["Cpad"], g["Cpad"], _ptr(ts.A_skip[0]), _ptr(ts.A_skip[1]), _ptr(ts.bias_skip),
                          1 if i else 0, st2)
            _lib.call("t2s_wg_skip_sum", _ptr(ts.A_skip[0]), _ptr(ts.A_skip[1]), _ptr(ts.bias_skip), _ptr(ts.AF[k][0]),
                      _ptr(ts.AF[k][1]), nl * xc, ts.act_bchunks, _ptr(ts.skip[k]), B, C, L, Lp, halo, ts.Mskip, st2)
            _lib.call("t2s_small_wgrad", None, None, _ptr(ts.skip[k]), _ptr(d_out), _ptr(dW_end), None, _ptr(ts.sw_scratch_side), B, xc,
                      Lp, halo, L, C, nj, nj, 0, 1, st2)
        else:
            _lib.call("t2s_small_wgrad", None, None, _ptr(ts.skip[k]), _ptr(d_out), _ptr(dW_end), None, _ptr(ts.sw_scratch), B, xc, Lp,
                      halo, L, C, nj, nj, 0, 1, st)
        _lib.call("t2s_rows_sum", _ptr(d_out), B, nj, 0, nj, L, _ptr(db_end), st)
        grads[id(wn.end.weight)] = dW_end
        grads[id(wn.end.bias)] = db_end
        w_endT = new(C, nj)
        _lib.call("t2s_transpose", _ptr(w_end), _ptr(w_endT), nj, C, st)
        keep.extend([w_end, w_endT])
        _lib.call("t2s_wg_start", _ptr(d_out), _ptr(w_endT), _ptr(zb), B, nj, 0, nj, C, L, Lp, halo, _ptr(DS[0]),
                  _ptr(DS[1]), st)
        # d_skip rows of every layer's d_rs are the same: transpose once per flow
        for i in reversed(range(nl)):
            last = i == nl - 1
            sv = lay[i]
            rows2 = C if last else 2 * C
            Mrs = _lib.padded_rows(rows2)
            conv_rs, conv_in, conv_c = wn.res_skip_layers[i], wn.in_layers[i], wn.cond_layers[i]
            # 1. d_pre = gate'(T,G) * (W_rs^T [dx ; dS])                                                     [main]
            A_rsT, A_inT, A_cT = ts.A_rsT[k][i], ts.A_inT[k][i], ts.A_cT[k]
            # (one event per layer on the data-gradient stream when the res/skip weight gradient runs there: each record is a ~7 us
            # bubble in front of the next kernel of the recording stream; T2S_WG_FEW_EVENTS=0: one per consumer as before)
            few_ev = cl and p1_main and os.environ.get("T2S_WG_FEW_EVENTS", "1") != "0"
            ev_in = None
            if not few_ev:
                ev_in = torch.cuda.Event()      # DX / DS of this layer are final (last written on the main stream)
                ev_in.record(main_s)
            # this layer's slice of the flow-wide d_pre planes (bytes from the start of each plane)
            dp_off = 2 * i * 2 * xc * Lp * 32
            dp_h, dp_l = _lib.c_vp(DP[0].data_ptr() + dp_off), _lib.c_vp(DP[1].data_ptr() + dp_off)
            a_h, a_l = _act_ptrs(ts, k, i, "A")
            g_h, g_l = _act_ptrs(ts, k, i, "G")
            ev_dx = None
            if p1_third:
                ev_dx = torch.cuda.Event()          # DX / DS of this layer are final: the third stream's res/skip weight gradient may read them
                ev_dx.record(main_s)
            _lib.call("t2s_wg_bwd_gate_dgrad", _ptr(A_rsT[0]), _ptr(A_rsT[1]), _ptr(zb),
                      None if last else _ptr(ts.DX[0]), None if last else _ptr(ts.DX[1]), _ptr(DS[0]), _ptr(DS[1]),
                      a_h, a_l, g_h, g_l, ts.act_bchunks, dp_h, dp_l, ts.dp_chunks, B, C, L, Lp, halo, ts.Mc, ts.p8_x, st)
            ev_dp = None
            if not few_ev:
                ev_dp = torch.cuda.Event()
                ev_dp.record(main_s)
            # 2. dW_rs = [dx ; dS] . acts^T  (+ bias column)                                                 [side, or main]
            d = 2 ** i
            P1 = ts.P1
            ev_tdrs = None
            if cl and p1_third:
                # third stream: needs DX / DS final (the record sits in front of the gate backward on the main stream) and a free slab set
                ta1, tb1, ta2, tb2 = _cl_tables(ts, (k, i), sv, last, xc, sc, ks, d, dev, i)
                ks1 = ts.ks1_last if last else ts.ks1
                pb = p1_count[0] & 1
                p1_count[0] += 1
                P1 = ts.P1b if pb else ts.P1
                p1_s.wait_event(ev_dx)
                if p1_free[pb] is not None:
                    p1_s.wait_event(p1_free[pb])
                _lib.call("t2s_wgrad_cl", _ptr(ta1), ta1.size(0), _ptr(tb1), tb1.size(0), _ptr(P1), B, rows2, ts.N1g, ts.ld1,
                          ts.k0, ts.k1, ks1, ts.bias_cols, st3)
                ev_p1_done = torch.cuda.Event()
                ev_p1_done.record(p1_s)
                side_s.wait_event(ev_p1_done)
                ev_p1 = torch.cuda.Event()          # d_pre of this layer is final (gate backward): the weight-gradient stream's cue
                ev_p1.record(main_s)
                side_s.wait_event(ev_p1)
            elif cl and p1_main:
                # On the data-gradient stream, between the two GEMMs that bracket it there: the chain then never waits for the
                # weight-gradient stream before it updates DX (that wait was 135 us of the 463 us a layer took,
                # profiles/r03_wg_train_timeline_before.md), and 74 us of half-chip work leave the longer stream.  Its slab
                # reduction stays on the side stream; two slab sets, so the chain only waits for the reduction of two layers ago.
                ta1, tb1, ta2, tb2 = _cl_tables(ts, (k, i), sv, last, xc, sc, ks, d, dev, i)
                ks1 = ts.ks1_last if last else ts.ks1
                pb = p1_count[0] & 1
                p1_count[0] += 1
                P1 = ts.P1b if pb else ts.P1
                if p1_free[pb] is not None:
                    main_s.wait_event(p1_free[pb])
                _lib.call("t2s_wgrad_cl", _ptr(ta1), ta1.size(0), _ptr(tb1), tb1.size(0), _ptr(P1), B, rows2, ts.N1g, ts.ld1,
                          ts.k0, ts.k1, ks1, ts.bias_cols, st)
                ev_p1 = torch.cuda.Event()
                ev_p1.record(main_s)
                side_s.wait_event(ev_p1)
            elif cl:
                side_s.wait_event(ev_in)
                ta1, tb1, ta2, tb2 = _cl_tables(ts, (k, i), sv, last, xc, sc, ks, d, dev, i)
                ks1 = ts.ks1_last if last else ts.ks1
                _lib.call("t2s_wgrad_cl", _ptr(ta1), ta1.size(0), _ptr(tb1), tb1.size(0), _ptr(ts.P1), B, rows2, ts.N1g, ts.ld1,
                          ts.k0, ts.k1, ks1, ts.bias_cols, st2)
                ev_tdrs = torch.cuda.Event()        # DX / DS have been read: the chain may update DX in place
                ev_tdrs.record(side_s)
            else:
                side_s.wait_event(ev_in)
                if not last:
                    _lib.call("t2s_plane_transpose", _ptr(ts.DX[0]), _ptr(ts.DX[1]), B, xc, xc, Lp, 0, _ptr(ts.TM_drs[0]),
                              _ptr(ts.TM_drs[1]), Mrs, 0, st2)
                _lib.call("t2s_plane_transpose", _ptr(DS[0]), _ptr(DS[1]), B, xc, xc, Lp, 0, _ptr(ts.TM_drs[0]),
                          _ptr(ts.TM_drs[1]), Mrs, 0 if last else C, st2)
                ev_tdrs = torch.cuda.Event()        # DX / DS have been read: the chain may update DX in place
                ev_tdrs.record(side_s)
                _lib.call("t2s_plane_transpose", a_h, a_l, B, ts.act_bchunks or xc, xc, Lp, 0, _ptr(ts.TM_act[0]),
                          _ptr(ts.TM_act[1]), ts.N1pad, 0, st2)
                _lib.call("t2s_wgrad_gemm_flat", _ptr(ts.TM_drs[0]), _ptr(ts.TM_drs[1]), _ptr(ts.TM_act[0]), _ptr(ts.TM_act[1]),
                          _ptr(zb), _ptr(ts.P1), B, rows2, ts.N1, Mrs, ts.N1pad, nt, ts.k0, ts.k1, ts.ks1, st2)
            wn_grads(conv_rs, P1, (ts.ks1_last if last else ts.ks1) if cl else ts.ks1, rows2, ts.ld1, 0, 0, C, rows2, C, 1,
                     stream=st2, nb=4 if ts.bias_cols and cl else 1)
            if cl and p1_main:
                p1_free[pb] = torch.cuda.Event()
                p1_free[pb].record(side_s)
            # 3. dW_in, dW_cond = d_pre . [x taps | spect | 1]^T                                             [side]
            if ev_dp is not None:
                side_s.wait_event(ev_dp)        # (few_ev: the wait for ev_p1 above covers d_pre too - recorded later on the same stream)
            if cl:
                _lib.call("t2s_wgrad_cl", _ptr(ta2), ta2.size(0), _ptr(tb2), tb2.size(0), _ptr(ts.P2), B, 2 * C, ts.N2, ts.ld2,
                          ts.k0, ts.k1, ts.ks2, 0, st2)
            else:
                _lib.call("t2s_plane_transpose", dp_h, dp_l, B, ts.dp_chunks, 2 * xc, Lp, 0, _ptr(ts.TM_dp[0]),
                          _ptr(ts.TM_dp[1]), ts.M2pad, 0, st2)
                for tap in range(ks):
                    _lib.call("t2s_plane_transpose", _ptr(sv["X"][0]), _ptr(sv["X"][1]), B, xc, xc, Lp, (tap - ks // 2) * d,
                              _ptr(ts.TM_x[0]), _ptr(ts.TM_x[1]), ts.N2pad, tap * C, st2)
                _lib.call("t2s_wgrad_gemm_flat", _ptr(ts.TM_dp[0]), _ptr(ts.TM_dp[1]), _ptr(ts.TM_x[0]), _ptr(ts.TM_x[1]),
                          _ptr(zb), _ptr(ts.P2), B, 2 * C, ts.N2, ts.M2pad, ts.N2pad, nt, ts.k0, ts.k1, ts.ks2, st2)
            wn_grads(conv_in, ts.P2, ts.ks2, 2 * C, ts.ld2, 0, C, ts.N2 - 1, 2 * C, C, ks, stream=st2)
            wn_grads(conv_c, ts.P2, ts.ks2, 2 * C, ts.ld2, ks * C, 0, ts.N2 - 1, 2 * C, n_cond, 1, stream=st2)
            # 4. dx (+)= W_in^T (*) d_pre ;  d_spect += W_cond^T d_pre                                       [main]
            if ev_tdrs is not None:
                main_s.wait_event(ev_tdrs)  # (events of one stream complete in order: this covers every earlier read of DX too)
            if p1_third and ev_p1_done is not None:
                main_s.wait_event(ev_p1_done)       # the third stream has read DX: W_in^T may update it in place
            _lib.call("t2s_conv_accumulate", _ptr(A_inT[0]), _ptr(A_inT[1]), _ptr(zb), dp_h, dp_l, ts.dp_chunks,
                      _ptr(ts.DX[0]), _ptr(ts.DX[1]), B, 2 * C, C, ks, d, 1 if last else 0, L, Lp, halo, ts.Mc, ts.p8_x, st)
            # (W_cond,i^T sits in K-chunks [2 xc i, 2 xc (i + 1)) of the flow's conditioning-gradient operand A_cT)
            if per_layer_cond:          # A/B switch (T2S_WCOND_PER_LAYER=1): the round-2 form, one accumulate per layer
                a_off = 2 * i * 2 * xc * ts.Ms * 32
                _lib.call("t2s_conv_accumulate", _lib.c_vp(A_cT[0].data_ptr() + a_off), _lib.c_vp(A_cT[1].data_ptr() + a_off),
                          _ptr(zb), dp_h, dp_l, ts.dp_chunks, _ptr(ts.DSp[0]), _ptr(ts.DSp[1]), B, 2 * C, n_cond, 1, 1, dsp_init,
                          L, Lp, halo, ts.Ms, ts.p8_c, st)
                dsp_init = 0
        # d_spect (+)= [W_cond,0^T | ... | W_cond,nl-1^T] [d_pre_0 ; ... ; d_pre_nl-1]: one GEMM per flow, K = nl * 2C       [main]
        if not per_layer_cond:
            # (T2S_WG_WCOND_SIDE=1: on the weight-gradient stream - it feeds the upsampler's gradient only, not the chain; that
            # stream has waited for every d_pre of the flow by now, and it owns the d_pre planes' reuse through side_done)
            _lib.call("t2s_conv_accumulate", _ptr(ts.A_cT[k][0]), _ptr(ts.A_cT[k][1]), _ptr(zb), _ptr(DP[0]), _ptr(DP[1]), 0,
                      _ptr(ts.DSp[0]), _ptr(ts.DSp[1]), B, nl * 2 * C, n_cond, 1, 1, dsp_init, L, Lp, halo, ts.Ms, ts.p8_c,
                      st2 if wcond_side else st)
            dsp_init = 0
        # ---- WN.start ----
        dW_eff = new(C, n_half)
        db_start = bucket.take(C)
        _lib.call("t2s_small_wgrad", _ptr(ts.DX[0]), _ptr(ts.DX[1]), None, _ptr(zw), _ptr(dW_eff), _ptr(db_start),
                  _ptr(ts.sw_scratch), B, xc, Lp,
                  halo, L, C, n_half, G, c_off, 0, st)
        wn_grads(wn.start, dW_eff, 1, C, n_half, 0, 0, 0, C, n_half, 1, with_bias=False)
        grads[id(wn.start.bias)] = db_start
        _lib.call("t2s_wg_start_dgrad", _ptr(ts.DX[0]), _ptr(ts.DX[1]), _ptr(fl["w_start"]), _ptr(dz), B, G, c_off, n_half, C,
                  L, Lp, halo, st)
        # ---- invertible 1x1 conv ----
        Wk = _f32c(m.convinv[k].conv.weight)
        Winv = ts.Winv[k]                   # from the forward's batched launch
        WT = new(n_rem, n_rem)
        _lib.call("t2s_transpose", _ptr(Wk), _ptr(WT), n_rem, n_rem, st)
        _lib.call("t2s_wg_convinv", _ptr(zw), _ptr(Winv), B, G, c_off, n_rem, L, st)          # zw <- flow input
        dW = bucket.take(*m.convinv[k].conv.weight.shape)
        gp = _ptr(g_log_det[k])
        _lib.call("t2s_wg_convinv_wgrad", _ptr(dz), _ptr(zw), _ptr(Winv), gp, float(B * L), B, G, c_off, n_rem, L, _ptr(dW), st)
        _lib.call("t2s_wg_convinv", _ptr(dz), _ptr(WT), B, G, c_off, n_rem, L, st)            # dz <- W^T dz
        grads[id(m.convinv[k].conv.weight)] = dW
        keep.extend([Wk, Winv, WT, d_out])
        # this flow's gradients are complete once both streams get here; the bucket ships from the communication stream behind
        # these two events, and the data-gradient chain goes straight on to the next flow (no join of the two streams)
        ev_side = torch.cuda.Event()
        ev_side.record(side_s)
        side_done[k] = ev_side
        ev_main = torch.cuda.Event()
        ev_main.record(main_s)
        bucket.ship((ev_main, ev_side))
    # ---- upsampler ----
    if wcond_side:
        main_s.wait_stream(side_s)          # d_spect is complete on the weight-gradient stream
    up = m.upsample
    bucket = _Bucket(up.parameters())
    dW_up = bucket.take(*up.weight.shape)
    db_up = bucket.take(up.out_channels)
    _lib.call("t2s_wg_upsample_wgrad", _ptr(ts.DSp[0]), _ptr(ts.DSp[1]), _ptr(ts.mel), B, up.in_channels, ts.mel.size(2),
              up.kernel_size[0], up.stride[0], G, L, Lp, halo, _ptr(dW_up), _ptr(db_up), st)
    grads[id(up.weight)] = dW_up
    grads[id(up.bias)] = db_up
    ev_main = torch.cuda.Event()
    ev_main.record(main_s)
    bucket.ship((ev_main,))
    main_s.wait_stream(side_s)              # every gradient has been written before autograd hands them on
    if sync is not None:
        sync.finish()
    ts.keep_bwd = keep
    return grads


class _WaveGlowFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, mel, audio, *params):
        eng = model._eng()
        with torch.no_grad():
            z, log_s_list, log_det, ts = forward_train(eng, mel, audio)
        ctx.model = model
        ctx.ts = ts
        ctx.n = len(log_s_list)
        ctx.params = params
        # log_det_W as n_flows 0-dim views of one device array: an indexing node per flow on the caller's side would cost a
        # zero-fill, a scatter and an add per flow in autograd's backward
        return (z, *log_s_list, *log_det.unbind(0))

    @staticmethod
    def backward(ctx, gz, *rest):
        n = ctx.n
        g_log_s = list(rest[:n])
        g_log_det = list(rest[n:2 * n])
        eng = ctx.model._eng()
        with torch.no_grad():
            grads = backward_train(eng, ctx.ts, gz, g_log_s, g_log_det)
        out = []
        for p in ctx.params:
            gr = grads.get(id(p))
            out.append(None if gr is None else gr.view_as(p).to(p.dtype))
        return (None, None, None, *out)


def waveglow_forward_with_grad(model, mel, audio):
    params = [p for p in model.parameters()]
    outs = _WaveGlowFn.apply(model, mel, audio, *params)
    n = model.n_flows
    return outs[0], list(outs[1:1 + n]), list(outs[1 + n:1 + 2 * n])

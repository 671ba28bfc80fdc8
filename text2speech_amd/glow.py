"""MI355X-native WaveGlow with the reference module API.

Drop-in for reference ``waveglow/glow.py``: same class names, constructor
signatures, attribute names (``upsample``, ``WN``, ``convinv``,
``n_remaining_channels`` ...) and ``state_dict`` keys, so checkpoints and the
reference's ``train.py`` / ``inference.py`` / ``denoiser.py`` call sites keep
working.  The math does not run through these ``nn.Module``
containers: ``forward`` / ``infer`` drive the hand-written gfx950 kernels of
``libt2s_hip.so`` through the C ABI (include/t2s_hip.h).  There is no CPU or
eager-PyTorch fallback: without the library, or off-GPU, the calls raise.

Reference lines are cited per method.
"""
import ctypes
import math

import os

import torch
import torch.nn.functional as F  # noqa: F401  (kept for API parity with the reference module)

from . import _lib


class _WaveGlowLossFn(torch.autograd.Function):
    """loss = (sum z^2 / (2 sigma^2) - sum_k sum log_s_k - sum_k log_det_k) / numel(z) in one fused HIP pass
    (csrc/loss_ops.hip); d/dz = z / (sigma^2 N) comes out of the same pass, d/dlog_s = d/dlog_det = -1/N.  No eager operator
    in forward or backward: the upstream gradient (a 0-dim device tensor) is applied by t2s_scale_by_scalar."""

    @staticmethod
    def forward(ctx, sigma, n_flows, z, *rest):
        log_s = [t.detach().to(torch.float32).contiguous() for t in rest[:n_flows]]
        dets = [t.detach() for t in rest[n_flows:]]
        # the model hands out log_det_W as the n_flows elements of ONE device array (views): use it in place
        d0 = dets[0]
        if all(t.dtype == torch.float32 and t.dim() == 0 and t.untyped_storage().data_ptr() == d0.untyped_storage().data_ptr()
               and t.storage_offset() == d0.storage_offset() + k for k, t in enumerate(dets)):
            log_det = d0.as_strided((n_flows,), (1,))
        else:
            log_det = torch.stack([t.to(torch.float32).reshape(()) for t in dets])
        zc = z.detach().to(torch.float32).contiguous()
        dev = zc.device
        d_z = torch.empty_like(zc) if z.requires_grad else None
        partial = torch.empty(256 * 2, dtype=torch.float64, device=dev)
        out = torch.empty(1, dtype=torch.float32, device=dev)
        ptrs = (ctypes.c_void_p * n_flows)(*[t.data_ptr() for t in log_s])
        counts = (ctypes.c_size_t * n_flows)(*[t.numel() for t in log_s])
        _lib.call("t2s_waveglow_loss", _lib.ptr(zc), zc.numel(), ptrs, counts, n_flows, _lib.ptr(log_det), float(sigma),
                  _lib.ptr(d_z), _lib.ptr(partial), _lib.ptr(out), _lib.current_stream())
        ctx.d_z, ctx.n_flows, ctx.inv_n = d_z, n_flows, 1.0 / zc.numel()
        ctx.meta = [(t.shape, t.requires_grad) for t in rest]
        ctx.z_shape = z.shape
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        g32 = g.detach().to(torch.float32).contiguous()
        st = _lib.current_stream()
        gz = None
        if ctx.d_z is not None:
            gz = torch.empty_like(ctx.d_z)
            _lib.call("t2s_scale_by_scalar", _lib.ptr(ctx.d_z), gz.numel(), _lib.ptr(g32), 1.0, _lib.ptr(gz), st)
            gz = gz.view(ctx.z_shape)
        # the constant -1/N times the upstream gradient: ONE device float, broadcast to every log_s / log_det (views, no kernel)
        neg = torch.empty(1, dtype=torch.float32, device=g32.device)
        _lib.call("t2s_scale_by_scalar", None, 1, _lib.ptr(g32), -ctx.inv_n, _lib.ptr(neg), st)
        outs = [neg.view(()).expand(shape) if need else None for shape, need in ctx.meta]
        return (None, None, gz, *outs)


class WaveGlowLoss(torch.nn.Module):
    """Reference glow.py:43-59: scalar NLL of the flow output, computed by the fused HIP kernel t2s_waveglow_loss.  Like every
    other entry point of this build it needs tensors in HBM; there is no CPU path (tests use oracle.waveglow_oracle's loss)."""

    def __init__(self, sigma=1.0):
        super().__init__()
        self.sigma = sigma

    def forward(self, model_output):
        z, log_s_list, log_det_W_list = model_output
        if not z.is_cuda:
            raise _lib.T2SError("WaveGlowLoss (MI355X build) needs tensors in HBM; there is no CPU path")
        if len(log_s_list) > 16 or len(log_s_list) != len(log_det_W_list):
            raise ValueError("WaveGlowLoss: at most 16 flows, one log_det_W per log_s")
        return _WaveGlowLossFn.apply(float(self.sigma), len(log_s_list), z, *log_s_list, *log_det_W_list)


class Invertible1x1Conv(torch.nn.Module):
    """Parameter container for the invertible 1x1 convolution (reference
    glow.py:62-80: QR-orthonormal init with det forced to +1).  The conv itself
    and log|det W| are computed by ``t2s_wg_convinv`` / ``t2s_small_logdet_inv``."""

    def __init__(self, c):
        super().__init__()
        self.conv = torch.nn.Conv1d(c, c, kernel_size=1, stride=1, padding=0, bias=False)
        W = torch.linalg.qr(torch.randn(c, c))[0]
        if torch.det(W) < 0:
            W[:, 0] = -1 * W[:, 0]
        self.conv.weight.data = W.contiguous().view(c, c, 1)

    def forward(self, z, reverse=False):
        """Reference glow.py:82-102 for callers that use the module directly (WaveGlow.forward / infer drive the same kernels
        through the engine): z [B, c, T] -> (W z, B * T * log|det W|), or with reverse=True -> W^-1 z, the inverse computed on
        first use and cached as ``W_inverse`` exactly as the reference caches it."""
        if not z.is_cuda:
            raise _lib.T2SError("Invertible1x1Conv (MI355X build) needs tensors in HBM; there is no CPU path")
        B, c, T = z.shape
        st = _lib.current_stream()
        out = z.detach().to(torch.float32).contiguous().clone()
        W = _f32c(self.conv.weight).view(c, c)
        with torch.no_grad():
            if reverse:
                if not hasattr(self, "W_inverse"):
                    Winv = torch.empty(c, c, dtype=torch.float32, device=z.device)
                    _lib.call("t2s_small_logdet_inv", _lib.ptr(W), c, 1.0, None, _lib.ptr(Winv), st)
                    self.W_inverse = Winv.view(c, c, 1)
                Wi = self.W_inverse.detach().to(torch.float32).contiguous()
                _lib.call("t2s_wg_convinv", _lib.ptr(out), _lib.ptr(Wi), B, c, 0, c, T, st)
                return out.to(z.dtype)
            log_det = torch.empty(1, dtype=torch.float32, device=z.device)
            _lib.call("t2s_small_logdet_inv", _lib.ptr(W), c, float(B * T), _lib.ptr(log_det), None, st)
            _lib.call("t2s_wg_convinv", _lib.ptr(out), _lib.ptr(W), B, c, 0, c, T, st)
            return out.to(z.dtype), log_det[0]


class WN(torch.nn.Module):
    """Parameter container for the coupling network (reference glow.py:105-152):
    weight-normed ``start``, zero-initialised ``end``, and per layer a dilated
    ``in_layers[i]``, a 1x1 ``cond_layers[i]`` and a 1x1 ``res_skip_layers[i]``."""

    def __init__(self, n_in_channels, n_mel_channels, n_layers, n_channels, kernel_size):
        super().__init__()
        assert kernel_size % 2 == 1
        assert n_channels % 2 == 0
        self.n_layers = n_layers
        self.n_channels = n_channels
        self.kernel_size = kernel_size
        self.in_layers = torch.nn.ModuleList()
        self.res_skip_layers = torch.nn.ModuleList()
        self.cond_layers = torch.nn.ModuleList()
        wn = torch.nn.utils.weight_norm
        self.start = wn(torch.nn.Conv1d(n_in_channels, n_channels, 1), name="weight")
        end = torch.nn.Conv1d(n_channels, 2 * n_in_channels, 1)
        end.weight.data.zero_()
        end.bias.data.zero_()
        self.end = end
        for i in range(n_layers):
            dilation = 2 ** i
            padding = (kernel_size * dilation - dilation) // 2
            self.in_layers.append(wn(torch.nn.Conv1d(n_channels, 2 * n_channels, kernel_size,
                                                     dilation=dilation, padding=padding), name="weight"))
            self.cond_layers.append(wn(torch.nn.Conv1d(n_mel_channels, 2 * n_channels, 1), name="weight"))
            rs = 2 * n_channels if i < n_layers - 1 else n_channels
            self.res_skip_layers.append(wn(torch.nn.Conv1d(n_channels, rs, 1), name="weight"))

    def __getstate__(self):            # the back-reference to the owning WaveGlow (a weak reference) stays out of pickles / copies
        d = self.__dict__.copy()
        d.pop("_owner", None)
        return d

    def forward(self, forward_input):
        """Reference glow.py:154-175: (audio [B, n_in, L], spect [B, n_mel * n_group, L]) -> WN.end's output [B, 2 n_in, L]
        ((b ; log_s) of the coupling).  Runs on the engine of the WaveGlow this module belongs to (forward only)."""
        ref = self.__dict__.get("_owner")
        owner = ref() if ref is not None else None
        if owner is None:
            raise _lib.T2SError("WN.forward runs on the engine of the WaveGlow it belongs to; build it through WaveGlow(...)")
        audio, spect = forward_input
        with torch.no_grad():
            return owner._eng().wn_forward(self.__dict__["_flow"], audio, spect)


def _vg(conv):
    """(v, g) of a weight-normed conv, or (weight, None) after remove_weightnorm."""
    if hasattr(conv, "weight_v"):
        return conv.weight_v, conv.weight_g
    return conv.weight, None


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class _Engine:
    """Owns the HBM-resident state of one WaveGlow: packed (hi, lo) bf16 weight
    planes and the activation workspaces, and issues the kernel sequence."""

    def __init__(self, model):
        self.m = model
        self.packed = None
        self.packed_key = None
        self.ws = {}
        self.use_fold = True        # no-grad forward / infer: WN.end folded into the skip path (t2s_wg_*_fold)
        self.grad_sync = None       # distributed.GradSync: bucketed RCCL all-reduce issued from inside backward
        self.gemm_events = None     # bench.py: list of (start, end) torch.cuda.Event pairs around the gate GEMM
        self.gemm_event_stride = 1  # bench.py: time every n-th gate-GEMM launch (an event pair costs ~1.5 us of stream time)
        self._gemm_launch_no = 0

    # ------------------------------------------------------------------ geometry
    def geom(self):
        m = self.m
        wn0 = m.WN[0]
        C, nl, ks = wn0.n_channels, wn0.n_layers, wn0.kernel_size
        n_cond = m.upsample.out_channels * m.n_group
        # halo = the largest dilated tap offset, rounded up to a whole 32-row block: the channel-last weight-gradient GEMM
        # (csrc/wgrad_cl.hip) walks whole 32-row K-blocks starting at row `halo` and shifts them by up to +-halo rows, which stays
        # inside the plane only if halo % 32 == 0 (n_layers <= 5 gives 2^(nl-1) < 32)
        halo = -(-((2 ** (nl - 1)) * (ks // 2)) // 32) * 32
        g = dict(C=C, nl=nl, ks=ks, n_cond=n_cond, Cpad=-(-C // 32) * 32, Spad=-(-n_cond // 32) * 32,
                 halo=halo, Mpad1=-(-C // 128) * 256)
        g["nk1"] = ks * g["Cpad"] // 32 + g["Spad"] // 32
        g["nk2"] = g["Cpad"] // 32
        return g

    # ------------------------------------------------------------------ weights
    def pack_weights(self, device, force=True, flow_events=None, res_pair8=False):
        """res_pair8: pack the residual rows of every res/skip convolution in the 8-consecutive-channels order the folded no-grad
        path's residual GEMM wants (t2s_wg_res_only(pair8 = 1)); the training path keeps the identity order."""
        m = self.m
        res_pair8 = bool(res_pair8) and self.geom()["C"] % 32 == 0 and os.environ.get("T2S_RES_PAIR8") != "0"
        key = tuple(p._version for p in m.parameters()) + (str(device), res_pair8)
        if not force and self.packed is not None and self.packed_key == key:
            return self.packed
        g = self.geom()
        C, nl, ks = g["C"], g["nl"], g["ks"]
        st = _lib.current_stream()
        if self.packed is None or self.packed["device"] != device:
            flows = []
            for k in range(m.n_flows):
                n_half = m.WN[k].start.in_channels
                layers = []
                for i in range(nl):
                    rows2 = 2 * C if i < nl - 1 else C
                    Mpad2 = _lib.padded_rows(rows2)
                    layers.append(dict(
                        A1h=torch.zeros(g["nk1"], g["Mpad1"], 32, dtype=torch.bfloat16, device=device),
                        A1l=torch.zeros(g["nk1"], g["Mpad1"], 32, dtype=torch.bfloat16, device=device),
                        b1=torch.zeros(g["Mpad1"], dtype=torch.float32, device=device),
                        A2h=torch.zeros(g["nk2"], Mpad2, 32, dtype=torch.bfloat16, device=device),
                        A2l=torch.zeros(g["nk2"], Mpad2, 32, dtype=torch.bfloat16, device=device),
                        b2=torch.zeros(Mpad2, dtype=torch.float32, device=device), Mpad2=Mpad2,
                        s_in=torch.empty(2 * C, dtype=torch.float32, device=device),
                        s_cond=torch.empty(2 * C, dtype=torch.float32, device=device),
                        s_rs=torch.empty(rows2, dtype=torch.float32, device=device),
                        fold_A=torch.zeros(-(-C // 128) * 8192, dtype=torch.bfloat16, device=device)))
                flows.append(dict(layers=layers, n_half=n_half, bes=torch.zeros(nl, 8, dtype=torch.float32, device=device),
                                  w_start=torch.empty(C, n_half, dtype=torch.float32, device=device),
                                  w_inv=None))
            self.packed = dict(flows=flows, device=device)
        # One table-driven launch packs all 3 * n_layers * n_flows convolutions (weight-norm + split + permute).
        srcs = []       # f32 source tensors, in job order; their data_ptrs key the cached job table
        specs = []
        for k in range(m.n_flows):
            wn = m.WN[k]
            fl = self.packed["flows"][k]
            for i in range(nl):
                ly = fl["layers"][i]
                v, gg = _vg(wn.in_layers[i])
                vc, gc = _vg(wn.cond_layers[i])
                vr, gr = _vg(wn.res_skip_layers[i])
                t = [_f32c(v), None if gg is None else _f32c(gg), _f32c(wn.in_layers[i].bias), _f32c(wn.cond_layers[i].bias),
                     _f32c(vc), None if gc is None else _f32c(gc),
                     _f32c(vr), None if gr is None else _f32c(gr), _f32c(wn.res_skip_layers[i].bias)]
                srcs += t
                # (v, g, bias, bias2, A_hi, A_lo, bias_out, O, Cin, Kt, perm, C_gate, Mpad, koff, Cin_pad)
                specs.append((t[0], t[1], t[2], t[3], ly["A1h"], ly["A1l"], ly["b1"], 2 * C, C, ks, 1, C, g["Mpad1"], 0, g["Cpad"],
                              ly["s_in"]))
                specs.append((t[4], t[5], None, None, ly["A1h"], ly["A1l"], None, 2 * C, g["n_cond"], 1, 1, C, g["Mpad1"],
                              ks * g["Cpad"], g["Spad"], ly["s_cond"]))
                # residual rows (the first C of 2C; the last layer has none) optionally in the PERM_PAIR8 order
                p8 = res_pair8 and i < nl - 1
                specs.append((t[6], t[7], t[8], None, ly["A2h"], ly["A2l"], ly["b2"], t[6].size(0), C, 1, 2 if p8 else 0,
                              C if p8 else 0, ly["Mpad2"], 0, g["Cpad"], ly["s_rs"]))
        ptr_key = tuple(0 if t is None else t.data_ptr() for t in srcs) + (res_pair8,)
        jpf = 3 * nl                                   # jobs per flow: (in, cond, res_skip) x layers
        if self.packed.get("job_key") != ptr_key:
            rows, flow_rows, row_start = [], [], 0
            dp = lambda t: 0 if t is None else t.data_ptr()
            for n, (v, gg, b1, b2, Ah, Al, bo, O, Cin, Kt, perm, Cg, Mpad, koff, Cin_pad, so) in enumerate(specs):
                if n % jpf == 0 and n:                  # row_start restarts per flow: every flow's jobs are a table of their own
                    flow_rows.append(row_start)
                    row_start = 0
                rows.append([dp(v), dp(gg), dp(b1), dp(b2), dp(Ah), dp(Al), dp(bo), row_start,
                             O, Cin, Kt, perm, Cg, Mpad, koff, Cin_pad, 0, 0, dp(so)])
                row_start += -(-O // 16)          # the table kernel packs 16 rows per workgroup
            flow_rows.append(row_start)
            self.packed["jobs"] = torch.tensor(rows, dtype=torch.int64).to(device)
            self.packed["flow_rows"] = flow_rows
            self.packed["job_key"] = ptr_key
        # WN.end folded into the skip path: (W_end . W_skip_i)^T per layer, from the scales the pack just wrote
        fsrc = []
        for k in range(m.n_flows):
            wn = m.WN[k]
            fl = self.packed["flows"][k]
            w_end = _f32c(wn.end.weight)
            for i in range(nl):
                vr = _f32c(_vg(wn.res_skip_layers[i])[0])
                br = _f32c(wn.res_skip_layers[i].bias)
                r0 = C if i < nl - 1 else 0
                fsrc.append((w_end, vr, br, r0, fl["layers"][i], fl["bes"], i, 2 * fl["n_half"]))
        fkey = tuple((a.data_ptr(), b.data_ptr(), c.data_ptr()) for a, b, c, *_ in fsrc)
        if self.packed.get("fold_key") != fkey:
            rows = []
            for (w_end, vr, br, r0, ly, bes, i, nj) in fsrc:
                rows.append([w_end.data_ptr(), vr.data_ptr() + 4 * r0 * C, ly["s_rs"].data_ptr() + 4 * r0,
                             br.data_ptr() + 4 * r0, ly["fold_A"].data_ptr(), bes.data_ptr() + 4 * 8 * i, nj, C])
            self.packed["fold_jobs"] = torch.tensor(rows, dtype=torch.int64).to(device)
            self.packed["fold_key"] = fkey
        # One table-driven launch per flow packs its 3 * n_layers convolutions (weight-norm + split + permute), one more
        # builds its folded WN.end matrices, a third its `start` weights.  With `flow_events` (the no-grad forward) the per-flow
        # work is enqueued on the caller's current stream - a side stream there - and an event per flow lets the main stream
        # start flow k as soon as ITS weights are packed: the pack is HBM-bound (2.1 GB per forward), the GEMMs are not.
        jobs_ptr, fold_ptr = self.packed["jobs"].data_ptr(), self.packed["fold_jobs"].data_ptr()
        starts = []
        for k in range(m.n_flows):
            wn = m.WN[k]
            fl = self.packed["flows"][k]
            _lib.call("t2s_pack_conv_weight_table", _lib.c_vp(jobs_ptr + k * jpf * 19 * 8), jpf, self.packed["flow_rows"][k], st)
            _lib.call("t2s_wg_endfold_weights", _lib.c_vp(fold_ptr + k * nl * 8 * 8), nl, C, st)
            v, gg = _vg(wn.start)
            v, gg = _f32c(v), (None if gg is None else _f32c(gg))
            starts += [v, gg]
            _lib.call("t2s_weightnorm_small", _lib.ptr(v), _lib.ptr(gg), C, fl["n_half"], _lib.ptr(fl["w_start"]), st)
            fl["w_inv"] = None
            if flow_events is not None:
                ev = torch.cuda.Event()
                ev.record()
                flow_events.append(ev)
        keep = srcs + [t for tup in fsrc for t in tup[:3]] + starts
        self.packed_key = key
        self.packed["res_pair8"] = res_pair8
        self._keep = keep
        return self.packed

    # ------------------------------------------------------------------ composed conditioning (inverse flow)
    def compose_geom(self):
        """(P, nlag, K2) when the conditioning path can be composed with the upsampler, else None."""
        m, g = self.m, self.geom()
        up = m.upsample
        ksz, stride, G = up.kernel_size[0], up.stride[0], m.n_group
        if os.environ.get("T2S_COND_COMPOSE") != "1" or not self.use_fold:      # opt-in: see DESIGN.md section 5 for the measurements
            return None
        if ksz % stride or stride % G or ((ksz // stride) * up.in_channels) % 32 or g["C"] % 16:
            return None
        return stride // G, ksz // stride, (ksz // stride) * up.in_channels

    def compose_cond(self, device):
        """(W_cond,i . U_phi) for every layer and phase as A-operand planes, and the biases with the upsampler's bias folded in.
        Built from the packed weights, so it is keyed like them; 42 MB per layer at config.json defaults (4 GB: sized for 288 GB)."""
        if self.packed.get("compose_key") == self.packed_key:
            return
        m, g = self.m, self.geom()
        P, nlag, K2 = self.compose_geom()
        C, nl, ks = g["C"], g["nl"], g["ks"]
        up = m.upsample
        st = _lib.current_stream()
        ncols = P * K2 + 1
        Lp_u = _lib.plane_rows(ncols, 0)
        sc = g["Spad"] // 32
        bf = dict(dtype=torch.bfloat16, device=device)
        U_h, U_l = torch.zeros(1, sc, Lp_u, 32, **bf), torch.zeros(1, sc, Lp_u, 32, **bf)
        W, bias = _f32c(up.weight), _f32c(up.bias)
        _lib.call("t2s_wg_upsample_basis", _lib.ptr(W), _lib.ptr(bias), up.in_channels, up.kernel_size[0], up.stride[0],
                  m.n_group, Lp_u, 0, _lib.ptr(U_h), _lib.ptr(U_l), st)
        tmp = torch.empty(2 * C, ncols, dtype=torch.float32, device=device)
        zb = torch.zeros(g["Mpad1"], dtype=torch.float32, device=device)
        cond_off = (ks * g["Cpad"] // 32) * g["Mpad1"] * 32 * 2          # bytes: the conditioning K-chunks of the packed gate weights
        mc = K2 // 32
        for k in range(m.n_flows):
            for i in range(nl):
                ly = self.packed["flows"][k]["layers"][i]
                if "Ach" not in ly:
                    ly["Ach"] = torch.zeros(P, mc, g["Mpad1"], 32, **bf)
                    ly["Acl"] = torch.zeros(P, mc, g["Mpad1"], 32, **bf)
                    ly["b1c"] = torch.zeros(g["Mpad1"], dtype=torch.float32, device=device)
                _lib.call("t2s_conv_bias_act", _lib.c_vp(ly["A1h"].data_ptr() + cond_off), _lib.c_vp(ly["A1l"].data_ptr() + cond_off),
                          _lib.ptr(zb), _lib.ptr(U_h), _lib.ptr(U_l), None, None, _lib.ptr(tmp), 0, 1, g["n_cond"], 2 * C, 1, 1, 0,
                          ncols, Lp_u, 0, g["Mpad1"], st)
                _lib.call("t2s_wg_compose_cond", _lib.ptr(tmp), _lib.ptr(ly["b1"]), 2 * C, g["Mpad1"], P, K2, ncols,
                          _lib.ptr(ly["Ach"]), _lib.ptr(ly["Acl"]), _lib.ptr(ly["b1c"]), st)
        self._keep_compose = (U_h, U_l, tmp, zb, W, bias)
        self.packed["compose_key"] = self.packed_key

    # ------------------------------------------------------------------ workspaces
    def workspace(self, B, L, device):
        key = (B, L, str(device))
        w = self.ws.get(key)
        if w is None:
            g = self.geom()
            Lp = _lib.plane_rows(L, g["halo"])
            bf = dict(dtype=torch.bfloat16, device=device)
            xc, sc = g["Cpad"] // 32, g["Spad"] // 32
            w = dict(Lp=Lp,
                     Xh=torch.zeros(B, xc, Lp, 32, **bf), Xl=torch.zeros(B, xc, Lp, 32, **bf),
                     Ah=torch.zeros(B, xc, Lp, 32, **bf), Al=torch.zeros(B, xc, Lp, 32, **bf),
                     Sh=torch.zeros(B, sc, Lp, 32, **bf), Sl=torch.zeros(B, sc, Lp, 32, **bf),
                     skip=torch.zeros(B, xc, Lp, 32, dtype=torch.float32, device=device),
                     fold_acc=torch.zeros(_lib.load().t2s_wg_gate_fold_slots(B, g["C"], L), B, 8, L, dtype=torch.float32,
                                          device=device))
            self.ws = {key: w}      # keep one shape resident
        return w

    # ------------------------------------------------------------------ stages
    def _check_inputs(self, *tensors):
        for t in tensors:
            if not t.is_cuda:
                raise _lib.T2SError("WaveGlow (MI355X build) needs CUDA/HIP tensors; got a %s tensor - there is no "
                                    "CPU fallback" % t.device)

    def _upsample(self, mel, B, L, w):
        m, g = self.m, self.geom()
        up = m.upsample
        mel32 = _f32c(mel)
        W, bias = _f32c(up.weight), _f32c(up.bias)
        _lib.call("t2s_wg_upsample_squeeze", _lib.ptr(mel32), _lib.ptr(W), _lib.ptr(bias), B, up.in_channels,
                  mel32.size(2), up.kernel_size[0], up.stride[0], m.n_group, L, w["Lp"], g["halo"],
                  _lib.ptr(w["Sh"]), _lib.ptr(w["Sl"]), _lib.current_stream())
        self._keep_up = (mel32, W, bias)

    def _wn(self, k, z, B, L, w, c_off, n_half, ph=None):
        """start -> n_layers x (in+cond+gate, res/skip); leaves the skip sum in w['skip'].  ph = (M_hi, M_lo, Fp, P, K2):
        conditioning through the composed weights and the mel-window planes (inverse flow)."""
        m, g = self.m, self.geom()
        C, nl, ks = g["C"], g["nl"], g["ks"]
        fl = self.packed["flows"][k]
        st = _lib.current_stream()
        wn = m.WN[k]
        b_start = _f32c(wn.start.bias)
        self._keep_wn = [b_start]
        _lib.call("t2s_wg_start", _lib.ptr(z), _lib.ptr(fl["w_start"]), _lib.ptr(b_start), B, m.n_group, c_off, n_half,
                  C, L, w["Lp"], g["halo"], _lib.ptr(w["Xh"]), _lib.ptr(w["Xl"]), st)
        for i in range(nl):
            ly = fl["layers"][i]
            timed = self.gemm_events is not None and self._gemm_launch_no % self.gemm_event_stride == 0
            self._gemm_launch_no += 1
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if ph is not None:
                Mh, Ml, Fp, P, K2 = ph
                _lib.call("t2s_wg_in_melwin_gate_fold", _lib.ptr(ly["A1h"]), _lib.ptr(ly["A1l"]), _lib.ptr(ly["Ach"]),
                          _lib.ptr(ly["Acl"]), _lib.ptr(ly["b1c"]), _lib.ptr(w["Xh"]), _lib.ptr(w["Xl"]), _lib.ptr(Mh), _lib.ptr(Ml),
                          _lib.ptr(w["Ah"]), _lib.ptr(w["Al"]), _lib.ptr(ly["fold_A"]), _lib.ptr(w["fold_acc"]),
                          1 if i == 0 else 0, B, C, K2, ks, 2 ** i, L, w["Lp"], g["halo"], g["Mpad1"], P, Fp, st)
            elif self.use_fold:
                _lib.call("t2s_wg_in_cond_gate_fold", _lib.ptr(ly["A1h"]), _lib.ptr(ly["A1l"]), _lib.ptr(ly["b1"]),
                          _lib.ptr(w["Xh"]), _lib.ptr(w["Xl"]), _lib.ptr(w["Sh"]), _lib.ptr(w["Sl"]),
                          _lib.ptr(w["Ah"]), _lib.ptr(w["Al"]), _lib.ptr(ly["fold_A"]), _lib.ptr(w["fold_acc"]),
                          1 if i == 0 else 0, B, C, g["n_cond"], ks, 2 ** i, L, w["Lp"], g["halo"], g["Mpad1"], st)
            else:
                _lib.call("t2s_wg_in_cond_gate", _lib.ptr(ly["A1h"]), _lib.ptr(ly["A1l"]), _lib.ptr(ly["b1"]),
                          _lib.ptr(w["Xh"]), _lib.ptr(w["Xl"]), _lib.ptr(w["Sh"]), _lib.ptr(w["Sl"]),
                          _lib.ptr(w["Ah"]), _lib.ptr(w["Al"]), B, C, g["n_cond"], ks, 2 ** i, L, w["Lp"], g["halo"],
                          g["Mpad1"], st)
            if timed:
                e1.record()
                self.gemm_events.append((e0, e1))
            n_res = C if i < nl - 1 else 0
            if self.use_fold:
                if n_res:       # the last layer has no residual half, and its skip half lives in the fold
                    _lib.call("t2s_wg_res_only", _lib.ptr(ly["A2h"]), _lib.ptr(ly["A2l"]), _lib.ptr(ly["b2"]),
                              _lib.ptr(w["Ah"]), _lib.ptr(w["Al"]), _lib.ptr(w["Xh"]), _lib.ptr(w["Xl"]), B, C, L, w["Lp"],
                              g["halo"], ly["Mpad2"], 1 if self.packed.get("res_pair8") else 0, st)
            else:
                _lib.call("t2s_wg_res_skip", _lib.ptr(ly["A2h"]), _lib.ptr(ly["A2l"]), _lib.ptr(ly["b2"]),
                          _lib.ptr(w["Ah"]), _lib.ptr(w["Al"]), _lib.ptr(w["Xh"]), _lib.ptr(w["Xl"]), _lib.ptr(w["skip"]),
                          B, C, n_res, 1 if i == 0 else 0, L, w["Lp"], g["halo"], ly["Mpad2"], st)

    def wn_forward(self, k, audio, spect):
        """WN[k].forward((audio, spect)) (reference glow.py:154-175) on the no-grad kernels: start, n_layers x (gate GEMM with
        WN.end folded in, residual GEMM), then WN.end's output (b ; log_s) without applying the coupling."""
        m = self.m
        self._check_inputs(audio, spect)
        dev = audio.device
        B, n_in, L = audio.shape
        c_off, n_rem, n_half = self._flow_geom(k)
        g = self.geom()
        if n_in != n_half or spect.size(1) != g["n_cond"] or spect.size(2) != L:
            raise ValueError("WN[%d] takes audio [B, %d, L] and spect [B, %d, L]" % (k, n_half, g["n_cond"]))
        self.pack_weights(dev, force=False, res_pair8=self.use_fold)
        w = self.workspace(B, L, dev)
        st = _lib.current_stream()
        spect32 = _f32c(spect)
        _lib.call("t2s_f32_to_planes", _lib.ptr(spect32), B, g["n_cond"], L, w["Lp"], g["halo"], _lib.ptr(w["Sh"]), _lib.ptr(w["Sl"]), st)
        z = torch.zeros(B, m.n_group, L, dtype=torch.float32, device=dev)
        z[:, c_off:c_off + n_half] = audio.detach().to(torch.float32)
        self._wn(k, z, B, L, w, c_off, n_half)
        wn_out = torch.empty(B, 2 * n_half, L, dtype=torch.float32, device=dev)
        if self.use_fold:
            w2 = dict(w)
            w2["wn_out"] = wn_out
            self._end(k, z, None, B, L, w2, c_off, n_half, reverse=False)
        else:
            self._end(k, z, None, B, L, w, c_off, n_half, reverse=False, wn_out=wn_out)
        self._keep_wnf = (spect32, z)
        return wn_out.to(audio.dtype)

    def _end(self, k, z, log_s, B, L, w, c_off, n_half, reverse, wn_out=None, skip=None):
        m, g = self.m, self.geom()
        wn = m.WN[k]
        w_end, b_end = _f32c(wn.end.weight), _f32c(wn.end.bias)
        self._keep_end = (w_end, b_end)
        if self.use_fold and wn_out is None and skip is None:
            fl = self.packed["flows"][k]
            _lib.call("t2s_wg_end_fold_affine", _lib.ptr(w["fold_acc"]), w["fold_acc"].size(0), _lib.ptr(fl["bes"]),
                      g["nl"], _lib.ptr(b_end), _lib.ptr(z), _lib.ptr(log_s), _lib.ptr(w.get("wn_out")), B, m.n_group, c_off,
                      n_half, L, 1 if reverse else 0, _lib.current_stream())
            return
        _lib.call("t2s_wg_end_affine", _lib.ptr(w["skip"] if skip is None else skip), _lib.ptr(w_end), _lib.ptr(b_end),
                  _lib.ptr(z), _lib.ptr(log_s), _lib.ptr(wn_out), B, m.n_group, c_off, n_half, g["C"], L, w["Lp"], g["halo"], 1 if reverse else 0,
                  _lib.current_stream())

    def _flow_geom(self, k):
        m = self.m
        c_off = m.n_early_size * (k // m.n_early_every)
        n_rem = m.n_group - c_off
        return c_off, n_rem, n_rem // 2

    # ------------------------------------------------------------------ forward / infer
    def forward(self, mel, audio):
        m = self.m
        self._check_inputs(mel, audio)
        dev = audio.device
        B, T = audio.shape
        G = m.n_group
        L = T // G
        up = m.upsample
        if (mel.size(2) - 1) * up.stride[0] + up.kernel_size[0] < T:
            raise AssertionError("upsampled spectrogram shorter than audio (reference glow.py:216)")
        w = self.workspace(B, L, dev)
        # The input-side work (conditioning upsampler: compute-bound; audio squeeze; 12 log-determinants) does not depend on
        # the per-forward weight pack (HBM-bound): it runs on a second HIP stream next to the pack and joins before flow 0.
        main = torch.cuda.current_stream(dev)
        side = self.side_stream if getattr(self, "side_stream", None) is not None else torch.cuda.Stream(device=dev)
        self.side_stream = side
        if os.environ.get("T2S_NO_SIDE_STREAM"):          # A/B switch: everything on the caller's stream
            side = main
        audio32 = _f32c(audio)
        z = torch.empty(B, G, L, dtype=torch.float32, device=dev)
        log_s_list, log_det_list = [], []
        log_det = torch.empty(m.n_flows, dtype=torch.float32, device=dev)
        Ws = [_f32c(m.convinv[k].conv.weight) for k in range(m.n_flows)]
        keep = list(Ws)
        # B*L*logdet(W_k) of all flows in one launch (reference glow.py:100)
        use_val = m.n_flows <= 16          # the table travels as a kernel argument: no per-forward host -> device copy
        jobs = torch.tensor([[Ws[k].data_ptr(), log_det.data_ptr() + 4 * k, 0, self._flow_geom(k)[1]]
                             for k in range(m.n_flows)], dtype=torch.int64)
        if not use_val:
            jobs = jobs.to(dev)
        keep.append(jobs)
        side.wait_stream(main)               # inputs, the job table and earlier users of the workspace are ordered before
        with torch.cuda.stream(side):
            st2 = _lib.current_stream()
            _lib.call("t2s_wg_audio_squeeze", _lib.ptr(audio32), _lib.ptr(z), B, T, G, L, 0, st2)
            self._upsample(mel, B, L, w)
            ev_inputs = torch.cuda.Event()       # flow 0 needs z and the conditioning planes; the log-determinants are only
            ev_inputs.record()                   # outputs and join at the end
            if use_val:
                _lib.call("t2s_small_logdet_inv_batch_host", ctypes.c_void_p(jobs.data_ptr()), m.n_flows, float(B * L), st2)
            else:
                _lib.call("t2s_small_logdet_inv_batch", _lib.ptr(jobs), m.n_flows, float(B * L), st2)
        # The per-forward weight pack (weight_norm recompute + split + permute, HBM-bound: 1.07 GB in, 1.07 GB out) runs flow by
        # flow on a third stream; the main stream waits for flow k's event only, so all but the first flow's share of the pack
        # hides under the GEMMs of the flows before it.
        pack_s = self.pack_stream if getattr(self, "pack_stream", None) is not None else torch.cuda.Stream(device=dev)
        self.pack_stream = pack_s
        if os.environ.get("T2S_NO_SIDE_STREAM") or os.environ.get("T2S_PACK_OVERLAP") == "0":
            pack_s = main                   # A/B switch: the whole pack in front of flow 0 on the caller's stream (round 1)
        pack_events = []
        pack_s.wait_stream(main)
        with torch.cuda.stream(pack_s):
            self.pack_weights(dev, force=True, flow_events=pack_events, res_pair8=self.use_fold)
        if side is not main:
            main.wait_event(ev_inputs)
        st = _lib.current_stream()
        for k in range(m.n_flows):
            main.wait_event(pack_events[k])
            c_off, n_rem, n_half = self._flow_geom(k)
            Wk = Ws[k]
            _lib.call("t2s_wg_convinv", _lib.ptr(z), _lib.ptr(Wk), B, G, c_off, n_rem, L, st)
            self._wn(k, z, B, L, w, c_off, n_half)
            log_s = torch.empty(B, n_half, L, dtype=torch.float32, device=dev)
            self._end(k, z, log_s, B, L, w, c_off, n_half, reverse=False)
            log_s_list.append(log_s)
            log_det_list.append(log_det[k])
        main.wait_stream(side)               # log_det
        self._keep_fwd = (audio32, keep)
        return z, log_s_list, log_det_list

    def infer(self, mel, sigma, noise):
        m = self.m
        self._check_inputs(mel)
        dev = mel.device
        B, _, frames = mel.shape
        G = m.n_group
        up = m.upsample
        # reference glow.py:254-255: drop the last (kernel - stride) upsampled samples
        T = (frames - 1) * up.stride[0] + up.kernel_size[0] - (up.kernel_size[0] - up.stride[0])
        L = T // G
        self.pack_weights(dev, force=False, res_pair8=self.use_fold)
        w = self.workspace(B, L, dev)
        st = _lib.current_stream()
        # Weights are packed once here, so the conditioning path can be composed with the upsampler (K = 640 -> 320 in the gate
        # GEMM, no upsampler launch) - wherever the grid is large enough for the 256-row ping-pong tiles anyway.  Opt-in
        # (T2S_COND_COMPOSE=1): with the activations in time-major planes the phase tiles read rows 2 KB apart and the gate GEMM
        # gains 3.5 % instead of 14.7 % (1000 frames: 32.5 -> 31.7 ms; 300-400 frames lose to tile padding)
        cg = self.compose_geom()
        C = self.geom()["C"]
        ph = None
        if cg is not None and _lib.load().t2s_wg_gate_tile_rows(B, C, L) == 256:      # the library's own tile-height decision
            P, nlag, K2 = cg
            self.compose_cond(dev)
            Fp = -(-frames // 256) * 256
            key = ("melwin", B, Fp, str(dev))
            mw = self.ws.get(key)
            if mw is None:
                mw = (torch.zeros(B, K2 // 32, Fp, 32, dtype=torch.bfloat16, device=dev),
                      torch.zeros(B, K2 // 32, Fp, 32, dtype=torch.bfloat16, device=dev))
                self.ws[key] = mw
            mel32 = _f32c(mel)
            _lib.call("t2s_wg_melwin_planes", _lib.ptr(mel32), B, mel32.size(1), frames, nlag, Fp, _lib.ptr(mw[0]), _lib.ptr(mw[1]), st)
            self._keep_up = (mel32,)
            ph = (mw[0], mw[1], Fp, P, K2)
        else:
            self._upsample(mel, B, L, w)
        # All Gaussian draws of glow.py:260-267,284-289 live in one [B, G, L] buffer: the final
        # n_remaining channels, and in front of them the n_early_size channels re-attached at each early flow.
        z = torch.empty(B, G, L, dtype=torch.float32, device=dev)
        n_rem_final = m.n_remaining_channels
        early_ks = [k for k in reversed(range(m.n_flows)) if k % m.n_early_every == 0 and k > 0]
        if noise is None:
            noise_final = torch.randn(B, n_rem_final, L, dtype=torch.float32, device=dev)
            noise_early = [torch.randn(B, m.n_early_size, L, dtype=torch.float32, device=dev) for _ in early_ks]
        else:
            noise_final, noise_early = noise
        z[:, G - n_rem_final:] = sigma * noise_final.to(dev, torch.float32)
        for k, ne in zip(early_ks, noise_early):
            c_off = m.n_early_size * (k // m.n_early_every)
            z[:, c_off - m.n_early_size:c_off] = sigma * ne.to(dev, torch.float32)
        for k in reversed(range(m.n_flows)):
            c_off, n_rem, n_half = self._flow_geom(k)
            fl = self.packed["flows"][k]
            if fl["w_inv"] is None:
                Wk = _f32c(m.convinv[k].conv.weight)
                fl["w_inv"] = torch.empty(n_rem, n_rem, dtype=torch.float32, device=dev)
                _lib.call("t2s_small_logdet_inv", _lib.ptr(Wk), n_rem, 1.0, None, _lib.ptr(fl["w_inv"]), st)
                fl["_Wk"] = Wk
            self._wn(k, z, B, L, w, c_off, n_half, ph=ph)
            self._end(k, z, None, B, L, w, c_off, n_half, reverse=True)
            _lib.call("t2s_wg_convinv", _lib.ptr(z), _lib.ptr(fl["w_inv"]), B, G, c_off, n_rem, L, st)
        audio = torch.empty(B, L * G, dtype=torch.float32, device=dev)
        _lib.call("t2s_wg_audio_squeeze", _lib.ptr(audio), _lib.ptr(z), B, L * G, G, L, 1, st)
        return audio


class WaveGlow(torch.nn.Module):
    """Reference glow.py:178-310 API on the MI355X kernels."""

    def __init__(self, n_mel_channels, n_flows, n_group, n_early_every, n_early_size, WN_config):
        super().__init__()
        self.upsample = torch.nn.ConvTranspose1d(n_mel_channels, n_mel_channels, 1024, stride=256)
        assert n_group % 2 == 0
        self.n_flows = n_flows
        self.n_group = n_group
        self.n_early_every = n_early_every
        self.n_early_size = n_early_size
        self.WN = torch.nn.ModuleList()
        self.convinv = torch.nn.ModuleList()
        n_half = n_group // 2
        n_remaining_channels = n_group
        for k in range(n_flows):
            if k % self.n_early_every == 0 and k > 0:
                n_half = n_half - self.n_early_size // 2
                n_remaining_channels = n_remaining_channels - self.n_early_size
            self.convinv.append(Invertible1x1Conv(n_remaining_channels))
            self.WN.append(WN(n_half, n_mel_channels * n_group, **WN_config))
        self.n_remaining_channels = n_remaining_channels
        self.__dict__["_engine"] = None
        self._adopt()

    def _adopt(self):
        """WN[k].forward reaches the engine through a weak reference to its owner (kept out of the module tree)."""
        import weakref
        for k, wn_k in enumerate(self.WN):
            wn_k.__dict__["_owner"] = weakref.ref(self)
            wn_k.__dict__["_flow"] = k

    def __setstate__(self, state):
        super().__setstate__(state)
        if "_engine" not in self.__dict__:
            self.__dict__["_engine"] = None
        self._adopt()

    def _eng(self):
        if self.__dict__.get("_engine") is None:
            self.__dict__["_engine"] = _Engine(self)
        return self.__dict__["_engine"]

    def __getstate__(self):            # checkpoints pickle the module object (reference waveglow/train.py:52-60)
        d = self.__dict__.copy()
        d["_engine"] = None
        return d

    def forward(self, forward_input):
        """forward_input = (mel [B, n_mel, frames], audio [B, T]) -> (z, [log_s], [log_det_W])
        (reference glow.py:207-249)."""
        spect, audio = forward_input
        f16 = _lib.operand_format() == 1
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if f16:
                raise _lib.T2SError("the fp16-operand diagnostic build runs the no-grad forward / infer only: gradient planes "
                                    "underflow fp16 (use the shipped library for training)")
            from .glow_autograd import waveglow_forward_with_grad
            return waveglow_forward_with_grad(self, spect, audio)
        out = self._eng().forward(spect, audio)
        if f16:
            self._refuse_overflow(out[0])
        return out

    @staticmethod
    def _refuse_overflow(t):
        """fp16-operand build only: a plane element beyond fp16's range (65504) became inf inside the flow and shows as a non-finite
        output - an error, not a result (the shipped bf16 planes have f32's exponent range and need no such check)."""
        import os
        if os.environ.get("T2S_F16_GUARD", "1") == "0":     # timing runs only: the check is a device read-back per call
            return
        if not bool(torch.isfinite(t).all()):
            raise _lib.T2SError("fp16 operand planes overflowed (|x| > 65504 somewhere in the flow): result refused; "
                                "use the shipped split-bf16 library for this checkpoint / input")

    def infer(self, spect, sigma=1.0, noise=None):
        """mel [B, n_mel, frames] -> audio [B, 256*frames] (reference glow.py:251-292).
        ``noise`` = (final [B, n_remaining, L], [early draws in the reference's order]) makes the
        Gaussian draws explicit for parity tests; by default they are drawn on the device."""
        with torch.no_grad():
            out = self._eng().infer(spect, float(sigma), noise)
        if _lib.operand_format() == 1:
            self._refuse_overflow(out)
        return out.to(spect.dtype) if spect.dtype in (torch.float16, torch.bfloat16) else out

    @staticmethod
    def remove_weightnorm(model):
        """Fold (g, v) into plain ``weight`` tensors (reference glow.py:294-310)."""
        waveglow = model
        for wn in waveglow.WN:
            wn.start = torch.nn.utils.remove_weight_norm(wn.start)
            wn.in_layers = remove(wn.in_layers)
            wn.cond_layers = remove(wn.cond_layers)
            wn.res_skip_layers = remove(wn.res_skip_layers)
        if waveglow.__dict__.get("_engine") is not None:
            waveglow.__dict__["_engine"].packed_key = None
        return waveglow


def remove(conv_list):
    new_conv_list = torch.nn.ModuleList()
    for old_conv in conv_list:
        new_conv_list.append(torch.nn.utils.remove_weight_norm(old_conv))
    return new_conv_list

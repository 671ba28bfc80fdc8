"""The one-launch attention backward (att_bwd_fused_kernel: sdot from the saved context, carries scattered into three slots per
position) against the three-launch form of the same step, through the C ABI (t2s_taco_att_bwd), on random inputs: several row
lengths (whole and partial 32-position chunks), both location-kernel sizes, ragged lengths, garbage in the carry slots that do
not exist for a position.  Reference: tacotron/tacotron.py:124-166,379 (the forward both differentiate)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(B, T, KS, seed):
    from text2speech_amd import _lib
    from text2speech_amd.tacotron.autograd import _AttBwd, _p

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(seed)
    E, AD, F = 512, 128, 32
    nch = (T + 31) // 32
    pad = KS // 2
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    lengths = torch.tensor([T, max(1, T - 7), max(1, T // 2)][:B] + [T] * max(0, B - 3), dtype=torch.int32)
    energies = torch.randn(B, T, generator=g)
    for b in range(B):
        energies[b, lengths[b]:] = float("-inf")
    w_cur = torch.softmax(energies, 1).to(dev)
    memory = r(B, T, E)
    ctx = torch.einsum("bt,bte->be", w_cur, memory).contiguous()
    w_prev, wc_prev = torch.softmax(r(B, T), 1), torch.rand(B, T, generator=g).to(dev) * 2
    q, pmem = r(B, AD), r(B, T, AD)
    w_conv, w_dense, w_v = r(F, 2, KS, sc=0.2), r(AD, F, sc=0.2), r(AD, sc=0.3)
    dctx = [r(B, E, sc=0.1) for _ in range(3)]
    carry = [r(B, T, sc=0.05), r(B, T, sc=0.05)]            # totals: dL/dw_t, dL/dwc_t coming from step t + 1
    st = _lib.current_stream()

    def common(a):
        a.dctx1, a.sc1, a.dctx2, a.sc2, a.dctx3, a.sc3 = _p(dctx[0]), E, _p(dctx[1]), E, _p(dctx[2]), E
        a.w_cur, a.s_wcur = _p(w_cur), T
        a.w_prev, a.wc_prev, a.s_wprev, a.s_wcprev = _p(w_prev), _p(wc_prev), T, T
        a.q, a.pmem, a.memory, a.lengths = _p(q), _p(pmem), _p(memory), _p(lengths_d)
        a.w_loc_conv, a.w_loc_dense, a.w_v = _p(w_conv), _p(w_dense), _p(w_v)
        a.B, a.T, a.att_dim, a.enc_dim, a.loc_f, a.loc_ks = B, T, AD, E, F, KS

    lengths_d = lengths.to(dev)
    out = {}
    for form in ("three", "one"):
        z = lambda *s: torch.zeros(*s, device=dev)
        d_q, d_pmem = z(B, AD), z(B, T, AD) + 0.5          # (d_pmem is accumulated into: start from a known non-zero value)
        dD, dK, dv = z(B * nch, AD * F) + 0.25, z(B * nch, F * 2 * KS) + 0.25, z(B * nch, AD) + 0.25
        dw_buf, df_buf, dq_part, dctx_out = z(B, T), z(B, T, 32), z(B, nch, AD), z(B, E)
        a = _AttBwd()
        common(a)
        a.d_q, a.d_pmem, a.d_memory = _p(d_q), _p(d_pmem), None
        a.dD_part, a.dK_part, a.dv_part = _p(dD), _p(dK), _p(dv)
        a.dw_buf, a.df_buf, a.dq_part, a.dctx_out = _p(dw_buf), _p(df_buf), _p(dq_part), _p(dctx_out)
        if form == "three":
            cw, cwc = carry[0].clone(), carry[1].clone()
            a.dw_carry, a.dwc_carry = _p(cw), _p(cwc)
            _lib.call("t2s_taco_att_bwd", ctypes.byref(a), st)
            torch.cuda.synchronize()
            out[form] = dict(d_q=d_q, d_pmem=d_pmem, dD=dD, dK=dK, dv=dv, dctx=dctx_out, cw=cw, cwc=cwc)
        else:
            # the totals split over the slots that exist for a position, garbage in those that do not
            t = torch.arange(T)
            l, t0 = t % 32, t - t % 32
            has1 = ((l >= 32 - pad) & (t0 + 32 < T)).to(dev)
            has2 = ((l < pad) & (t0 > 0)).to(dev)
            ins = []
            for tot in carry:
                s1, s2 = r(B, T, sc=0.05), r(B, T, sc=0.05)
                s0 = tot - torch.where(has1, s1, torch.zeros_like(s1)) - torch.where(has2, s2, torch.zeros_like(s2))
                ins.append(torch.stack([s0, torch.where(has1, s1, s1 * 0 + 123.0), torch.where(has2, s2, s2 * 0 - 77.0)]).contiguous())
            outs = [torch.full((3, B, T), float("nan"), device=dev) for _ in range(2)]
            a.dw_carry, a.dwc_carry = _p(ins[0]), _p(ins[1])
            a.ctx, a.s_ctx, a.dw_carry_out, a.dwc_carry_out = _p(ctx), E, _p(outs[0]), _p(outs[1])
            _lib.call("t2s_taco_att_bwd", ctypes.byref(a), st)
            torch.cuda.synchronize()
            tot = []
            for o in outs:
                v = o[0].clone()
                v += torch.where(has1, o[1], torch.zeros_like(v))
                v += torch.where(has2, o[2], torch.zeros_like(v))
                assert torch.isfinite(v).all(), "a carry slot that exists for a position was not written"
                tot.append(v)
            out[form] = dict(d_q=dq_part.sum(1), d_pmem=d_pmem, dD=dD, dK=dK, dv=dv, dctx=dctx_out, cw=tot[0], cwc=tot[1])
    return out


@pytest.mark.parametrize("B,T,KS", [(3, 70, 31), (2, 33, 31), (3, 256, 31), (3, 96, 5), (1, 20, 31), (2, 600, 31)])
def test_one_launch_matches_three_launches(B, T, KS):
    out = _run(B, T, KS, seed=100 + T + KS)
    for k, ref in out["three"].items():
        got = out["one"][k]
        err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        assert err < 2e-5, "%s: max error %.2e of max |ref| (B=%d T=%d KS=%d)" % (k, err, B, T, KS)

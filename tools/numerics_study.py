#!/usr/bin/env python3
"""CPU study: how many MFMA products per MAC does WaveGlow parity need?

The gate GEMM is MFMA-bound with THREE bf16 products per algorithmic MAC (split-bf16: a_hi b_hi + a_hi b_lo + a_lo b_hi).  Before
any kernel is written, the candidate cheaper operand formats are emulated here inside the CPU oracle: the three GEMM-shaped
convolutions of every WN layer (in_layers, cond_layers, res_skip_layers - the ones the product runs on the matrix cores) are
replaced by f32 convolutions over ROUNDED operands, everything else stays exact f32, and z / every log_s / infer() audio are
compared with the exact-f32 oracle on the full 8 x 16000 config and on a stress state dict (WN.end std 0.04 instead of 0.02,
all weight-norm gains x 1.25: max |log_s| 3-4; see stress_state for why not harsher).

Cost model (MI355X_MICROARCH.md, matrix cores): bf16 / fp16 MFMA = 1 unit per product; block-scaled fp8 (e4m3) MFMA = 0.5.

    python tools/numerics_study.py [--quick] > profiles/r03_numerics.md
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import waveglow_oracle as O  # noqa: E402
from text2speech_amd import synth  # noqa: E402

bf16 = lambda t: t.to(torch.bfloat16).float()
fp16 = lambda t: t.to(torch.float16).float()


def fp8(t):
    """e4m3 with ONE power-of-two scale per tensor that puts max|t| in [128, 256) (a block-scaled MFMA would use one scale per 32
    elements, so this is the pessimistic end); returns the de-scaled values."""
    m = float(t.abs().max())
    if m == 0.0:
        return t.clone()
    if m != m or m == float("inf"):          # an fp16 operand overflowed upstream: the scheme fails, let it show as nan
        return t * float("nan")
    s = 2.0 ** (7 - int(torch.floor(torch.log2(torch.tensor(m)))))
    return (t * s).to(torch.float8_e4m3fn).float() / s


def fp8b(t, dim=1):
    """e4m3 with one power-of-two scale per block of 32 along `dim` (the contraction index: input channels), i.e. what the
    block-scaled MFMA (v_mfma_scale_f32_16x16x128_f8f6f4, E8M0 scale per 32 elements) applies in hardware."""
    t2 = t.movedim(dim, -1)
    shp = t2.shape
    pad = (-shp[-1]) % 32
    if pad:
        t2 = F.pad(t2, (0, pad))
    blk = t2.reshape(*t2.shape[:-1], -1, 32)
    m = blk.abs().amax(dim=-1, keepdim=True)
    if not bool(torch.isfinite(m).all()):
        return t * float("nan")
    e = torch.floor(torch.log2(torch.clamp(m, min=1e-30)))
    sc = torch.exp2(7 - e)
    q = (blk * sc).to(torch.float8_e4m3fn).float() / sc
    q = torch.where(m > 0, q, torch.zeros_like(q)).reshape(*t2.shape)
    if pad:
        q = q[..., :shp[-1]]
    return q.movedim(-1, dim)


SCHEMES = {
    # name: (cost in bf16-MFMA units per MAC, fn(x, w) -> list of (x_operand, w_operand) products to sum)
    "exact f32": (16.0, lambda x, w: [(x, w)]),
    "bf16 x3 (shipped)": (3.0, lambda x, w: [(bf16(x), bf16(w)), (bf16(x), bf16(w - bf16(w))), (bf16(x - bf16(x)), bf16(w))]),
    "fp16 x3": (3.0, lambda x, w: [(fp16(x), fp16(w)), (fp16(x), fp16(w - fp16(w))), (fp16(x - fp16(x)), fp16(w))]),
    "fp16 weights split x fp16 acts (2)": (2.0, lambda x, w: [(fp16(x), fp16(w)), (fp16(x), fp16(w - fp16(w)))]),
    "bf16 weights split x bf16 acts (2)": (2.0, lambda x, w: [(bf16(x), bf16(w)), (bf16(x), bf16(w - bf16(w)))]),
    "fp16 main + 2 fp8 cross (2)": (2.0, lambda x, w: [(fp16(x), fp16(w)), (fp8(x), fp8(w - fp16(w))), (fp8(x - fp16(x)), fp8(w))]),
    "fp16 main + 2 fp8 cross, scale per 32-block (2)": (2.0, lambda x, w: [(fp16(x), fp16(w)), (fp8b(x), fp8b(w - fp16(w))),
                                                                      (fp8b(x - fp16(x)), fp8b(w))]),
    "fp16 main + fp8 weight-residual cross (1.5)": (1.5, lambda x, w: [(fp16(x), fp16(w)), (fp8(x), fp8(w - fp16(w)))]),
    "bf16 main + 2 fp8 cross (2)": (2.0, lambda x, w: [(bf16(x), bf16(w)), (fp8(x), fp8(w - bf16(w))), (fp8(x - bf16(x)), fp8(w))]),
    "fp16 x1": (1.0, lambda x, w: [(fp16(x), fp16(w))]),
    "bf16 x1": (1.0, lambda x, w: [(bf16(x), bf16(w))]),
}


class Emulate:
    """Route the oracle's GEMM-shaped WN convolutions (weights with >= 64 input channels: in / cond / res_skip; `start`, `end`
    and the 1x1 invertible convolutions have at most 8 and stay exact, as in the product) through a scheme."""

    def __init__(self, fn):
        self.fn, self.real = fn, F.conv1d

    def conv1d(self, x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if self.fn is None or w.size(1) < 64:
            return self.real(x, w, b, stride, padding, dilation, groups)
        y = None
        for xo, wo in self.fn(x, w):
            t = self.real(xo, wo, None, stride, padding, dilation, groups)
            y = t if y is None else y + t
        return y if b is None else y + b.view(1, -1, 1)

    def __enter__(self):
        O.F = type("Fshim", (), {k: getattr(F, k) for k in dir(F) if not k.startswith("__")})
        O.F.conv1d = self.conv1d
        return self

    def __exit__(self, *a):
        O.F = F


def stress_state(cfg, end_std=0.03, gain=1.25):
    """WN.end std 0.03 (seeded default 0.02) and every weight-norm gain of the WN layers x 1.25: max |log_s| 3-4 (the seeded
    weights already reach 1.3), |z| up to a few hundred.  The flow multiplies by exp(log_s) twelve times over, so its condition
    number grows exponentially with |log_s|: at std 0.04 / gains x 1.5 (max |log_s| 6.3, std(z) 2000) the SHIPPED split-bf16
    path is already 47 % off the f32 oracle and every fp16 operand overflows, and at VERDICT r2's suggestion (std 0.1, gains x 4)
    the exact-f32 reference itself returns nan (max |log_s| 42).  This setting is the strongest at which 1e-3 is still a
    meaningful bar for ANY 16-bit-class scheme."""
    return synth.waveglow_state(cfg, end_std=end_std, wn_gain=gain)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def mx(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300))


def run(cfg, sd, mel, audio, mel_inf, noise, names):
    with torch.no_grad():
        ref = O.waveglow_forward(sd, cfg, mel, audio)
        ref_audio = O.waveglow_infer(sd, cfg, mel_inf, noise[0], noise[1], sigma=0.666)
    ls_abs = max(float(ls.abs().max()) for ls in ref[1])
    rows = []
    for name in names:
        cost, fn = SCHEMES[name]
        t0 = time.time()
        with torch.no_grad(), Emulate(fn):
            out = O.waveglow_forward(sd, cfg, mel, audio)
            aud = O.waveglow_infer(sd, cfg, mel_inf, noise[0], noise[1], sigma=0.666)
        worst_ls = max(rel(a, b) for a, b in zip(out[1], ref[1]))
        worst_ls_abs = max(float((a - b).abs().max()) for a, b in zip(out[1], ref[1]))
        rows.append((name, cost, rel(out[0], ref[0]), mx(out[0], ref[0]), worst_ls, worst_ls_abs, rel(aud, ref_audio), mx(aud, ref_audio)))
        print("  [%s: %.0f s]" % (name, time.time() - t0), file=sys.stderr, flush=True)
    return rows, ls_abs, float(ref[0].std())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="2 x 4096 samples instead of 8 x 16000 (minutes -> seconds)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    cfg = synth.WAVEGLOW_DEFAULT
    B, T = (2, 4096) if args.quick else (8, 16000)
    mel, audio = synth.waveglow_inputs(B, T, seed=31)
    gen = torch.Generator().manual_seed(5)
    frames = 24 if args.quick else 120
    mel_inf = torch.randn(1, 80, frames, generator=gen)
    L = frames * 256 // 8
    noise = (torch.randn(1, 4, L, generator=gen), [torch.randn(1, 2, L, generator=gen) for _ in range(2)])
    names = [n for n in SCHEMES if n != "exact f32"]
    print("# Products per MAC vs parity: CPU emulation inside the oracle (tools/numerics_study.py)\n")
    print("Forward %d x %d, config.json defaults (512 channels); infer: 1 x %d frames, sigma 0.666.  Reference = the exact-f32 oracle; "
          "errors are relative (L2, and max |diff| / max |ref|).  Parity bar 1e-3; a scheme earns a kernel only at <= 3e-4 worst case."
          % (B, T, frames))
    for title, sd in (("seeded weights (`synth.waveglow_state`: WN.end std 0.02, gains 0.75-1.25)", synth.waveglow_state(cfg)),
                      ("stress weights (WN.end std 0.03, weight-norm gains x 1.25)", stress_state(cfg))):
        rows, ls_abs, zstd = run(cfg, sd, mel, audio, mel_inf, noise, names)
        print("\n## %s\n\nmax |log_s| = %.2f, std(z) = %.3f\n" % (title, ls_abs, zstd))
        print("| scheme | MFMA units / MAC | z rel-L2 | z max | worst log_s rel-L2 | worst log_s abs | infer audio rel-L2 | audio max |")
        print("|---|---|---|---|---|---|---|---|")
        for r in rows:
            print("| %s | %.1f | %.1e | %.1e | %.1e | %.1e | %.1e | %.1e |" % r)


if __name__ == "__main__":
    main()

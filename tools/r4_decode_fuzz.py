#!/usr/bin/env python3
"""Fuzz of the B <= 8 autoregressive decode: random batch sizes, encoder lengths (1 .. 520: both sides of the fused-attention limit
of 512) and step counts, the round-4 chain (streamed gates + folded prenet + location term one launch early) against the round-3
chain on the same inputs.  Prints one JSON line; exits non-zero on the first mismatch."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.tacotron import Tacotron  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    m = Tacotron(dict(synth.TACOTRON_HPARAMS), 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.cuda().eval()
    eng = m._eng()
    gen = torch.Generator().manual_seed(2026)
    worst, cases = 0.0, []
    m.decoder.gate_threshold = 2.0
    for c in range(n_cases):
        B = int(torch.randint(1, 9, (1,), generator=gen))
        T_in = int(torch.randint(1, 521, (1,), generator=gen)) if c % 4 else int(torch.randint(1, 40, (1,), generator=gen))
        n = int(torch.randint(3, 70, (1,), generator=gen))
        ids = torch.randint(2, 80, (B, T_in), generator=gen).cuda()
        masks = (torch.rand(n, B, 2, 256, generator=gen) < 0.5).to(torch.uint8)
        m.decoder.max_decoder_steps = n
        eng.decode_stream = True
        a = m.inference(ids, None, prenet_masks=masks)
        eng.decode_stream = False
        b = m.inference(ids, None, prenet_masks=masks)
        torch.cuda.synchronize()
        r = max(rel(x, y) for x, y in zip(a, b))
        ok = all(bool(torch.isfinite(x).all()) for x in a) and r < 1e-4
        cases.append((B, T_in, n, r))
        worst = max(worst, r)
        if not ok:
            print(json.dumps({"failed": [B, T_in, n, r]}))
            sys.exit(1)
    eng.decode_stream = True
    print(json.dumps({"cases": len(cases), "worst_rel": worst, "max_T_in": max(c[1] for c in cases), "min_T_in": min(c[1] for c in cases)}))


if __name__ == "__main__":
    main()

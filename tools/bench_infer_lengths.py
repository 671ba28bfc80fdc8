#!/usr/bin/env python3
"""WaveGlow.infer at B=1 for several utterance lengths (config.json defaults, synthetic weights): ms per call and samples/s.
Short utterances leave most of the chip idle with 256-row gate tiles; T2S_GATE_TILE=256 / 128 forces a tile height for A/B."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.glow import WaveGlow  # noqa: E402


def main():
    cfg = synth.WAVEGLOW_DEFAULT
    m = WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg))
    m = m.cuda().eval()
    out = {"gate_tile": os.environ.get("T2S_GATE_TILE", "auto")}
    for frames in (100, 200, 256, 300, 400, 1000):
        mel = torch.randn(1, 80, frames, generator=torch.Generator().manual_seed(frames)).cuda()
        for _ in range(2):
            m.infer(mel, sigma=0.666)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            m.infer(mel, sigma=0.666)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["frames_%d" % frames] = {"ms": round(dt * 1e3, 3), "samples_per_s": round(frames * 256 / dt)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

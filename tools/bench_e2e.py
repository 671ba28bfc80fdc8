#!/usr/bin/env python3
"""End-to-end inference timing on one MI355X (BASELINE configs[4]): Tacotron-2 decode of ~1000 frames then
WaveGlow infer (22.05 kHz, 256 samples per frame)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.glow import WaveGlow  # noqa: E402
from text2speech_amd.tacotron import Tacotron  # noqa: E402


def main():
    hp = dict(synth.TACOTRON_HPARAMS)
    taco = Tacotron(hp, 80, num_speakers=2)
    taco.load_state_dict(synth.tacotron_state())
    taco = taco.cuda().eval()
    cfg = synth.WAVEGLOW_DEFAULT
    wg = WaveGlow(**cfg)
    wg.load_state_dict(synth.waveglow_state(cfg))
    wg = WaveGlow.remove_weightnorm(wg).cuda().eval()
    ids = (torch.arange(128) % 78 + 2)[None].cuda()
    n = 1000
    taco.decoder.gate_threshold, taco.decoder.max_decoder_steps = 2.0, n

    def run():
        t0 = time.perf_counter()
        out = taco.inference(ids, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        audio = wg.infer(out[1], sigma=0.666)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return t1 - t0, t2 - t1, audio

    run()
    ts = [run()[:2] for _ in range(3)]
    ta = sorted(t[0] for t in ts)[1]
    tw = sorted(t[1] for t in ts)[1]
    seconds_of_audio = n * 256 / 22050.0
    print(json.dumps({"frames": n, "audio_samples": n * 256, "tacotron_ms": ta * 1e3, "waveglow_infer_ms": tw * 1e3,
                      "total_ms": (ta + tw) * 1e3, "mel_frames_per_s": n / ta, "vocoder_samples_per_s": n * 256 / tw,
                      "real_time_factor": seconds_of_audio / (ta + tw)}))


if __name__ == "__main__":
    main()

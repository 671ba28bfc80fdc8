#!/usr/bin/env python3
"""B = 1 autoregressive decode, streamed gate partials (ABI v4 gate_part) on / off, alternating in one process on one box:
us per decoder step = slope between 200 and 1000 forced frames.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.tacotron import Tacotron  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    m = Tacotron(dict(synth.TACOTRON_HPARAMS), 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.cuda().eval()
    ids = (torch.arange(64) % 78 + 2)[None].repeat(B, 1).cuda()
    m.decoder.gate_threshold = 2.0
    eng = m._eng()

    def run(n, reps=3):
        m.decoder.max_decoder_steps = n
        m.inference(ids, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            m.inference(ids, None)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    out = {"B": B, "runs": []}
    for rep in range(3):
        for mode in (True, False):
            eng.decode_stream = mode
            t200, t1000 = run(200), run(1000)
            out["runs"].append({"stream": mode, "ms_1000": t1000 * 1e3, "decode_us_per_step": (t1000 - t200) / 800 * 1e6})
    for mode in (True, False):
        v = sorted(r["decode_us_per_step"] for r in out["runs"] if r["stream"] == mode)
        out["stream_on" if mode else "stream_off"] = v[len(v) // 2]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Tacotron-2 throughput (mel frames/s) on one MI355X: autoregressive inference (BASELINE configs[4] front
half: B=1, 1000 forced frames) and the teacher-forced eval forward at configs[1] shapes (B=32, T_in=256,
T_out=800).  Optionally times the CPU oracle on a bounded sample."""
import argparse
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--cpu", action="store_true")
    args = ap.parse_args()
    hp = dict(synth.TACOTRON_HPARAMS)
    sd = synth.tacotron_state()
    m = Tacotron(hp, 80, num_speakers=2)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    out = {}
    ids = (torch.arange(64) % 78 + 2)[None].cuda()
    m.decoder.gate_threshold = 2.0
    for n in (200, 1000):
        m.decoder.max_decoder_steps = n
        m.inference(ids, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            m.inference(ids, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out["inference_B1_%dframes" % n] = {"frames_per_s": n / dt, "ms": dt * 1e3, "us_per_step": dt / n * 1e6}
    B, T_in, T_out = 32, 256, 800
    gen = torch.Generator().manual_seed(21)
    text = torch.randint(2, 80, (B, T_in), generator=gen).cuda()
    mel = torch.randn(B, 80, T_out, generator=gen).cuda()
    il = torch.full((B,), T_in, dtype=torch.long).cuda()
    ol = torch.full((B,), T_out, dtype=torch.long).cuda()
    inp = (text, il, mel, T_in, torch.zeros(B).cuda(), ol)
    m(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        m(inp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["forward_B32_Tin256_Tout800"] = {"frames_per_s": B * T_out / dt, "ms": dt * 1e3, "us_per_step": dt / T_out * 1e6}
    if args.cpu:
        from oracle import tacotron_oracle as O
        torch.set_num_threads(16)
        n = 100
        masks = (torch.rand(n, 1, 2, 256) < 0.5).float()
        with torch.no_grad():
            O.tacotron_inference(sd, hp, ids.cpu(), 10, masks)
            t0 = time.perf_counter()
            O.tacotron_inference(sd, hp, ids.cpu(), n, masks)
            dt = time.perf_counter() - t0
        out["cpu_oracle_inference_B1"] = {"frames_per_s": n / dt, "sample": "%d frames, 16 threads" % n}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Tacotron-2 training step on one MI355X at the BASELINE configs[1] shape: batch 32, T_in 256, T_out 800,
teacher-forced (zero_grad -> forward -> Tacotron2Loss -> backward -> Adam)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.optim import FusedAdam  # noqa: E402
from text2speech_amd.tacotron import Tacotron  # noqa: E402
from text2speech_amd.tacotron.loss_function import Tacotron2Loss  # noqa: E402


def main():
    B, T_in, T_out = int(os.environ.get("B", 32)), 256, int(os.environ.get("T_OUT", 800))
    hp = dict(synth.TACOTRON_HPARAMS)
    m = Tacotron(hp, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.cuda().train()
    opt = FusedAdam([p for p in m.parameters()], lr=1e-4, weight_decay=1e-6)      # train.py:187-189
    gen = torch.Generator().manual_seed(21)
    text = torch.randint(2, 80, (B, T_in), generator=gen).cuda()
    mel = torch.randn(B, 80, T_out, generator=gen).cuda()
    gate = torch.zeros(B, T_out).cuda()
    gate[:, -1] = 1
    il = torch.full((B,), T_in, dtype=torch.long).cuda()
    ol = torch.full((B,), T_out, dtype=torch.long).cuda()
    inp = (text, il, mel, T_in, torch.zeros(B).cuda(), ol)

    crit = Tacotron2Loss()

    def step():
        m.zero_grad(set_to_none=True)
        out = m(inp)
        loss = crit(out, (mel, gate))                 # reference call: criterion(y_pred, y), train.py:219-221
        loss.backward()
        opt.step()
        return loss

    l0 = float(step())
    for _ in range(2):          # the pooled save / scratch buffers exist in two sets (one is held until the next backward ends)
        step()
    torch.cuda.synchronize()
    print("[bench] warm-up done, loss %.4f" % l0, file=sys.stderr, flush=True)
    from text2speech_amd import _lib
    _lib.HOST_TIMES.clear()
    n = 3
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter()
        l = step()
        host += time.perf_counter() - h0            # time the host needs to ENQUEUE a step (no synchronisation inside)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"metric": "Tacotron-2 train-step mel frames/sec (B=%d, T_in=%d, T_out=%d, fwd+loss+bwd+Adam)" % (B, T_in, T_out),
                      "value": B * T_out / dt, "ms_per_step": dt * 1e3, "host_enqueue_ms_per_step": host / n * 1e3, "loss_first": l0, "loss_last": float(l),
                      "max_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))


def host_report(n=3):
    """T2S_HOST_TIMING=1: where the host's enqueue time of a step goes (ms per step, calls per step), to stderr."""
    from text2speech_amd import _lib
    rows = sorted(_lib.HOST_TIMES.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print("[host] inside the library's entry points: %.2f ms per step in %d calls" % (tot / n * 1e3, sum(v[0] for _, v in rows) // n),
          file=sys.stderr)
    for k, v in rows[:12]:
        print("[host]   %-32s %7.2f ms  %5d calls" % (k, v[1] / n * 1e3, v[0] // n), file=sys.stderr)


if __name__ == "__main__":
    main()
    if os.environ.get("T2S_HOST_TIMING"):
        host_report()

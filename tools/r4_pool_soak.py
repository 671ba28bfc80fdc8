#!/usr/bin/env python3
"""Soak of the Tacotron training step over batches padded to many different lengths (the reference's collate pads every batch to its
own maximum): the engine's buffer pool must stay under its cap, device memory must stop growing once the cap is reached, and a
step on a much-used model must equal the same step on a fresh model bit for bit.  Prints one JSON line."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.tacotron import Tacotron, Tacotron2Loss  # noqa: E402
from text2speech_amd.tacotron.tacotron import BufferPool  # noqa: E402


def batch(B, T_in, T_out, gen):
    text = torch.randint(2, 80, (B, T_in), generator=gen)
    mel = torch.randn(B, 80, T_out, generator=gen)
    il = torch.tensor([T_in - (i % 5) for i in range(B)])
    ol = torch.tensor([T_out - (3 * i) % 11 for i in range(B)])
    il[0], ol[0] = T_in, T_out
    gate = torch.zeros(B, T_out)
    for b in range(B):
        text[b, il[b]:] = 0
        mel[b, :, ol[b]:] = 0
        gate[b, ol[b] - 1:] = 1
    return text, il, mel, gate, ol


def main():
    dev = "cuda:0"
    B = 16
    gen = torch.Generator().manual_seed(3)
    crit = Tacotron2Loss()

    def fresh(cap_gb=None):
        m = Tacotron(dict(synth.TACOTRON_HPARAMS), 80, num_speakers=2)
        m.load_state_dict(synth.tacotron_state())
        m = m.to(dev).train()
        if cap_gb is not None:
            m._eng().pool = BufferPool(cap_bytes=int(cap_gb * (1 << 30)))
        return m

    def step(m, bt, seed):
        text, il, mel, gate, ol = bt
        m.load_state_dict(synth.tacotron_state())
        m.zero_grad(set_to_none=True)
        torch.manual_seed(seed)
        out = m((text.to(dev), il.to(dev), mel.to(dev), int(il.max()), torch.zeros(B, device=dev), ol.to(dev)))
        loss = crit(out, (mel.to(dev), gate.to(dev)))
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}

    m = fresh(cap_gb=1.0)
    shapes = [(60 + (7 * i) % 40, 120 + (37 * i) % 150) for i in range(24)]
    batches = [batch(B, ti, to, gen) for ti, to in shapes]
    mem, free_bytes, losses = [], [], []
    for i, bt in enumerate(batches):
        l, _ = step(m, bt, 100 + i)
        losses.append(l)
        mem.append(torch.cuda.memory_allocated() / 2 ** 30)
        free_bytes.append(m._eng().pool.free_bytes / 2 ** 30)
    pool = m._eng().pool
    # the last batch again on the much-used model and on a fresh one: bit-identical
    l_used, g_used = step(m, batches[5], 777)
    l_new, g_new = step(fresh(), batches[5], 777)
    same = l_used == l_new and all(torch.equal(g_used[n], g_new[n]) for n in g_new)
    print(json.dumps({"steps": len(batches), "distinct_shapes": len(set(shapes)), "pool_cap_GB": pool.cap_bytes / 2 ** 30,
                      "pool_free_GB_max": max(free_bytes), "pool_evicted": pool.evicted,
                      "mem_allocated_GB_first_half_max": max(mem[:12]), "mem_allocated_GB_second_half_max": max(mem[12:]),
                      "losses_finite": all(l == l and abs(l) < 1e6 for l in losses), "used_model_equals_fresh_model_bitwise": bool(same)}))
    assert same and max(free_bytes) <= pool.cap_bytes / 2 ** 30 + 1e-9


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Diagnostic (needs a library built with T2S_BUILD_DEFINES=-DT2S_GEMM_STAMPS): where one launch of the ping-pong gate GEMM
spends its cycles - prologue / main loop / epilogue per workgroup, and the in-kernel clock (s_memtime vs s_memrealtime)."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import _lib, synth  # noqa: E402
from text2speech_amd.glow import WaveGlow  # noqa: E402


def main():
    lib = _lib.load()
    cfg = synth.WAVEGLOW_DEFAULT
    m = WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg))
    m = m.cuda().eval()
    mel, audio = synth.waveglow_inputs(8, 16000, seed=1234)
    mel, audio = mel.cuda(), audio.cuda()
    with torch.no_grad():
        for _ in range(12):           # ~0.25 s of back-to-back launches: the clock has settled
            m((mel, audio))
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    fn = lib.t2s_debug_read_pp_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rc = fn(buf, 1024 * 8)
    assert rc == 0, rc
    st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8)[:256].astype(np.int64)     # the last launch: 256 workgroups
    pro, loop, epi, tot = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2], st[:, 3] - st[:, 0]
    real = (st[:, 5] - st[:, 4]).astype(np.float64) * 10.0      # ns (100 MHz)
    clk = tot / real                                             # cycles per ns = GHz
    q = lambda x: [float(np.percentile(x, p)) for p in (5, 50, 95)]
    skew = (st[:, 4].max() - st[:, 4].min()) * 10.0
    out = {"cycles_p5_p50_p95": {"prologue": q(pro), "main_loop": q(loop), "epilogue": q(epi), "total": q(tot)},
           "ns_total_p5_p50_p95": q(real), "in_kernel_clock_GHz_p5_p50_p95": q(clk),
           "entry_skew_ns_first_to_last_workgroup": float(skew),
           "span_ns_first_entry_to_last_exit": float((st[:, 5].max() - st[:, 4].min()) * 10.0)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Diagnostic (library built with T2S_BUILD_DEFINES=-DT2S_CLOCK_PROBE): per launch of the B=1 decoder chain, the body time
of workgroup 0 (100 MHz real-time counter), the in-kernel shader clock (s_memtime / s_memrealtime) and the gap between the end
of one launch's workgroup 0 and the start of the next one's."""
import ctypes
import json
import os
import sys
from collections import defaultdict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import _lib, synth  # noqa: E402
from text2speech_amd.tacotron import Tacotron  # noqa: E402


def main():
    lib = _lib.load()
    m = Tacotron(dict(synth.TACOTRON_HPARAMS), 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    m = m.cuda().eval()
    n_sym = int(sys.argv[1]) if len(sys.argv) > 1 else 64       # usage: decode_probe.py [symbols]
    ids = (torch.arange(n_sym) % 78 + 2)[None].cuda()
    n = 400
    m.decoder.gate_threshold, m.decoder.max_decoder_steps = 2.0, n
    for _ in range(3):
        m.inference(ids, None)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (12 * (1 << 16)))()
    cnt = ctypes.c_uint(0)
    fn = lib.t2s_debug_read_probe
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert fn(buf, ctypes.byref(cnt)) == 0
    total = cnt.value
    a = np.frombuffer(buf, dtype=np.uint64).reshape(1 << 16, 12).astype(np.int64)
    last = min(total, 1 << 16)
    # the ring holds the most recent launches; take the last 5 * 300 in issue order
    idx = [(total - k - 1) & 0xffff for k in range(min(last, 1500))][::-1]
    rows = a[idx]
    per = defaultdict(list)
    gaps = defaultdict(list)
    mids = defaultdict(list)
    for i, r in enumerate(rows):
        kid, t0, r0, t1, r1 = r[:5]
        mids[int(kid)].append([(x - r0) * 10.0 for x in r[5:10]])
        if r1 <= r0:
            continue
        per[int(kid)].append(((r1 - r0) * 10.0, (t1 - t0) / ((r1 - r0) * 10.0)))
        if i + 1 < len(rows) and rows[i + 1][2] > r1:
            gaps[int(kid)].append((rows[i + 1][2] - r1) * 10.0)
    names = {301: "att_fused_mfma (attention role, workgroup 0; stamps: loads staged / query / features / energies / softmax)",
             210: "lstm_cell_p2 attention cell + sparse prenet layer (round 4)", 201: "lstm_cell<1> decoder cell, context columns only (round 4)",
             101: "gemv<1> prenet layer 2 (256x256)", 107: "gemv<7> projection + prenet layer 1 (337x1536)", 300: "att_fused",
             202: "lstm_cell<2> attention cell (4096x1792)", 203: "lstm_cell<3> decoder cell (4096x2560)"}
    out = {}
    for k in sorted(per):
        b = np.array(per[k])
        g = np.array(gaps.get(k, [0.0]))
        out[names.get(k, str(k))] = {"launches": len(b), "body_ns_wg0_median": float(np.median(b[:, 0])),
                                     "clock_GHz_median": float(np.median(b[:, 1])),
                                     "gap_to_next_launch_ns_median": float(np.median(g)),
                                     "mid_stamps_ns_from_entry_median": [float(v) for v in np.median(np.array(mids[k]), axis=0)]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

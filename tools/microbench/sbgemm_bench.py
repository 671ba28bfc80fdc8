#!/usr/bin/env python3
"""Back-to-back launches of the small-batch GEMM (sbgemm_plain_kernel through t2s_gemv) at the Tacotron-2 B = 32 shapes:
us per launch and the weight-stream rate.  usage: sbgemm_bench.py [lib.so to copy over the in-tree library first]"""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    shutil.copy(sys.argv[1], os.path.join(ROOT, "text2speech_amd", "libt2s_hip.so"))
import torch  # noqa: E402
from text2speech_amd import _lib  # noqa: E402

dev = "cuda:0"
out = {"lib": sys.argv[1] if len(sys.argv) > 1 else "in-tree"}
for rows, K, items in ((4096, 2560, 32), (2560, 2560, 32), (4096, 1792, 32), (1792, 2560, 32)):
    W = torch.randn(rows, K, device=dev)
    x = torch.randn(items, K, device=dev)
    y = torch.empty(items, rows, device=dev)
    # a second weight set so that consecutive launches do not find their weights in L2 (the decoder alternates two cells)
    W2 = torch.randn(rows, K, device=dev)
    st = _lib.current_stream()

    def call(Wm):
        _lib.call("t2s_gemv", _lib.ptr(Wm), K, K, None, 0, 0, _lib.ptr(x), K, K, None, 0, 0, None, 0, 0, None, None, _lib.ptr(y),
                  rows, 1, rows, items, 0, None, 0, 1.0, st)
    for _ in range(20):
        call(W); call(W2)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        call(W); call(W2)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (2 * n) * 1e6
    ref = x @ W2.t()
    err = float((y - ref).abs().max() / ref.abs().max())
    out["%dx%d" % (rows, K)] = {"us": round(us, 2), "weight_TBps": round(rows * K * 4 / us / 1e6, 2), "max_rel_err": err}
print(json.dumps(out))

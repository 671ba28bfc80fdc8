#!/usr/bin/env python3
"""Write a legacy-layout WaveGlow checkpoint with the REFERENCE code: `{'model': <pickled glow.WaveGlow object>, ...}` exactly
as waveglow/train.py:52-60 does (tiny configuration so the fixture stays small).  Runs only in the build container; the
output tests/golden/legacy_waveglow_ckpt.pt is data (a pickle naming classes, no source).

usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_checkpoint.py
"""
import os
import sys
import warnings

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/waveglow")

import glow as ref_glow  # noqa: E402  (the reference)
from text2speech_amd import synth  # noqa: E402

CFG = dict(n_mel_channels=8, n_flows=4, n_group=8, n_early_every=2, n_early_size=2,
           WN_config=dict(n_layers=2, n_channels=8, kernel_size=3))


def main():
    torch.manual_seed(0)
    m = ref_glow.WaveGlow(**CFG)
    m.load_state_dict(synth.waveglow_state(CFG, seed=77), strict=True)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    out = os.path.join(ROOT, "tests", "golden", "legacy_waveglow_ckpt.pt")
    torch.save({"model": m, "iteration": 4321, "optimizer": opt.state_dict(), "learning_rate": 1e-4}, out)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()

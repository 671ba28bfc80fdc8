#!/usr/bin/env python3
"""Child process of tests/test_waveglow_gpu.py::test_stress_weights_infer_120_frames_both_splits: the WaveGlow forward (2 x 4096)
and `infer` (1 x 120 frames, sigma 0.666) at the stress weights on whatever library T2S_LIB_PATH names (the library is loaded at
import, so a second operand format needs a second process).  Writes z, audio and the outcome of the range-guard probe (audio x 1e7)
to the .npz given as argv[1]."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from text2speech_amd import synth  # noqa: E402
from text2speech_amd.glow import WaveGlow  # noqa: E402


def stress_case():
    cfg = synth.WAVEGLOW_DEFAULT
    sd = synth.waveglow_state(cfg, end_std=0.03, wn_gain=1.25)
    mel, audio = synth.waveglow_inputs(2, 4096, seed=31)
    gen = torch.Generator().manual_seed(5)
    frames = 120
    mel_inf = torch.randn(1, 80, frames, generator=gen)
    L = frames * 256 // 8
    noise = (torch.randn(1, 4, L, generator=gen), [torch.randn(1, 2, L, generator=gen) for _ in range(2)])
    return cfg, sd, mel, audio, mel_inf, noise


def main():
    cfg, sd, mel, audio, mel_inf, noise = stress_case()
    m = WaveGlow(**cfg)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        z, _, _ = m((mel.cuda(), audio.cuda()))
        a = m.infer(mel_inf.cuda(), sigma=0.666, noise=noise)
    torch.cuda.synchronize()
    # range guard: audio far outside fp16's range - the fp16-operand build must REFUSE with an error, the shipped bf16 planes (f32's
    # exponent range) just compute
    from text2speech_amd import _lib
    refused, finite = 0, 1
    try:
        with torch.no_grad():
            zz, _, _ = m((mel.cuda(), (audio * 1e7).cuda()))
        torch.cuda.synchronize()
        finite = int(bool(torch.isfinite(zz).all()))
    except _lib.T2SError as e:
        print("refused:", e)
        refused = 1
    np.savez(sys.argv[1], z=z.cpu().numpy(), audio=a.cpu().numpy(), overflow_refused=refused, overflow_finite=finite,
             operand_format=_lib.operand_format())


if __name__ == "__main__":
    main()

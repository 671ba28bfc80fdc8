#!/usr/bin/env python3
"""Generate STFT golden vectors by running the REFERENCE class utils/stft.py:STFT on CPU.

Runs only in the build container (needs /root/reference); writes data only (tests/golden/audio_stft.npz).
Harness shims (none touches the reference's arithmetic):
  * librosa is absent.  utils/stft.py and utils/audio_processing.py import pad_center, tiny and normalize from it; in the
    configuration used everywhere in the reference (win_length == filter_length, norm=None) pad_center and normalize are
    the identity - the stand-ins below ASSERT that they are only ever asked for the identity - and tiny is numpy's
    smallest normal number of the array's dtype;
  * STFT.transform hard-codes .cuda() on its operands (utils/stft.py:85-89): Tensor.cuda is rebound to a no-op so the
    conv1d runs on CPU.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_audio.py
"""
import os
import sys
import types
import warnings

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _pad_center(data, size, **kw):
    assert len(data) == size, "stand-in only covers the identity case"
    return data


def _normalize(x, norm=None, **kw):
    assert norm is None, "stand-in only covers the identity case"
    return x


def _tiny(x):
    return np.finfo(np.asarray(x).dtype).tiny


for name in ["librosa", "librosa.util", "librosa.filters"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["librosa"].util = sys.modules["librosa.util"]
sys.modules["librosa.util"].pad_center = _pad_center
sys.modules["librosa.util"].tiny = _tiny
sys.modules["librosa.util"].normalize = _normalize
sys.modules["librosa.filters"].mel = None
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, "/root/reference")

from utils.stft import STFT  # noqa: E402  (the reference)

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    gen = torch.Generator().manual_seed(33)
    out = {}
    for tag, (n_fft, hop, B, T) in {"a": (1024, 256, 2, 4096), "b": (1024, 256, 1, 5000), "c": (256, 64, 3, 1000)}.items():
        audio = torch.rand(B, T, generator=gen) * 1.6 - 0.8
        st = STFT(filter_length=n_fft, hop_length=hop, win_length=n_fft)
        with torch.no_grad():
            mag, ph = st.transform(audio)
            rec = st.inverse(mag, ph)
        out[tag + "_cfg"] = np.array([n_fft, hop, B, T])
        out[tag + "_audio"] = audio.numpy()
        out[tag + "_mag"] = mag.numpy()
        out[tag + "_cos"] = torch.cos(ph).numpy().astype(np.float16)
        out[tag + "_sin"] = torch.sin(ph).numpy().astype(np.float16)
        out[tag + "_rec"] = rec.numpy()
        if tag == "c":
            out["c_fwd_basis"] = st.forward_basis.numpy()
            out["c_inv_basis"] = st.inverse_basis.numpy()
    np.savez_compressed(os.path.join(OUT, "audio_stft.npz"), **out)
    print("wrote audio_stft.npz", {k: v.shape for k, v in out.items()})
    denoiser_and_griffin_lim()


def denoiser_and_griffin_lim():
    """The reference's Denoiser (waveglow/denoiser.py:7-40) on the reference's WaveGlow with this repo's seeded small
    weights, and the reference's griffin_lim (utils/audio_processing.py:51-67) -> tests/golden/audio_denoise_gl.npz.

    Further shims, device placement / RNG capture only: torch.nn.Module.cuda is a no-op (denoiser.py:15 moves its STFT to the
    GPU), torch.cuda.FloatTensor = torch.FloatTensor for WaveGlow.infer's noise constructors (sigma = 0 here, so the draws
    are multiplied by zero), and griffin_lim's numpy draw of the initial phase is replayed from a seeded numpy RNG so that
    the same angles can be handed to the oracle / the HIP path."""
    from text2speech_amd import synth
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    sys.path.insert(0, "/root/reference/waveglow")
    import glow as ref_glow          # noqa: E402  (the reference)
    import denoiser as ref_denoiser  # noqa: E402  (the reference)
    from utils.audio_processing import griffin_lim as ref_griffin_lim  # noqa: E402

    cfg = synth.WAVEGLOW_SMALL
    torch.manual_seed(0)
    wg = ref_glow.WaveGlow(**cfg)
    wg.load_state_dict(synth.waveglow_state(cfg), strict=True)
    wg.eval()
    out = {}
    den = ref_denoiser.Denoiser(wg)                    # mode='zeros': bias spectrum of the vocoder at a zero mel
    out["bias_spec"] = den.bias_spec.numpy()           # [1, 513, 1]
    gen = torch.Generator().manual_seed(44)
    clip = torch.rand(2, 4096, generator=gen) * 0.6 - 0.3
    with torch.no_grad():
        out["denoised_s01"] = den(clip, strength=0.1).numpy()
        out["denoised_s10"] = den(clip, strength=1.0).numpy()      # strong enough for the clamp at zero to bite
    # griffin_lim: magnitudes of a seeded clip, 6 iterations, the reference's own initial-phase draw captured
    st = STFT(filter_length=1024, hop_length=256, win_length=1024)
    sig = torch.rand(1, 6000, generator=gen) * 1.2 - 0.6
    with torch.no_grad():
        mag, _ = st.transform(sig)
    np.random.seed(1234)
    angles = np.angle(np.exp(2j * np.pi * np.random.rand(*mag.size()))).astype(np.float32)
    np.random.seed(1234)                               # the call below draws the same numbers
    with torch.no_grad():
        rec = ref_griffin_lim(mag, st, n_iters=6)
    out["gl_mag"] = mag.numpy()
    out["gl_angles"] = angles
    out["gl_signal"] = rec.numpy()
    np.savez_compressed(os.path.join(OUT, "audio_denoise_gl.npz"), **out)
    print("wrote audio_denoise_gl.npz", {k: v.shape for k, v in out.items()},
          "bias max %.4f" % float(np.abs(out["bias_spec"]).max()))


if __name__ == "__main__":
    main()

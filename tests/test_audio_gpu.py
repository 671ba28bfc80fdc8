"""GPU parity of the audio front-end / back-end: STFT / inverse STFT against outputs of the
reference's own class (tests/golden/audio_stft.npz), log-mel and the Denoiser against the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from text2speech_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu().flatten()
    b = torch.as_tensor(b).double().cpu().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_stft_vs_reference_golden(golden_dir, tag):
    from text2speech_amd.audio import STFT
    _lib.load()
    g = np.load(os.path.join(golden_dir, "audio_stft.npz"))
    n_fft, hop, B, T = [int(v) for v in g[tag + "_cfg"]]
    st = STFT(filter_length=n_fft, hop_length=hop, win_length=n_fft).to(DEV)
    audio = torch.from_numpy(g[tag + "_audio"]).to(DEV)
    mag, ph = st.transform(audio)
    assert tuple(mag.shape) == g[tag + "_mag"].shape
    assert _rel(mag, g[tag + "_mag"]) < 1e-5
    w = torch.from_numpy(g[tag + "_mag"])
    assert _rel(torch.cos(ph).cpu() * w, torch.from_numpy(g[tag + "_cos"].astype(np.float32)) * w) < 2e-3
    assert _rel(torch.sin(ph).cpu() * w, torch.from_numpy(g[tag + "_sin"].astype(np.float32)) * w) < 2e-3
    rec = st.inverse(mag, ph)
    assert tuple(rec.shape) == g[tag + "_rec"].shape
    assert _rel(rec, g[tag + "_rec"]) < 1e-4
    # module forward = transform -> inverse; reconstructs the input away from the edges
    rec2 = st(audio)
    assert _rel(rec2[:, 0, n_fft:-n_fft], audio[:, n_fft:rec2.size(2) - n_fft]) < 1e-3


@pytest.mark.parametrize("B,T", [(8, 16000), (1, 1500), (3, 2048)])
def test_mel_spectrogram_vs_oracle(B, T):
    """B=8, T=16000 is the WaveGlow training batch of BASELINE configs[3] (63 frames); T=1500 gives 6 frames (GEMV path)."""
    from oracle import audio_oracle as A
    from text2speech_amd.audio import TacotronSTFT, mel_filterbank
    _lib.load()
    gen = torch.Generator().manual_seed(B * 1000 + T)
    audio = torch.rand(B, T, generator=gen) * 1.9 - 0.95
    ts = TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0).to(DEV)
    mel = ts.mel_spectrogram(audio.to(DEV))
    fwd, _ = A.stft_basis(1024, 256, 1024)
    basis = A.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert np.abs(basis - mel_filterbank(22050, 1024, 80, 0.0, 8000.0)).max() < 1e-7
    want = A.mel_spectrogram(audio, fwd, basis)
    assert tuple(mel.shape) == tuple(want.shape) == (B, 80, T // 256 + 1)
    assert float((mel.cpu() - want).abs().max()) < 1e-4          # log domain: absolute


def test_mel_range_assert_and_cpu_tensor_raise():
    from text2speech_amd.audio import STFT, TacotronSTFT
    ts = TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0).to(DEV)
    with pytest.raises(AssertionError):
        ts.mel_spectrogram(torch.full((1, 4096), 1.5, device=DEV))
    with pytest.raises(RuntimeError):
        STFT(1024, 256, 1024).to(DEV).transform(torch.zeros(1, 4096))      # host tensor: no CPU path


def test_denoiser_vs_oracle():
    """Denoiser(waveglow) end to end: bias spectrum from the HIP vocoder at zero input, spectral subtraction, inverse."""
    from oracle import audio_oracle as A
    import text2speech_amd.glow as glow
    from text2speech_amd.audio import Denoiser
    _lib.load()
    cfg = synth.WAVEGLOW_SMALL
    m = glow.WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg), strict=True)
    m = m.to(DEV).eval()
    dn = Denoiser(m).to(DEV)
    assert tuple(dn.bias_spec.shape) == (1, 513, 1)
    with torch.no_grad():
        bias_audio = m.infer(torch.zeros(1, 80, 88, device=DEV), sigma=0.0).float().cpu()
    fwd, inv = A.stft_basis(1024, 256, 1024)
    bias_o = A.stft_transform(bias_audio, fwd)[0][:, :, 0][:, :, None]
    assert _rel(dn.bias_spec, bias_o) < 1e-4
    gen = torch.Generator().manual_seed(9)
    audio = torch.rand(2, 8192, generator=gen) * 0.2 - 0.1
    got = dn(audio.to(DEV), strength=0.1)
    want = A.denoise(audio, bias_o, fwd, inv, strength=0.1)
    assert tuple(got.shape) == tuple(want.shape) == (2, 1, 8192)
    assert _rel(got, want) < 1e-3


def test_griffin_lim_vs_oracle():
    """Five alternating-projection iterations from the same initial phase (the iteration amplifies rounding differences, so the
    comparison is on the spectrogram the result reproduces, plus a loose waveform check)."""
    from oracle import audio_oracle as A
    from text2speech_amd.audio import STFT, griffin_lim
    _lib.load()
    gen = torch.Generator().manual_seed(17)
    audio = torch.rand(1, 4096, generator=gen) * 1.2 - 0.6
    fwd, inv = A.stft_basis(1024, 256, 1024)
    mag, _ = A.stft_transform(audio, fwd)
    ang = torch.rand(mag.shape, generator=gen) * 6.28 - 3.14
    want = A.griffin_lim(mag, ang, fwd, inv, n_iters=5)
    st = STFT(1024, 256, 1024).to(DEV)
    got = griffin_lim(mag.to(DEV), st, n_iters=5, angles=ang.to(DEV))
    assert tuple(got.shape) == tuple(want.shape)
    assert _rel(got, want) < 2e-2
    mg, _ = st.transform(got)
    mw, _ = A.stft_transform(want, fwd)
    assert _rel(mg, mw) < 5e-3


def test_denoiser_and_griffin_lim_vs_reference_golden(golden_dir):
    """HIP Denoiser / griffin_lim against what the REFERENCE's own classes produced (tests/golden/audio_denoise_gl.npz):
    the bias spectrum of the small vocoder, two denoised clips (strength 0.1 and 1.0) and a 6-iteration Griffin-Lim signal
    from the reference's captured initial phase."""
    import os
    import numpy as np
    import text2speech_amd.glow as glow
    from text2speech_amd.audio import STFT, Denoiser, griffin_lim
    _lib.load()
    g = np.load(os.path.join(golden_dir, "audio_denoise_gl.npz"))
    cfg = synth.WAVEGLOW_SMALL
    m = glow.WaveGlow(**cfg)
    m.load_state_dict(synth.waveglow_state(cfg), strict=True)
    m = m.to(DEV).eval()
    dn = Denoiser(m).to(DEV)
    assert _rel(dn.bias_spec, g["bias_spec"]) < 1e-3
    gen = torch.Generator().manual_seed(44)
    clip = torch.rand(2, 4096, generator=gen) * 0.6 - 0.3
    for key, strength in (("denoised_s01", 0.1), ("denoised_s10", 1.0)):
        got = dn(clip.to(DEV), strength=strength)
        assert tuple(got.shape) == g[key].shape
        assert _rel(got, g[key]) < 1e-3, key
    st = STFT(1024, 256, 1024).to(DEV)
    sig = griffin_lim(torch.from_numpy(g["gl_mag"]).to(DEV), st, n_iters=6, angles=torch.from_numpy(g["gl_angles"]).to(DEV))
    assert tuple(sig.shape) == g["gl_signal"].shape
    assert _rel(sig, g["gl_signal"]) < 2e-2                 # the iteration amplifies rounding differences
    mg, _ = st.transform(sig)
    assert _rel(mg, st.transform(torch.from_numpy(g["gl_signal"]).to(DEV))[0]) < 5e-3

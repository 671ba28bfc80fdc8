"""CPU: the audio oracle (oracle/audio_oracle.py) against outputs of the reference's own STFT class
(tests/golden/audio_stft.npz, made by tools/gen_golden_audio.py), plus self-consistency of the pieces that have no
reference-side pin (mel filterbank: librosa is absent and unpinned upstream)."""
import os

import numpy as np
import pytest
import torch

from oracle import audio_oracle as A


def _rel(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_stft_transform_and_inverse_vs_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "audio_stft.npz"))
    n_fft, hop, B, T = [int(v) for v in g[tag + "_cfg"]]
    fwd, inv = A.stft_basis(n_fft, hop, n_fft)
    audio = torch.from_numpy(g[tag + "_audio"])
    mag, ph = A.stft_transform(audio, fwd, n_fft, hop)
    assert mag.shape == g[tag + "_mag"].shape
    assert _rel(mag, g[tag + "_mag"]) < 1e-6
    # phase compared through cos / sin (atan2 is discontinuous at +-pi), weighted by magnitude
    w = torch.from_numpy(g[tag + "_mag"])
    assert _rel(torch.cos(ph) * w, torch.from_numpy(g[tag + "_cos"].astype(np.float32)) * w) < 2e-3
    assert _rel(torch.sin(ph) * w, torch.from_numpy(g[tag + "_sin"].astype(np.float32)) * w) < 2e-3
    rec = A.stft_inverse(mag, ph, inv, n_fft, hop, n_fft)
    assert rec.shape == g[tag + "_rec"].shape
    assert _rel(rec, g[tag + "_rec"]) < 1e-5
    if tag == "c":
        assert _rel(fwd, g["c_fwd_basis"]) < 1e-7
        assert _rel(inv, g["c_inv_basis"]) < 1e-6


def test_reconstruction_property():
    """transform -> inverse returns the input away from the edges (the property the reference's STFT.forward relies on)."""
    gen = torch.Generator().manual_seed(3)
    audio = torch.rand(2, 8192, generator=gen) - 0.5
    fwd, inv = A.stft_basis(1024, 256, 1024)
    mag, ph = A.stft_transform(audio, fwd)
    rec = A.stft_inverse(mag, ph, inv)[:, 0]
    assert rec.shape[1] == 8192
    assert _rel(rec[:, 1024:-1024], audio[:, 1024:-1024]) < 1e-4


def test_mel_filterbank_shape_and_normalisation():
    w = A.mel_filterbank(22050, 1024, 80, 0.0, 8000.0)
    assert w.shape == (80, 513) and w.dtype == np.float32
    assert (w >= 0).all()
    # Slaney area normalisation: every triangle integrates to ~1 Hz^-1 * bin width
    freqs = np.linspace(0, 22050 / 2, 513)
    area = (w * (freqs[1] - freqs[0])).sum(1)
    assert np.allclose(area, 1.0, atol=0.08)
    assert w[:, freqs > 8000.0 + 22].sum() == 0       # nothing above fmax
    peaks = freqs[w.argmax(1)]
    assert (np.diff(peaks) > 0).all()


def test_mel_filterbank_published_example_value():
    """The only published number available for `librosa.filters.mel` (absent here, unpinned upstream): its documentation prints
    `melfb = librosa.filters.mel(sr=22050, n_fft=2048)` with melfb[0][1] shown as 0.016 (three decimals), zeros at both ends of every
    row.  The restated Slaney filterbank gives 0.01618 there.  A weak pin - one rounded value - and labelled as such: the 80 x 513
    table the model uses stays "parity unpinned" (DESIGN.md section 6b, N3)."""
    w = A.mel_filterbank(22050, 2048, 128, 0.0, 11025.0)
    assert w.shape == (128, 1025)
    assert abs(round(float(w[0, 1]), 3) - 0.016) < 1e-9
    assert float(abs(w[:, 0]).max()) == 0.0 and float(abs(w[1:, 1]).max()) == 0.0 and float(w[-1, -1]) == 0.0


def test_mel_spectrogram_and_denoise_shapes():
    gen = torch.Generator().manual_seed(4)
    audio = torch.rand(2, 4096, generator=gen) * 1.8 - 0.9
    fwd, inv = A.stft_basis(1024, 256, 1024)
    mel = A.mel_spectrogram(audio, fwd, A.mel_filterbank(22050, 1024))
    assert mel.shape == (2, 80, 17) and float(mel.min()) >= np.log(1e-5) - 1e-6
    bias = torch.rand(1, 513, 1, generator=gen) * 0.01
    out = A.denoise(audio, bias, fwd, inv, strength=0.1)
    assert out.shape == (2, 1, 4096)
    # zero strength = plain reconstruction
    out0 = A.denoise(audio, bias, fwd, inv, strength=0.0)
    assert _rel(out0[:, 0, 1024:-1024], audio[:, 1024:-1024]) < 1e-4


def test_product_filterbank_and_bases_match_oracle_on_cpu():
    """Host-side construction in text2speech_amd/audio.py (mel filterbank, windowed Fourier bases) against the oracle's own
    restatement - no GPU involved; every call that would compute raises without one."""
    from text2speech_amd.audio import STFT, TacotronSTFT, mel_filterbank
    for sr, n_fft, n_mels, fmin, fmax in [(22050, 1024, 80, 0.0, 8000.0), (16000, 512, 40, 50.0, 7600.0)]:
        assert np.abs(mel_filterbank(sr, n_fft, n_mels, fmin, fmax) - A.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)).max() < 1e-7
    st = STFT(1024, 256, 1024)
    fwd, inv = A.stft_basis(1024, 256, 1024)
    assert torch.equal(st.forward_basis, fwd) and torch.equal(st.inverse_basis, inv)
    ts = TacotronSTFT(1024, 256, 1024, 80, 22050, 0.0, 8000.0)
    assert tuple(ts.mel_basis.shape) == (80, 513)
    with pytest.raises(RuntimeError):
        st.transform(torch.zeros(1, 4096))            # host tensor: there is no CPU path
    with pytest.raises(RuntimeError):
        ts.mel_spectrogram(torch.zeros(1, 4096))


def test_denoiser_and_griffin_lim_vs_reference(golden_dir):
    """The reference's own Denoiser (waveglow/denoiser.py:7-40) and griffin_lim (utils/audio_processing.py:51-67) outputs
    (tests/golden/audio_denoise_gl.npz, tools/gen_golden_audio.py) against the oracle's restatements, with the bias spectrum
    derived through the WaveGlow oracle exactly as Denoiser.__init__ derives it (zero mel [1, 80, 88], sigma = 0)."""
    from oracle import waveglow_oracle as O
    from text2speech_amd import synth
    g = np.load(os.path.join(golden_dir, "audio_denoise_gl.npz"))
    cfg = synth.WAVEGLOW_SMALL
    sd = synth.waveglow_state(cfg)
    fwd, inv = A.stft_basis(1024, 256, 1024)
    L = 88 * 256 // cfg["n_group"]
    with torch.no_grad():
        bias_audio = O.waveglow_infer(sd, cfg, torch.zeros(1, 80, 88), torch.zeros(1, 4, L),
                                      [torch.zeros(1, 2, L) for _ in range(2)], sigma=0.0)
        bias = A.stft_transform(bias_audio, fwd)[0][:, :, 0][:, :, None]
    assert bias.shape == g["bias_spec"].shape
    assert _rel(bias, g["bias_spec"]) < 1e-5
    gen = torch.Generator().manual_seed(44)
    clip = torch.rand(2, 4096, generator=gen) * 0.6 - 0.3
    for key, strength in (("denoised_s01", 0.1), ("denoised_s10", 1.0)):
        out = A.denoise(clip, torch.from_numpy(g["bias_spec"]), fwd, inv, strength=strength)
        assert out.shape == g[key].shape
        assert _rel(out, g[key]) < 1e-4, key
    # the clamp at zero is active at strength 1.0: the two outputs differ by more than a scale
    assert _rel(g["denoised_s10"], g["denoised_s01"]) > 1e-2
    # griffin_lim from the reference's captured initial phase
    sig = A.griffin_lim(torch.from_numpy(g["gl_mag"]), torch.from_numpy(g["gl_angles"]), fwd, inv, n_iters=6)
    assert sig.shape == g["gl_signal"].shape
    assert _rel(sig, g["gl_signal"]) < 2e-3

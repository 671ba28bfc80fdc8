"""CPU restatement of the reference's audio front-end / back-end.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product path (text2speech_amd/audio.py fails loudly without the HIP library).

What it restates, and where parity is pinned:
  * stft_basis / stft_transform / stft_inverse  <- utils/stft.py:36-134 (STFT as a strided conv1d against a windowed Fourier
    basis, inverse as conv_transpose1d + window-sum-square normalisation, utils/audio_processing.py:7-47).
    PINNED: tests/golden/audio_stft.npz holds outputs of the reference class itself, made by tools/gen_golden_audio.py.
    The reference imports three helpers from librosa (absent here, unpinned upstream): pad_center, tiny, normalize.  In the
    only configuration the repository uses (win_length == filter_length, norm=None) pad_center and normalize are the
    identity and tiny is numpy's float32 smallest normal; the generator supplies exactly those, nothing else.
  * mel_filterbank <- librosa.filters.mel as called at utils/layers.py:49-50 (Slaney scale, area-normalised triangles).
    librosa is an unpinned, un-vendored dependency: restated from its published algorithm, PARITY UNPINNED for the
    filterbank values (SURVEY.md 8c says the same).  mel_spectrogram on top of it (utils/layers.py:63-79) is a matmul and
    log(clamp(x, 1e-5)) (utils/audio_processing.py:78-84).
  * griffin_lim <- utils/audio_processing.py:50-72 (alternating projections over the two pinned transforms).
  * denoise <- waveglow/denoiser.py:10-40 (bias spectrum of the vocoder at zero input, spectral subtraction, inverse).
    The reference class hard-codes .cuda(); pinned through its two pinned pieces (transform, inverse).
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import get_window


def stft_basis(filter_length=1024, hop_length=256, win_length=1024, window="hann"):
    """utils/stft.py:41-70 -> (forward_basis [2*cutoff, 1, n_fft], inverse_basis [2*cutoff, 1, n_fft])."""
    scale = filter_length / hop_length
    fourier = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fourier = np.vstack([np.real(fourier[:cutoff, :]), np.imag(fourier[:cutoff, :])])
    fwd = torch.FloatTensor(fourier[:, None, :])
    inv = torch.FloatTensor(np.linalg.pinv(scale * fourier).T[:, None, :])
    if window is not None:
        assert filter_length >= win_length
        w = get_window(window, win_length, fftbins=True)
        lpad = (filter_length - win_length) // 2
        w = np.pad(w, (lpad, filter_length - win_length - lpad))
        w = torch.from_numpy(w).float()
        fwd = fwd * w
        inv = inv * w
    return fwd.float(), inv.float()


def stft_transform(audio, fwd_basis, filter_length=1024, hop_length=256):
    """utils/stft.py:72-99: audio [B, T] -> (magnitude, phase) [B, cutoff, 1 + T // hop]."""
    B, T = audio.shape
    x = F.pad(audio.view(B, 1, 1, T), (filter_length // 2, filter_length // 2, 0, 0), mode="reflect").squeeze(1)
    ft = F.conv1d(x, fwd_basis, stride=hop_length, padding=0)
    cutoff = filter_length // 2 + 1
    re, im = ft[:, :cutoff], ft[:, cutoff:]
    return torch.sqrt(re ** 2 + im ** 2), torch.atan2(im, re)


def window_sumsquare(window, n_frames, hop_length, win_length, n_fft):
    """utils/audio_processing.py:7-47 (norm=None)."""
    n = n_fft + hop_length * (n_frames - 1)
    x = np.zeros(n, dtype=np.float32)
    win_sq = get_window(window, win_length, fftbins=True) ** 2
    lpad = (n_fft - win_length) // 2
    win_sq = np.pad(win_sq, (lpad, n_fft - win_length - lpad))
    for i in range(n_frames):
        s = i * hop_length
        x[s:min(n, s + n_fft)] += win_sq[:max(0, min(n_fft, n - s))]
    return x


def stft_inverse(magnitude, phase, inv_basis, filter_length=1024, hop_length=256, win_length=1024, window="hann"):
    """utils/stft.py:101-129: -> audio [B, 1, hop * (frames - 1)]."""
    rc = torch.cat([magnitude * torch.cos(phase), magnitude * torch.sin(phase)], dim=1)
    out = F.conv_transpose1d(rc, inv_basis, stride=hop_length, padding=0)
    if window is not None:
        ws = window_sumsquare(window, magnitude.size(-1), hop_length, win_length, filter_length)
        nz = torch.from_numpy(np.where(ws > np.finfo(ws.dtype).tiny)[0])
        ws = torch.from_numpy(ws)
        out[:, :, nz] /= ws[nz]
        out *= float(filter_length) / hop_length
    out = out[:, :, filter_length // 2:]
    out = out[:, :, :-(filter_length // 2)]
    return out


def _hz_to_mel(f):
    """Slaney (auditory toolbox) scale: linear below 1 kHz, logarithmic above."""
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels=80, fmin=0.0, fmax=8000.0):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (htk=False, Slaney area normalisation)."""
    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def mel_spectrogram(audio, fwd_basis, mel_basis, filter_length=1024, hop_length=256):
    """utils/layers.py:63-79: audio [B, T] in [-1, 1] -> log-mel [B, n_mel, frames]."""
    assert float(audio.min()) >= -1 and float(audio.max()) <= 1
    mag, _ = stft_transform(audio, fwd_basis, filter_length, hop_length)
    mel = torch.matmul(torch.as_tensor(mel_basis), mag)
    return torch.log(torch.clamp(mel, min=1e-5))


def denoise(audio, bias_spec, fwd_basis, inv_basis, strength=0.1, filter_length=1024, hop_length=256, win_length=1024):
    """waveglow/denoiser.py:34-40: bias_spec [1, cutoff, 1]."""
    mag, ph = stft_transform(audio, fwd_basis, filter_length, hop_length)
    mag = torch.clamp(mag - bias_spec * strength, 0.0)
    return stft_inverse(mag, ph, inv_basis, filter_length, hop_length, win_length)


def griffin_lim(magnitudes, angles, fwd_basis, inv_basis, n_iters=30, filter_length=1024, hop_length=256, win_length=1024):
    """utils/audio_processing.py:50-72 with the initial phase given explicitly (the reference draws it with numpy)."""
    signal = stft_inverse(magnitudes, angles, inv_basis, filter_length, hop_length, win_length).squeeze(1)
    for _ in range(n_iters):
        _, angles = stft_transform(signal, fwd_basis, filter_length, hop_length)
        signal = stft_inverse(magnitudes, angles, inv_basis, filter_length, hop_length, win_length).squeeze(1)
    return signal

"""Audio front-end / back-end on the MI355X, behind the reference's own class surface:

  STFT(filter_length, hop_length, win_length, window)        utils/stft.py:36-134      .transform / .inverse / .forward
  TacotronSTFT(filter_length, hop_length, win_length, n_mel_channels, sampling_rate, mel_fmin, mel_fmax)
                                                             utils/layers.py:42-79     .mel_spectrogram
  Denoiser(waveglow, filter_length, n_overlap, win_length, mode)   waveglow/denoiser.py:7-40   .forward(audio, strength)
  dynamic_range_compression / _decompression                 utils/audio_processing.py:78-93

All arithmetic after construction runs in libt2s_hip.so (csrc/audio_ops.hip + the GEMV / f32 matrix-core GEMM of the
Tacotron path); there is no CPU or eager fallback - without the library every call raises.  The reference bounces the
waveform host -> GPU -> host inside STFT.transform (utils/stft.py:85-89); here input and output stay in HBM.
Bases are built on the host at construction exactly as the reference builds them (numpy FFT of the identity, pinv, scipy
window); the mel filterbank restates librosa.filters.mel's defaults (Slaney scale, area normalisation) because librosa is
an unpinned dependency that this image does not have.
"""
import numpy as np
import torch
from scipy.signal import get_window

from . import _lib

__all__ = ["STFT", "TacotronSTFT", "Denoiser", "griffin_lim", "mel_filterbank", "dynamic_range_compression",
           "dynamic_range_decompression"]


def _ru(a, b):
    return -(-a // b) * b


def _need_cuda(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError("%s: expected a tensor in HBM (cuda); there is no CPU path" % what)


def dynamic_range_compression(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression(x, C=1):
    return torch.exp(x) / C


def mel_filterbank(sr, n_fft, n_mels=80, fmin=0.0, fmax=None):
    """Triangular mel filters [n_mels, 1 + n_fft/2], Slaney scale and area normalisation (librosa.filters.mel defaults)."""
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    f_sp, brk = 200.0 / 3.0, 1000.0
    step = np.log(6.4) / 27.0

    def to_mel(hz):
        hz = np.asarray(hz, dtype=np.float64)
        return np.where(hz < brk, hz / f_sp, brk / f_sp + np.log(np.maximum(hz, 1e-10) / brk) / step)

    def to_hz(mel):
        mel = np.asarray(mel, dtype=np.float64)
        return np.where(mel < brk / f_sp, mel * f_sp, brk * np.exp(step * (mel - brk / f_sp)))

    edges = to_hz(np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2))          # n_mels + 2 band edges in Hz
    bins = np.linspace(0.0, float(sr) / 2, 1 + n_fft // 2)
    rise = (bins[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
    fall = (edges[2:, None] - bins[None, :]) / (edges[2:] - edges[1:-1])[:, None]
    tri = np.clip(np.minimum(rise, fall), 0.0, None)
    tri *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return tri.astype(np.float32)


def griffin_lim(magnitudes, stft_fn, n_iters=30, angles=None):
    """Phase reconstruction by alternating projections (reference utils/audio_processing.py:50-72): every transform and
    inverse runs on the device.  `angles` (radians, same shape as `magnitudes`) replaces the reference's random initial phase
    when given, which makes the result reproducible."""
    _need_cuda(magnitudes, "griffin_lim")
    if angles is None:
        angles = torch.rand(magnitudes.shape, device=magnitudes.device, dtype=torch.float32) * (2 * np.pi) - np.pi
    signal = stft_fn.inverse(magnitudes, angles).squeeze(1)
    for _ in range(n_iters):
        _, angles = stft_fn.transform(signal)
        signal = stft_fn.inverse(magnitudes, angles).squeeze(1)
    return signal


class STFT(torch.nn.Module):
    """Short-time Fourier transform as a strided contraction with a windowed Fourier basis (reference utils/stft.py:36)."""

    def __init__(self, filter_length=800, hop_length=200, win_length=800, window="hann"):
        super().__init__()
        if filter_length % 16 or hop_length % 4:
            raise ValueError("filter_length must be a multiple of 16 and hop_length a multiple of 4")
        self.filter_length, self.hop_length, self.win_length, self.window = filter_length, hop_length, win_length, window
        self.forward_transform = None
        scale = filter_length / hop_length
        fourier = np.fft.fft(np.eye(filter_length))
        cutoff = filter_length // 2 + 1
        fourier = np.vstack([fourier[:cutoff].real, fourier[:cutoff].imag])
        fwd = torch.FloatTensor(fourier[:, None, :])
        inv = torch.FloatTensor(np.linalg.pinv(scale * fourier).T[:, None, :])
        win_sq = None
        if window is not None:
            if filter_length < win_length:
                raise ValueError("filter_length < win_length")
            w = get_window(window, win_length, fftbins=True)
            lpad = (filter_length - win_length) // 2
            w = np.pad(w, (lpad, filter_length - win_length - lpad))        # librosa.util.pad_center
            wt = torch.from_numpy(w).float()
            fwd = fwd * wt
            inv = inv * wt
            win_sq = torch.from_numpy((w ** 2).astype(np.float32))
        self.register_buffer("forward_basis", fwd.float())
        self.register_buffer("inverse_basis", inv.float())
        # device-side operands: inverse basis transposed [n_fft][ld_rc] (zero padded), squared window
        self._ld_rc = _ru(2 * cutoff, 16)
        inv_t = torch.zeros(filter_length, self._ld_rc)
        inv_t[:, :2 * cutoff] = inv[:, 0, :].t()
        self.register_buffer("_inv_basis_t", inv_t, persistent=False)
        self.register_buffer("_win_sq", win_sq if win_sq is not None else torch.zeros(0), persistent=False)

    # -- helpers ------------------------------------------------------------------------------------------------------
    def _transform(self, input_data, want_mag=True, want_phase=True, ld_mt=0):
        _need_cuda(input_data, "STFT.transform")
        _need_cuda(self.forward_basis, "STFT (call .cuda() on the module)")
        x = input_data.detach().to(torch.float32).contiguous()
        B, T = x.shape
        n_fft, hop = self.filter_length, self.hop_length
        c, F = n_fft // 2 + 1, T // hop + 1
        dev = x.device
        ldp = _ru(T + n_fft, 4)
        xp = torch.empty(B, ldp, device=dev)
        ft = torch.empty(B, F, 2 * c, device=dev)
        mag = torch.empty(B, c, F, device=dev) if want_mag else None
        ph = torch.empty(B, c, F, device=dev) if want_phase else None
        magT = torch.empty(B * F, ld_mt, device=dev) if ld_mt else None
        p = _lib.ptr
        _lib.call("t2s_stft_transform", p(x), B, T, p(self.forward_basis), n_fft, hop, p(xp), ldp, p(ft),
                  p(mag) if mag is not None else None, p(ph) if ph is not None else None,
                  p(magT) if magT is not None else None, ld_mt, _lib.current_stream())
        self._keep = (x, xp, ft)
        return mag, ph, magT, F

    # -- reference surface --------------------------------------------------------------------------------------------
    def transform(self, input_data):
        """audio [B, T] -> (magnitude, phase), each [B, filter_length/2 + 1, 1 + T // hop_length]"""
        self.num_samples = input_data.size(1)
        mag, ph, _, _ = self._transform(input_data)
        return mag, ph

    def inverse(self, magnitude, phase, _bias=None, _strength=0.0):
        """(magnitude, phase) -> audio [B, 1, hop_length * (frames - 1)]"""
        _need_cuda(magnitude, "STFT.inverse")
        mag = magnitude.detach().to(torch.float32).contiguous()
        ph = phase.detach().to(torch.float32).contiguous()
        B, c, F = mag.shape
        n_fft, hop = self.filter_length, self.hop_length
        dev = mag.device
        rc = torch.empty(B * F, self._ld_rc, device=dev)
        frames = torch.empty(B, F, n_fft, device=dev)
        out = torch.empty(B, 1, hop * (F - 1), device=dev)
        p = _lib.ptr
        has_win = self.window is not None
        _lib.call("t2s_stft_inverse", p(mag), p(ph), B, F, n_fft, hop, p(self._inv_basis_t), self._ld_rc,
                  p(_bias) if _bias is not None else None, float(_strength), p(self._win_sq) if has_win else None,
                  float(np.finfo(np.float32).tiny), p(rc), p(frames), p(out), _lib.current_stream())
        self._keep_inv = (mag, ph, rc, frames, _bias)
        return out

    def forward(self, input_data):
        self.magnitude, self.phase = self.transform(input_data)
        return self.inverse(self.magnitude, self.phase)


class TacotronSTFT(torch.nn.Module):
    """log-mel spectrogram of the reference (utils/layers.py:42-79)."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=44800, mel_fmin=0.0,
                 mel_fmax=8000.0):
        super().__init__()
        self.n_mel_channels, self.sampling_rate = n_mel_channels, sampling_rate
        self.stft_fn = STFT(filter_length, hop_length, win_length)
        mel_basis = torch.from_numpy(mel_filterbank(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)).float()
        self.register_buffer("mel_basis", mel_basis)
        self._ld_mt = _ru(filter_length // 2 + 1, 16)
        mp = torch.zeros(n_mel_channels, self._ld_mt)
        mp[:, :mel_basis.size(1)] = mel_basis
        self.register_buffer("_mel_basis_p", mp, persistent=False)

    def spectral_normalize(self, magnitudes):
        return dynamic_range_compression(magnitudes)

    def spectral_de_normalize(self, magnitudes):
        return dynamic_range_decompression(magnitudes)

    def mel_spectrogram(self, y):
        """y [B, T] in [-1, 1] (in HBM) -> log-mel [B, n_mel_channels, 1 + T // hop]"""
        _need_cuda(y, "TacotronSTFT.mel_spectrogram")
        # the reference asserts the range on the host (utils/layers.py:72-73); one fused device reduction here
        lo, hi = torch.aminmax(y.detach())
        if float(lo) < -1 or float(hi) > 1:
            raise AssertionError("audio outside [-1, 1]")
        _, _, magT, F = self.stft_fn._transform(y, want_mag=False, want_phase=False, ld_mt=self._ld_mt)
        B = y.size(0)
        mel = torch.empty(B, self.n_mel_channels, F, device=y.device)
        p = _lib.ptr
        _lib.call("t2s_mel_from_mag", p(magT), self._ld_mt, B, F, p(self._mel_basis_p), self.n_mel_channels, 1e-5, p(mel),
                  _lib.current_stream())
        self._keep = magT
        return mel


class Denoiser(torch.nn.Module):
    """Removes the vocoder's bias spectrum from its output (waveglow/denoiser.py:7-40)."""

    def __init__(self, waveglow, filter_length=1024, n_overlap=4, win_length=1024, mode="zeros"):
        super().__init__()
        w = waveglow.upsample.weight
        self.stft = STFT(filter_length=filter_length, hop_length=int(filter_length / n_overlap), win_length=win_length).to(w.device)
        if mode == "zeros":
            mel_input = torch.zeros((1, 80, 88), dtype=w.dtype, device=w.device)
        elif mode == "normal":
            mel_input = torch.randn((1, 80, 88), dtype=w.dtype, device=w.device)
        else:
            raise Exception("Mode {} if not supported".format(mode))
        with torch.no_grad():
            bias_audio = waveglow.infer(mel_input, sigma=0.0).float()
            bias_spec, _ = self.stft.transform(bias_audio)
        self.register_buffer("bias_spec", bias_spec[:, :, 0][:, :, None].contiguous())

    def forward(self, audio, strength=0.1):
        _need_cuda(audio, "Denoiser.forward")
        mag, ph = self.stft.transform(audio.float())
        # spectral subtraction + clamp are fused into the recombination kernel of the inverse transform
        return self.stft.inverse(mag, ph, _bias=self.bias_spec.reshape(-1).contiguous(), _strength=strength)

"""BASELINE configs[4]: end-to-end inference on one MI355X - Tacotron-2 autoregressive decode feeding the
WaveGlow vocoder (reference inference.py:80-95: model.inference(sequence) -> waveglow.infer(mel_postnet,
sigma=0.666)) - against the two CPU oracles chained the same way, with identical dropout masks and noise."""
import pytest
import torch

from text2speech_amd import _lib, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_text_to_audio_matches_chained_oracles():
    from oracle import tacotron_oracle as TO
    from oracle import waveglow_oracle as WO
    from text2speech_amd.glow import WaveGlow
    from text2speech_amd.tacotron import Tacotron
    _lib.load()
    hp = dict(synth.TACOTRON_HPARAMS)
    n_frames = 96
    tsd = synth.tacotron_state()
    cfg = synth.WAVEGLOW_DEFAULT
    wsd = synth.waveglow_state(cfg)
    taco = Tacotron(hp, 80, num_speakers=2)
    taco.load_state_dict(tsd)
    taco = taco.to(DEV).eval()
    wg = WaveGlow(**cfg)
    wg.load_state_dict(wsd)
    wg = WaveGlow.remove_weightnorm(wg).to(DEV).eval()      # as reference waveglow/inference.py:38 does
    ids = (torch.arange(48) % 78 + 2)[None]
    gen = torch.Generator().manual_seed(17)
    masks = (torch.rand(n_frames, 1, 2, 256, generator=gen) < 0.5)
    L = n_frames * 256 // 8
    nf = torch.randn(1, 4, L, generator=gen)
    ne = [torch.randn(1, 2, L, generator=gen) for _ in range(2)]
    taco.decoder.gate_threshold, taco.decoder.max_decoder_steps = 2.0, n_frames
    mel, mel_post, gate, align = taco.inference(ids.to(DEV), None, prenet_masks=masks.to(torch.uint8))
    audio = wg.infer(mel_post, sigma=0.666, noise=(nf, ne))
    assert tuple(audio.shape) == (1, n_frames * 256)
    with torch.no_grad():
        o = TO.tacotron_inference(tsd, hp, ids, n_frames, masks.float())
        # remove_weightnorm must not change the function: the oracle runs on the weight-normed state
        audio_o = WO.waveglow_infer(wsd, cfg, o[1], nf, ne, sigma=0.666)
    assert _rel(mel_post, o[1]) < 1e-3
    assert _rel(audio, audio_o) < 1e-3
    assert float((audio.cpu().double() - audio_o.double()).abs().max() / audio_o.double().abs().max()) < 2e-3

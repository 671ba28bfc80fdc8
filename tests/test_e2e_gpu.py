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


def test_half_precision_call_order_of_the_reference_script():
    """The reference's inference.py:59-94 call order, verbatim: Tacotron(...).cuda().eval().half(); the vocoder .half() with
    its invertible 1x1 convolutions put back to float (inference.py:73-74), `padding_mode` set on every Conv module
    (inference.py:69-71), Denoiser(waveglow), model.inference(sequence, speaker_id) -> waveglow.infer(mel_postnet, 0.666).
    The kernels compute in f32 from the half-rounded weights, so the yardstick is the pair of f32 oracles run on the same
    half-rounded weights; outputs come back as half tensors like the reference's."""
    import numpy as np
    from oracle import tacotron_oracle as TO
    from oracle import waveglow_oracle as WO
    from text2speech_amd.audio import Denoiser
    from text2speech_amd.glow import WaveGlow
    from text2speech_amd.tacotron import Tacotron
    from text2speech_amd.text import symbols, text_to_sequence
    _lib.load()
    hp = dict(synth.TACOTRON_HPARAMS)
    n_frames = 40
    tsd = synth.tacotron_state()
    cfg = synth.WAVEGLOW_SMALL
    wsd = synth.waveglow_state(cfg)
    model = Tacotron(hp, len(symbols), num_speakers=2)
    model.load_state_dict(tsd)
    _ = model.cuda().eval().half()
    waveglow = WaveGlow(**cfg)
    waveglow.load_state_dict(wsd)
    waveglow.cuda().eval().half()
    for m in waveglow.modules():
        if "Conv" in str(type(m)):
            setattr(m, "padding_mode", "zeros")
    for k in waveglow.convinv:
        k.float()
    denoiser = Denoiser(waveglow)
    assert denoiser.bias_spec.dtype == torch.float32 and bool(torch.isfinite(denoiser.bias_spec).all())
    sequence = np.array(text_to_sequence("존경하는 국민 여러분, 2017년 9월 12일입니다."))[None, :]
    sequence = torch.from_numpy(sequence).cuda().long()
    speaker_id = torch.from_numpy(np.array([0]).reshape(1, -1)).cuda().long()
    gen = torch.Generator().manual_seed(23)
    masks = (torch.rand(n_frames, 1, 2, 256, generator=gen) < 0.5)
    L = n_frames * 256 // 8
    nf = torch.randn(1, 4, L, generator=gen)
    ne = [torch.randn(1, 2, L, generator=gen) for _ in range(2)]
    model.decoder.gate_threshold, model.decoder.max_decoder_steps = 2.0, n_frames
    mel_outputs, mel_outputs_postnet, _, alignments = model.inference(sequence, speaker_id, prenet_masks=masks.to(torch.uint8))
    assert mel_outputs_postnet.dtype == torch.float16 and alignments.dtype == torch.float16
    with torch.no_grad():
        audio = waveglow.infer(mel_outputs_postnet, sigma=0.666, noise=(nf, ne))
    assert audio.dtype == torch.float16 and tuple(audio.shape) == (1, n_frames * 256)
    den = denoiser(audio, strength=0.01)
    assert tuple(den.shape) == (1, 1, n_frames * 256) and bool(torch.isfinite(den).all())
    # oracles on the half-rounded weights (convinv stays float, as in the script)
    h = lambda sd, keep=(): {k: (v if (not v.is_floating_point() or any(s in k for s in keep)) else v.half().float())
                             for k, v in sd.items()}
    with torch.no_grad():
        o = TO.tacotron_inference(h(tsd), hp, sequence.cpu(), n_frames, masks.float())
        assert _rel(mel_outputs_postnet.float(), o[1]) < 2e-3
        audio_o = WO.waveglow_infer(h(wsd, keep=("convinv",)), cfg, mel_outputs_postnet.float().cpu(), nf, ne, sigma=0.666)
    assert _rel(audio.float(), audio_o) < 2e-3

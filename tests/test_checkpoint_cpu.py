"""CPU: checkpoint I/O (text2speech_amd/checkpoint.py) - state_dict layout round trip for both models, and loading a
legacy WaveGlow checkpoint that the reference's own code pickled as a whole object (tests/golden/legacy_waveglow_ckpt.pt,
written by tools/gen_golden_checkpoint.py)."""
import os

import torch

from text2speech_amd import checkpoint, synth

LEGACY_CFG = dict(n_mel_channels=8, n_flows=4, n_group=8, n_early_every=2, n_early_size=2,
                  WN_config=dict(n_layers=2, n_channels=8, kernel_size=3))


def test_legacy_object_pickle_loads_into_native_model(golden_dir):
    import sys
    from text2speech_amd.glow import WaveGlow
    path = os.path.join(golden_dir, "legacy_waveglow_ckpt.pt")
    before = sys.modules.get("glow")
    m = checkpoint.load_waveglow(path, device="cpu")
    assert sys.modules.get("glow") is before                  # the alias does not leak
    assert isinstance(m, WaveGlow)
    want = synth.waveglow_state(LEGACY_CFG, seed=77)
    got = m.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    assert m.n_flows == 4 and m.n_group == 8 and m.n_early_every == 2 and m.n_early_size == 2
    # reference-style loader into an existing model + optimizer
    m2 = WaveGlow(**LEGACY_CFG)
    opt = torch.optim.Adam(m2.parameters(), lr=1e-4)
    m2, opt, it = checkpoint.load_checkpoint(path, m2, opt)
    assert it == 4321
    assert torch.equal(m2.state_dict()["WN.1.start.weight_v"], want["WN.1.start.weight_v"])


def test_config_recovered_from_shapes_only():
    sd = synth.waveglow_state(synth.WAVEGLOW_SMALL)
    cfg = checkpoint.waveglow_config_from_state_dict(sd)
    assert cfg == synth.WAVEGLOW_SMALL


def test_state_dict_round_trip_waveglow(tmp_path):
    from text2speech_amd.glow import WaveGlow
    m = WaveGlow(**LEGACY_CFG)
    m.load_state_dict(synth.waveglow_state(LEGACY_CFG, seed=5))
    opt = torch.optim.Adam(m.parameters(), lr=3e-4)
    p = str(tmp_path / "wg.pt")
    checkpoint.save_checkpoint(m, opt, 3e-4, 17, p, config=LEGACY_CFG)
    assert not os.path.exists(p + ".tmp")
    ck = torch.load(p, map_location="cpu", weights_only=True)     # plain tensors and containers only
    assert ck["format"] == checkpoint.FORMAT and ck["iteration"] == 17
    m2 = checkpoint.load_waveglow(p, device="cpu")
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v), k


def test_state_dict_round_trip_tacotron(tmp_path):
    from text2speech_amd.tacotron import Tacotron
    hp = synth.TACOTRON_HPARAMS
    m = Tacotron(hp, 80, num_speakers=2)
    m.load_state_dict(synth.tacotron_state())
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    p = str(tmp_path / "taco.pt")
    checkpoint.save_checkpoint(m, opt, 1e-3, 99, p)
    m2 = Tacotron(hp, 80, num_speakers=2)
    m2, _, it = checkpoint.load_checkpoint(p, m2, torch.optim.Adam(m2.parameters(), lr=1e-3))
    assert it == 99
    for k, v in m.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v), k
    # the reference's own Tacotron layout (train.py:72-75) loads too
    p2 = str(tmp_path / "ref_layout.pt")
    torch.save({"iteration": 5, "state_dict": m.state_dict(), "optimizer": opt.state_dict(), "learning_rate": 1e-3}, p2)
    _, _, it2 = checkpoint.load_checkpoint(p2, Tacotron(hp, 80, num_speakers=2), None)
    assert it2 == 5

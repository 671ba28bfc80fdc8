"""The CPU oracle (oracle/waveglow_oracle.py) against vectors produced by the
reference itself (tools/gen_golden_waveglow.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import waveglow_oracle as O
from text2speech_amd import synth


def _rel(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("name,batch,n,seed", [
    ("waveglow_small_fwd", 2, 4096, 31),
    ("waveglow_small_ragged_fwd", 3, 2400, 32),
])
def test_forward_small(golden_dir, name, batch, n, seed):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = synth.WAVEGLOW_SMALL
    sd = synth.waveglow_state(cfg)
    mel, audio = synth.waveglow_inputs(batch, n, seed=seed)
    with torch.no_grad():
        z, log_s, log_det = O.waveglow_forward(sd, cfg, mel, audio)
        loss = O.waveglow_loss((z, log_s, log_det))
    assert _rel(z, g["z"]) < 2e-6
    assert abs(float(loss) - float(g["loss"])) < 1e-6
    np.testing.assert_allclose([float(d) for d in log_det], g["log_det"], rtol=1e-5, atol=1e-3)
    for k, ls in enumerate(log_s):
        step = max(1, ls.size(2) // 64)
        assert _rel(ls[:, :, ::step], g[f"log_s_{k}"]) < 5e-6


@pytest.mark.parametrize("name,sigma", [("waveglow_small_infer_s0", 0.0), ("waveglow_small_infer_s0666", 0.666)])
def test_infer_small(golden_dir, name, sigma):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = synth.WAVEGLOW_SMALL
    sd = synth.waveglow_state(cfg)
    gen = torch.Generator().manual_seed(41)
    mel = torch.randn(2, 80, 12, generator=gen)
    early = [torch.from_numpy(g[f"noise_early_{i}"]) for i in range(2)]
    with torch.no_grad():
        audio = O.waveglow_infer(sd, cfg, mel, torch.from_numpy(g["noise_final"]), early, sigma=sigma)
    assert audio.shape == g["audio"].shape
    assert _rel(audio, g["audio"]) < 5e-6


def test_roundtrip_property():
    """forward o infer == identity (holds to ~4e-7 on the reference)."""
    cfg = synth.WAVEGLOW_SMALL
    sd = synth.waveglow_state(cfg)
    gen = torch.Generator().manual_seed(5)
    mel = torch.randn(1, 80, 8, generator=gen)
    L = 8 * 256 // 8
    nf = torch.randn(1, 4, L, generator=gen)
    ne = [torch.randn(1, 2, L, generator=gen) for _ in range(2)]
    with torch.no_grad():
        audio = O.waveglow_infer(sd, cfg, mel, nf, ne, sigma=1.0)
        # forward needs spect trimmed to the audio length: mel has 8 frames -> 2816 >= 2048
        z, _, _ = O.waveglow_forward(sd, cfg, mel, audio)
    want = torch.cat([ne[1], ne[0], nf], 1)
    assert _rel(z, want) < 1e-5


@pytest.mark.slow
def test_grads_config_defaults_vs_reference(golden_dir):
    """config.json defaults (512 channels), B=2 x 4096: loss and the gradient of EVERY parameter (sum and sum of squares)
    plus strided samples of one tensor of every kind in flows 0, 5, 11 against the reference's own backward
    (tests/golden/waveglow_full_grads.npz, tools/gen_golden_waveglow.py)."""
    g = np.load(os.path.join(golden_dir, "waveglow_full_grads.npz"))
    cfg = synth.WAVEGLOW_DEFAULT
    sd = {k: v.clone().requires_grad_(True) for k, v in synth.waveglow_state(cfg).items()}
    mel, audio = synth.waveglow_inputs(2, 4096, seed=33)
    out = O.waveglow_forward(sd, cfg, mel, audio)
    loss = O.waveglow_loss(out)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    names = [str(n) for n in g["all_names"]]
    assert len(names) == len(sd) == 938
    for n, gs, gq in zip(names, g["all_gradsum"], g["all_gradsq"]):
        gr = sd[n].grad.double()
        assert abs(float((gr ** 2).sum()) - gq) <= 2e-4 * gq + 1e-12, n
        assert abs(float(gr.sum()) - gs) <= 2e-3 * float(gr.abs().sum()) + 1e-9, n
    for key in g.files:
        if key.startswith("grad::"):
            name = key[len("grad::"):]
            flat = sd[name].grad.flatten()
            step = max(1, flat.numel() // 4096)
            assert _rel(flat[::step], g[key]) < 2e-4, name


def full_infer_case(golden_dir):
    """Inputs of tests/golden/waveglow_full_infer_1000.npz (tools/gen_golden_waveglow.py --infer-full) regenerated from its seed:
    mel from a seeded generator, the Gaussian draws from the global CPU RNG in the reference's draw order (glow.py:260-267, 284-289),
    checked against the fixture's checksums before they are used."""
    g = np.load(os.path.join(golden_dir, "waveglow_full_infer_1000.npz"))
    cfg = synth.WAVEGLOW_DEFAULT
    seed, frames = int(g["seed"]), int(g["frames"])
    mel = torch.randn(1, cfg["n_mel_channels"], frames, generator=torch.Generator().manual_seed(seed))
    L = frames * 256 // cfg["n_group"]
    torch.manual_seed(seed + 1)
    n_rem = cfg["n_group"] - cfg["n_early_size"] * sum(1 for k in range(cfg["n_flows"]) if k % cfg["n_early_every"] == 0 and k > 0)
    draws = [torch.FloatTensor(1, n_rem, L).normal_()]
    for k in reversed(range(cfg["n_flows"])):
        if k % cfg["n_early_every"] == 0 and k > 0:
            draws.append(torch.FloatTensor(1, cfg["n_early_size"], L).normal_())
    for i, t in enumerate(draws):       # last-bit differences between CPU vector paths are fine; a different stream is not
        assert abs(float(t.double().sum()) - g["noise_sum"][i]) < 1e-3 and abs(float((t.double() ** 2).sum()) - g["noise_sq"][i]) < 1e-5 * g["noise_sq"][i], \
            "the CPU RNG on this box does not reproduce the reference's Gaussian draws"
    return g, cfg, mel, draws[0], draws[1:]


def check_full_infer(audio, g, tol):
    audio = torch.as_tensor(audio).detach().cpu().reshape(-1)
    n = audio.numel()
    assert n == 256 * int(g["frames"])
    stride = int(g["stride"])
    errs = {"strided": _rel(audio[::stride], g["audio_strided"]), "head": _rel(audio[:2048], g["audio_head"]),
            "tail": _rel(audio[-2048:], g["audio_tail"])}
    sq = np.array([float((audio[i:i + 4096].double() ** 2).sum()) for i in range(0, n, 4096)])
    errs["block_energy"] = float(np.max(np.abs(sq - g["audio_sq_by_4096"]) / g["audio_sq_by_4096"]))
    errs["max_sample"] = float((audio[::stride].double() - torch.from_numpy(g["audio_strided"]).double()).abs().max() / float(g["audio_absmax"]))
    assert errs["strided"] < tol and errs["head"] < tol and errs["tail"] < tol and errs["max_sample"] < tol, errs
    assert errs["block_energy"] < 2 * tol, errs
    return errs


@pytest.mark.slow
def test_infer_benchmarked_length_vs_reference(golden_dir):
    """The oracle's `infer` at the length bench.py and tools/bench_e2e.py time (B = 1, 512 channels, 1000 frames, sigma 0.666)
    against the reference's own audio (strided samples, head, tail, per-4096-sample energies)."""
    g, cfg, mel, nf, ne = full_infer_case(golden_dir)
    with torch.no_grad():
        audio = O.waveglow_infer(synth.waveglow_state(cfg), cfg, mel, nf, ne, sigma=float(g["sigma"]))
    print(check_full_infer(audio, g, 1e-5))

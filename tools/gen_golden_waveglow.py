#!/usr/bin/env python3
"""Generate WaveGlow golden vectors by running the REFERENCE implementation.

Runs only in the build container (needs /root/reference).  The reference is
imported read-only with PYTHONDONTWRITEBYTECODE=1; nothing of it is copied:
the outputs written to tests/golden/*.npz are data (inputs are regenerated from
seeds by text2speech_amd.synth, so only expected outputs are stored).

Shims applied (device placement only, never arithmetic):
  * torch.cuda.FloatTensor = torch.FloatTensor for WaveGlow.infer
    (reference glow.py:261-267,286-288 build their noise on CUDA types).

usage: PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden_waveglow.py [--full]
"""
import argparse
import os
import sys
import warnings

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/waveglow")

import glow as ref_glow  # noqa: E402  (the reference)
from text2speech_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def build_ref(cfg, sd):
    torch.manual_seed(0)
    m = ref_glow.WaveGlow(**cfg)
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    return m


GRAD_KEYS_SMALL = ["WN.0.in_layers.0.weight_v", "WN.0.in_layers.0.weight_g", "WN.3.cond_layers.7.weight_v",
                   "WN.11.res_skip_layers.2.weight_v", "WN.5.start.weight_g", "WN.7.end.weight",
                   "convinv.2.conv.weight", "upsample.weight", "WN.11.in_layers.7.bias"]


def grad_keys_full():
    """config.json defaults (512 channels): one tensor of every kind in flows 0, 5 and 11, first / middle / last layer."""
    keys = ["upsample.weight", "upsample.bias"]
    for k in (0, 5, 11):
        keys += [f"convinv.{k}.conv.weight", f"WN.{k}.start.weight_v", f"WN.{k}.start.weight_g", f"WN.{k}.start.bias",
                 f"WN.{k}.end.weight", f"WN.{k}.end.bias"]
        for i in (0, 3, 7):
            for kind in ("in_layers", "cond_layers", "res_skip_layers"):
                keys += [f"WN.{k}.{kind}.{i}.weight_v", f"WN.{k}.{kind}.{i}.weight_g", f"WN.{k}.{kind}.{i}.bias"]
    return keys


def run_forward(name, cfg, batch, n_samples, seed, store_z=True, grads=False, grad_keys=None, sample=32768, all_norms=False):
    sd = synth.waveglow_state(cfg)
    mel, audio = synth.waveglow_inputs(batch, n_samples, seed=seed)
    m = build_ref(cfg, sd)
    out = {}
    if grads:
        m.train()
        z, log_s, log_det = m((mel, audio))
        loss = ref_glow.WaveGlowLoss(1.0)((z, log_s, [d.clone() for d in log_det]))
        loss.backward()
        named = dict(m.named_parameters())
        # torch >= 2 exposes weight_norm params under their original names
        for key in (grad_keys or GRAD_KEYS_SMALL):
            gflat = named[key].grad.detach().flatten()
            step = max(1, gflat.numel() // sample)       # strided sample keeps fixtures small
            out["grad::" + key] = gflat[::step].contiguous().numpy()
            out["gradsum::" + key] = np.float64(gflat.double().sum().item())
            out["gradsq::" + key] = np.float64((gflat.double() ** 2).sum().item())
        if all_norms:      # every parameter's gradient is pinned by its sum and sum of squares (two scalars per tensor)
            names = sorted(named)
            out["all_names"] = np.array(names)
            out["all_gradsum"] = np.array([named[n].grad.double().sum().item() for n in names], dtype=np.float64)
            out["all_gradsq"] = np.array([(named[n].grad.double() ** 2).sum().item() for n in names], dtype=np.float64)
        out["loss"] = np.float64(loss.item())
    else:
        with torch.no_grad():
            z, log_s, log_det = m((mel, audio))
            loss = ref_glow.WaveGlowLoss(1.0)((z, log_s, [d.clone() for d in log_det]))
        out["loss"] = np.float64(loss.item())
    if store_z:
        out["z"] = z.detach().numpy()
    out["z_sum"] = np.float64(z.double().sum().item())
    out["z_sqsum"] = np.float64((z.double() ** 2).sum().item())
    out["log_det"] = np.array([float(d) for d in log_det], dtype=np.float64)
    out["log_s_sum"] = np.array([float(ls.double().sum()) for ls in log_s], dtype=np.float64)
    # a strided sample of every flow's log_s pins each flow individually
    for k, ls in enumerate(log_s):
        out[f"log_s_{k}"] = ls.detach()[:, :, ::max(1, ls.size(2) // 64)].contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", out["loss"], "z std", float(z.std()))


def run_train_full(name, cfg, batch, n_samples, seed, sample=2048):
    """BASELINE configs[3] at the BENCHMARKED shape (8 x 16000, config.json defaults): loss and every parameter gradient of the
    reference.  WaveGlow has no coupling between batch entries (no BatchNorm; the loss is a sum over entries divided by the
    element count), so the batch gradient is the mean of the per-entry gradients: `batch` reference backward passes at B = 1
    (a few GB each instead of ~40 GB at once), accumulated in f64."""
    sd = synth.waveglow_state(cfg)
    mel, audio = synth.waveglow_inputs(batch, n_samples, seed=seed)
    m = build_ref(cfg, sd)
    m.train()
    named = dict(m.named_parameters())
    names = sorted(named)
    acc = {n: torch.zeros_like(named[n], dtype=torch.float64) for n in names}
    loss_sum = 0.0
    for b in range(batch):
        m.zero_grad(set_to_none=True)
        z, log_s, log_det = m((mel[b:b + 1], audio[b:b + 1]))
        loss = ref_glow.WaveGlowLoss(1.0)((z, log_s, [d.clone() for d in log_det]))
        loss.backward()
        loss_sum += float(loss)
        for n in names:
            acc[n] += named[n].grad.double()
        print("  entry", b, "loss", float(loss), flush=True)
    out = {"loss": np.float64(loss_sum / batch), "all_names": np.array(names)}
    out["all_gradsum"] = np.array([(acc[n] / batch).sum().item() for n in names], dtype=np.float64)
    out["all_gradsq"] = np.array([((acc[n] / batch) ** 2).sum().item() for n in names], dtype=np.float64)
    for key in grad_keys_full():
        gflat = (acc[key] / batch).flatten()
        step = max(1, gflat.numel() // sample)
        out["grad::" + key] = gflat[::step].float().contiguous().numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "loss", out["loss"])


def run_infer(name, cfg, batch, frames, seed, sigma):
    sd = synth.waveglow_state(cfg)
    gen = torch.Generator().manual_seed(seed)
    mel = torch.randn(batch, cfg["n_mel_channels"], frames, generator=gen)
    m = build_ref(cfg, sd)
    torch.cuda.FloatTensor = torch.FloatTensor      # device shim only
    L = frames * 256 // cfg["n_group"]
    # capture the reference's draws: replay the same global-RNG stream
    torch.manual_seed(seed + 1)
    noise_final = torch.FloatTensor(batch, m.n_remaining_channels, L).normal_()
    noise_early = []
    for k in reversed(range(cfg["n_flows"])):
        if k % cfg["n_early_every"] == 0 and k > 0:
            noise_early.append(torch.FloatTensor(batch, cfg["n_early_size"], L).normal_())
    torch.manual_seed(seed + 1)
    with torch.no_grad():
        audio = m.infer(mel, sigma=sigma)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), audio=audio.numpy(),
                        noise_final=noise_final.numpy(),
                        **{f"noise_early_{i}": n.numpy() for i, n in enumerate(noise_early)})
    print(name, "audio std", float(audio.std()), tuple(audio.shape))


def run_infer_full(name, cfg, frames, seed, sigma, sample=4096):
    """BASELINE configs[4], vocoder half, at the BENCHMARKED length (B = 1, config.json defaults, `frames` mel frames ->
    256 * frames samples): the reference's own `infer`.  The fixture holds what pins the audio without shipping it: strided
    samples, the first / last 2048 samples, the energy of every 4096-sample block, and checksums of the Gaussian draws (the test
    regenerates them from the seed, in the reference's draw order, and checks these before using them)."""
    sd = synth.waveglow_state(cfg)
    gen = torch.Generator().manual_seed(seed)
    mel = torch.randn(1, cfg["n_mel_channels"], frames, generator=gen)
    m = build_ref(cfg, sd)
    torch.cuda.FloatTensor = torch.FloatTensor      # device shim only
    L = frames * 256 // cfg["n_group"]
    torch.manual_seed(seed + 1)
    draws = [torch.FloatTensor(1, m.n_remaining_channels, L).normal_()]
    for k in reversed(range(cfg["n_flows"])):
        if k % cfg["n_early_every"] == 0 and k > 0:
            draws.append(torch.FloatTensor(1, cfg["n_early_size"], L).normal_())
    torch.manual_seed(seed + 1)
    with torch.no_grad():
        audio = m.infer(mel, sigma=sigma)[0]
    n = audio.numel()
    wsum = lambda t: float((t.flatten().double() * (torch.arange(t.numel(), dtype=torch.float64) % 9973 + 1)).sum())
    out = {"seed": np.int64(seed), "sigma": np.float64(sigma), "frames": np.int64(frames),
           "noise_sum": np.array([float(t.double().sum()) for t in draws]),
           "noise_sq": np.array([float((t.double() ** 2).sum()) for t in draws]),
           "noise_wsum": np.array([wsum(t) for t in draws]),
           "audio_strided": audio[::max(1, n // sample)].contiguous().numpy(), "stride": np.int64(max(1, n // sample)),
           "audio_head": audio[:2048].numpy(), "audio_tail": audio[-2048:].numpy(),
           "audio_sq_by_4096": np.array([float((audio[i:i + 4096].double() ** 2).sum()) for i in range(0, n, 4096)]),
           "audio_absmax": np.float64(float(audio.abs().max()))}
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "audio std", float(audio.std()), "absmax", float(audio.abs().max()), n)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--infer-full", action="store_true", help="ONLY the 512-channel, 1000-frame infer fixture (~1 min)")
    ap.add_argument("--full", action="store_true", help="also run the 512-channel 8x16000 config (~1 min)")
    ap.add_argument("--train-full", action="store_true", help="ONLY the 8x16000 training-step fixture (8 reference backward passes)")
    args = ap.parse_args()
    if args.infer_full:
        torch.set_num_threads(8)
        run_infer_full("waveglow_full_infer_1000", synth.WAVEGLOW_DEFAULT, 1000, seed=51, sigma=0.666)
        sys.exit(0)
    if args.train_full:
        torch.set_num_threads(8)
        run_train_full("waveglow_train_full_grads", synth.WAVEGLOW_DEFAULT, 8, 16000, seed=31)
        sys.exit(0)
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    run_forward("waveglow_small_fwd", synth.WAVEGLOW_SMALL, 2, 4096, seed=31)
    run_forward("waveglow_small_ragged_fwd", synth.WAVEGLOW_SMALL, 3, 2400, seed=32)
    run_forward("waveglow_small_grads", synth.WAVEGLOW_SMALL, 2, 4096, seed=31, store_z=False, grads=True)
    run_infer("waveglow_small_infer_s0", synth.WAVEGLOW_SMALL, 2, 12, seed=41, sigma=0.0)
    run_infer("waveglow_small_infer_s0666", synth.WAVEGLOW_SMALL, 2, 12, seed=41, sigma=0.666)
    # BASELINE configs[3] arithmetic (config.json defaults, 512 channels) on a short segment: loss + gradients
    run_forward("waveglow_full_grads", synth.WAVEGLOW_DEFAULT, 2, 4096, seed=33, store_z=False, grads=True,
                grad_keys=grad_keys_full(), sample=4096, all_norms=True)
    if args.full:
        run_forward("waveglow_full_fwd", synth.WAVEGLOW_DEFAULT, 8, 16000, seed=31)

"""ORACLE — test infrastructure, not product code.

CPU restatement (stock PyTorch fp32/fp64 ops, no custom kernels) of the
reference WaveGlow path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product package
``text2speech_amd`` never does.

Parity status: PINNED.  ``tools/gen_golden_waveglow.py`` imports the reference
(`/root/reference/waveglow/glow.py`) in the build container, loads the same
seeded weights (``text2speech_amd.synth``) and writes ``tests/golden/waveglow_*.npz``;
``tests/test_oracle_waveglow_golden.py`` checks this restatement against those vectors.

Each function cites the reference lines it restates.  It is written as pure
functions over a ``state_dict`` (reference key names), not as a module tree.
"""
import torch
import torch.nn.functional as F


def _flow_sizes(cfg):
    # reference glow.py:193-205
    n_half = cfg["n_group"] // 2
    n_rem = cfg["n_group"]
    out = []
    for k in range(cfg["n_flows"]):
        if k % cfg["n_early_every"] == 0 and k > 0:
            n_half -= cfg["n_early_size"] // 2
            n_rem -= cfg["n_early_size"]
        out.append((n_rem, n_half))
    return out


def effective_weight(sd, prefix):
    """weight_norm over dim 0: w = g * v / ||v|| (reference glow.py:123,138,142,151;
    torch.nn.utils.weight_norm default dim=0).  Falls back to a plain ``weight``
    (after remove_weightnorm, glow.py:294-310)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    v = sd[prefix + ".weight_v"]
    g = sd[prefix + ".weight_g"]
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def upsample_and_squeeze(sd, cfg, mel, n_samples=None, trim_tail=False):
    """ConvTranspose1d(k=1024, stride=256) then group-by-n_group channel fold.
    forward: glow.py:215-223 (trim to audio length); infer: glow.py:252-258
    (drop the last kernel-stride samples)."""
    spect = F.conv_transpose1d(mel, sd["upsample.weight"], sd["upsample.bias"], stride=256)
    if trim_tail:
        spect = spect[:, :, :-(1024 - 256)]
    else:
        assert spect.size(2) >= n_samples
        spect = spect[:, :, :n_samples]
    g = cfg["n_group"]
    B, M, T = spect.shape
    L = T // g
    # channel index = mel*g + phase  (glow.py:220-221)
    spect = spect[:, :, :L * g].reshape(B, M, L, g).permute(0, 1, 3, 2).reshape(B, M * g, L)
    return spect


def wn_forward(sd, cfg, k, audio_0, spect, taps=None):
    """WN coupling network, reference glow.py:154-175 (start, n_layers x
    [dilated conv + cond 1x1 -> tanh*sigmoid gate -> res/skip 1x1], end)."""
    wn = cfg["WN_config"]
    C, nl, ks = wn["n_channels"], wn["n_layers"], wn["kernel_size"]
    p = f"WN.{k}."
    x = F.conv1d(audio_0, effective_weight(sd, p + "start"), sd[p + "start.bias"])
    skip = None
    for i in range(nl):
        d = 2 ** i
        a = F.conv1d(x, effective_weight(sd, p + f"in_layers.{i}"), sd[p + f"in_layers.{i}.bias"],
                     dilation=d, padding=(ks * d - d) // 2)
        c = F.conv1d(spect, effective_weight(sd, p + f"cond_layers.{i}"), sd[p + f"cond_layers.{i}.bias"])
        s = a + c
        acts = torch.tanh(s[:, :C]) * torch.sigmoid(s[:, C:])          # glow.py:33-40
        rs = F.conv1d(acts, effective_weight(sd, p + f"res_skip_layers.{i}"), sd[p + f"res_skip_layers.{i}.bias"])
        if i < nl - 1:
            x = x + rs[:, :C]
            sk = rs[:, C:]
        else:
            sk = rs
        skip = sk if skip is None else skip + sk
        if taps is not None:
            taps.append((acts, x, skip))
    return F.conv1d(skip, sd[p + "end.weight"], sd[p + "end.bias"])


def waveglow_forward(sd, cfg, mel, audio):
    """Reference WaveGlow.forward, glow.py:207-249.
    Returns (z [B,n_group,L], [log_s per flow], [log_det_W per flow])."""
    B, T = audio.shape
    g = cfg["n_group"]
    spect = upsample_and_squeeze(sd, cfg, mel, n_samples=T)
    L = T // g
    z = audio[:, :L * g].reshape(B, L, g).permute(0, 2, 1)       # channel = sample phase (glow.py:223)
    outs, log_s_list, log_det_list = [], [], []
    for k, (n_rem, n_half) in enumerate(_flow_sizes(cfg)):
        if k % cfg["n_early_every"] == 0 and k > 0:
            outs.append(z[:, :cfg["n_early_size"]])
            z = z[:, cfg["n_early_size"]:]
        W = sd[f"convinv.{k}.conv.weight"]
        log_det_list.append(B * L * torch.logdet(W.squeeze(-1)))      # glow.py:100
        z = F.conv1d(z, W)
        a0, a1 = z[:, :n_half], z[:, n_half:]
        out = wn_forward(sd, cfg, k, a0, spect)
        log_s, b = out[:, n_half:], out[:, :n_half]                   # glow.py:241-242
        a1 = torch.exp(log_s) * a1 + b
        log_s_list.append(log_s)
        z = torch.cat([a0, a1], 1)
    outs.append(z)
    return torch.cat(outs, 1), log_s_list, log_det_list


def waveglow_loss(outputs, sigma=1.0):
    """Reference WaveGlowLoss.forward, glow.py:48-59."""
    z, log_s_list, log_det_list = outputs
    log_s_total = sum(torch.sum(ls) for ls in log_s_list)
    log_det_total = sum(log_det_list)
    loss = torch.sum(z * z) / (2 * sigma * sigma) - log_s_total - log_det_total
    return loss / (z.size(0) * z.size(1) * z.size(2))


def waveglow_infer(sd, cfg, mel, noise_final, noise_early, sigma=1.0):
    """Reference WaveGlow.infer, glow.py:251-292, with the Gaussian draws made
    explicit inputs: ``noise_final`` [B,n_remaining,L] replaces the draw at
    glow.py:260-267, ``noise_early`` (list, in the order the reference draws
    them: k = 8 then k = 4 for the default config) replaces glow.py:284-289."""
    spect = upsample_and_squeeze(sd, cfg, mel, trim_tail=True)
    sizes = _flow_sizes(cfg)
    audio = sigma * noise_final
    early = list(noise_early)
    for k in reversed(range(cfg["n_flows"])):
        n_half = audio.size(1) // 2
        a0, a1 = audio[:, :n_half], audio[:, n_half:]
        out = wn_forward(sd, cfg, k, a0, spect)
        s, b = out[:, n_half:], out[:, :n_half]
        a1 = (a1 - b) / torch.exp(s)
        audio = torch.cat([a0, a1], 1)
        W = sd[f"convinv.{k}.conv.weight"].squeeze(-1)
        audio = F.conv1d(audio, torch.linalg.inv(W).unsqueeze(-1))     # glow.py:88-96
        if k % cfg["n_early_every"] == 0 and k > 0:
            audio = torch.cat((sigma * early.pop(0), audio), 1)
    B = audio.size(0)
    return audio.permute(0, 2, 1).reshape(B, -1)                      # glow.py:291

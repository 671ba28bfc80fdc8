#!/usr/bin/env python3
"""Which eager (aten) operators does one step launch, and from which source line?  CPU-side torch profiler around one step on
the GPU box; the hand-written kernels go through ctypes and do not show here, so every aten:: row that touches device memory is
an eager kernel inside the step (there should be none in a train step).

    python tools/prof_ops.py [forward|waveglow_train|tacotron_train]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from text2speech_amd import synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "forward"
NOISE = ('aten::empty', 'aten::view', 'aten::detach', 'aten::select', 'aten::slice', 'aten::as_strided', 'aten::empty_strided',
         'aten::to', 'aten::alias', 'aten::lift_fresh', 'aten::detach_', 'aten::contiguous', 'aten::reshape',
         'aten::_unsafe_view', 'aten::expand', 'aten::permute', 'aten::transpose', 'aten::unsqueeze', 'aten::squeeze', 'aten::t',
         'aten::empty_like', 'aten::view_as', 'aten::result_type', 'aten::item', 'aten::_local_scalar_dense', 'aten::is_nonzero',
         'aten::resolve_conj', 'aten::resolve_neg', 'aten::unbind', 'aten::numpy_T', 'aten::narrow', 'aten::chunk', 'aten::split')

if mode == "forward":
    from text2speech_amd.glow import WaveGlow
    cfg = synth.WAVEGLOW_DEFAULT
    m = WaveGlow(**cfg); m.load_state_dict(synth.waveglow_state(cfg)); m = m.cuda().eval()
    mel, audio = synth.waveglow_inputs(8, 16000, seed=1); mel, audio = mel.cuda(), audio.cuda()

    def step():
        with torch.no_grad():
            m((mel, audio))
elif mode == "waveglow_train":
    from text2speech_amd.glow import WaveGlow, WaveGlowLoss
    from text2speech_amd.optim import FusedAdam
    cfg = synth.WAVEGLOW_DEFAULT
    m = WaveGlow(**cfg); m.load_state_dict(synth.waveglow_state(cfg)); m = m.cuda().train()
    mel, audio = synth.waveglow_inputs(8, 16000, seed=1); mel, audio = mel.cuda(), audio.cuda()
    crit, opt = WaveGlowLoss(1.0), FusedAdam(m.parameters(), lr=1e-4)

    def step():
        m.zero_grad(set_to_none=True)
        crit(m((mel, audio))).backward()
        opt.step()
else:
    from text2speech_amd.optim import FusedAdam
    from text2speech_amd.tacotron import Tacotron, Tacotron2Loss
    B, T_in, T_out = 32, 256, 800
    m = Tacotron(synth.TACOTRON_HPARAMS, 80, num_speakers=2); m.load_state_dict(synth.tacotron_state()); m = m.cuda().train()
    gen = torch.Generator().manual_seed(21)
    text = torch.randint(2, 80, (B, T_in), generator=gen).cuda()
    mel_t = torch.randn(B, 80, T_out, generator=gen).cuda()
    gate_t = torch.zeros(B, T_out); gate_t[:, -1] = 1; gate_t = gate_t.cuda()
    il, ol = torch.full((B,), T_in).cuda(), torch.full((B,), T_out).cuda()
    x = (text, il, mel_t, T_in, torch.zeros(B).cuda(), ol)
    crit, opt = Tacotron2Loss(), FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-6)

    def step():
        m.zero_grad(set_to_none=True)
        crit(m(x), (mel_t, gate_t)).backward()
        opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ops = {}
for e in prof.events():
    if e.name.startswith('aten::') and e.name not in NOISE:
        stack = [s for s in (e.stack or []) if "text2speech_amd" in s or "bench" in s or "prof_ops" in s]
        key = (e.name, str(e.input_shapes)[:70], stack[0][-70:] if stack else "?")
        ops[key] = ops.get(key, 0) + 1
tot = 0
for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:60]:
    print(v, k)
    tot += v
print("aten ops listed:", tot)
# device-side view of the same step (roctracer): every kernel / copy that is not one of the library's own, with its launching op
try:
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof2:
        step()
        torch.cuda.synchronize()
    dev = {}
    for e in prof2.events():
        if str(e.device_type).endswith("CUDA") or str(e.device_type).endswith("PrivateUse1"):
            nm = e.name
            if "t2s" in nm or "_kernel" in nm and "at::" not in nm and "rocclr" not in nm:
                continue
            parent = e.cpu_parent.name if getattr(e, "cpu_parent", None) is not None else "?"
            dev[(nm[:70], parent[:40])] = dev.get((nm[:70], parent[:40]), 0) + 1
    print("device-side events that are not library kernels:")
    for k, v in sorted(dev.items(), key=lambda kv: -kv[1])[:30]:
        print(v, k)
    print(prof2.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=60)[:9000])
except Exception as ex:      # noqa: BLE001
    print("device-side profile unavailable:", ex)

import sys, torch
sys.path.insert(0, '.')
from text2speech_amd import _lib, synth
from text2speech_amd.glow import WaveGlow
from torch.profiler import profile, ProfilerActivity
cfg = synth.WAVEGLOW_DEFAULT
m = WaveGlow(**cfg); m.load_state_dict(synth.waveglow_state(cfg)); m = m.cuda().eval()
mel, audio = synth.waveglow_inputs(8, 16000, seed=1); mel, audio = mel.cuda(), audio.cuda()
with torch.no_grad():
    for _ in range(2): m((mel, audio))
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
        m((mel, audio))
        torch.cuda.synchronize()
ops = {}
for e in prof.events():
    if e.name.startswith('aten::') and e.name not in ('aten::empty', 'aten::view', 'aten::detach', 'aten::select', 'aten::slice', 'aten::as_strided', 'aten::empty_strided', 'aten::to', 'aten::_to_copy', 'aten::alias', 'aten::lift_fresh', 'aten::detach_', 'aten::contiguous', 'aten::reshape', 'aten::_unsafe_view'):
        key = (e.name, str(e.input_shapes)[:80])
        ops[key] = ops.get(key, 0) + 1
for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:25]:
    print(v, k)
print(prof.key_averages(group_by_stack_n=4).table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=40)[:6000])

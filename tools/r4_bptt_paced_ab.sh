#!/bin/bash
# Same-box A/B: the BPTT loop's helper chain paced by a device word behind the attention backward's launch (T2S_BPTT_PACED=1), with its
# per-step GEMM on the 96 KB ring (so that it shares CUs with that launch) or on the 144 KB ring (T2S_BPTT_PACED_NARROW=0), alternating.
for rep in 1 2; do for envs in "A=1" "T2S_BPTT_PACED=1" "T2S_BPTT_PACED=1 T2S_BPTT_PACED_NARROW=0"; do
  echo -n "$envs : train ms "; env $envs python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

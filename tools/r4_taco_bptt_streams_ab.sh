#!/bin/bash
# Same-box A/Bs of how the BPTT loop's two chains share the chip (Tacotron-2 train step, B = 32, 256 / 800), alternating.
for rep in 1 2; do for envs in "A=1" "T2S_BPTT_SIDE_NARROW=1" "T2S_BPTT_SPLIT_ROWS=1" "T2S_BPTT_ONE_STREAM=1"; do
  echo -n "$envs : train ms "; env $envs python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

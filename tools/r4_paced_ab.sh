#!/bin/bash
# Same-box A/B: teacher-forced decoder cells paced by a device word behind the attention cell's launch (default) against chunks of 16
# steps behind the chain (T2S_DECODE_PACED=0), Tacotron-2 B = 32, 256 / 800, alternating.
for rep in 1 2; do for v in 1 0; do
  echo -n "DECODE_PACED=$v : fwd B32 ms "; T2S_DECODE_PACED=$v python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'], end='')"
  echo -n "  train ms "; T2S_DECODE_PACED=$v python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

#!/bin/bash
# Same-box A/B of the energies kernel's block order (T2S_ENERGY_XCD=0: tile-major as launched; default: all tiles of an item on one XCD):
# teacher-forced eval forward and the train step at B = 32, 256 / 800, alternating.
for rep in 1 2; do for v in 1 0; do
  echo -n "ENERGY_XCD=$v fwd B32 ms "; T2S_ENERGY_XCD=$v python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'])"
  echo -n "ENERGY_XCD=$v train ms "; T2S_ENERGY_XCD=$v python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

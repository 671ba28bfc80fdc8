#!/bin/bash
# Same-box A/Bs of how the Tacotron-2 B = 32 loops share the chip between the serial chain and the helper stream:
#  T2S_HELPER_PRIO=1   helper streams at low priority      T2S_SB_STEP=16   small-batch GEMMs with the 96 KB ring (64-byte fragment rows)
for rep in 1 2; do for envs in "A=1" "T2S_HELPER_PRIO=1" "T2S_SB_STEP=16" "T2S_HELPER_PRIO=1 T2S_SB_STEP=16"; do
  echo -n "$envs : fwd B32 ms "; env $envs python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'], end='')"
  echo -n "  train ms "; env $envs python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

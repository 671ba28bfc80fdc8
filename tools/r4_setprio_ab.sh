#!/bin/bash
# Same-box A/B: s_setprio 3 in the teacher-forced attention kernel (shipped) against none (build/noprio), alternating
R=$(pwd)
for rep in 1 2; do for v in shipped noprio; do
  lib=""; [ $v != shipped ] && lib=$R/build/$v/libt2s_hip.so
  echo -n "$v : fwd B32 ms "; T2S_LIB_PATH=$lib python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'], end='')"
  echo -n "  train ms "; T2S_LIB_PATH=$lib python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

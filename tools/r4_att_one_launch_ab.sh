#!/bin/bash
# Same-box A/B: energies + softmax + context as ONE launch (tiles of an element exchange energies through t2s_taco_decoder::att_xbuf;
# default) against two launches (T2S_ATT_ONE_LAUNCH=0), Tacotron-2 B = 32, 256 / 800, alternating.
for rep in 1 2; do for v in 1 0; do
  echo -n "ATT_ONE_LAUNCH=$v : fwd B32 ms "; T2S_ATT_ONE_LAUNCH=$v python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'], end='')"
  echo -n "  train ms "; T2S_ATT_ONE_LAUNCH=$v python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

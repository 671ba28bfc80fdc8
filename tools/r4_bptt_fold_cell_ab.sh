#!/bin/bash
# Same-box A/B: the attention LSTMCell's pointwise backward inside the one-launch attention backward (chunks of an element exchange
# their partial d_q through t2s_taco_bptt::att_xbuf; default) against a launch of its own (T2S_BPTT_FOLD_CELL=0), alternating.
for rep in 1 2; do for v in 1 0; do
  echo -n "BPTT_FOLD_CELL=$v : train ms "; T2S_BPTT_FOLD_CELL=$v python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

#!/bin/bash
# Same-box A/B of the encoder BiLSTM recurrence: W_hh resident over four workgroups per (element, direction) (default) against the
# one-workgroup kernels that stream it every step (T2S_LSTM_SEQ_SPLIT=0), alternating.
for rep in 1 2; do for v in 1 0; do
  echo -n "LSTM_SEQ_SPLIT=$v : "; T2S_LSTM_SEQ_SPLIT=$v python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd B32 ms %.2f  B=1 200 frames ms %.2f  1000 frames ms %.2f' % (d['forward_B32_Tin256_Tout800']['ms'], d['inference_B1_200frames']['ms'], d['inference_B1_1000frames']['ms']), end='')"
  echo -n "  train ms "; T2S_LSTM_SEQ_SPLIT=$v python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

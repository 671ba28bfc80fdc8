#!/bin/bash
# Same-box A/B: the decoder cells' input product per chunk of 16 steps (t2s_taco_decoder::dec_in_part; T2S_DECODE_CHUNK_GEMM=1, default: per step),
# with and without the BPTT loop's counterpart (T2S_BPTT_SPLIT_ROWS=1), alternating.
for rep in 1 2; do for envs in "T2S_DECODE_CHUNK_GEMM=1" "A=1" "T2S_DECODE_CHUNK_GEMM=1 T2S_BPTT_SPLIT_ROWS=1" "T2S_BPTT_SPLIT_ROWS=1"; do
  echo -n "$envs : fwd B32 ms "; env $envs python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'], end='')"
  echo -n "  train ms "; env $envs python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

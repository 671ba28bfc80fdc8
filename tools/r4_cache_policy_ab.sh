#!/bin/bash
# Same-box A/Bs of load cache policies (diagnostic builds under build/, see text2speech_amd/build.py --variant):
#  sbnt      -DT2S_SB_NT_WEIGHTS     small-batch GEMM weight rows by nt LDS-DMA (aux = 2)            -> forward / train step at B = 32
#  cellplain -DT2S_CELL_PLAIN_LOADS  default-policy weight loads in the two B <= 8 cells only        -> B = 1 decode step
#  allplain  -DT2S_LSTM_PLAIN_LOADS  default-policy loads in the cells and the gate-stream role      -> B = 1 decode step
R=$(pwd)
for rep in 1 2; do for v in shipped sbnt; do
  lib=""; [ $v != shipped ] && lib=$R/build/$v/libt2s_hip.so
  echo -n "$v fwd B32 ms "; T2S_LIB_PATH=$lib python tools/bench_tacotron.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['forward_B32_Tin256_Tout800']['ms'])"
  echo -n "$v train ms "; T2S_LIB_PATH=$lib python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done
for rep in 1 2; do for v in shipped cellplain allplain; do
  lib=""; [ $v != shipped ] && lib=$R/build/$v/libt2s_hip.so
  echo -n "$v B=1 decode us/step "; T2S_LIB_PATH=$lib python tools/r4_decode_ab.py 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream on %.2f off %.2f' % (d['stream_on'], d['stream_off']))"
done; done

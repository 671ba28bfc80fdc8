R=$(pwd)
for rep in 1 2; do for v in "shipped" "attbprio" "chunk8" "chunk32"; do
  lib=""; envs="A=1"
  [ $v = attbprio ] && lib=$R/build/attbprio/libt2s_hip.so
  [ $v = chunk8 ] && envs="T2S_BPTT_CHUNK=8"
  [ $v = chunk32 ] && envs="T2S_BPTT_CHUNK=32"
  echo -n "$v : train ms "; env $envs T2S_LIB_PATH=$lib python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

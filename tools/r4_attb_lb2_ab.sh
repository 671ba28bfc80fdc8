R=$(pwd)
for rep in 1 2; do for v in shipped attblb2; do
  lib=""; [ $v != shipped ] && lib=$R/build/$v/libt2s_hip.so
  echo -n "$v : train ms "; T2S_LIB_PATH=$lib python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

for rep in 1 2; do for envs in "A=1" "T2S_POOL_CAP_GB=64"; do
  echo -n "$envs : train ms "; env $envs python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f  max_mem %.1f GB' % (d['ms_per_step'], d['max_mem_GB']))"
done; done

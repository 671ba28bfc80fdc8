for rep in 1 2; do for envs in "A=1" "T2S_BPTT_SIDE_FULL=1" "T2S_HELPER_PRIO=1"; do
  echo -n "$envs : train ms "; env $envs python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"
done; done

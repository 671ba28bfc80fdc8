#!/bin/bash
# B = 1 decode: us per step for a list of environment settings, alternating, two rounds (one box).  usage: r4_decode_env_ab.sh "A=1" "B=0 C=1" ...
set -uo pipefail
for round in 1 2; do
  for envs in "$@"; do
    echo -n "$envs : "
    env $envs python tools/r4_decode_ab.py 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream_on %.2f us/step, stream_off %.2f us/step' % (d['stream_on'], d['stream_off']))"
  done
done

#!/bin/bash
# A/B/A of two builds of libt2s_hip.so on one box: tools/ab_libs.sh <libA> <libB> [bench args...]; prints ms_per_step and the
# dominant kernel's average launch for A, B, A, B.
set -euo pipefail
A=$1; B=$2; shift 2
L=text2speech_amd/libt2s_hip.so
for lib in "$A" "$B" "$A" "$B"; do
  cp "$lib" $L
  python bench.py --no-cpu-baseline --no-tacotron --no-train --steps 30 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$lib', round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms']*1e3,2))"
done

#!/bin/bash
# WaveGlow forward 8 x 16000: the shipped split-bf16 library against the fp16-operand diagnostic build, alternating on one box.
# T2S_F16_GUARD=0: without the fp16 build's per-call overflow read-back (a host synchronisation that would otherwise be the difference).
set -uo pipefail
R=$(pwd)
for round in 1 2; do
  for lib in "" "$R/build/f16x3/libt2s_hip.so"; do
    name=${lib:+fp16x3}; name=${name:-bf16x3}
    T2S_F16_GUARD=0 T2S_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-tacotron --no-train --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$name', 'ms_per_step %.3f' % d['ms_per_step'], 'gate GEMM us %.2f' % (d['roofline']['avg_launch_ms']*1e3))"
  done
done

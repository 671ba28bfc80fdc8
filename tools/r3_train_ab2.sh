#!/bin/bash
# Round 3: (1) one-stream kernel tables of the WaveGlow train step for the lockstep / ping-pong weight-gradient kernel: standalone
# durations, no concurrency between the two backward streams; (2) un-profiled A/B of the per-flow conditioning-gradient GEMM.
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/train_ab2}")
mkdir -p "$OUT"
T="python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train"
for rep in 1 2; do
  T2S_WCOND_PER_LAYER=1 $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/cond_per_layer_$rep.json"
  $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/cond_per_flow_$rep.json"
  T2S_WG_BWD_ONE_STREAM=1 $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/one_stream_$rep.json"
done
python3 - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        print(os.path.basename(f), "ms_per_step %.2f" % json.loads(open(f).read())["ms_per_step"])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
export T2S_WG_BWD_ONE_STREAM=1
for pp in 0 1; do
  export T2S_WGRAD_PP=$pp
  rocprofv3 --kernel-trace --stats -d "$OUT/kt$pp" -o k -- $T --steps 3 --warmup 1 > /dev/null 2> "$OUT/kt$pp.err"
  python3 $R/tools/rocpd_by_grid.py "$OUT/kt$pp/k_results.db" 4 26 > "$OUT/one_stream_pp${pp}_by_grid.md"
  rm -rf "$OUT/kt$pp"
done
echo done

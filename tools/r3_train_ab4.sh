#!/bin/bash
# Round 3: un-profiled A/B of the backward's accumulate / gate-backward GEMMs on 256-row ping-pong tiles (T2S_BWD_PP256=0: the
# lockstep 128-row tiles of round 2), then kernel tables (two streams, by grid) of both.
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/train_ab4}")
mkdir -p "$OUT"
T="python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train"
for rep in 1 2; do
  T2S_BWD_PP256=0 $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/pp256_off_$rep.json"
  $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/pp256_on_$rep.json"
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
for v in 0 1; do
  T2S_BWD_PP256=$v rocprofv3 --kernel-trace --stats -d "$OUT/kt$v" -o k -- $T --steps 3 --warmup 1 > /dev/null 2> "$OUT/kt$v.err"
  python3 $R/tools/rocpd_by_grid.py "$OUT/kt$v/k_results.db" 4 16 > "$OUT/two_stream_pp256_${v}_by_grid.md"
  rm -rf "$OUT/kt$v"
done
echo done

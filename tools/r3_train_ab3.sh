#!/bin/bash
# Round 3: un-profiled A/B of the training forward on the fold path (T2S_TRAIN_NO_FOLD=1 = the round-2 forward with the f32 skip
# accumulator), then the one-stream kernel table of the default.
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/train_ab3}")
mkdir -p "$OUT"
T="python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train"
for rep in 1 2; do
  T2S_TRAIN_NO_FOLD=1 $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/no_fold_$rep.json"
  $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/fold_$rep.json"
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
T2S_WG_BWD_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o k -- $T --steps 3 --warmup 1 > /dev/null 2> "$OUT/kt.err"
python3 $R/tools/rocpd_by_grid.py "$OUT/kt/k_results.db" 4 30 > "$OUT/one_stream_by_grid.md"
rm -rf "$OUT/kt"
echo done

#!/bin/bash
# Round 3: second pass of the split-K width A/B (see r3_train_ab5.sh) around the first pass's winner (256 / 128).
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/train_ab6}")
mkdir -p "$OUT"
T="python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train"
for rep in 1 2; do
  for cfg in 256_128 256_192 256_160 256_96 192_128 224_128; do
    f2=${cfg%_*}; f1=${cfg#*_}
    T2S_WGRAD_FILL2=$f2 T2S_WGRAD_FILL1=$f1 $T --steps 10 --warmup 2 2> /dev/null | tail -1 > "$OUT/fill_${cfg}_$rep.json"
  done
done
python3 - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        print(os.path.basename(f), "ms_per_step %.2f" % json.loads(open(f).read())["ms_per_step"])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
echo done

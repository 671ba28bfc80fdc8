#!/bin/bash
# Round 3: un-profiled A/B of the WaveGlow train step on one box (T2S_WGRAD_PP / T2S_WGRAD_BIAS_COL), then a HIP-API trace of two
# steps to find who issues the device copies that show up as __amd_rocclr_copyBuffer.
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/train_ab}")
mkdir -p "$OUT"
T="python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train --steps 10 --warmup 2"
for rep in 1 2; do
  T2S_WGRAD_PP=0 $T 2> /dev/null | tail -1 > "$OUT/pp0_$rep.json"
  T2S_WGRAD_BIAS_COL=ones $T 2> /dev/null | tail -1 > "$OUT/pp1_ones_$rep.json"
  $T 2> /dev/null | tail -1 > "$OUT/pp1_$rep.json"
done
python3 - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pp*.json"))):
    try:
        d = json.loads(open(f).read())
        print(os.path.basename(f), "ms_per_step %.2f" % d["ms_per_step"])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --memory-copy-trace --stats --output-format csv -d "$OUT/hip" -o h -- python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train --steps 2 --warmup 1 > /dev/null 2> "$OUT/hip.err"
ls "$OUT/hip" > "$OUT/hip_files.txt" 2>&1
for f in "$OUT"/hip/*stats*.csv "$OUT"/hip/*/*stats*.csv; do [ -f "$f" ] && cp "$f" "$OUT/"; done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "hip", "**", "*memory_copy_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    c = collections.Counter((r.get("Direction"), r.get("Bytes") or r.get("Size")) for r in rows)
    with open(os.path.join(out, "memcpy_summary.txt"), "w") as g:
        g.write("%d copies\n" % len(rows))
        for k, v in c.most_common(40):
            g.write("%s %s\n" % (v, k))
PY
rm -rf "$OUT/hip"
echo done

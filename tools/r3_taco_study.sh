#!/bin/bash
# Round 3, Tacotron train step: the kernel table of the step, then counters of the per-decoder-step kernels
# (LDS instruction / bank-conflict / busy counters and L2 request counters, their own passes) - evidence before any change.
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/taco_study}")
TAG="${2:-before}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T="python3 $R/tools/bench_tacotron_train.py"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o k -- $T > "$OUT/train_${TAG}.json" 2> "$OUT/kt.err"
python3 $R/tools/rocpd_stats.py "$OUT/kt/k_results.db" 6 24 > "$OUT/taco_train_kernels_${TAG}.md"
rm -rf "$OUT/kt"
export T_OUT=96
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_ANY -d "$OUT/sq" -o s -- $T > /dev/null 2> "$OUT/sq.err"
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d "$OUT/tcc" -o t -- $T > /dev/null 2> "$OUT/tcc.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/f" -o f -- $T > /dev/null 2> "$OUT/f.err"
python3 - "$OUT" "$TAG" "$R" <<'PY'
import json, subprocess, sys
out, tag, R = sys.argv[1:4]
merged = {}
for db in ("sq/s_results.db", "tcc/t_results.db", "f/f_results.db"):
    try:
        txt = subprocess.run([sys.executable, R + "/tools/pmc_counters.py", out + "/" + db], capture_output=True, text=True).stdout
        for k, v in json.loads(txt).items():
            merged.setdefault(k, {}).update(v)
    except Exception as e:
        print("skip", db, e)
keep = {k: v for k, v in merged.items() if any(s in k for s in ("att_", "sbgemm", "lstm_cell", "gemv_rows"))}
json.dump(keep, open(out + "/taco_step_kernel_counters_%s.json" % tag, "w"), indent=1, sort_keys=True)
PY
rm -rf "$OUT/sq" "$OUT/tcc" "$OUT/f"
echo done

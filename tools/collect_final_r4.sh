#!/bin/bash
# The subset of tools/collect_profiles.sh that the late round-4 changes (Tacotron B = 32 loops, encoder recurrence) moved:
# bench line, end-to-end, Tacotron tables.  ~6 GPU-minutes.  Output: gpurun_out/final4/
set -uo pipefail
R=$(pwd)
OUT=$R/gpurun_out/final4
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/taco_inf" -o ti -- python3 $R/tools/bench_tacotron.py > "$OUT/taco_inf_under_rocprof.json" 2> "$OUT/taco_inf.err"
rocprofv3 --kernel-trace --stats -d "$OUT/taco_train" -o tt -- python3 $R/tools/bench_tacotron_train.py > "$OUT/taco_train_under_rocprof.json" 2> "$OUT/taco_train.err"
cd "$R"
python3 tools/rocpd_stats.py "$OUT/taco_inf/ti_results.db" 1 14 > "$OUT/taco_inf_kernels.md"
python3 tools/rocpd_stats.py "$OUT/taco_train/tt_results.db" 6 24 > "$OUT/taco_train_kernels.md"
python3 tools/rocpd_phases.py "$OUT/taco_train/tt_results.db" > "$OUT/taco_train_phases.md"
python3 tools/rocpd_timeline.py "$OUT/taco_train/tt_results.db" sbgemm_lstm 0.80 24 > "$OUT/taco_timeline_fwd.md"
rm -rf "$OUT/taco_inf" "$OUT/taco_train"
python3 tools/bench_e2e.py > "$OUT/e2e.json" 2> /dev/null
python3 tools/bench_tacotron.py > "$OUT/bench_tacotron.json" 2> /dev/null
python3 tools/bench_tacotron_train.py > "$OUT/bench_tacotron_train.json" 2> /dev/null
python3 tools/prof_ops.py tacotron_train > "$OUT/ops_tacotron_train.txt" 2> /dev/null
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "wrote $OUT"

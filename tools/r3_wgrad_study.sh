#!/bin/bash
# Round 3: the weight-gradient GEMM before / after the ping-pong schedule (T2S_WGRAD_PP=0 / 1) inside the WaveGlow train step.
# Kernel-trace passes for the per-kernel times, then counter passes (their own runs, no trace domains next to --pmc).
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/wgrad_study}")
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
T="python3 $R/bench.py --mode train --no-cpu-baseline --no-tacotron --no-train"
for pp in 0 1; do
  export T2S_WGRAD_PP=$pp
  rocprofv3 --kernel-trace --stats -d "$OUT/kt$pp" -o k -- $T --steps 4 --warmup 1 > "$OUT/train_pp$pp.json" 2> "$OUT/kt$pp.err"
  python3 $R/tools/rocpd_by_grid.py "$OUT/kt$pp/k_results.db" 4 22 > "$OUT/train_pp${pp}_by_grid.md"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d "$OUT/sq$pp" -o s -- $T --steps 2 --warmup 1 > /dev/null 2> "$OUT/sq$pp.err"
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/f$pp" -o f -- $T --steps 2 --warmup 1 > /dev/null 2> "$OUT/f$pp.err"
  rocprofv3 --pmc WRITE_SIZE -d "$OUT/w$pp" -o w -- $T --steps 2 --warmup 1 > /dev/null 2> "$OUT/w$pp.err"
  python3 $R/tools/pmc_counters.py --match wgrad "$OUT/sq$pp/s_results.db" "$OUT/f$pp/f_results.db" "$OUT/w$pp/w_results.db" > "$OUT/wgrad_counters_pp$pp.json"
  rm -rf "$OUT/kt$pp" "$OUT/sq$pp" "$OUT/f$pp" "$OUT/w$pp"
done
unset T2S_WGRAD_PP
cd "$R"
python3 tools/prof_ops.py waveglow_train > "$OUT/ops_waveglow_train.txt" 2> "$OUT/ops_wg.err"
python3 tools/prof_ops.py tacotron_train > "$OUT/ops_tacotron_train.txt" 2> "$OUT/ops_taco.err"
echo done

#!/bin/bash
# Reproduce the evidence under profiles/ on a 1-GPU MI355X box (run from the repo root; ~10 GPU-minutes).
# rocprofv3 7.2 writes a rocpd SQLite database; tools/rocpd_stats.py, tools/pmc_traffic.py and tools/pmc_counters.py turn
# it into the tables / JSON that profiles/r02_summary.md quotes.  Counters are collected in their own passes (no trace
# domains next to --pmc).  The program after `--` is always python3 itself (never env / bash -c: MI355X pool rule).
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/collect}")
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-tacotron --no-train"
rocprofv3 --kernel-trace --stats -d "$OUT/fwd" -o fw -- $B --steps 5 --warmup 1 > "$OUT/fwd_under_rocprof.json" 2> "$OUT/fwd.err"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o f -- $B --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_f.err"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w -- $B --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_w.err"
rocprofv3 --kernel-trace --stats -d "$OUT/train" -o wt -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-train --no-cpu-baseline --no-tacotron > "$OUT/train_under_rocprof.json" 2> "$OUT/train.err"
T2S_WG_BWD_ONE_STREAM=1 rocprofv3 --kernel-trace --stats -d "$OUT/train1" -o w1 -- python3 $R/bench.py --mode train --steps 3 --warmup 1 --no-train --no-cpu-baseline --no-tacotron > /dev/null 2> "$OUT/train1.err"
rocprofv3 --kernel-trace --stats -d "$OUT/taco_inf" -o ti -- python3 $R/tools/bench_tacotron.py > "$OUT/taco_inf_under_rocprof.json" 2> "$OUT/taco_inf.err"
rocprofv3 --kernel-trace --stats -d "$OUT/taco_train" -o tt -- python3 $R/tools/bench_tacotron_train.py > "$OUT/taco_train_under_rocprof.json" 2> "$OUT/taco_train.err"
cd "$R"
python3 tools/rocpd_stats.py "$OUT/fwd/fw_results.db" 6 16 > "$OUT/fwd_kernels.md"
python3 tools/rocpd_stats.py "$OUT/train/wt_results.db" 4 16 > "$OUT/train_kernels.md"
python3 tools/rocpd_by_grid.py "$OUT/train/wt_results.db" 4 24 > "$OUT/train_two_streams_by_grid.md"
python3 tools/rocpd_by_grid.py "$OUT/train1/w1_results.db" 4 30 > "$OUT/train_one_stream_by_grid.md"
python3 tools/rocpd_stats.py "$OUT/taco_inf/ti_results.db" 1 14 > "$OUT/taco_inf_kernels.md"
python3 tools/rocpd_stats.py "$OUT/taco_train/tt_results.db" 6 20 > "$OUT/taco_train_kernels.md"
python3 tools/rocpd_timeline.py "$OUT/taco_train/tt_results.db" sbgemm_lstm 0.80 24 > "$OUT/taco_timeline_fwd.md"
python3 tools/rocpd_timeline.py "$OUT/taco_train/tt_results.db" att_bwd_fused 0.92 30 > "$OUT/taco_timeline_bwd.md"
python3 tools/pmc_traffic.py "$OUT/pmc_fetch/f_results.db" "$OUT/pmc_write/w_results.db" > "$OUT/pmc_traffic.json"
rm -rf "$OUT"/train1
rm -rf "$OUT"/fwd "$OUT"/train "$OUT"/taco_inf "$OUT"/taco_train "$OUT"/pmc_fetch "$OUT"/pmc_write     # databases are large
python3 tools/bench_e2e.py > "$OUT/e2e.json" 2> /dev/null
python3 tools/bench_infer_lengths.py > "$OUT/infer_lengths.json" 2> /dev/null
python3 tools/microbench/sbgemm_bench.py > "$OUT/sbgemm_bench.json" 2> /dev/null
python3 tools/prof_ops.py waveglow_train > "$OUT/ops_waveglow_train.txt" 2> /dev/null
python3 tools/prof_ops.py tacotron_train > "$OUT/ops_tacotron_train.txt" 2> /dev/null
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
# round 4: the B = 1 decode chain (streamed gates / folded prenet / location term one launch early, each against its switch), the
# operand-format A/B and - when the diagnostic builds exist (python -m text2speech_amd.build --variant ...) - the role ablation and the
# in-kernel stamps of the attention role
python3 tools/r4_decode_ab.py 1 > "$OUT/decode_stream_ab.json" 2> /dev/null
bash tools/r4_decode_env_ab.sh "T2S_DECODE_FOLD_PRE2=1" "T2S_DECODE_FOLD_PRE2=0" > "$OUT/decode_fold_ab.txt" 2>&1
bash tools/r4_decode_env_ab.sh "T2S_DECODE_PLOC=1" "T2S_DECODE_PLOC=0" > "$OUT/decode_ploc_ab.txt" 2>&1
[ -f build/f16x3/libt2s_hip.so ] && bash tools/r4_operand_ab.sh > "$OUT/operand_format_ab.txt" 2>&1
[ -f build/attstream_ablate/libt2s_hip.so ] && bash tools/r4_attstream_ablate.sh > /dev/null 2>&1 && cp gpurun_out/r4_attstream/summary.md "$OUT/attstream_role_ablation.md"
if [ -f build/probe/libt2s_hip.so ]; then
  for n in 64 128 256; do echo "== $n symbols"; T2S_LIB_PATH=$R/build/probe/libt2s_hip.so python3 tools/decode_probe.py $n 2> /dev/null; done > "$OUT/att_role_probe.txt"
fi
# late round 4 (Tacotron B = 32 loops): each of these is a same-box alternating A/B that prints a few lines; their outputs are the
# profiles/r04_*_ab.txt files of the same name
#   tools/r4_energy_xcd_ab.sh  r4_lstm_split_ab.sh  r4_att_one_launch_ab.sh  r4_bptt_fold_cell_ab.sh  r4_paced_ab.sh  r4_setprio_ab.sh
#   r4_cache_policy_ab.sh  r4_taco_sched_ab.sh  r4_taco_bptt_streams_ab.sh  r4_taco_chunk_gemm_ab.sh  r4_bptt_paced_ab.sh  r4_bptt_side_full_ab.sh
# and tools/collect_final_r4.sh re-collects the bench line, the end-to-end number and the Tacotron tables / phases / timeline
echo "wrote $OUT"

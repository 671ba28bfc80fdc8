#!/bin/bash
# Round-2 study of the gate GEMM on one MI355X box: (1) timing-only ablations of the ping-pong kernel, (2) SQ counters of the
# lockstep kernel (T2S_GEMM_PP=0) and the ping-pong kernel (default), each in its own --pmc pass.  Run from the repo root.
set -uo pipefail
R=$(pwd)
OUT=$R/gpurun_out/gg_study
mkdir -p "$OUT"
B="python3 $R/bench.py --no-cpu-baseline --no-tacotron --no-train --steps 10 --warmup 2"
line() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']
print(sys.argv[2], 'ms/step %.3f' % d['ms_per_step'], 'gate us %.1f' % (r['avg_launch_ms']*1e3))" "$1" "$2"; }
# ---- ablation build in a scratch copy of the library ----
cp text2speech_amd/libt2s_hip.so /tmp/libt2s_hip.so.keep
T2S_BUILD_DEFINES="-DT2S_GEMM_ABLATE" python3 -c "
from text2speech_amd import build; build.build(force=True)" || exit 1
for dbg in 0 1 2 3 4 5 6 7; do
  T2S_DBG_GEMM=$dbg $B > "$OUT/abl_$dbg.json" 2>/dev/null && line "$OUT/abl_$dbg.json" "ablate=$dbg"
done | tee "$OUT/ablations.txt"
cp /tmp/libt2s_hip.so.keep text2speech_amd/libt2s_hip.so
# ---- counters ----
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1 || true
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
P3="SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE"
for pp in 0 1; do
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    T2S_GEMM_PP=$pp rocprofv3 --pmc $P -d "$OUT/pmc_pp${pp}_$i" -o c -- python3 $R/bench.py --no-cpu-baseline --no-tacotron --no-train --steps 2 --warmup 1 > /dev/null 2> "$OUT/pmc_pp${pp}_$i.err" || echo "pmc pass pp=$pp set $i failed"
  done
  python3 $R/tools/pmc_counters.py --match gemm $OUT/pmc_pp${pp}_1/c_results.db $OUT/pmc_pp${pp}_2/c_results.db $OUT/pmc_pp${pp}_3/c_results.db > "$OUT/counters_pp$pp.json" 2> "$OUT/counters_pp$pp.err" || true
done
rm -rf $OUT/pmc_pp*/  # the databases are large; the JSON summaries are what is kept
echo done

#!/bin/bash
# Where the time of the role-specialised attention launch goes: kernel-trace averages of att_fused_mfma_kernel<true> with both
# roles, with the attention role returning at once (T2S_DBG_ATTSTREAM=1) and with the gate-stream role returning at once (=2).
# Needs the diagnostic build: python -m text2speech_amd.build --variant attstream_ablate "-DT2S_ATTSTREAM_ABLATE"
set -uo pipefail
R=$(pwd)
OUT=$R/gpurun_out/r4_attstream
mkdir -p $OUT
export T2S_LIB_PATH=$R/build/attstream_ablate/libt2s_hip.so
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2; do
  T2S_DBG_ATTSTREAM=$dbg rocprofv3 --kernel-trace --stats -d $OUT/d$dbg -o t -- python3 $R/tools/prof_taco_utterance.py 600 > /dev/null 2> $OUT/err$dbg.txt
  echo "## T2S_DBG_ATTSTREAM=$dbg" >> $OUT/summary.md
  python3 $R/tools/rocpd_stats.py $OUT/d$dbg/t_results.db 1 5 >> $OUT/summary.md
  rm -rf $OUT/d$dbg
done
cat $OUT/summary.md

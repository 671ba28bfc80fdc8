#!/bin/bash
# Timing-only ablations of the residual GEMM (conv_gemm_kernel<EPI_RESSKIP,128>) on one MI355X box: T2S_DBG_GEMM bit 0 = no DMA
# in the K loop, bit 1 = no MFMA (results are wrong; -DT2S_GEMM_ABLATE build in a scratch copy).  Prints the kernel's average
# launch time from a rocprofv3 kernel trace for each setting.  Run from the repo root.
set -uo pipefail
R=$(pwd)
OUT=$R/gpurun_out/res_study
mkdir -p "$OUT"
cp text2speech_amd/libt2s_hip.so /tmp/libt2s_hip.so.keep
T2S_BUILD_DEFINES="-DT2S_GEMM_ABLATE" python3 -c "
from text2speech_amd import build; build.build(force=True)" || exit 1
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2 3; do
  export T2S_DBG_GEMM=$dbg
  rocprofv3 --kernel-trace --stats -d "$OUT/k$dbg" -o k -- python3 $R/bench.py --no-cpu-baseline --no-tacotron --no-train --steps 4 --warmup 1 > /dev/null 2> "$OUT/k$dbg.err"
  echo "ablate=$dbg $(python3 $R/tools/rocpd_stats.py $OUT/k$dbg/k_results.db 5 6 | grep -E 'conv_gemm_kernelILi1ELi128|gate_gemm_pp' | tr '\n' ' ')"
  rm -rf "$OUT/k$dbg"
done | tee "$OUT/ablations.txt"
cp /tmp/libt2s_hip.so.keep $R/text2speech_amd/libt2s_hip.so

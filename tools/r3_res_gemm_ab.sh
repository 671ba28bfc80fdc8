#!/bin/bash
# Round 3, VERDICT r2 item 6: the residual GEMM on 256-row tiles (half the workgroups, each with whole 256-row panels of x) against
# the shipped 128-row tiles, both with identity row order (T2S_RES_PAIR8=0; the 16-byte PAIR8 epilogue exists for 128 rows only),
# and the shipped PAIR8 form: forward ms/step un-profiled, then the kernel's average duration under rocprofv3.
set -uo pipefail
R=$(pwd)
OUT=$(realpath -m "${1:-$R/gpurun_out/res_ab}")
mkdir -p "$OUT"
B="python3 $R/bench.py --no-cpu-baseline --no-tacotron --no-train --steps 20 --warmup 3"
python3 -m pytest $R/tests/test_waveglow_gpu.py -x -q -m gpu -k "forward" > "$OUT/parity_default.log" 2>&1
T2S_RES_PAIR8=0 T2S_RES_TILE=256 python3 -m pytest $R/tests/test_waveglow_gpu.py -x -q -m gpu -k "forward" > "$OUT/parity_tile256.log" 2>&1
tail -1 "$OUT/parity_default.log"; tail -1 "$OUT/parity_tile256.log"
for rep in 1 2; do
  $B 2> /dev/null | tail -1 > "$OUT/pair8_t128_$rep.json"
  T2S_RES_PAIR8=0 $B 2> /dev/null | tail -1 > "$OUT/ident_t128_$rep.json"
  T2S_RES_PAIR8=0 T2S_RES_TILE=256 $B 2> /dev/null | tail -1 > "$OUT/ident_t256_$rep.json"
done
cd /tmp && export TMPDIR=/tmp
for cfg in ident_t128 ident_t256; do
  if [ $cfg = ident_t256 ]; then export T2S_RES_TILE=256; fi
  T2S_RES_PAIR8=0 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$cfg" -o k -- $B --steps 5 > /dev/null 2> "$OUT/kt_$cfg.err"
  python3 $R/tools/rocpd_stats.py "$OUT/kt_$cfg/k_results.db" 8 4 > "$OUT/fwd_kernels_$cfg.md"
  rm -rf "$OUT/kt_$cfg"
done
python3 - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    print(os.path.basename(f), "ms_per_step %.3f" % json.loads(open(f).read())["ms_per_step"])
PY
grep conv_gemm "$OUT"/fwd_kernels_*.md

#!/bin/bash
# train step + kernel table after the BatchNorm kernels' plane I/O went through LDS and rows_to_tm's grid order was swapped
for rep in 1 2; do echo -n "train ms "; python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"; done

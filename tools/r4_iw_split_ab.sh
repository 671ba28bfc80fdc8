#!/bin/bash
# Tacotron train step: split-K slab count of the all-items weight-gradient products (T2S_IW_WGS = target workgroups; 4096 = always 16 slabs)
for rep in 1 2; do for v in 768 4096 256 1536; do echo -n "T2S_IW_WGS=$v train ms "; T2S_IW_WGS=$v python tools/bench_tacotron_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"; done; done

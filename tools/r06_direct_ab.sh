#!/bin/bash
# fused backward: the table path against plain atomics for every block (an experiment build), t1m and the headline
for r in 1 2; do
  for v in main direct; do
    if [ $v == direct ]; then export NVDR_LIB_PATH=$PWD/nvdiffrast_amd/libnvdr_hip_direct.so; else unset NVDR_LIB_PATH; fi
    python tools/bench_regimes.py t1m t1m_shuffled ch --no-check 2>/dev/null | grep regime | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$v', d['regime'], d['ms_per_step'], d['kernels_ms'].get('interp_raster_grad'))"
  done
done

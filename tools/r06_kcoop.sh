#!/bin/bash
mkdir -p gpurun_out
python tools/ab_raster.py --rounds 30 --scenes ch,c2,s10k,dense main k8 k6 k3 > gpurun_out/r06e_kcoop_ab.log 2>&1
tail -30 gpurun_out/r06e_kcoop_ab.log
python -m pytest tests/test_gpu_raster_interp.py tests/test_gpu_fuzz.py tests/test_gpu_bin_lists.py tests/test_gpu_edge_cases.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
timeout 200 python tools/fuzz_soak.py 120 5000 2>&1 | tail -3

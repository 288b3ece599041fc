#!/bin/bash
# round-5 closing run on the GPU box: suite, suite subset with flag verification, bench, rocprofv3 stats + PMC traffic (ch, c3),
# flag-order time at 16384 bins.  Everything under gpurun_out/r05k.
O=gpurun_out/${RUN_TAG:-r05k}; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-200
NVDR_VERIFY_TILE_FLAGS=1 python -m pytest tests/test_gpu_tile_flags.py tests/test_gpu_work_order.py tests/test_gpu_texture_aa.py tests/test_gpu_fused_backward.py tests/test_gpu_raster_interp.py tests/test_gpu_end_to_end.py tests/test_gpu_plugin_fused_backward.py -m gpu -q > $O/tests_verify_flags.log 2>&1; tail -1 $O/tests_verify_flags.log | cut -c1-200
python bench.py > $O/bench.json 2> $O/bench.err; cp bench_detail.json $O/
bash tools/profile_gpu.sh ${RUN_TAG:-r05k}_ch --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/prof_ch.log 2>&1
bash tools/profile_gpu.sh ${RUN_TAG:-r05k}_c3 --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/prof_c3.log 2>&1
python bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/c4_1gpu.json 2> $O/c4.err
tail -c 300 $O/bench.json

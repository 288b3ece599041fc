#!/bin/bash
mkdir -p gpurun_out
( time python bench.py --detail gpurun_out/r06m_bench_detail.json > gpurun_out/r06m_bench.json 2> gpurun_out/r06m_bench.err ) 2> gpurun_out/r06m_bench_time.log
tail -3 gpurun_out/r06m_bench_time.log
python -m pytest tests/test_gpu_bench.py tests/test_gpu_full_size.py tests/test_gpu_host_layer.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4 | cut -c1-300

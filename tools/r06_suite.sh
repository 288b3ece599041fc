#!/bin/bash
# round 6: the whole GPU suite + the default bench line (tag in $1)
tag=${1:-r06x}
mkdir -p gpurun_out
( time python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > gpurun_out/${tag}_gpu_tests.log 2>&1
tail -4 gpurun_out/${tag}_gpu_tests.log | cut -c1-600
python bench.py --detail gpurun_out/${tag}_bench_detail.json > gpurun_out/${tag}_bench.line 2> gpurun_out/${tag}_bench.err
tail -c 3000 gpurun_out/${tag}_bench.line

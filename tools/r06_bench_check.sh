#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_host_layer.py -q -m gpu -p no:cacheprovider -k "leaks or threads" > gpurun_out/r06n_tests.log 2>&1
tail -4 gpurun_out/r06n_tests.log | cut -c1-300; grep -n "^E  .*AssertionError" gpurun_out/r06n_tests.log | head -20 | cut -c1-600

mkdir -p gpurun_out/r04n
python -m pytest tests/test_gpu_raster_interp.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_tile_flags.py -m gpu -q -x > gpurun_out/r04n/tests.log 2>&1; tail -2 gpurun_out/r04n/tests.log | cut -c1-200
for K in 0 1 2 4 8; do for O in 0 1; do
  echo "K=$K ordered=$O"; NVDR_TUNE_IPFWD_K=$K NVDR_TUNE_IPFWD_ORDERED=$O python tools/bench_regimes.py ch dense 2>/dev/null | grep -o '"regime": "[a-z]*"\|"interp_fwd": [0-9.]*' | paste - - ;
done; done

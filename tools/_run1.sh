mkdir -p gpurun_out; rm -f gpurun_out/clipexp.log
python -m pytest tests/test_gpu_raster_interp.py tests/test_gpu_fuzz.py tests/test_gpu_reference_direct.py tests/test_gpu_edge_cases.py -x -q -m gpu 2>&1 | tail -1 > gpurun_out/clipexp.log
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[0])
print(d['ms_per_step'], d['kernels']['raster_fine']['avg_ms'], d['parity']['tri_id_mismatches'])" >> gpurun_out/clipexp.log; done
python tools/bench_stress.py 2>/dev/null | tail -1 >> gpurun_out/clipexp.log
cat gpurun_out/clipexp.log

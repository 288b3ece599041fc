#!/bin/bash
# round 6: the compiled host layer on the GPU -- its tests, the host time of one step with and without launches (both layers),
# and BASELINE configs[1] eager against its hipGraph replay.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_host_layer.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r06a_host_tests.log
for h in 1 0; do
  echo "== NVDR_HOST=$h, launches off" >> gpurun_out/r06a_host_profile.log
  NVDR_HOST=$h NVDR_DEBUG=2097152 python tools/host_profile.py 3000 2>&1 | grep -E "us per step|function calls" >> gpurun_out/r06a_host_profile.log
  echo "== NVDR_HOST=$h, launches on" >> gpurun_out/r06a_host_profile.log
  NVDR_HOST=$h python tools/host_profile.py 3000 2>&1 | grep -E "us per step|function calls" >> gpurun_out/r06a_host_profile.log
done
for h in 1 0; do
  NVDR_HOST=$h python bench.py --workload c2 --steps 50 --no-cpu-baseline --no-extra-configs --detail gpurun_out/r06a_c2_host$h.json > gpurun_out/r06a_c2_host$h.line 2>gpurun_out/r06a_c2_host$h.err
  NVDR_HOST=$h python bench.py --workload c2 --steps 50 --graph --no-cpu-baseline --no-extra-configs --detail gpurun_out/r06a_c2g_host$h.json > gpurun_out/r06a_c2g_host$h.line 2>>gpurun_out/r06a_c2_host$h.err
done
tail -5 gpurun_out/r06a_host_tests.log; cat gpurun_out/r06a_host_profile.log
for f in gpurun_out/r06a_c2*_host?.line; do echo $f; python -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; done

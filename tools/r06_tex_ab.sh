#!/bin/bash
# interleaved A/B of two builds on config 3's per-kernel times (bench.py --workload c3, the library's own hipEvent brackets)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_texture_aa.py tests/test_gpu_fuzz.py tests/test_gpu_work_order.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 | cut -c1-200
for r in 1 2 3; do
  for v in old new; do
    if [ $v == old ]; then export NVDR_LIB_PATH=$PWD/nvdiffrast_amd/libnvdr_hip_old.so; else unset NVDR_LIB_PATH; fi
    python bench.py --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --detail gpurun_out/_ab.json > /dev/null 2>&1
    python -c "
import json; d=json.load(open('gpurun_out/_ab.json')); k=d['kernels']
print('$v', d['ms_per_step'], {n: round(k[n]['avg_ms'],4) for n in ('tex_fwd','tex_grad','tex_grad_light','aa_discontinuity') if n in k})" | tee -a gpurun_out/r06p_tex_ab.log
  done
done

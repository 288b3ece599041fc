#!/bin/bash
# round-6 closing run on the GPU box: suite, bench (default line), host time of one step on both host layers, rocprofv3 stats +
# PMC traffic (headline, config 3), config 4 on one GPU, the forced-collective line.  Everything under gpurun_out/$RUN_TAG.
O=gpurun_out/${RUN_TAG:-r06k}; mkdir -p $O
( time python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/tests.log 2>&1; tail -5 $O/tests.log | cut -c1-200
python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err
for h in 1 0; do
  echo "== NVDR_HOST=$h (1 = compiled host layer, 0 = Python), tools/host_profile.py 3000: launches on" >> $O/host_profile.log
  NVDR_HOST=$h python tools/host_profile.py 3000 2>&1 | grep -E "us per step|function calls" >> $O/host_profile.log
  echo "== NVDR_HOST=$h, launches off (NVDR_DEBUG=2097152)" >> $O/host_profile.log
  NVDR_HOST=$h NVDR_DEBUG=2097152 python tools/host_profile.py 3000 2>&1 | grep -E "us per step|function calls" >> $O/host_profile.log
done
bash tools/profile_gpu.sh ${RUN_TAG:-r06k}_ch --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs > $O/prof_ch.log 2>&1
bash tools/profile_gpu.sh ${RUN_TAG:-r06k}_c3 --workload c3 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/prof_c3.log 2>&1
python bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --detail $O/c4_detail.json > $O/c4_1gpu.json 2> $O/c4.err
HSA_ENABLE_IPC_MODE_LEGACY=0 python bench.py --force-collectives --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --detail $O/collective_detail.json > $O/collective_one_rank.json 2> $O/collective.err
cat $O/host_profile.log; tail -c 400 $O/bench.json

#!/bin/bash
# SQ / TA / TCP counters of every kernel of the headline step (separate PMC passes, kernel-trace only).
# usage: tools/pmc_step.sh [kernel-substring ...]   (default: all k_* kernels)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_step
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline"
pass() { leg=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$leg -o $leg --output-format csv -- $CMD > $OUT/$leg.log 2>&1; }
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass c TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
pass d TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python - "$@" <<PY
import csv, glob, collections, sys
want = sys.argv[1:]
for leg in "abcd":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % leg, recursive=True)
    if not f: print("no csv for", leg); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        if "k_" in k and (not want or any(w in k for w in want)):
            print(leg, k, {c: round(sorted(v)[len(v)//2]) for c, v in d.items()})
PY

#!/bin/bash
# SQ counters of every kernel of a step (separate PMC passes, kernel-trace only; PMC_CMD overrides the workload).
# TA_* counters are left out on purpose: that pass aborted rocprofv3 on this image and hung until the time limit.
# usage: tools/pmc_step.sh [kernel-substring ...]   (default: all k_* kernels)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_step
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline"}
pass() { leg=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$leg -o $leg --output-format csv -- $CMD > $OUT/$leg.log 2>&1; }
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass c SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_WAVE_CYCLES
pass b SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
python - "$@" <<PY
import csv, glob, collections, sys
want = sys.argv[1:]
for leg in "abc":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % leg, recursive=True)
    if not f: print("no csv for", leg); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0][-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        if "k_" in k and (not want or any(w in k for w in want)):
            print(leg, k, {c: round(sorted(v)[len(v)//2]) for c, v in d.items()})
PY

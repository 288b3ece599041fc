#!/bin/bash
# L2 (TCC) counters of every kernel of a step: requests to the fabric with their summed in-flight levels (average latency =
# LEVEL / REQ), hits / misses, stalls.  Separate PMC passes, kernel-trace only.  TA_* and TCP_* counters are left out on
# purpose: those passes abort rocprofv3 on this image and hang until the time limit (12 GPU-minutes lost finding out).
# usage: tools/pmc_mem.sh [kernel-substring ...]      PMC_CMD overrides the workload
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_mem
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $ROOT/bench.py --steps 4 --warmup 2 --windows 1 --no-cpu-baseline --no-extra-configs"}
pass() { leg=$1; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$leg -o $leg --output-format csv -- $CMD > $OUT/$leg.log 2>&1; }
pass c TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
pass d TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
python - "$@" <<PY
import csv, glob, collections, sys
want = sys.argv[1:]
for leg in "cd":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % leg, recursive=True)
    if not f: print("no csv for", leg); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in sorted(acc.items()):
        if "nvdr" in k and (not want or any(w in k for w in want)):
            print(leg, k, {c: round(sorted(v)[len(v)//2]) for c, v in d.items()})
PY

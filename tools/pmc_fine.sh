#!/bin/bash
# SQ counters of the rasterizer kernels (one PMC pass, kernel-trace only).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_fine
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $OUT/a -o a --output-format csv -- python $ROOT/tools/raster_times.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA -d $OUT/b -o b --output-format csv -- python $ROOT/tools/raster_times.py > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for leg in "ab":
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % leg, recursive=True)
    if not f: print("no csv for", leg); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0][-28:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "k_fine" in k or "k_setup" in k:
            print(k, {c: round(sorted(v)[len(v)//2]) for c, v in d.items()})
PY

#!/bin/bash
# r05 first GPU call: suite on the new library, then A/B of the k_fine variants (tools/build_ab.sh) on ch / c2 / s10k / dense
O=gpurun_out/r05a; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log | cut -c1-300
for v in main r04 q nb nz c1; do
  if [ $v == main ]; then unset NVDR_LIB_PATH; else export NVDR_LIB_PATH=$PWD/nvdiffrast_amd/libnvdr_hip_$v.so; fi
  python tools/bench_regimes.py ch c2 s10k dense $([ $v == main ] || echo --no-check) > $O/reg_$v.jsonl 2> $O/reg_$v.err
  echo "== $v"; python - <<PY
import json
for l in open("$O/reg_$v.jsonl"):
    d=json.loads(l); k=d["kernels_ms"]
    print(d["regime"], d["ms_per_step"], "fine", k.get("raster_fine"), "setup", k.get("raster_setup"), "order", k.get("raster_order"), "mism", d.get("tri_id_mismatches_item0"))
PY
done

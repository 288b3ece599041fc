#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; tail -2 $O/tests.log | cut -c1-200
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real; cp bench_detail.json $O/ 2>/dev/null
wc -c $O/bench.json; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"].get("kernel_avg_ms"), "literal", d.get("value_literal_step"))
print("kernels", {k:v.get("ms", v) if isinstance(v,dict) else v for k,v in d["kernels"].items()})
print("parity", d["parity"])
for k,v in d["configs"].items(): print(k, {kk:v.get(kk) for kk in ("ms","ms_per_step","v","value","err","error","it_s","iters_per_s")})
PY

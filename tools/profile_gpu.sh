#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + two separate PMC passes
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they need 3 + 2).
#   tools/profile_gpu.sh <tag> [bench args...]
# Output: gpurun_out/prof_<tag>/{stats,fetch,write}/...; summarise with tools/summarize_profile.py.
set -u
TAG=${1:-run}; shift || true
ARGS=${*:---steps 10 --warmup 3 --no-cpu-baseline}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o "$TAG" --output-format csv -- python "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o "$TAG" --output-format csv -- python "$ROOT/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o "$TAG" --output-format csv -- python "$ROOT/bench.py" $ARGS > "$OUT/write.log" 2>&1
python "$ROOT/tools/summarize_profile.py" "$OUT" "$TAG"

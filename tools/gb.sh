#!/bin/bash
# build, then (only if the build succeeded) run a command on the GPU box:  tools/gb.sh '<command>'
set -e
cd "$(dirname "$0")/.."
python -m nvdiffrast_amd._build > /tmp/nvdr_build.log 2>&1 || { grep -E "error" -A6 /tmp/nvdr_build.log | head -40; echo "BUILD FAILED"; exit 1; }
/usr/local/graft/bin/gpurun --timeout ${GB_TIMEOUT:-900} -- "$1" 2>&1 | tail -${GB_TAIL:-25}

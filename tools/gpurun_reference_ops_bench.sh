#!/bin/bash
# tools/bench_reference_ops.py on the GPU box with the reference's own ops.py shipped inside the command line (never written
# into the repository).  Log -> gpurun_out/reference_ops_ch_step.log (copy to profiles/).
set -e
B64=$(base64 -w0 /root/reference/nvdiffrast/torch/ops.py)
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p /tmp/refops && echo $B64 | base64 -d > /tmp/refops/ops.py && (NVDR_REFERENCE_OPS=/tmp/refops/ops.py python tools/bench_reference_ops.py ch; NVDR_REFERENCE_OPS=/tmp/refops/ops.py python tools/bench_reference_ops.py c2) 2>/dev/null | tee gpurun_out/reference_ops_ch_step.log"

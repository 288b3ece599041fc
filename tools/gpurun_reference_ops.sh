#!/bin/bash
# Runs tests/test_gpu_reference_ops.py on the GPU box WITH the reference's own nvdiffrast/torch/ops.py available:
# the file travels inside the command line (base64) to /tmp on the box -- it is never written into the repository --
# and NVDR_REFERENCE_OPS points the test at it.  Log -> gpurun_out/reference_ops_on_plugin.log (copy to profiles/).
set -e
B64=$(base64 -w0 /root/reference/nvdiffrast/torch/ops.py)
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p /tmp/refops && echo $B64 | base64 -d > /tmp/refops/ops.py && NVDR_REFERENCE_OPS=/tmp/refops/ops.py python -m pytest tests/test_gpu_reference_ops.py -m gpu -v -rs 2>&1 | grep -v '^oracle pinned' | tail -20 | tee gpurun_out/reference_ops_on_plugin.log"

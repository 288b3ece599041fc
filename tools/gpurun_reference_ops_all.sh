#!/bin/bash
# Everything that needs the reference's own nvdiffrast/torch/ops.py on the GPU box, in ONE call: the file travels inside the
# command line (never written into the repository); tests first, then tools/bench_reference_ops.py at the headline and config-2
# batches.  Logs -> gpurun_out/reference_ops_{tests,step}_$TAG.log (copy to profiles/).
set -e
TAG=${RUN_TAG:-r05p}
B64=$(base64 -w0 /root/reference/nvdiffrast/torch/ops.py)
/usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p /tmp/refops gpurun_out && echo $B64 | base64 -d > /tmp/refops/ops.py && export NVDR_REFERENCE_OPS=/tmp/refops/ops.py && (timeout 600 python -m pytest tests/test_gpu_plugin_fused_backward.py tests/test_gpu_fused_backward.py tests/test_gpu_reference_ops.py -m gpu -q -rs 2>&1 | grep -v '^oracle pinned' | tail -25 | tee gpurun_out/reference_ops_tests_$TAG.log); (timeout 200 python tools/bench_reference_ops.py ch; timeout 200 python tools/bench_reference_ops.py c2) 2>&1 | grep -v Warning | tail -12 | tee gpurun_out/reference_ops_step_$TAG.log"

#!/bin/bash
# Runs tests/test_gpu_reference_samples.py on the GPU box WITH the reference's own sample programs and fixtures:
# /root/reference/samples is copied into .refdata/ (git-ignored, so it is never committed, but it travels with the
# gpurun snapshot), the tests run with NVDR_REFERENCE_SAMPLES pointing at it, and the copy is deleted again whatever
# happens.  Log -> gpurun_out/reference_samples.log (copy to profiles/).
cd "$(dirname "$0")/.."
trap 'rm -rf .refdata' EXIT
rm -rf .refdata && mkdir -p .refdata && cp -r /root/reference/samples .refdata/samples
/usr/local/graft/bin/gpurun --timeout 1500 -- 'NVDR_REFERENCE_SAMPLES=$PWD/.refdata/samples python -m pytest tests/test_gpu_reference_samples.py -m gpu -v -s -rs 2>&1 | grep -v "^oracle pinned\|amdgpu.ids" | grep -v "^iter=\|^rep=" | tail -40 | tee gpurun_out/reference_samples.log'

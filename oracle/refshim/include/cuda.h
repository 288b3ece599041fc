// Stand-in for the CUDA toolkit header of the same name (test infrastructure, see nvdr_cuda_shim.h).
#pragma once
#include "nvdr_cuda_shim.h"

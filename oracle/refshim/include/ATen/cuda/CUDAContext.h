// Stand-in for the PyTorch/pybind11 header of the same name (test infrastructure, see torch/extension.h).
#pragma once
#include <torch/extension.h>

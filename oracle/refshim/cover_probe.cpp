// Probe into the reference's fine-raster coverage functions (cudaraster/impl/Util.inl:214-359), compiled from the
// reference's own (PTX-patched, see build.py) Util.inl + TriangleSetup.inl: for a batch of edges it returns the 8x8
// coverage mask of the LIVE path (cover8x8_exact_fast: flip bits from cover8x8_selectFlips + the 768-entry LUT built
// by cover8x8_setupLUT) and of the reference's own non-LUT statement of the same rule (cover8x8_exact_noLUT).
// SURVEY Appendix A3 asks for the equivalence to be kept as a property test (tests/test_ref_pins_oracle.py).
// TEST INFRASTRUCTURE ONLY.
#include <cuda_runtime.h>
#include "../CudaRaster.hpp"
#include "PrivateDefs.hpp"
#include "Constants.hpp"
#include "Util.inl"

namespace CR
{
#include "TriangleSetup.inl"
}

struct CoverProbeParams
{
    const int*          edges;      // n x (ox, oy, dx, dy): vertex relative to the tile's first pixel centre, edge vector; subpixels
    unsigned long long* masks;      // n x (LUT path, non-LUT path)
    int                 n;
    int                 pad[3];
};

void coverProbeKernel(const CoverProbeParams p)
{
    __shared__ volatile CR::U64 s_lut[CR_COVER8X8_LUT_SIZE];
    CR::cover8x8_setupLUT(s_lut);
    __syncthreads();
    int tid = threadIdx.x + blockDim.x * threadIdx.y;
    int nthreads = blockDim.x * blockDim.y;
    for (int i = tid; i < p.n; i += nthreads)
    {
        int ox = p.edges[4 * i + 0], oy = p.edges[4 * i + 1], dx = p.edges[4 * i + 2], dy = p.edges[4 * i + 3];
        CR::U32 flips = CR::cover8x8_selectFlips(dx, dy);
        p.masks[2 * i + 0] = CR::cover8x8_exact_fast(ox, oy, dx, dy, flips, s_lut);
        p.masks[2 * i + 1] = CR::cover8x8_exact_noLUT(ox, oy, dx, dy);
    }
}

extern "C" int nvdr_ref_cover8x8_probe(const int* edges, unsigned long long* masks, int n)
{
    CoverProbeParams p;
    memset(&p, 0, sizeof(p));
    p.edges = edges; p.masks = masks; p.n = n;
    void* args[] = {&p};
    return (int)cudaLaunchKernel((void*)coverProbeKernel, dim3(1, 1, 1), dim3(32, 2, 1), args, 0, 0);
}

// Workgroup dispatch rate and float4 copy ceiling at the interpolate-forward shape (16 B in, 16 B out per pixel).
// hipcc --offload-arch=gfx950 -O3 tools/dispatch_bench.hip -o tools/dispatch_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

template <int LDSB>
__global__ void k_empty_lds(int* p)
{
    __shared__ int s[LDSB / 4];
    if (threadIdx.x == 9999) { s[0] = 1; p[0] = s[1]; }
}

template <int PER>
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, long n)
{
    long i = ((long)blockIdx.x * 256 + threadIdx.x);
    long stride = (long)gridDim.x * 256;
    float4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { long j = i + k * stride; if (j < n) v[k] = a[j]; }
#pragma unroll
    for (int k = 0; k < PER; ++k) { long j = i + k * stride; if (j < n) b[j] = v[k]; }
}

__global__ __launch_bounds__(256) void k_copy_loop(const float4* __restrict__ a, float4* __restrict__ b, long n)
{
    long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) b[i] = a[i];
}

template <class F>
static float timeit(F f, int reps = 20)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2] * 1000.f;
}

int main()
{
    int* d; hipMalloc(&d, 64);
    for (int threads : {64, 256, 512, 1024})
        for (int blocks : {256, 2560, 10240, 65536, 262144}) {
            float us = timeit([&] { k_empty<<<blocks, threads>>>(d); });
            printf("empty  thr %4d blocks %6d : %8.2f us  (%.1f waves/us)\n", threads, blocks, us,
                   blocks * (threads / 64.0) / us);
        }
    for (int blocks : {2560, 65536}) {
        float us = timeit([&] { k_empty_lds<40960><<<blocks, 256>>>(d); });
        printf("empty+40KB LDS thr 256 blocks %6d : %8.2f us\n", blocks, us);
    }
    long n = 64l * 512 * 512;
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16);
    auto rep = [&](const char* name, float us) { printf("copy %-18s : %8.2f us  %.2f TB/s\n", name, us, 2.0 * n * 16 / us * 1e-6); };
    rep("1/thread", timeit([&] { k_copy<1><<<(n + 255) / 256, 256>>>(a, b, n); }));
    rep("2/thread", timeit([&] { k_copy<2><<<(n / 2 + 255) / 256, 256>>>(a, b, n); }));
    rep("4/thread", timeit([&] { k_copy<4><<<(n / 4 + 255) / 256, 256>>>(a, b, n); }));
    rep("8/thread", timeit([&] { k_copy<8><<<(n / 8 + 255) / 256, 256>>>(a, b, n); }));
    rep("loop 2048 blocks", timeit([&] { k_copy_loop<<<2048, 256>>>(a, b, n); }));
    rep("loop 4096 blocks", timeit([&] { k_copy_loop<<<4096, 256>>>(a, b, n); }));
    rep("loop 8192 blocks", timeit([&] { k_copy_loop<<<8192, 256>>>(a, b, n); }));
    rep("hipMemcpyDtoD", timeit([&] { hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); }));
    return 0;
}

// Development microbenchmark: global f32 atomic-add throughput on gfx950 by scope bits, footprint
// and address pattern.  Informs the gradient-scatter design (texture grad, vertex grads).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// MODE: 0 = no scope bits, 1 = sc1, 2 = sc0 sc1, 3 = f64 atomic, 4 = plain store (baseline), 5 = nt
template <int MODE, int PATTERN>
__global__ __launch_bounds__(256) void k_atom(float* buf, size_t n, int iters)
{
    const int lane = threadIdx.x & 63;
    uint32_t h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t hw = ((blockIdx.x * 4u + (threadIdx.x >> 6)) + 1u) * 2246822519u;   // per-wave stream
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        hw = hw * 1664525u + 1013904223u;
        size_t a;
        if (PATTERN == 0) a = (size_t)(h >> 4) % n;                               // every lane random
        else if (PATTERN == 1) a = (((size_t)(hw >> 4) % (n / 64)) * 64) + lane;  // wave = 256 contiguous bytes
        else a = (((size_t)(hw >> 4) % (n / 64)) * 64) + (lane & 7) + 8 * ((h >> 9) & 7);  // 8x conflicts inside a wave, 1 line
        float* p = buf + a;
        float v = 1.0f;
        if (MODE == 0) asm volatile("global_atomic_add_f32 %0, %1, off" :: "v"(p), "v"(v) : "memory");
        if (MODE == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
        if (MODE == 2) asm volatile("global_atomic_add_f32 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
        if (MODE == 3) { double d = 1.0; double* q = (double*)(buf + (a & ~(size_t)1)); asm volatile("global_atomic_add_f64 %0, %1, off" :: "v"(q), "v"(d) : "memory"); }
        if (MODE == 4) asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
        if (MODE == 5) asm volatile("global_atomic_add_f32 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    }
}

int main()
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {256u << 10, 4u << 20, 64u << 20};        // floats: 1 MB, 16 MB, 256 MB
    float* buf; hipMalloc(&buf, sizes[2] * 4); hipMemset(buf, 0, sizes[2] * 4);
    const int blocks = 256 * 8, iters = 256;
    auto run = [&](auto kern, const char* name, size_t n) {
        kern<<<blocks, 256>>>(buf, n, 8);
        hipDeviceSynchronize();
        hipEventRecord(e0); kern<<<blocks, 256>>>(buf, n, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double ops = (double)blocks * 256 * iters;
        printf("%-34s %4zu MB: %8.3f ms  %7.1f G lane-ops/s\n", name, n * 4 >> 20, ms, ops / ms / 1e6);
    };
    for (size_t n : sizes) {
        run(k_atom<0, 0>, "add_f32        random", n);
        run(k_atom<0, 1>, "add_f32        wave-contiguous", n);
        run(k_atom<0, 2>, "add_f32        8-way same addr", n);
        run(k_atom<1, 0>, "add_f32 sc1    random", n);
        run(k_atom<1, 1>, "add_f32 sc1    wave-contiguous", n);
        run(k_atom<2, 1>, "add_f32 sc1 nt wave-contiguous", n);
        run(k_atom<5, 1>, "add_f32 nt     wave-contiguous", n);
        run(k_atom<3, 0>, "add_f64        random", n);
        run(k_atom<3, 1>, "add_f64        wave-contiguous", n);
        run(k_atom<4, 0>, "store_dword    random", n);
        run(k_atom<4, 1>, "store_dword    wave-contiguous", n);
    }
    return 0;
}

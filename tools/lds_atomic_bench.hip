// Development microbenchmark: throughput of LDS f32 atomic adds / u64 atomic min / CAS on gfx950
// as a function of same-address conflict degree, and of global f32 atomics (packed vs single lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float* out, int iters, int conflict)
{
    __shared__ float s[4096];
    __shared__ unsigned long long s64[1024];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    for (int i = threadIdx.x; i < 1024; i += 256) s64[i] = ~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // `conflict` lanes share one address
    int idx = wave * 64 + (lane / conflict);
    float v = 1.0f + lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            int a = (idx + k * 256 + it) & 4095;
            if (MODE == 0) unsafeAtomicAdd(&s[a], v);
            if (MODE == 1) atomicMin(&s64[a & 1023], ((unsigned long long)(it + k) << 32) | lane);
            if (MODE == 2) atomicCAS((unsigned int*)&s[a], 0u, (unsigned)lane + 1u);
            if (MODE == 3) s[a] = v;                       // plain store baseline
            if (MODE == 4) atomicAdd(&s64[a & 1023], (unsigned long long)(it + lane));
            if (MODE == 5) atomicAdd((unsigned int*)&s[a], (unsigned)lane);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[5] + (float)s64[7];
}

template <int MODE>
__global__ __launch_bounds__(256) void k_glob(float* buf, int n, int iters)
{
    const int lane = threadIdx.x & 63;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < iters; it++) {
        h = h * 1664525u + 1013904223u;
        int a = (h >> 8) % n;
        if (MODE == 0) unsafeAtomicAdd(&buf[a], 1.0f);                    // all 64 lanes, random addresses
        if (MODE == 1) { if (lane == 63) unsafeAtomicAdd(&buf[a], 1.0f); } // single lane
        if (MODE == 2) unsafeAtomicAdd(&buf[(a & ~63) + lane], 1.0f);      // 64 lanes, 256 contiguous bytes
    }
}

int main()
{
    float* out; hipMalloc(&out, 1 << 20);
    float* buf; const int n = 5151 * 4 * 64; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8, iters = 200;
    auto run = [&](auto kern, const char* name, int conflict) {
        kern<<<blocks, 256>>>(out, 10, conflict);
        hipDeviceSynchronize();
        hipEventRecord(a); kern<<<blocks, 256>>>(out, iters, conflict); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double wave_instr = (double)blocks * 4 * iters * 12;
        printf("%-14s conflict %2d: %.3f ms  %.2f G wave-instr/s  = %.1f cycles/instr/CU (2.4GHz)\n", name, conflict, ms,
               wave_instr / ms / 1e6, 2.4e9 / (wave_instr / (ms * 1e-3) / 256));
    };
    for (int c : {1, 2, 4, 8, 16, 64}) run(k_lds<0>, "ds_add_f32", c);
    for (int c : {1, 4, 16, 64}) run(k_lds<1>, "ds_min_u64", c);
    for (int c : {1, 4, 16}) run(k_lds<2>, "ds_cmpst", c);
    run(k_lds<3>, "ds_write", 1);
    for (int c : {1, 4, 16}) run(k_lds<4>, "ds_add_u64", c);
    for (int c : {1, 4, 16}) run(k_lds<5>, "ds_add_u32", c);
    auto rung = [&](auto kern, const char* name, int lanes) {
        kern<<<blocks, 256>>>(buf, n, 10);
        hipDeviceSynchronize();
        const int it = 500;
        hipEventRecord(a); kern<<<blocks, 256>>>(buf, n, it); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double instr = (double)blocks * 4 * it;
        printf("%-22s: %.3f ms  %.2f G wave-instr/s  %.1f G lane-ops/s\n", name, ms, instr / ms / 1e6, instr * lanes / ms / 1e6);
    };
    rung(k_glob<0>, "global f32 64 random", 64);
    rung(k_glob<1>, "global f32 1 lane", 1);
    rung(k_glob<2>, "global f32 64 contig", 64);
    return 0;
}

// Runtime of the CUDA-on-CPU shim (see include/nvdr_cuda_shim.h): fibre scheduler that executes one CUDA
// block at a time with 32-lane warps, plus the handful of CUDA runtime calls the reference's host code makes.
// TEST INFRASTRUCTURE ONLY.
#include "nvdr_cuda_shim.h"
#include <stdio.h>
#include <sys/mman.h>
#include <vector>

thread_local uint3 threadIdx;
thread_local uint3 blockIdx;
thread_local dim3  blockDim;
thread_local dim3  gridDim;
#define NVDR_SHIM_DYN_SMEM_FLOATS (1 << 16)
thread_local float s_texelAccum[NVDR_SHIM_DYN_SMEM_FLOATS];

//------------------------------------------------------------------------ context switch (x86-64 SysV)
// Saves the callee-saved registers on the current stack, stores the stack pointer through `save`, installs
// `load` and restores.  A fresh fibre's stack is laid out as if it had been suspended at the entry of
// fibre_entry().
extern "C" void nvdr_shim_switch(void** save, void* load);
asm(R"(
    .text
    .globl nvdr_shim_switch
    .type nvdr_shim_switch,@function
nvdr_shim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size nvdr_shim_switch,.-nvdr_shim_switch
)");

namespace
{
enum { ST_RUN = 0, ST_WAIT_WARP, ST_WAIT_BLOCK, ST_DONE };
const size_t STACK_BYTES = 256u << 10;
const int    MAX_THREADS = 1024;
const int    MAX_WARPS   = MAX_THREADS / 32;

struct ParamBlob { unsigned char bytes[NVDR_SHIM_MAX_PARAM_BYTES]; };
typedef void (*KernelFn)(ParamBlob);

struct Fibre
{
    void*    sp;
    uint3    tid;
    int      lane, warp;
    int      state;
    int      op;
    unsigned mask, value, aux, result;
};

struct Warp { unsigned live, arrived, runnable; };

struct BlockState
{
    Fibre     fibres[MAX_THREADS];
    Warp      warps[MAX_WARPS];
    int       numThreads, numWarps, numLive, numAtBarrier;
    Fibre*    cur;
    void*     mainSp;
    KernelFn  kernel;
    ParamBlob params;
    char*     stacks;
    bool      deadlock;
};

thread_local BlockState* g_bs = 0;

inline void set_state(BlockState* bs, Fibre* f, int st)
{
    f->state = st;
    unsigned bit = 1u << f->lane;
    if (st == ST_RUN) bs->warps[f->warp].runnable |= bit;
    else              bs->warps[f->warp].runnable &= ~bit;
}

// Next runnable fibre after `from`: same warp first (lanes above, then below), then the following warps.
Fibre* pick_next(BlockState* bs, Fibre* from)
{
    int w0 = from ? from->warp : 0;
    if (from)
    {
        unsigned r = bs->warps[w0].runnable & ~(1u << from->lane);
        unsigned above = r & ~((2u << from->lane) - 1u);
        if (above) return &bs->fibres[w0 * 32 + __builtin_ctz(above)];
        if (r)     return &bs->fibres[w0 * 32 + __builtin_ctz(r)];
    }
    for (int i = from ? 1 : 0; i <= bs->numWarps; i++)
    {
        int w = (w0 + i) % bs->numWarps;
        unsigned r = bs->warps[w].runnable;
        if (from && w == w0) r &= ~(1u << from->lane);
        if (r) return &bs->fibres[w * 32 + __builtin_ctz(r)];
    }
    return 0;
}

void switch_to(BlockState* bs, Fibre* from, Fibre* to)
{
    bs->cur = to;
    threadIdx = to->tid;
    nvdr_shim_switch(&from->sp, to->sp);
}

// Give up the processor until this fibre is runnable again.
void wait_until_runnable(BlockState* bs, Fibre* me)
{
    while (me->state != ST_RUN)
    {
        Fibre* next = pick_next(bs, me);
        if (!next)
        {
            bs->deadlock = true;                         // nobody can make progress: report from the main context
            nvdr_shim_switch(&me->sp, bs->mainSp);
            continue;
        }
        switch_to(bs, me, next);
    }
}

// Complete the warp collective `f` is waiting on if every live lane of its mask has arrived.
bool try_complete_warp(BlockState* bs, Fibre* f)
{
    Warp& w = bs->warps[f->warp];
    unsigned need = f->mask & w.live;
    if ((w.arrived & need) != need)
        return false;
    Fibre* base = &bs->fibres[f->warp * 32];
    unsigned ballot = 0;
    for (unsigned m = need; m; m &= m - 1)
    {
        int l = __builtin_ctz(m);
        if (base[l].value) ballot |= 1u << l;
    }
    for (unsigned m = need; m; m &= m - 1)
    {
        int l = __builtin_ctz(m);
        Fibre& g = base[l];
        unsigned r = 0;
        switch (g.op)
        {
        case nvdr_shim::OP_SYNC:   r = 0; break;
        case nvdr_shim::OP_BALLOT: r = ballot; break;
        case nvdr_shim::OP_ANY:    r = (ballot != 0); break;
        case nvdr_shim::OP_ALL:    r = (ballot == need); break;
        case nvdr_shim::OP_MATCH_ANY:
            for (unsigned k = need; k; k &= k - 1)
            {
                int j = __builtin_ctz(k);
                if (base[j].value == g.value) r |= 1u << j;
            }
            break;
        case nvdr_shim::OP_SHFL:   r = (need >> g.aux & 1u) ? base[g.aux].value : g.value; break;
        }
        g.result = r;
    }
    for (unsigned m = need; m; m &= m - 1)
        set_state(bs, &base[__builtin_ctz(m)], ST_RUN);
    w.arrived &= ~need;
    return true;
}

void release_barrier_if_complete(BlockState* bs)
{
    if (bs->numAtBarrier == 0 || bs->numAtBarrier != bs->numLive)
        return;
    for (int i = 0; i < bs->numThreads; i++)
        if (bs->fibres[i].state == ST_WAIT_BLOCK)
            set_state(bs, &bs->fibres[i], ST_RUN);
    bs->numAtBarrier = 0;
}

void fibre_exit(BlockState* bs, Fibre* me)
{
    set_state(bs, me, ST_DONE);
    Warp& w = bs->warps[me->warp];
    w.live &= ~(1u << me->lane);
    bs->numLive--;
    // An exited thread counts as arrived: pending collectives of its warp and the block barrier may complete.
    for (bool again = true; again; )
    {
        again = false;
        for (unsigned m = w.arrived; m; m &= m - 1)
            if (try_complete_warp(bs, &bs->fibres[me->warp * 32 + __builtin_ctz(m)])) { again = true; break; }
    }
    release_barrier_if_complete(bs);
    Fibre* next = pick_next(bs, me);
    if (next)
        switch_to(bs, me, next);
    else
    {
        if (bs->numLive != 0) bs->deadlock = true;
        nvdr_shim_switch(&me->sp, bs->mainSp);
    }
    abort();    // a finished fibre is never resumed
}

extern "C" void nvdr_shim_fibre_entry(void)
{
    BlockState* bs = g_bs;
    Fibre* me = bs->cur;
    bs->kernel(bs->params);
    fibre_exit(bs, me);
}

void run_block(BlockState* bs)
{
    int n = bs->numThreads;
    bs->numWarps = (n + 31) / 32;
    bs->numLive = n;
    bs->numAtBarrier = 0;
    bs->deadlock = false;
    for (int w = 0; w < bs->numWarps; w++)
    {
        int cnt = (n - w * 32) < 32 ? (n - w * 32) : 32;
        unsigned m = (cnt == 32) ? 0xffffffffu : ((1u << cnt) - 1u);
        bs->warps[w].live = m;
        bs->warps[w].arrived = 0;
        bs->warps[w].runnable = m;
    }
    for (int i = 0; i < n; i++)
    {
        Fibre& f = bs->fibres[i];
        f.tid.x = i % blockDim.x;
        f.tid.y = (i / blockDim.x) % blockDim.y;
        f.tid.z = i / (blockDim.x * blockDim.y);
        f.lane = i & 31;
        f.warp = i >> 5;
        f.state = ST_RUN;
        f.op = 0; f.mask = 0; f.value = 0; f.aux = 0; f.result = 0;
        // Stack image of a suspended fibre: six callee-saved registers, then the address `ret` jumps to.
        // After that `ret` the stack pointer is 8 mod 16, as at any function entry.
        uintptr_t top = ((uintptr_t)(bs->stacks + (size_t)(i + 1) * STACK_BYTES)) & ~(uintptr_t)15;
        void** s = (void**)top;
        *--s = 0;                                   // fake return address of the entry function
        *--s = (void*)&nvdr_shim_fibre_entry;
        for (int k = 0; k < 6; k++) *--s = 0;
        f.sp = (void*)s;
    }
    Fibre* first = &bs->fibres[0];
    bs->cur = first;
    threadIdx = first->tid;
    nvdr_shim_switch(&bs->mainSp, first->sp);
    if (bs->deadlock || bs->numLive != 0)
    {
        fprintf(stderr, "nvdr_cuda_shim: deadlock in block (%u,%u,%u): %d live threads, %d at __syncthreads\n",
                blockIdx.x, blockIdx.y, blockIdx.z, bs->numLive, bs->numAtBarrier);
        for (int i = 0; i < n && i < 64; i++)
            fprintf(stderr, "  t%d state %d op %d mask %08x\n", i, bs->fibres[i].state, bs->fibres[i].op, bs->fibres[i].mask);
        abort();
    }
}

BlockState* get_block_state(void)
{
    if (!g_bs)
    {
        g_bs = new BlockState();
        void* p = mmap(0, STACK_BYTES * MAX_THREADS, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { perror("nvdr_cuda_shim: mmap"); abort(); }
        g_bs->stacks = (char*)p;
    }
    return g_bs;
}

thread_local cudaError_t g_lastError = cudaSuccess;
} // namespace

//------------------------------------------------------------------------ device-side entry points
namespace nvdr_shim
{
void block_barrier(void)
{
    BlockState* bs = g_bs;
    Fibre* me = bs->cur;
    set_state(bs, me, ST_WAIT_BLOCK);
    bs->numAtBarrier++;
    release_barrier_if_complete(bs);
    wait_until_runnable(bs, me);
}

unsigned warp_collective(int op, unsigned mask, unsigned value, unsigned aux)
{
    BlockState* bs = g_bs;
    Fibre* me = bs->cur;
    me->op = op; me->mask = mask; me->value = value; me->aux = aux;
    set_state(bs, me, ST_WAIT_WARP);
    bs->warps[me->warp].arrived |= 1u << me->lane;
    try_complete_warp(bs, me);
    wait_until_runnable(bs, me);
    return me->result;
}

unsigned lane_id(void) { return (unsigned)g_bs->cur->lane; }
}

//------------------------------------------------------------------------ host-side runtime API
cudaError_t cudaGetLastError(void) { cudaError_t e = g_lastError; g_lastError = cudaSuccess; return e; }
cudaError_t cudaGetDevice(int* dev) { *dev = 0; return cudaSuccess; }
// One "SM", one resident block per kernel: persistent kernels are launched with one block per image.
cudaError_t cudaDeviceGetAttribute(int* value, cudaDeviceAttr, int) { *value = 1; return cudaSuccess; }
cudaError_t cudaFuncGetAttributes(cudaFuncAttributes* attr, const void*)
{
    memset(attr, 0, sizeof(*attr));
    attr->maxThreadsPerBlock = MAX_THREADS;
    return cudaSuccess;
}
cudaError_t cudaFuncSetCacheConfig(const void*, cudaFuncCache) { return cudaSuccess; }
cudaError_t cudaOccupancyMaxActiveBlocksPerMultiprocessor(int* numBlocks, const void*, int, size_t) { *numBlocks = 1; return cudaSuccess; }
cudaError_t cudaMalloc(void** p, size_t bytes)
{
    *p = 0;
    if (posix_memalign(p, 256, bytes ? bytes : 1)) return g_lastError = cudaErrorMemoryAllocation;
    memset(*p, 0xcd, bytes);                        // device memory is not zero-initialised
    return cudaSuccess;
}
cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
cudaError_t cudaMallocHost(void** p, size_t bytes) { return cudaMalloc(p, bytes); }
cudaError_t cudaFreeHost(void* p) { free(p); return cudaSuccess; }
cudaError_t cudaMemcpyAsync(void* dst, const void* src, size_t bytes, cudaMemcpyKind, cudaStream_t) { memmove(dst, src, bytes); return cudaSuccess; }
cudaError_t cudaMemcpy(void* dst, const void* src, size_t bytes, cudaMemcpyKind) { memmove(dst, src, bytes); return cudaSuccess; }
cudaError_t cudaMemsetAsync(void* dst, int value, size_t bytes, cudaStream_t) { memset(dst, value, bytes); return cudaSuccess; }
cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
cudaError_t cudaDeviceSynchronize(void) { return cudaSuccess; }

cudaError_t cudaLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t sharedMem, cudaStream_t)
{
    size_t nthreads = (size_t)block.x * block.y * block.z;
    if (!func || !args || nthreads == 0 || nthreads > (size_t)MAX_THREADS || sharedMem > sizeof(float) * NVDR_SHIM_DYN_SMEM_FLOATS)
        return g_lastError = cudaErrorInvalidValue;
    if ((size_t)grid.x * grid.y * grid.z == 0)
        return g_lastError = cudaErrorInvalidValue;
    BlockState* bs = get_block_state();
    bs->kernel = (KernelFn)func;
    memcpy(bs->params.bytes, args[0], NVDR_SHIM_MAX_PARAM_BYTES);
    bs->numThreads = (int)nthreads;
    blockDim = block;
    gridDim = grid;
    for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
    for (unsigned x = 0; x < grid.x; x++)
    {
        blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
        run_block(bs);
    }
    return cudaSuccess;
}

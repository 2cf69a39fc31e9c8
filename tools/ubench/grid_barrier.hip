// Micro-benchmark: what a grid-wide barrier costs on gfx950 (the price of one phase boundary inside a persistent kernel, to set beside
// the 4.8-5.0 us device-time floor of a separate launch).  Flat = one agent-scope counter; hier = one counter per XCD (blockIdx % 8)
// whose last arriver bumps the chip-wide one.  Also: a chain of K empty launches for the floor itself.
// Build: hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned ld_acq(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

template <bool HIER>
__global__ __launch_bounds__(256) void bar_kernel(unsigned* ctr, int rounds, float* sink, int work) {
    const unsigned nb = gridDim.x;
    float acc = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int w = 0; w < work; ++w) acc = fmaf(acc, 0.999f, 0.5f);
        __syncthreads();
        if (threadIdx.x == 0) {
            if (!HIER) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (ld_acq(ctr) < (unsigned)r * nb) __builtin_amdgcn_s_sleep(1);
            } else {
                const unsigned x = blockIdx.x & 7u, per = (nb + 7u - x) / 8u;     // blocks with blockIdx % 8 == x
                unsigned* local = ctr + 32 * (1 + x);
                unsigned old = __hip_atomic_fetch_add(local, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == (unsigned)r * per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (ld_acq(ctr) < (unsigned)r * 8u) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// slots: every workgroup stores its round number into its own word (no read-modify-write anywhere); workgroup 0's 256 threads watch
// the slots and one of them publishes the round in a flag on its own cache line; everybody else polls only that flag
__global__ __launch_bounds__(256) void bar_slots_kernel(unsigned* ctr, int rounds, float* sink, int work, int sleep) {
    const unsigned nb = gridDim.x;
    unsigned* slots = ctr + 1024;          // nb words
    unsigned* flag = ctr;                  // own line
    float acc = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int w = 0; w < work; ++w) acc = fmaf(acc, 0.999f, 0.5f);
        __syncthreads();
        if (blockIdx.x == 0) {
            bool ok;
            do {
                ok = true;
                for (unsigned i = threadIdx.x; i < nb; i += 256) if (i && ld_acq(slots + i) < (unsigned)r) ok = false;
            } while (!__syncthreads_and(ok));
            if (threadIdx.x == 0) __hip_atomic_store(flag, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (threadIdx.x == 0) {
            __hip_atomic_store(slots + blockIdx.x, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (ld_acq(flag) < (unsigned)r) if (sleep) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// counter + flag: arrivals are one atomic add each, nobody polls the counter: the last arriver publishes the round in a flag
__global__ __launch_bounds__(256) void bar_flag_kernel(unsigned* ctr, int rounds, float* sink, int work) {
    const unsigned nb = gridDim.x;
    unsigned* flag = ctr + 64;
    float acc = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int w = 0; w < work; ++w) acc = fmaf(acc, 0.999f, 0.5f);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (unsigned)r * nb) __hip_atomic_store(flag, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else while (ld_acq(flag) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

// the same with RELAXED atomics (no L2 write-back / invalidate: what is left when the data between phases moves with sc1 stores and
// loads of its own): the pure cost of the rendezvous
__global__ __launch_bounds__(256) void bar_relaxed_kernel(unsigned* ctr, int rounds, float* sink, int work) {
    const unsigned nb = gridDim.x;
    unsigned* flag = ctr + 64;
    float acc = threadIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        for (int w = 0; w < work; ++w) acc = fmaf(acc, 0.999f, 0.5f);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned old = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (unsigned)r * nb) __hip_atomic_store(flag, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void empty_kernel(float* sink, int work) {
    float acc = threadIdx.x;
    for (int w = 0; w < work; ++w) acc = fmaf(acc, 0.999f, 0.5f);
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    unsigned* ctr; float* sink;
    hipMalloc(&ctr, 16384); hipMalloc(&sink, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int rounds = 2000;
    for (int hier = 0; hier < 2; ++hier)
        for (int blocks : {64, 128, 256, 512, 1024})
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(ctr, 0, 16384);
                hipEventRecord(e0);
                if (hier) hipLaunchKernelGGL(bar_kernel<true>, dim3(blocks), dim3(256), 0, 0, ctr, rounds, sink, 0);
                else hipLaunchKernelGGL(bar_kernel<false>, dim3(blocks), dim3(256), 0, 0, ctr, rounds, sink, 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("%s barrier, %4d workgroups: %.2f us per barrier\n", hier ? "hier" : "flat", blocks, ms * 1e3 / rounds);
            }
    for (int kind = 0; kind < 4; ++kind)
        for (int blocks : {64, 128, 256, 512, 1024})
            for (int rep = 0; rep < 2; ++rep) {
                hipMemset(ctr, 0, 16384);
                hipEventRecord(e0);
                if (kind == 3) hipLaunchKernelGGL(bar_relaxed_kernel, dim3(blocks), dim3(256), 0, 0, ctr, rounds, sink, 0);
                else if (kind == 2) hipLaunchKernelGGL(bar_flag_kernel, dim3(blocks), dim3(256), 0, 0, ctr, rounds, sink, 0);
                else hipLaunchKernelGGL(bar_slots_kernel, dim3(blocks), dim3(256), 0, 0, ctr, rounds, sink, 0, kind);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("%s barrier, %4d workgroups: %.2f us per barrier\n", kind == 3 ? "counter+flag, relaxed atomics" : kind == 2 ? "counter+flag" : kind ? "slots (sleep)" : "slots (spin)", blocks, ms * 1e3 / rounds);
            }
    for (int blocks : {256, 1024})
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < rounds; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, 0, sink, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("empty launches, %4d workgroups: %.2f us per launch (back to back on one stream)\n", blocks, ms * 1e3 / rounds);
        }
    return 0;
}

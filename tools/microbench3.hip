// Microbench 3: raw LDS atomic throughput on gfx950 (no global memory in the loop).
// Each workgroup hammers a 8192-cell table in LDS with pseudo-random cell indices.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int CELLS = 8192;
// MODE: 0 u32 add, 1 u64 add, 2 f32 add, 3 f64 add, 4 f64 max, 5 u32 add returning, 6 f64 add via CAS loop,
//       7 u32 + f64 (count+sum), 8 u32+u32+f64, 9 f64 add returning, 10 two f64 adds
template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, unsigned long long *sink, int spread) {
    __shared__ __attribute__((aligned(16))) unsigned long long t64[CELLS];
    __shared__ unsigned int t32[CELLS];
    __shared__ unsigned int t32b[CELLS];
    for (int c = threadIdx.x; c < CELLS; c += blockDim.x) { t64[c] = 0; t32[c] = 0; t32b[c] = 0; }
    __syncthreads();
    uint32_t s = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const uint32_t c = (s >> 8) % (uint32_t)spread;
        const double w = (double)(s & 0xffff) * 1e-3;
        if (MODE == 0) atomicAdd(&t32[c], 1u);
        if (MODE == 1) atomicAdd(&t64[c], 1ull);
        if (MODE == 2) unsafeAtomicAdd((float *)&t32[c], (float)w);
        if (MODE == 3) unsafeAtomicAdd((double *)&t64[c], w);
        if (MODE == 4) __hip_atomic_fetch_max((double *)&t64[c], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 5) acc += atomicAdd(&t32[c], 1u);
        if (MODE == 6) {
            unsigned long long *p = &t64[c];
            unsigned long long old = *p, assumed;
            do { assumed = old; old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(__longlong_as_double((long long)assumed) + w)); } while (old != assumed);
        }
        if (MODE == 7) { atomicAdd(&t32[c], 1u); unsafeAtomicAdd((double *)&t64[c], w); }
        if (MODE == 8) { atomicAdd(&t32[c], 1u); atomicAdd(&t32b[c], 1u); unsafeAtomicAdd((double *)&t64[c], w); }
        if (MODE == 9) acc += (unsigned long long)unsafeAtomicAdd((double *)&t64[c], w);
        if (MODE == 10) { unsafeAtomicAdd((double *)&t64[c], w); unsafeAtomicAdd((double *)&t64[(c + 4096) & (CELLS - 1)], w); }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = t64[1] + t32[2] + t32b[3] + acc;
}

template <int MODE>
void run(const char *name, int blocks, int bs, int spread, unsigned long long *sink) {
    const int iters = 2048;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<blocks, bs>>>(iters, sink, spread); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
        CK(hipEventRecord(a)); k<MODE><<<blocks, bs>>>(iters, sink, spread); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    double ops = (double)blocks * bs * iters;
    printf("%-34s blocks=%4d x%4d spread=%5d  %8.3f ms  %8.1f G iters/s  %6.2f clk/CU per iter (2.4GHz)\n", name, blocks, bs, spread, best, ops / best / 1e6, best * 1e-3 * 2.4e9 * 256 / ops);
    fflush(stdout);
}

int main() {
    unsigned long long *sink; CK(hipMalloc(&sink, 8 * 4096));
    for (int bs : {1024, 512}) {
        int blocks = 256 * (1024 / bs);
        for (int spread : {8192, 64}) {
            run<0>("u32 add", blocks, bs, spread, sink);
            run<1>("u64 add", blocks, bs, spread, sink);
            run<2>("f32 add", blocks, bs, spread, sink);
            run<3>("f64 add", blocks, bs, spread, sink);
            run<4>("f64 max", blocks, bs, spread, sink);
            run<5>("u32 add returning", blocks, bs, spread, sink);
            run<9>("f64 add returning", blocks, bs, spread, sink);
            run<6>("f64 add via CAS loop", blocks, bs, spread, sink);
            run<7>("u32 add + f64 add", blocks, bs, spread, sink);
            run<8>("u32 + u32 + f64", blocks, bs, spread, sink);
            run<10>("f64 + f64", blocks, bs, spread, sink);
        }
    }
    return 0;
}

// Microbench 2: can S slab-owner workgroups on one XCD share their input through the XCD's L2?
// (v1 showed zero reuse with contiguous — load-imbalanced — slabs: total CU ingest stuck at ~6 TB/s.)
// Variants: interleaved slabs (cell % S), optional soft throttle (no block runs more than D tiles ahead
// of the slowest of its group), block sizes; plus the raw LDS f64-atomic rate.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e = (x);                                                                            \
        if (e != hipSuccess) {                                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__);               \
            exit(1);                                                                                   \
        }                                                                                              \
    } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__global__ void gen_normal(double *out, uint64_t n, uint64_t seed, double mu, double sigma) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t a = mix(i * 2 + seed), b = mix(i * 2 + 1 + seed * 7919);
        double u1 = ((a >> 11) + 1) * (1.0 / 9007199254740993.0), u2 = (b >> 11) * (1.0 / 9007199254740992.0);
        out[i] = mu + sigma * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
}

template <int BINS>
__device__ __forceinline__ uint32_t sub_index(double v) {
    double s = (v + 4.0) * 0.125;
    uint32_t idx = 0;
    if (s != s) {
    } else if (s < 0) idx = 1;
    else if (s >= 1) idx = BINS + 2;
    else idx = (uint32_t)((int)(s * (double)BINS) + 2);
    return idx;
}
template <int BINS>
__device__ __forceinline__ uint32_t cell_of(double x, double y) { return sub_index<BINS>(x) + (BINS + 3) * sub_index<BINS>(y); }

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// ---- slabs, interleaved: cell c belongs to slab c & (S-1), local index c >> log2S -------------------
// MODE 1 count u32, 2 count u32 + sum f64, 3 sum f64 only
// block b: xcd = b & 7 (observed dispatch order), local = b >> 3, slab = local & (S-1), group = (local >> LOG2S) * 8 + xcd
template <int MODE, int THROTTLE>
__global__ void __launch_bounds__(1024) k_slab(const double *x, const double *y, const double *v, uint64_t n, unsigned long long *cnt, double *sum, int log2S, int ngroups, int U,
                                               unsigned int *progress, int D, unsigned int *xcc_census) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CELLS = 259 * 259;
    const int S = 1 << log2S;
    const int slab_cells = (CELLS + S - 1) >> log2S;
    unsigned int *hc = (unsigned int *)smem;
    double *hs = (double *)(smem + (MODE == 3 ? 0 : (((size_t)slab_cells * 4 + 15) & ~(size_t)15)));
    for (int c = threadIdx.x; c < slab_cells; c += blockDim.x) {
        if (MODE != 3) hc[c] = 0;
        if (MODE != 1) hs[c] = 0.0;
    }
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const uint32_t slab = local & (S - 1);
    const int group = (local >> log2S) * 8 + xcd;
    if (threadIdx.x == 0 && xcc_census) atomicAdd(&xcc_census[group * 8 + xcc_id()], 1u); // where did the group's blocks land?
    __syncthreads();
    const uint64_t tile = (uint64_t)blockDim.x * U;
    uint64_t base = (uint64_t)group * tile;
    const uint64_t stride = (uint64_t)ngroups * tile;
    unsigned int t = 0;
    unsigned int *gp = progress + (size_t)group * 32;
    for (; base < n; base += stride, ++t) {
        if (THROTTLE && (t & 3) == 0 && t >= (unsigned)D) {
            // wait until every block of the group has finished tile t-D
            if (threadIdx.x < (unsigned)S) {
                int spins = 0; // bounded: a wrong placement assumption must never hang the GPU
                while (__hip_atomic_load(&gp[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + D < t && ++spins < 2000000) __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
        }
#pragma unroll 4
        for (int u = 0; u < U; ++u) {
            uint64_t i = base + (uint64_t)u * blockDim.x + threadIdx.x;
            if (i < n) {
                uint32_t c = cell_of<256>(x[i], y[i]);
                if ((c & (S - 1)) == slab) {
                    c >>= log2S;
                    if (MODE != 3) atomicAdd(&hc[c], 1u);
                    if (MODE != 1) { double w = v[i]; if (w == w) unsafeAtomicAdd(&hs[c], w); }
                }
            }
        }
        if (THROTTLE && (t & 3) == 3) {
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(&gp[slab], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (THROTTLE && threadIdx.x == 0) __hip_atomic_store(&gp[slab], 0x7fffff00u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    for (int c = threadIdx.x; c < slab_cells; c += blockDim.x) {
        uint32_t gc = ((uint32_t)c << log2S) + slab;
        if (gc < CELLS) {
            if (MODE != 3 && hc[c]) atomicAdd(&cnt[gc], (unsigned long long)hc[c]);
            if (MODE != 1 && hs[c] != 0.0) unsafeAtomicAdd(&sum[gc], hs[c]);
        }
    }
}

// ---- raw LDS atomic rates on a grid that fits: 128^2 (131^2 cells) ------------------------------------
// MODE 1 count u32; 3 sum f64; 4 count u64
template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(const double *x, const double *y, const double *v, uint64_t n, unsigned long long *cnt, double *sum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CELLS = 131 * 131;
    unsigned int *hc = (unsigned int *)smem;
    unsigned long long *hl = (unsigned long long *)smem;
    double *hs = (double *)smem;
    for (int c = threadIdx.x; c < CELLS; c += blockDim.x) {
        if (MODE == 1) hc[c] = 0;
        else hl[c] = 0;
    }
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t c = cell_of<128>(x[i], y[i]);
        if (MODE == 1) atomicAdd(&hc[c], 1u);
        else if (MODE == 4) atomicAdd(&hl[c], 1ull);
        else { double w = v[i]; if (w == w) unsafeAtomicAdd(&hs[c], w); }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < CELLS; c += blockDim.x) {
        if (MODE == 1) { if (hc[c]) atomicAdd(&cnt[c], (unsigned long long)hc[c]); }
        else if (MODE == 4) { if (hl[c]) atomicAdd(&cnt[c], hl[c]); }
        else if (hs[c] != 0.0) unsafeAtomicAdd(&sum[c], hs[c]);
    }
}

template <typename F>
float time_ms(F launch, int reps = 4) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CK(hipGetLastError());
    return best;
}

int main(int argc, char **argv) {
    int lg = argc > 1 ? atoi(argv[1]) : 28;
    const uint64_t n = 1ull << lg;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs=%d rows=2^%d\n", p.name, p.multiProcessorCount, lg);
    double *x, *y, *v;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&v, n * 8));
    gen_normal<<<4096, 256>>>(x, n, 1, 0, 1); gen_normal<<<4096, 256>>>(y, n, 2, 0, 1); gen_normal<<<4096, 256>>>(v, n, 3, 3, 2);
    CK(hipDeviceSynchronize());
    constexpr int CELLS = 259 * 259;
    unsigned long long *cnt; double *sum; unsigned int *progress, *census;
    CK(hipMalloc(&cnt, (size_t)CELLS * 8)); CK(hipMalloc(&sum, (size_t)CELLS * 8));
    CK(hipMalloc(&progress, 4096 * 32 * 4)); CK(hipMalloc(&census, 4096 * 8 * 4));
    CK(hipMemset(cnt, 0, (size_t)CELLS * 8)); CK(hipMemset(sum, 0, (size_t)CELLS * 8));
    auto report = [&](const char *name, float ms, int bpr) {
        printf("%-64s %8.3f ms %8.2f Grows/s %7.1f GB/s\n", name, ms, n / ms / 1e6, n * (double)bpr / ms / 1e6);
        fflush(stdout);
    };
#define SETLDS(K) CK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    SETLDS((k_lds<1>)); SETLDS((k_lds<3>)); SETLDS((k_lds<4>));
    for (int blocks : {256, 512}) {
        for (int bs : {512, 1024}) {
            char nm[128];
            snprintf(nm, sizeof nm, "LDS 128^2 count u32   blocks=%d x%d", blocks, bs);
            report(nm, time_ms([&] { k_lds<1><<<blocks, bs, 131 * 131 * 4>>>(x, y, v, n, cnt, sum); }), 16);
        }
    }
    report("LDS 128^2 count u64   blocks=256 x1024", time_ms([&] { k_lds<4><<<256, 1024, 131 * 131 * 8>>>(x, y, v, n, cnt, sum); }), 16);
    report("LDS 128^2 sum f64     blocks=256 x1024", time_ms([&] { k_lds<3><<<256, 1024, 131 * 131 * 8>>>(x, y, v, n, cnt, sum); }), 24);
    report("LDS 128^2 sum f64     blocks=256 x512", time_ms([&] { k_lds<3><<<256, 512, 131 * 131 * 8>>>(x, y, v, n, cnt, sum); }), 24);

    SETLDS((k_slab<1, 0>)); SETLDS((k_slab<1, 1>)); SETLDS((k_slab<2, 0>)); SETLDS((k_slab<2, 1>)); SETLDS((k_slab<3, 0>)); SETLDS((k_slab<3, 1>));
    auto run_slab = [&](int mode, int throttle, int log2S, int bs, int U, int D, int blocks_per_cu) {
        const int S = 1 << log2S;
        const int slab_cells = (CELLS + S - 1) >> log2S;
        size_t lds = 16 + (mode == 1 ? (size_t)slab_cells * 4 : mode == 3 ? (size_t)slab_cells * 8 : ((((size_t)slab_cells * 4 + 15) & ~(size_t)15) + (size_t)slab_cells * 8));
        if (lds * blocks_per_cu > 160 * 1024) return;
        const int blocks = 256 * blocks_per_cu;
        const int ngroups = blocks / S;
        CK(hipMemset(progress, 0, 4096 * 32 * 4));
        CK(hipMemset(census, 0, 4096 * 8 * 4));
        auto go = [&] {
            if (throttle) CK(hipMemsetAsync(progress, 0, 4096 * 32 * 4));
#define L(M, T) k_slab<M, T><<<blocks, bs, lds>>>(x, y, v, n, cnt, sum, log2S, ngroups, U, progress, D, census)
            if (mode == 1) { if (throttle) L(1, 1); else L(1, 0); }
            else if (mode == 2) { if (throttle) L(2, 1); else L(2, 0); }
            else { if (throttle) L(3, 1); else L(3, 0); }
#undef L
        };
        float ms = time_ms(go);
        char nm[160];
        snprintf(nm, sizeof nm, "slab %s S=%d bs=%d U=%d blk/CU=%d %s lds=%zuK", mode == 1 ? "count" : mode == 2 ? "count+sum" : "sum", S, bs, U, blocks_per_cu, throttle ? (D == 8 ? "thr D=8" : "thr D=16") : "free", lds >> 10);
        report(nm, ms, mode == 1 ? 16 : 24);
    };
    for (int throttle : {0, 1}) {
        for (int U : {4, 16}) {
            run_slab(1, throttle, 1, 1024, U, 8, 1); // count S=2
            run_slab(1, throttle, 2, 1024, U, 8, 1); // count S=4
            run_slab(1, throttle, 2, 512, U, 8, 2);
            run_slab(3, throttle, 2, 1024, U, 8, 1); // sum S=4
            run_slab(2, throttle, 3, 1024, U, 8, 1); // count+sum S=8
            run_slab(2, throttle, 3, 512, U, 8, 1);
            run_slab(2, throttle, 4, 512, U, 8, 2);  // count+sum S=16, 2 blocks/CU
        }
    }
    run_slab(2, 1, 3, 1024, 4, 16, 1);
    // placement census of the last run: for each group, how many of its blocks were on each XCD
    {
        unsigned int h[64 * 8];
        CK(hipMemcpy(h, census, sizeof h, hipMemcpyDeviceToHost));
        int pure = 0, total = 0;
        for (int g = 0; g < 32; g++) {
            int nz = 0;
            for (int k = 0; k < 8; k++) nz += h[g * 8 + k] != 0;
            if (nz) { total++; pure += nz == 1; }
        }
        printf("placement census: %d of %d groups had all their blocks on ONE XCD\n", pure, total);
    }
    return 0;
}

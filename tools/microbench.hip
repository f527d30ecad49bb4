// Design microbenchmarks for the scatter-add of the 2-D 256x256 (259x259-cell) binned count+mean
// pass on MI355X: how fast are HBM-side atomics vs LDS-private grids vs LDS slabs with L2-shared
// input?  Standalone: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -munsafe-fp-atomics
// tools/microbench.hip -o tools/microbench && tools/microbench [log2_rows]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

constexpr int BINS = 256, SH = BINS + 3, CELLS = SH * SH;

__device__ __forceinline__ uint32_t sub_index(double v, double vmin, double scale) {
    double s = (v - vmin) * scale;
    uint32_t idx = 0;
    if (s != s) {
    } else if (s < 0) idx = 1;
    else if (s >= 1) idx = BINS + 2;
    else idx = (uint32_t)((int)(s * (double)BINS) + 2);
    return idx;
}
__device__ __forceinline__ uint32_t cell_of(double x, double y) { return sub_index(x, -4.0, 0.125) + SH * sub_index(y, -4.0, 0.125); }

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

// ---- A: read-only ceiling ---------------------------------------------------------------------
template <int NCOL>
__global__ void __launch_bounds__(256) k_read(const double *x, const double *y, const double *v, uint64_t n, unsigned long long *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n; i += stride) {
        uint32_t c = cell_of(x[i], y[i]);
        if (NCOL == 3) acc += (unsigned long long)__double_as_longlong(v[i]);
        acc += c;
    }
    if (acc == 0x1234567) sink[0] = acc;
}
// 2 rows per lane (16-byte loads)
template <int NCOL>
__global__ void __launch_bounds__(256) k_read2(const double2 *x, const double2 *y, const double2 *v, uint64_t n2, unsigned long long *sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (; i < n2; i += stride) {
        double2 a = x[i], b = y[i];
        acc += cell_of(a.x, b.x) + cell_of(a.y, b.y);
        if (NCOL == 3) { double2 w = v[i]; acc += (unsigned long long)__double_as_longlong(w.x) ^ (unsigned long long)__double_as_longlong(w.y); }
    }
    if (acc == 0x1234567) sink[0] = acc;
}

// ---- B/C/D: global atomics -----------------------------------------------------------------------
// MODE 1: count u64. 2: count + sum f64. 3: count + sum + countv. 4: count u32 only. 5: sum f64 only
template <int MODE, bool XCC>
__global__ void __launch_bounds__(256) k_glob(const double *x, const double *y, const double *v, uint64_t n, unsigned long long *cnt, double *sum, unsigned long long *cntv, int R) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rep = XCC ? (xcc_id() * (R / 8) + (blockIdx.x >> 3) % (R / 8)) : (blockIdx.x % R);
    cnt += rep * CELLS; sum += rep * CELLS; cntv += rep * CELLS;
    unsigned int *cnt32 = (unsigned int *)cnt;
    for (; i < n; i += stride) {
        uint32_t c = cell_of(x[i], y[i]);
        if (MODE == 4) { atomicAdd(&cnt32[c], 1u); continue; }
        if (MODE != 5) atomicAdd(&cnt[c], 1ull);
        if (MODE == 2 || MODE == 3 || MODE == 5) {
            double w = v[i];
            if (w == w) {
                unsafeAtomicAdd(&sum[c], w);
                if (MODE == 3) atomicAdd(&cntv[c], 1ull);
            }
        }
    }
}
// AoS cell {count,sum}: both atomics of a row land in one 16-byte slot
struct Cell2 { unsigned long long c; double s; };
__global__ void __launch_bounds__(256) k_glob_aos(const double *x, const double *y, const double *v, uint64_t n, Cell2 *g, int R) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    g += (uint64_t)(blockIdx.x % R) * CELLS;
    for (; i < n; i += stride) {
        uint32_t c = cell_of(x[i], y[i]);
        double w = v[i];
        atomicAdd(&g[c].c, 1ull);
        if (w == w) unsafeAtomicAdd(&g[c].s, w);
    }
}

// ---- F: LDS-private count (131x131 u32) ----------------------------------------------------------
constexpr int SH1 = 131, CELLS1 = SH1 * SH1;
__device__ __forceinline__ uint32_t sub_index128(double v) {
    double s = (v + 4.0) * 0.125;
    uint32_t idx = 0;
    if (s != s) {
    } else if (s < 0) idx = 1;
    else if (s >= 1) idx = 130;
    else idx = (uint32_t)((int)(s * 128.0) + 2);
    return idx;
}
__global__ void __launch_bounds__(1024) k_lds128(const double *x, const double *y, uint64_t n, unsigned long long *cnt) {
    extern __shared__ unsigned int h[];
    for (int c = threadIdx.x; c < CELLS1; c += blockDim.x) h[c] = 0;
    __syncthreads();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) atomicAdd(&h[sub_index128(x[i]) + SH1 * sub_index128(y[i])], 1u);
    __syncthreads();
    for (int c = threadIdx.x; c < CELLS1; c += blockDim.x)
        if (h[c]) atomicAdd(&cnt[c], (unsigned long long)h[c]);
}

// ---- G: LDS slabs over the 259x259 grid; S slabs; slab owners read the same rows (L2-shared) ------
// block b: XCD = b%8 (hardware round robin), local = b/8; slab = local % S; group = local / S.
// all S blocks of a group walk the same row tiles.  MODE 1: count u32; 2: count u32 + sum f64
template <int MODE>
__global__ void __launch_bounds__(1024) k_slab(const double *x, const double *y, const double *v, uint64_t n, unsigned long long *cnt, double *sum, int S, int groups_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int slab_cells = (CELLS + S - 1) / S;
    unsigned int *hc = (unsigned int *)smem;
    double *hs = (double *)(smem + (((size_t)slab_cells * 4 + 15) & ~(size_t)15));
    for (int c = threadIdx.x; c < slab_cells; c += blockDim.x) { hc[c] = 0; if (MODE == 2) hs[c] = 0.0; }
    __syncthreads();
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int slab = local % S, group = xcd * groups_per_xcd + local / S;
    const int ngroups = 8 * groups_per_xcd;
    const uint32_t lo = slab * slab_cells;
    uint64_t i = (uint64_t)group * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)ngroups * blockDim.x;
    for (; i < n; i += stride) {
        uint32_t c = cell_of(x[i], y[i]) - lo;
        if (c < (uint32_t)slab_cells) {
            atomicAdd(&hc[c], 1u);
            if (MODE == 2) { double w = v[i]; if (w == w) unsafeAtomicAdd(&hs[c], w); }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < slab_cells; c += blockDim.x) {
        if (lo + c < CELLS && hc[c]) {
            atomicAdd(&cnt[lo + c], (unsigned long long)hc[c]);
            if (MODE == 2) unsafeAtomicAdd(&sum[lo + c], hs[c]);
        }
    }
}

template <typename F>
float time_ms(F launch, int reps = 5) {
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
    printf("device %s CUs=%d  rows=2^%d\n", p.name, p.multiProcessorCount, lg);
    double *x, *y, *v;
    CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&v, n * 8));
    gen_normal<<<4096, 256>>>(x, n, 1, 0, 1); gen_normal<<<4096, 256>>>(y, n, 2, 0, 1); gen_normal<<<4096, 256>>>(v, n, 3, 3, 2);
    CK(hipDeviceSynchronize());
    const int RMAX = 64;
    unsigned long long *cnt, *cntv, *sink; double *sum; Cell2 *aos;
    CK(hipMalloc(&cnt, (size_t)RMAX * CELLS * 8)); CK(hipMalloc(&cntv, (size_t)RMAX * CELLS * 8)); CK(hipMalloc(&sum, (size_t)RMAX * CELLS * 8));
    CK(hipMalloc(&aos, (size_t)RMAX * CELLS * 16)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(cnt, 0, (size_t)RMAX * CELLS * 8)); CK(hipMemset(cntv, 0, (size_t)RMAX * CELLS * 8)); CK(hipMemset(sum, 0, (size_t)RMAX * CELLS * 8)); CK(hipMemset(aos, 0, (size_t)RMAX * CELLS * 16));

    auto report = [&](const char *name, float ms, int bytes_per_row) {
        printf("%-44s %8.3f ms  %8.2f Grows/s  %7.1f GB/s (%d B/row)\n", name, ms, n / ms / 1e6, n * (double)bytes_per_row / ms / 1e6, bytes_per_row);
        fflush(stdout);
    };
    for (int blocks : {2048, 8192}) {
        char nm[96];
        snprintf(nm, sizeof nm, "read xy      8B loads  blocks=%d", blocks);
        report(nm, time_ms([&] { k_read<2><<<blocks, 256>>>(x, y, v, n, sink); }), 16);
        snprintf(nm, sizeof nm, "read xyv     8B loads  blocks=%d", blocks);
        report(nm, time_ms([&] { k_read<3><<<blocks, 256>>>(x, y, v, n, sink); }), 24);
        snprintf(nm, sizeof nm, "read xyv    16B loads  blocks=%d", blocks);
        report(nm, time_ms([&] { k_read2<3><<<blocks, 256>>>((double2 *)x, (double2 *)y, (double2 *)v, n / 2, sink); }), 24);
    }
    for (int R : {1, 8, 64}) {
        char nm[96];
        snprintf(nm, sizeof nm, "global count u64            R=%d", R);
        report(nm, time_ms([&] { k_glob<1, false><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 16);
        snprintf(nm, sizeof nm, "global count u32            R=%d", R);
        report(nm, time_ms([&] { k_glob<4, false><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 16);
        snprintf(nm, sizeof nm, "global sum f64 only         R=%d", R);
        report(nm, time_ms([&] { k_glob<5, false><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 24);
        snprintf(nm, sizeof nm, "global count+sum            R=%d", R);
        report(nm, time_ms([&] { k_glob<2, false><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 24);
        snprintf(nm, sizeof nm, "global count+sum+countv     R=%d", R);
        report(nm, time_ms([&] { k_glob<3, false><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 24);
        snprintf(nm, sizeof nm, "global AoS {count,sum}      R=%d", R);
        report(nm, time_ms([&] { k_glob_aos<<<2048, 256>>>(x, y, v, n, aos, R); }), 24);
    }
    for (int R : {8, 64}) {
        char nm[96];
        snprintf(nm, sizeof nm, "xcc-local count u64         R=%d", R);
        report(nm, time_ms([&] { k_glob<1, true><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 16);
        snprintf(nm, sizeof nm, "xcc-local count+sum+countv  R=%d", R);
        report(nm, time_ms([&] { k_glob<3, true><<<2048, 256>>>(x, y, v, n, cnt, sum, cntv, R); }), 24);
    }
    for (int blocks : {256, 512}) {
        char nm[96];
        snprintf(nm, sizeof nm, "LDS count 128^2 (u32)  blocks=%d x1024", blocks);
        report(nm, time_ms([&] { k_lds128<<<blocks, 1024, CELLS1 * 4>>>(x, y, n, cnt); }), 16);
    }
    CK(hipFuncSetAttribute((const void *)k_slab<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void *)k_slab<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int S : {2, 4}) {
        const int slab_cells = (CELLS + S - 1) / S;
        size_t lds = (size_t)slab_cells * 4 + 16;
        if (lds > 160 * 1024) continue;
        int per_xcd = 32 / S; // one block per CU
        char nm[96];
        snprintf(nm, sizeof nm, "LDS slabs count 256^2  S=%d (lds %zu KB)", S, lds >> 10);
        report(nm, time_ms([&] { k_slab<1><<<8 * per_xcd * S, 1024, lds>>>(x, y, v, n, cnt, sum, S, per_xcd); }), 16);
    }
    for (int S : {6, 8}) {
        const int slab_cells = (CELLS + S - 1) / S;
        size_t lds = (((size_t)slab_cells * 4 + 15) & ~(size_t)15) + (size_t)slab_cells * 8 + 16;
        if (lds > 160 * 1024) continue;
        int per_xcd = 32 / S;
        char nm[96];
        snprintf(nm, sizeof nm, "LDS slabs count+sum 256^2  S=%d (lds %zu KB)", S, lds >> 10);
        report(nm, time_ms([&] { k_slab<2><<<8 * per_xcd * S, 1024, lds>>>(x, y, v, n, cnt, sum, S, per_xcd); }), 24);
    }
    return 0;
}

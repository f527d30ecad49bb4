// Microbench 4 (round 3): does the 256 MiB Infinity Cache absorb partition-queue traffic, and what does a grid barrier cost?
//   A  write W bytes (16-byte coalesced stores), then read them back in a second kernel: read rate vs W
//   B  read W bytes twice: rate of the second read vs W
//   C  the partition pattern as separate launches per chunk: K1 reads 16 B/row of a big input and writes 10 B/row into a
//      queue (each workgroup its own contiguous slice), K2 reads the queue slice of ANOTHER workgroup; queue buffer reused
//      by every chunk ("reuse") or a fresh region per chunk ("fresh" = HBM traffic); chunk = 2^20 .. 2^26 rows
//   D  the same as ONE persistent kernel (256 workgroups x 1024) with a hand-rolled grid barrier per chunk (release /
//      acquire at agent scope), double-buffered queue; and the barrier alone
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench4 tools/microbench4.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef unsigned long long u64;

__global__ void __launch_bounds__(1024) k_write(uint4 *p, size_t n16, uint32_t tag) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(tag, (uint32_t)i, tag, (uint32_t)i);
}
__global__ void __launch_bounds__(1024) k_read(const uint4 *p, size_t n16, u64 *sink) {
    u64 acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x + v.w; }
    if (acc == 0x123456789ull) sink[0] = acc;
}

// C: one chunk.  rows are split evenly over the workgroups; workgroup w writes queue slice w, reads slice (w + shift) % wgs
__global__ void __launch_bounds__(1024) k_scatter(const double *x, const double *v, double *qv, uint16_t *qi, size_t row0, size_t rows) {
    const size_t per = rows / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) {
        const double a = x[row0 + lo + i], b = v[row0 + lo + i];
        qv[lo + i] = b;
        qi[lo + i] = (uint16_t)(int)(a * 1000.0);
    }
}
__global__ void __launch_bounds__(1024) k_reduce(const double *qv, const uint16_t *qi, size_t rows, u64 *sink, int shift) {
    const size_t per = rows / gridDim.x;
    const size_t lo = (size_t)((blockIdx.x + shift) % gridDim.x) * per;
    double acc = 0;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) acc += qv[lo + i] * (double)qi[lo + i];
    if (acc == 1.2345e300) sink[0] = 1;
}

// D: persistent.  bar[0] = arrivals (monotonic)
__device__ __forceinline__ void grid_barrier(unsigned int *bar, unsigned int &epoch, unsigned int wgs) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // writes of this workgroup's XCD L2 become visible to the other XCDs (buffer_wbl2)
        epoch += wgs;
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
template <int MODE> // 0: scatter + reduce; 1: barrier only; 2: scatter + reduce without barrier (wrong, for the cost of the barrier); 3: input read only
__global__ void __launch_bounds__(1024) k_persist(const double *x, const double *v, double *qv, uint16_t *qi, size_t total_rows, size_t chunk, size_t qstride, int fresh, unsigned int *bar, u64 *sink, int shift) {
    unsigned int epoch = 0;
    const unsigned int wgs = gridDim.x;
    const size_t per = chunk / wgs;
    double acc = 0;
    const size_t chunks = total_rows / chunk;
    for (size_t c = 0; c <= chunks; ++c) {
        if (MODE != 1 && c < chunks) { // scatter chunk c into buffer c % 2 (or a fresh region)
            const size_t q0 = fresh ? c * qstride : (c & 1) * qstride;
            const size_t lo = (size_t)blockIdx.x * per;
            for (size_t i = threadIdx.x; i < per; i += blockDim.x) {
                const double a = x[c * chunk + lo + i], b = v[c * chunk + lo + i];
                if (MODE == 3) { acc += a + b; continue; }
                qv[q0 + lo + i] = b;
                qi[q0 + lo + i] = (uint16_t)(int)(a * 1000.0);
            }
        }
        if (MODE != 1 && MODE != 3 && c > 0) { // reduce chunk c - 1
            const size_t q0 = fresh ? (c - 1) * qstride : ((c - 1) & 1) * qstride;
            const size_t lo = (size_t)((blockIdx.x + shift) % wgs) * per;
            for (size_t i = threadIdx.x; i < per; i += blockDim.x) acc += qv[q0 + lo + i] * (double)qi[q0 + lo + i];
        }
        if (MODE == 0 || MODE == 1) grid_barrier(bar, epoch, wgs);
    }
    if (acc == 1.2345e300) sink[0] = 1;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
    const size_t ROWS = 1ull << 28; // 2^28 rows x 16 B = 4 GiB of input
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    u64 *sink; CK(hipMalloc(&sink, 64));
    char *big; CK(hipMalloc(&big, 8ull << 30));
    CK(hipMemset(big, 1, 8ull << 30));
    printf("# A/B: W MiB   write GB/s   read-after-write GB/s   second read GB/s\n");
    for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
        const size_t n16 = (mb << 20) / 16;
        float w = 1e9, r = 1e9, r2 = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            // evict: stream 1 GiB elsewhere
            k_read<<<2048, 1024>>>((const uint4 *)(big + (7ull << 30)), (1ull << 30) / 16, sink);
            CK(hipEventRecord(e0)); k_write<<<2048, 1024>>>((uint4 *)big, n16, rep); CK(hipEventRecord(e1));
            k_read<<<2048, 1024>>>((const uint4 *)big, n16, sink); CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
            w = fminf(w, time_ms(e0, e1)); r = fminf(r, time_ms(e1, e2));
            k_read<<<2048, 1024>>>((const uint4 *)(big + (7ull << 30)), (1ull << 30) / 16, sink);
            k_read<<<2048, 1024>>>((const uint4 *)big, n16, sink);
            CK(hipEventRecord(e0)); k_read<<<2048, 1024>>>((const uint4 *)big, n16, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            r2 = fminf(r2, time_ms(e0, e1));
        }
        const double gb = (double)(mb << 20) / 1e9;
        printf("A %5zu MiB  write %7.0f  read-after-write %7.0f  second read %7.0f   (us: %.1f %.1f %.1f)\n", mb, gb / w * 1e3, gb / r * 1e3, gb / r2 * 1e3, w * 1e3, r * 1e3, r2 * 1e3);
        fflush(stdout);
    }
    // C / D
    double *x = (double *)big, *v = (double *)(big + ROWS * 8);
    double *qv = (double *)(big + ROWS * 16);               // 3 GiB of queue space behind the input
    const size_t QROWS = (3ull << 30) / 10;
    uint16_t *qi = (uint16_t *)(big + ROWS * 16 + QROWS * 8);
    unsigned int *bar; CK(hipMalloc(&bar, 64));
    printf("# C: separate launches per chunk, 2^28 rows.  GB/s = 16 B/row of input / time\n");
    for (int lg = 20; lg <= 26; ++lg) {
        const size_t chunk = 1ull << lg;
        for (int fresh = 0; fresh < 2; ++fresh) {
            if (fresh && (ROWS / chunk) * chunk > QROWS) { /* fresh regions wrap inside 2 GiB: still > 256 MiB apart in time */ }
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (size_t c = 0; c < ROWS / chunk; ++c) {
                    const size_t q0 = fresh ? (c * chunk) % (QROWS - chunk) : 0;
                    k_scatter<<<256, 1024>>>(x, v, qv + q0, qi + q0, c * chunk, chunk);
                    k_reduce<<<256, 1024>>>(qv + q0, qi + q0, chunk, sink, 37);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                best = fminf(best, time_ms(e0, e1));
            }
            printf("C chunk 2^%d rows  %s  %8.3f ms  %7.0f GB/s of input   (%zu launches x2)\n", lg, fresh ? "fresh" : "reuse", best, ROWS * 16.0 / best / 1e6, ROWS / chunk);
            fflush(stdout);
        }
    }
    printf("# D: one persistent kernel (256 x 1024), grid barrier per chunk\n");
    for (int lg = 18; lg <= 26; lg += 1) {
        const size_t chunk = 1ull << lg;
        for (int mode : {0, 1, 2, 3}) {
            for (int fresh = 0; fresh < 2; ++fresh) {
                if (mode != 0 && fresh) continue;
                const size_t qstride = fresh ? chunk : chunk; // fresh: region c; reuse: two regions
                if (fresh && (ROWS / chunk) * chunk > QROWS) continue;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipMemsetAsync(bar, 0, 64));
                    CK(hipEventRecord(e0));
                    if (mode == 0) k_persist<0><<<256, 1024>>>(x, v, qv, qi, ROWS, chunk, qstride, fresh, bar, sink, 37);
                    if (mode == 1) k_persist<1><<<256, 1024>>>(x, v, qv, qi, ROWS, chunk, qstride, fresh, bar, sink, 37);
                    if (mode == 2) k_persist<2><<<256, 1024>>>(x, v, qv, qi, ROWS, chunk, qstride, fresh, bar, sink, 37);
                    if (mode == 3) k_persist<3><<<256, 1024>>>(x, v, qv, qi, ROWS, chunk, qstride, fresh, bar, sink, 37);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    best = fminf(best, time_ms(e0, e1));
                }
                const char *names[] = {"scatter+reduce+barrier", "barrier only", "scatter+reduce, no barrier", "input read only"};
                printf("D chunk 2^%d rows  %-28s %s  %8.3f ms  %7.0f GB/s of input  (%zu barriers: %.2f us each if barrier only)\n", lg, names[mode], fresh ? "fresh" : "reuse", best, ROWS * 16.0 / best / 1e6,
                       ROWS / chunk + 1, mode == 1 ? best * 1e3 / (ROWS / chunk + 1) : 0.0);
                fflush(stdout);
            }
        }
    }
    return 0;
}

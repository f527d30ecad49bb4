// Microbench 9 (round 5): microbench8's shared-stream hash partition with its READS and WRITES gated by the chip-wide wall clock
// (s_memrealtime, 100 MHz): a tile's rows are requested only while (ticks % PN) < RN, its records copied out only in the rest of the
// period — every workgroup of the chip reads in the same window and writes in the same window.  Question: the mixed 16 B read + 16 B
// written per row runs at 4.8-5.1 TB/s free-running; does alternating pure-read and pure-write phases chip-wide run faster?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench9 tools/microbench9.hip
// (below: microbench 8) the hash partition of microbench5 (MODE 2: 16-byte AoS records, 1024 threads x 8 rows, S streams) with the record
// streams SHARED between workgroups: NG groups of workgroups (group = blockIdx % NG — blocks are dealt to the 8 XCDs round robin, so NG = 8
// is "one set of S streams per XCD", NG = 1 one set for the chip, NG = wgs the private streams of microbench5), every tile's segment of a
// stream reserved with ONE returning device atomic per (workgroup, stream, tile).  Question: gb_scatter's stores cost ~3.7 ms per 12 GB as
// 131072 private streams advancing 192 bytes a tile; do 4096 streams advancing by neighbouring segments of 32 workgroups cost less?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench8 tools/microbench8.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31; return x; }

struct Args {
    const long long *keys; const u64 *vals; u64 n;
    int s_log2, ng;    // streams per set, sets of streams
    u64 cap;           // records per (set, stream) region
    uint4 *q16; uint32_t *q12;
    u64 *gcount;       // [ng][S] records reserved so far
    int abl;           // 1: no stores; 8: 12-byte records
    int tick, pn, rn;  // gating: tick = realtime >> tick; reads while tick % pn < rn, writes otherwise (pn == 0: free-running)
    int gate;          // 1: both gated, 2: the copy-out only, 3: the requests only
};
__device__ __forceinline__ void wait_phase(const Args &A, bool want_read) {
    for (;;) {
        const uint32_t t = (uint32_t)(__builtin_amdgcn_s_memrealtime() >> A.tick) % (uint32_t)A.pn;
        if ((t < (uint32_t)A.rn) == want_read) return;
        __builtin_amdgcn_s_sleep(2);
    }
}

template <int R>
__global__ void __launch_bounds__(1024) scatter(const Args A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t THREADS = 1024, T = THREADS * R, NW = 16;
    const uint32_t S = 1u << A.s_log2;
    uint32_t *const cnt = (uint32_t *)lds;         // [S]
    uint32_t *const off = cnt + S;                 // [S]
    u64 *const gbase = (u64 *)(off + S);           // [S] where this tile's segment of stream s starts (record index inside the set's region)
    uint32_t *const s_wave = (uint32_t *)(gbase + S); // [16]
    u64 *const st_a = (u64 *)(s_wave + 16);        // [T] key
    u64 *const st_b = st_a + T;                    // [T] value
    uint16_t *const st_s = (uint16_t *)(st_b + T); // [T] stream
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t group = blockIdx.x % (uint32_t)A.ng;
    for (uint32_t s = tid; s < S; s += THREADS) cnt[s] = 0;
    __syncthreads();
    long long key[R], key_n[R];
    u64 val[R], val_n[R];
    const u64 n = A.n;
    auto request = [&](u64 tile, long long (&k)[R], u64 (&v)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u64 i = tile * T + (u64)r * THREADS + tid;
            if (i >= n) i = n - 1;
            k[r] = A.keys[i];
            v[r] = A.vals[i];
        }
    };
    if ((u64)blockIdx.x * T < n) { request(blockIdx.x, key_n, val_n);
#pragma unroll
        for (int r = 0; r < R; ++r) { key[r] = key_n[r]; val[r] = val_n[r]; } }
    for (u64 tile = blockIdx.x; tile * T < n; tile += gridDim.x) {
        const u64 next = tile + gridDim.x;
        const bool has_next = next * T < n;
        if (A.pn && (A.gate & 1) && has_next) wait_phase(A, true);
        if (has_next) request(next, key_n, val_n);
        uint32_t b[R], pos[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            b[r] = (uint32_t)(mix((u64)key[r]) >> (64 - A.s_log2));
            const bool ok = tile * T + (u64)r * THREADS + tid < n;
            pos[r] = ok ? __hip_atomic_fetch_add(&cnt[b[r]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0xffffffffu;
        }
        __syncthreads();
        uint32_t c = 0;
        if (tid < S) { // (S <= 1024: thread s owns stream s)
            c = cnt[tid];
            cnt[tid] = 0;
            // this tile's segment of the stream: one returning device atomic (looked at two barriers later)
            gbase[tid] = c ? atomicAdd(&A.gcount[(u64)group * S + tid], (u64)c) : 0ull;
        }
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64); if ((int)lane >= o) inc += t; }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w2 = 0; w2 < NW; ++w2) { const uint32_t x = s_wave[w2]; if (w2 < wave) before += x; total += x; }
        if (tid < S) off[tid] = before + inc - c;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (pos[r] == 0xffffffffu) continue;
            const uint32_t j = off[b[r]] + pos[r];
            st_a[j] = (u64)key[r]; st_b[j] = val[r]; st_s[j] = (uint16_t)b[r];
        }
        __syncthreads();
        if (A.pn && (A.gate == 1 || A.gate == 2)) wait_phase(A, false);
        for (uint32_t j = tid; j < total; j += THREADS) {
            const uint32_t s = st_s[j];
            const u64 at = gbase[s] + (j - off[s]);
            if (at >= A.cap || (A.abl & 1)) continue;
            const u64 dst = ((u64)group * S + s) * A.cap + at;
            const u64 a = st_a[j], c2 = st_b[j];
            if (A.abl & 8) { uint32_t *p = A.q12 + dst * 3; typedef unsigned int u3 __attribute__((ext_vector_type(3))); typedef u3 u3a __attribute__((aligned(4))); *(u3a *)p = u3a{(uint32_t)a, (uint32_t)c2, (uint32_t)(c2 >> 32)}; }
            else A.q16[dst] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c2, (uint32_t)(c2 >> 32));
        }
        if (has_next) {
#pragma unroll
            for (int r = 0; r < R; ++r) { key[r] = key_n[r]; val[r] = val_n[r]; }
        }
        __syncthreads();
    }
}

__global__ void gen(long long *k, u64 *v, u64 n, u64 card) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 h = mix(i * 0x9e3779b97f4a7c15ULL + 12345);
        k[i] = (long long)(((h % card) * 2654435761ull) % (1ull << 40));
        v[i] = (u64)__double_as_longlong((double)(h >> 40) * 1e-3);
    }
}

void run(const char *name, Args A, int ng, int abl, int tick = 0, int pn = 0, int rn = 0, int gate = 1) {
    A.tick = tick; A.pn = pn; A.rn = rn; A.gate = gate;
    const int wgs = 256;
    const uint32_t S = 1u << A.s_log2;
    constexpr int R = 8;
    const size_t T = 1024 * R;
    A.ng = ng; A.abl = abl;
    A.cap = (u64)((double)A.n / ng / S * 1.25) + 65536;
    if ((u64)ng * S * A.cap * 16 > (40ull << 30)) { printf("%-40s skipped: queue too large\n", name); return; }
    const size_t lds = (size_t)S * 16 + 64 + T * 16 + T * 2 + 32;
    CK(hipFuncSetAttribute((const void *)scatter<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(A.gcount, 0, (size_t)ng * S * 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((scatter<R>), dim3(wgs), dim3(1024), lds, 0, A);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms);
    }
    std::vector<u64> cnt((size_t)ng * S);
    CK(hipMemcpy(cnt.data(), A.gcount, cnt.size() * 8, hipMemcpyDeviceToHost));
    u64 tot = 0, mx = 0; for (auto f : cnt) { tot += f; mx = f > mx ? f : mx; }
    char nm[128]; snprintf(nm, sizeof nm, "%s tick=2^%d*10ns pn=%d rn=%d gate=%d", name, tick, pn, rn, gate); name = nm;
    printf("%-64s sets=%4d S=%4u abl=%d  %7.3f ms  %6.1f Grows/s  (records %llu%s, fullest stream %llu of cap %llu)\n", name, ng, S, abl, best, A.n / best / 1e6, tot, tot == A.n ? "" : " MISMATCH", mx, A.cap);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const u64 n = argc > 1 ? (u64)atof(argv[1]) : (1ull << 29);
    long long *ks; u64 *v;
    CK(hipMalloc(&ks, n * 8)); CK(hipMalloc(&v, n * 8));
    gen<<<4096, 256>>>(ks, v, n, 1000000);
    char *q; CK(hipMalloc(&q, 41ull << 30));
    u64 *gcount; CK(hipMalloc(&gcount, 256ull * 1024 * 8));
    CK(hipDeviceSynchronize());
    Args A{};
    A.n = n; A.keys = ks; A.vals = v; A.q16 = (uint4 *)q; A.q12 = (uint32_t *)q; A.gcount = gcount;
    printf("# rows %llu\n", n);
    A.s_log2 = 9;
    for (int abl : {0, 8}) {
        run("free-running", A, 8, abl);
        for (int gate : {1, 2}) {
            run("gated", A, 8, abl, 7, 8, 4, gate);   // 10.24 us period, half / half
            run("gated", A, 8, abl, 7, 9, 5, gate);   // 11.5 us
            run("gated", A, 8, abl, 7, 10, 5, gate);  // 12.8 us
            run("gated", A, 8, abl, 7, 12, 6, gate);  // 15.4 us
            run("gated", A, 8, abl, 7, 8, 5, gate);   // 10.24 us, 6.4 read / 3.84 write
            run("gated", A, 8, abl, 7, 16, 8, gate);  // 20.5 us (two natural tiles per period: one per phase pair would idle)
            run("gated", A, 8, abl, 6, 9, 5, gate);   // 5.8 us
        }
        run("... no stores", A, 8, abl | 1);
    }
    return 0;
}

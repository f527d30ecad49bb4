// Microbench 6 (round 3): what do the cold-record stores of the bench pass cost, and which store pattern is cheaper?
// One wave = 256-row tiles of three float64 columns (the bench pass's 24 B/row, two 16-byte loads per column per lane, the
// next tile requested before the current one is used), 16 waves per workgroup, one workgroup per CU.  Per tile ~24 of the
// 256 rows are "cold": a 12-byte record {value, local index} goes to one of the wave's S = 8 private streams.
//   MODE 0  no stores at all (the read loop alone)
//   MODE 1  production pattern: lanes 0..23 store one record each (dwordx3), neighbouring lanes to DIFFERENT streams
//   MODE 2  the same records with the lanes sorted by stream: 8 runs of 3 neighbouring lanes -> 36 contiguous bytes per run
//   MODE 3  as 2 with a structure-of-arrays queue: an 8-byte value store + a 2-byte index store per lane
//   MODE 4  records staged: every third tile 64 lanes store 72 records' worth (64) sorted by stream: runs of 8 lanes, dwordx3
//   MODE 5  as 4 with the structure-of-arrays queue
//   MODE 6  as 5 with ONE stream per flush (64 contiguous records: 512 B + 128 B)
//   MODE 7  every lane stores to the wave's sink record (one address)
//   MODE 8  as 6, a flush every sixth tile (half the records)
//   MODE 9  as 6 with whole, aligned lines (stream regions and fill levels multiples of 64 records)
//   MODE 10 as 9 with ONE stream per wave (4096 open streams instead of 32768)
//   MODE 11 as 10 with non-temporal stores
//   MODE 12 as 10 with ordinary (not `nt`) loads
//   MODE 13 as 10, the wave's stream only 64 records long (rewritten in place: the lines can stay in the L2)
//   MODE 14 as 10, the wave's stream wraps after 1536 records: 63 MB of queue in all (would fit the 256 MB Infinity Cache)
//   MODE 15 as 10, wraps after 384 records: 16 MB in all (would fit the L2s: 4 MB per XCD, 2 MB of queue per XCD)
//   MODE 16 as 10, but TIME-PHASED: a wave holds its flushes back (here: counts them — the records are synthetic) and issues them all
//           when bit 11 of the 100 MHz wall clock (s_memrealtime) flips, i.e. every wave of the chip writes in the same ~1 us window
//           every 20 us and only reads in between: does the memory side like its writes in bursts?
//   MODE 17 as 16 with an 82 us period (bit 13); MODE 18 as 17 with non-temporal stores; MODE 19 as 16 with a 328 us period (bit 15)
//   MODE 20 (round 4) compacted: four times per tile lanes 0..5 store six NEIGHBOURING 12-byte records of the wave's ONE stream
//           straight from the registers (positions from a ballot / mbcnt, no LDS ring, no flush); MODE 21 with non-temporal stores;
//           MODE 22 once per tile lanes 0..23 store 24 neighbouring records
// Lanes that have no record store to the sink (every lane executes every store, as in the production kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench6 tools/microbench6.hip ; run: tools/microbench6 [rows]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));

struct Args {
    const double *x, *y, *v;
    uint64_t n;
    uint32_t *q12;      // AoS queue: [waves][8][cap] x 12 bytes (+ sinks behind)
    uint64_t *qv;       // SoA queue values
    uint16_t *qi;       // SoA queue indices
    uint64_t cap;       // records per (wave, stream)
    double *out;        // keeps the loads alive
};

template <int MODE>
__global__ void __launch_bounds__(1024) pass(const Args A) {
    constexpr uint32_t TW = 256, S = 8;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const uint32_t ntiles = (uint32_t)((A.n + TW - 1) / TW), GW = gridDim.x * nwave;
    const uint32_t gwave = blockIdx.x * nwave + wave;
    uint32_t tile = gwave;
    if (tile >= ntiles) return;
    struct Raw { u32x4 b[3][2]; };
    auto request = [&](uint32_t t, Raw &raw) {
        const uint64_t r0 = (uint64_t)t * TW;
        const uint32_t rows = t + 1u == ntiles ? (uint32_t)(A.n - r0) : TW;
        const double *cols[3] = {A.x, A.y, A.v};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(cols[d] + r0), 0, (int)(rows * 8u), 0x00020000);
            raw.b[d][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane * 16u), 0, MODE == 12 ? 0 : 2);
            raw.b[d][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(lane * 16u), 1024, MODE == 12 ? 0 : 2);
        }
    };
    const uint64_t sbase = (uint64_t)gwave * S * A.cap;              // first record of the wave's stream 0
    const uint64_t sink = (uint64_t)GW * S * A.cap + (uint64_t)gwave * 16u;
    uint32_t filled = 0;   // records per stream so far (the same for every stream of the wave: 3 per tile)
    uint32_t phase = 0, flushes = 0, pending = 0;
    uint64_t epoch = 0;
    double acc = 0;
    auto store12 = [&](uint64_t dst, uint32_t local, uint64_t bits) {
        *(u32x3_a4 *)(A.q12 + dst * 3) = u32x3_a4{(uint32_t)bits, (uint32_t)(bits >> 32), local};
    };
    auto process = [&](const Raw &raw, uint32_t t) {
        uint64_t bits = ((uint64_t)raw.b[2][0][1] << 32) | raw.b[2][0][0];
        acc += __longlong_as_double((long long)(((uint64_t)(raw.b[0][0][1] ^ raw.b[1][0][3] ^ raw.b[0][1][1] ^ raw.b[1][1][3] ^ raw.b[2][1][1]) << 32) | raw.b[2][0][2]));
        const uint32_t local = (raw.b[0][0][0] ^ t) & 0x1fffu;
        if (MODE == 0) return;
        if (MODE == 7) { store12(sink, local, bits); return; }
        if (MODE >= 20 && MODE <= 22) {
            // round 4: COMPACTED records straight from the registers into ONE stream per wave — what a ballot / mbcnt position
            // gives: a row-step's cold rows (6 of 64 lanes) are neighbouring records of the wave's stream, no LDS ring, no flush
            constexpr uint32_t steps = MODE == 22 ? 1u : 4u, per = 24u / steps;
#pragma unroll
            for (uint32_t r = 0; r < steps; ++r) {
                const uint64_t dst = lane < per ? sbase + filled + lane : sink;
                if (MODE == 21) __builtin_nontemporal_store(u32x3_a4{(uint32_t)bits, (uint32_t)(bits >> 32), local + r}, (u32x3_a4 *)(A.q12 + dst * 3));
                else store12(dst, local + r, bits);
                filled += per;
            }
            return;
        }
        if (MODE == 1 || MODE == 2 || MODE == 3) {
            const bool live = lane < 24u;
            // MODE 1: lane -> stream lane % 8, record (lane / 8) of the tile's three; MODE 2/3: stream lane / 3, record lane % 3
            const uint32_t s = MODE == 1 ? (lane & 7u) : lane / 3u, k = MODE == 1 ? (lane >> 3) : lane % 3u;
            const uint64_t dst = live ? sbase + (uint64_t)s * A.cap + filled + k : sink;
            if (MODE == 3) {
                A.qv[dst] = bits;
                A.qi[dst] = (uint16_t)local;
            } else {
                store12(dst, local, bits);
            }
            filled += 3;
            return;
        }
        if (MODE >= 16) {
            if (++phase == 3u) { phase = 0; ++pending; }
            const uint64_t now = __builtin_amdgcn_s_memrealtime() >> (MODE == 16 ? 11 : (MODE == 19 ? 15 : 13));
            if (now != epoch) {
                epoch = now;
                while (pending) {
                    const uint64_t dst = sbase + filled + lane;
                    if (MODE == 18) {
                        __builtin_nontemporal_store(bits, A.qv + dst);
                        __builtin_nontemporal_store((uint16_t)local, A.qi + dst);
                    } else {
                        A.qv[dst] = bits;
                        A.qi[dst] = (uint16_t)local;
                    }
                    filled += 64u;
                    --pending;
                }
            }
            return;
        }
        // MODE 4/5/6: every third tile a flush of 64 records (the 72 of three tiles, minus 8 to keep it simple)
        if (++phase == (MODE == 8 ? 6u : 3u)) {
            phase = 0;
            const bool one = MODE >= 10;                       // one stream per wave
            const bool whole = MODE == 6 || MODE >= 8;          // 64 records of one stream per flush
            const uint32_t s = one ? 0u : (whole ? (flushes & 7u) : (lane >> 3)), k = whole ? lane : (lane & 7u);
            ++flushes;
            const uint64_t dst = sbase + (uint64_t)s * A.cap + (MODE == 13 ? 0u : (MODE == 14 ? filled % 1536u : (MODE == 15 ? filled % 384u : filled))) + k;
            if (MODE == 4) {
                store12(dst, local, bits);
            } else if (MODE == 11) {
                __builtin_nontemporal_store(bits, A.qv + dst);
                __builtin_nontemporal_store((uint16_t)local, A.qi + dst);
            } else {
                A.qv[dst] = bits;
                A.qi[dst] = (uint16_t)local;
            }
            filled += one ? 64u : (whole ? (s == 7u ? 64u : 0u) : 8u);
        }
    };
    Raw a, b;
    request(tile, a);
    for (;;) {
        uint32_t next = tile + GW;
        bool has_next = next < ntiles;
        request(has_next ? next : tile, b);
        process(a, tile);
        if (!has_next) break;
        tile = next;
        next = tile + GW;
        has_next = next < ntiles;
        request(has_next ? next : tile, a);
        process(b, tile);
        if (!has_next) break;
        tile = next;
    }
    if (MODE >= 16)
        while (pending) { // (what the last period left)
            A.qv[sbase + filled + lane] = (uint64_t)__double_as_longlong(acc);
            filled += 64u;
            --pending;
        }
    if (acc == 1.2345e-300) A.out[0] = acc;
}

int main(int argc, char **argv) {
    const uint64_t n = argc > 1 ? (uint64_t)atof(argv[1]) : 1000000000ull;
    const bool only_new = argc > 2; // a second argument: the round-4 modes next to their references (0, 1, 7, 10, 11)
    int dev_cus = 256;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    dev_cus = prop.multiProcessorCount;
    const uint32_t wgs = dev_cus, waves = wgs * 16;
    const uint64_t tiles_per_wave = (n / 256 + waves - 1) / waves + 2;
    const uint64_t cap = (tiles_per_wave * 3 + 128 + 63) / 64 * 64 * 1;   // records per (wave, stream); MODE 10+: the wave's eight regions are used as one
    const uint64_t recs = (uint64_t)waves * 8 * cap + (uint64_t)waves * 16 + 64;
    Args A{};
    A.n = n;
    A.cap = cap;
    double *cols;
    CK(hipMalloc(&cols, n * 8 * 3));
    CK(hipMemset(cols, 0x3c, n * 8 * 3));
    A.x = cols; A.y = cols + n; A.v = cols + 2 * n;
    CK(hipMalloc(&A.q12, recs * 12));
    CK(hipMalloc(&A.qv, recs * 8));
    CK(hipMalloc(&A.qi, recs * 2));
    CK(hipMalloc(&A.out, 8));
    printf("rows %.3g, %u workgroups x 16 waves, %.2f GB of 12-byte records per pass (9.4 %% of the rows)\n", (double)n, wgs, (double)n * 24 / 256 * 12 / 1e9);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[23] = {"read loop alone", "scattered dwordx3 (production)", "sorted by stream, runs of 3, dwordx3", "sorted, runs of 3, SoA 8+2 bytes",
                            "staged: 64 per flush, runs of 8, dwordx3", "staged: 64 per flush, runs of 8, SoA", "staged: 64 per flush, one stream, SoA", "every lane to the sink",
                            "one stream per flush, half the records", "one stream per flush, aligned whole lines", "aligned, ONE stream per wave", "... non-temporal stores", "... ordinary loads", "... stream rewritten in place (64 records)", "... stream wraps: 63 MB of queue in all", "... stream wraps: 16 MB of queue in all",
                            "time-phased: all waves flush every 20 us", "time-phased: every 82 us", "time-phased: every 82 us, non-temporal", "time-phased: every 328 us",
                            "compacted from registers, one stream, 4 x 6 records", "... non-temporal", "... 1 x 24 records per tile"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 23; ++mode) {
            if (only_new && !(mode == 0 || mode == 1 || mode == 7 || mode == 10 || mode == 11 || mode >= 20)) continue;
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                CK(hipEventRecord(e0));
                switch (mode) {
                case 0: hipLaunchKernelGGL(pass<0>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 1: hipLaunchKernelGGL(pass<1>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 2: hipLaunchKernelGGL(pass<2>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 3: hipLaunchKernelGGL(pass<3>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 4: hipLaunchKernelGGL(pass<4>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 5: hipLaunchKernelGGL(pass<5>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 6: hipLaunchKernelGGL(pass<6>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 7: hipLaunchKernelGGL(pass<7>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 8: hipLaunchKernelGGL(pass<8>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 9: hipLaunchKernelGGL(pass<9>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 10: hipLaunchKernelGGL(pass<10>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 11: hipLaunchKernelGGL(pass<11>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 12: hipLaunchKernelGGL(pass<12>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 13: hipLaunchKernelGGL(pass<13>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 14: hipLaunchKernelGGL(pass<14>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 15: hipLaunchKernelGGL(pass<15>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 16: hipLaunchKernelGGL(pass<16>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 17: hipLaunchKernelGGL(pass<17>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 18: hipLaunchKernelGGL(pass<18>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 19: hipLaunchKernelGGL(pass<19>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 20: hipLaunchKernelGGL(pass<20>, dim3(wgs), dim3(1024), 0, 0, A); break;
                case 21: hipLaunchKernelGGL(pass<21>, dim3(wgs), dim3(1024), 0, 0, A); break;
                default: hipLaunchKernelGGL(pass<22>, dim3(wgs), dim3(1024), 0, 0, A); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            if (rep) printf("MODE %2d  %-46s %7.3f ms  %6.2f TB/s of the 24 B/row\n", mode, names[mode], best, (double)n * 24 / best / 1e9);
        }
    return 0;
}

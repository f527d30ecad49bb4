// standalone check of the DPP-scan slab ranking against the ballot version (round 4 experiment)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ uint32_t wave_scan_add_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}
__global__ void k(const uint32_t *in, uint32_t *scan_out, uint32_t *rank_a, uint32_t *rank_b, unsigned long long *hdr_a, unsigned long long *hdr_b, int count) {
    const uint32_t lane = threadIdx.x & 63u, base = blockIdx.x * 64u;
    const bool live = (int)lane < count;
    const uint32_t ix = in[base + lane];
    const uint32_t sl = live ? (ix & 7u) : 0xffu;
    // ballots
    uint32_t start = 0, rank = 0; uint64_t hdr = 0;
    for (uint32_t s8 = 0; s8 < 8u; ++s8) {
        const unsigned long long bmask = __ballot(sl == s8);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bmask, 0u));
        if (sl == s8) rank = start + below;
        start += (uint32_t)__builtin_popcountll(bmask);
        hdr |= (uint64_t)start << (8u * s8);
    }
    rank_a[base + lane] = rank; if (lane == 0) hdr_a[blockIdx.x] = hdr;
    // dpp
    const uint32_t one = live ? 1u << (8u * (sl & 3u)) : 0u;
    const uint32_t sc_lo = wave_scan_add_u32(sl < 4u ? one : 0u), sc_hi = wave_scan_add_u32((sl >= 4u && sl < 8u) ? one : 0u);
    scan_out[base + lane] = sc_lo;
    const uint32_t tot_lo = (uint32_t)__builtin_amdgcn_readlane((int)sc_lo, 63), tot_hi = (uint32_t)__builtin_amdgcn_readlane((int)sc_hi, 63);
    const uint32_t end_lo = tot_lo * 0x01010101u, end_hi = tot_hi * 0x01010101u + (end_lo >> 24) * 0x01010101u;
    const uint64_t hdr2 = ((uint64_t)end_hi << 32) | end_lo;
    const uint64_t starts = hdr2 - (((uint64_t)tot_hi << 32) | tot_lo);
    const uint32_t within = ((sl < 4u ? sc_lo : sc_hi) >> (8u * (sl & 3u))) & 0xffu;
    const uint32_t rank2 = live ? ((uint32_t)(starts >> (8u * sl)) & 0xffu) + within - 1u : 0u;
    rank_b[base + lane] = live ? rank2 : 0; if (lane == 0) hdr_b[blockIdx.x] = hdr2;
}
int main() {
    const int NB = 64, N = NB * 64;
    uint32_t h[N]; for (int i = 0; i < N; i++) h[i] = (uint32_t)(i * 2654435761u >> 7);
    uint32_t *in, *sc, *ra, *rb; unsigned long long *ha, *hb;
    hipMalloc(&in, N * 4); hipMalloc(&sc, N * 4); hipMalloc(&ra, N * 4); hipMalloc(&rb, N * 4); hipMalloc(&ha, NB * 8); hipMalloc(&hb, NB * 8);
    hipMemcpy(in, h, N * 4, hipMemcpyHostToDevice);
    for (int count : {64, 37, 1}) {
        hipLaunchKernelGGL(k, dim3(NB), dim3(64), 0, 0, in, sc, ra, rb, ha, hb, count);
        static uint32_t A[N], B[N], SC[N]; static unsigned long long HA[NB], HB[NB];
        hipMemcpy(A, ra, N * 4, hipMemcpyDeviceToHost); hipMemcpy(B, rb, N * 4, hipMemcpyDeviceToHost); hipMemcpy(SC, sc, N * 4, hipMemcpyDeviceToHost);
        hipMemcpy(HA, ha, NB * 8, hipMemcpyDeviceToHost); hipMemcpy(HB, hb, NB * 8, hipMemcpyDeviceToHost);
        int bad = 0, badh = 0, first = -1;
        for (int i = 0; i < N; i++) if ((i & 63) < count && A[i] != B[i]) { bad++; if (first < 0) first = i; }
        for (int b = 0; b < NB; b++) if (HA[b] != HB[b]) badh++;
        printf("count %d: rank mismatches %d (first at %d: ballot %u dpp %u), header mismatches %d\n", count, bad, first, first >= 0 ? A[first] : 0, first >= 0 ? B[first] : 0, badh);
        if (bad) { printf("block 0 scan_lo: "); for (int i = 0; i < 64; i++) printf("%x ", SC[i]); printf("\nslabs: "); for (int i = 0; i < 64; i++) printf("%u ", h[i] & 7); printf("\n"); }
    }
    return 0;
}

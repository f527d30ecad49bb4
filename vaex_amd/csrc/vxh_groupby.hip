// K7 — hash groupby in one partitioned pass: df.groupby(<integer key>).agg({v: [count, sum, mean, var, std]}).
//
// vaex does this in two passes over the rows: pass 1 collects the distinct keys in an ordered_set (vaex/hash.py:152-171,
// src/hash_primitives.hpp:98-295), pass 2 maps every key to its ordinal (map_ordinal :611-691 — one random probe of a
// table of all keys per row) and bins the ordinals (BinnerOrdinal + AggCount / AggSum / AggSumMoment, vaex/cpu.py:678-786).
// On the GPU a table of 1e6 keys lives in Infinity Cache / HBM and a probe per row costs a random 64-byte sector: the
// round-1 path ran at 0.02-0.05 of the HBM roofline (profiles/r01_configs.txt).  This file restates the same result —
// per key: rows, and per value column count / sum / sum of squares of the non-NaN values — as a radix-partitioned
// aggregation whose hash table is probed in LDS:
//
//   gb_scatter   rows -> NB buckets by the TOP bits of splitmix64(key) (the reference's hash finaliser, src/hash.hpp:40-45).
//                One 1024-thread workgroup per CU bucket-sorts 1024*R-row tiles in LDS (returning ds_add = position
//                in bucket, workgroup scan = bucket offsets) and appends each bucket's segment to the bucket's record STREAM of
//                the workgroup's SET — round 4: the 32 workgroups of an XCD share one set of NB streams, a tile's segment is
//                reserved with one returning device atomic on the stream's counter by the thread that owns the bucket
//                (thread b <-> bucket b), so a stream grows by neighbouring segments written within a microsecond of each
//                other (rounds 2-3: every (workgroup, bucket) pair had a private block — 131072 streams growing 192 bytes a
//                tile, which the memory side absorbed at ~3 TB/s).  Records: 16 bytes {key, payload} or 12 {remainder,
//                payload} with one payload word, SoA key + W words otherwise (W = 1 per value column for rows; 4 / 7 for
//                partial results being merged).  Heavy keys named by the host are aggregated in LDS partials instead.
//   gb_reduce    one workgroup per bucket: insert-or-get in an open-addressing table IN LDS — round 3: BUCKETISED, four keys
//                per 32-byte line read with two ds_read_b128 and compared at once, a key lives in its home line or (when
//                that is full) in the next ones; 64-bit ds_cmpst claims an empty word.  A probe is one trip for ~99 % of the
//                records (round 2's one-key-per-trip linear probing kept every wave in a divergent, scalar-issue-bound
//                loop: 1.22e9 SALU vs 0.69e9 VALU wave-instructions per 1e9 records, profiles/r02_pmc_groupby_fused.txt).
//                Accumulators (rows, count, sum, sum of squares) sit in LDS arrays indexed by the slot (ds_add_u64 /
//                ds_add_f64), the waves walk the bucket's stream of every set in blocks of 2048 records, software-pipelined.  At the end the occupied slots are compacted
//                (workgroup scan) into the result arrays behind ONE atomic per bucket.
//   sort         rocPRIM radix sort of (key, position) + a gather: groups ascending by key, as vaex returns them.
//
// HBM traffic per row with one value column: 16 B read + 16 B written + 16 B read = 48 B (the two-pass scheme with a
// global table: 8 + 16 B of streams plus two random probes).  No ordinals, no global table, no host in the loop.
// Partial results (multi-chunk inputs, other ranks' results) are merged by the same two kernels in MERGE mode.
#include "vxh_internal.hpp"

#ifdef VXH_ABLATE
#define VXH_GB_ABL(G, mask) ((G).abl & (mask))
#else
#define VXH_GB_ABL(G, mask) 0
#endif

#include <string.h> // (rocPRIM's texture_cache_iterator.hpp calls memset without including it)

#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <vector>

namespace {

constexpr long long GB_EMPTY = (long long)0x8000000000000000ull; // never stored as a key: the key INT64_MIN has its own slot
constexpr int GB_MAX_NV = 2;
constexpr int GB_MAX_W = 1 + 3 * GB_MAX_NV; // MERGE payload: rows, then (count, sum, sum2) per value column

__device__ __forceinline__ uint64_t gb_mix(uint64_t x) { // splitmix64 finaliser (src/hash.hpp:40-45)
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ long long gb_load_key(const void *p, uint64_t i, int dt) {
    switch (dt) {
    case VXH_I64: case VXH_U64: return ((const long long *)p)[i];
    case VXH_I32: return ((const int32_t *)p)[i];
    case VXH_U32: return ((const uint32_t *)p)[i];
    case VXH_I16: return ((const int16_t *)p)[i];
    case VXH_U16: return ((const uint16_t *)p)[i];
    case VXH_I8: return ((const int8_t *)p)[i];
    case VXH_U8: return ((const uint8_t *)p)[i];
    default: return ((const uint8_t *)p)[i] ? 1 : 0;
    }
}
// a key as its load returned it (zero-extended) -> the int64 the reference's ordered_set<T> holds (sign extension, bool != 0).
// Kept apart from the load: arithmetic on a loaded value next to the load makes the wave wait for it before the next load leaves.
__device__ __forceinline__ long long gb_fix_key(long long raw, int dt) {
    switch (dt) {
    case VXH_I32: return (long long)(int32_t)(uint32_t)raw;
    case VXH_I16: return (long long)(int16_t)(uint16_t)raw;
    case VXH_I8: return (long long)(int8_t)(uint8_t)raw;
    case VXH_I64: case VXH_U64: case VXH_U32: case VXH_U16: case VXH_U8: return raw;
    default: return raw ? 1 : 0;
    }
}

// the filter's terms on a row's value columns (GbArgs::pred; the rule of vxh_kernels.hip pred_keep2 / vxh_select.hip cmp_f64: a term
// `x <op> c` holds iff bit (relation of x to c: 0 less, 1 equal, 2 greater, 3 unordered) of its code is set — numpy's NaN rules)
__device__ __forceinline__ bool gb_pred_keep(const PredDesc &Q, double x0, double x1) {
    uint32_t bits = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t < Q.nterms) {
            const double c = Q.c[t], x = Q.tcol[t] ? x1 : x0;
            const uint32_t rel = x < c ? 0u : (x == c ? 1u : (x > c ? 2u : 3u));
            bits |= ((Q.code[t] >> rel) & 1u) << t;
        }
    }
    return ((Q.truth >> bits) & 1u) != 0u;
}

struct GbArgs {
    // input rows
    const void *keys;
    int32_t key_dtype, nv, w, merge; // w payload words per record; merge: payload = partial results (rows, count, sum, sum2 ...)
    const uint64_t *payload[GB_MAX_W]; // RAW: the value columns (float64 bits); MERGE: rows, count_0, sum_0, sum2_0, ...
    const uint8_t *keep;               // RAW: one byte per row, 1 = the row takes part (a filter / selection over the whole call), or null
    // RAW (round 6, late): the filter as TERMS over the pass's own value columns (vxh_groupby_run_selected) — `pred.on`; term t reads value
    // column pred.tcol[t] (pred.col / col2 are unused: the payload words the row carries anyway are the columns).  gb_scatter evaluates
    // it on the rows it loads: no sel_eval pass in front, no keep byte written and read back.  With `keep` as well: both must hold.
    PredDesc pred;
    uint64_t n;
    // Queues (round 4, late): `ng` SETS of NB record streams; workgroup w appends to the streams of set w % ng — blocks are dealt to
    // the 8 XCDs round robin, so with ng = 8 the 32 workgroups of ONE XCD share a set.  Every tile's segment of a stream is reserved
    // with one returning device atomic on the stream's counter: neighbouring segments of a stream are written by the XCD's
    // workgroups within a microsecond of each other, the chip has 4096 streams growing line after line instead of 131072 private
    // ones growing 192 bytes a tile — which is what the memory side was slow at (tools/microbench8.hip: 4.8 -> 3.46 ms per 5.4e8
    // 16-byte records; 4 or 1 sets: 4.15 / 4.47; 32 sets: 3.76).  Stream (set, b) owns records [(set NB + b) cap, ... + cap).
    int32_t nb_log2, slots_log2;
    uint32_t ng;       // sets of streams
    uint64_t cap;      // records per stream
    const unsigned long long *gstart; // null: stream s owns records [s cap, (s + 1) cap); else [gstart[s], gstart[s + 1]) — the second attempt of a call
                                      // whose first found a stream too small lays the streams out from the counts that attempt measured
    unsigned long long *gcount; // [ng][NB] records reserved so far (may pass `cap`: the segments beyond it are dropped, flag 1, the host retries with more room)
    long long *qkey;        // [streams][cap]
    uint64_t *qw[GB_MAX_W]; // [streams][cap] each
    uint4 *qrec;            // W == 1: the queues hold 16-byte records {key, payload} instead (one store / one load per record)
    uint32_t lines;         // gb_reduce: 4-key lines of the LDS table (slots = 4 * lines; any count, not a power of two)
    // Compact records (round 4; W == 1, counting rows, the key RANGE known): key - kc_min is a kc_bits-bit number; an invertible
    // mix of it (multiply / xor-shift / multiply modulo 2^kc_bits) gives the bucket in its top nb_log2 bits and a remainder of
    // <= 32 bits below — what the bucket implies is not stored: records are 12 bytes {remainder, payload} instead of 16, in the
    // scatter's writes and the reduce's reads alike.  gb_reduce rebuilds the key of every GROUP (bucket | remainder, mixed back).
    int32_t kc_bits;        // 0: 16-byte records {key, payload}
    int32_t key32;          // compact records with a remainder of < 32 bits: gb_reduce keeps 32-bit keys in its table
    int32_t direct;         // compact records AND 2^(kc_bits - nb_log2) <= GB_DIRECT_SLOTS: gb_reduce indexes its table with the remainder (no keys, no probe)
    int32_t tag;            // compact records with a remainder of < 32 bits (no direct table): gb_reduce's TAG table (lines of four 4-byte entries indexed by the remainder's top bits, accumulators by group id)
    int32_t tag_idx_bits;   // ... log2 of its LINES of four entries (<= 12)
    uint32_t ids;           // ... group ids (accumulator sets) a bucket has room for (a multiple of 4, < 4095)
    // Heavy keys (round 4, the one-kernel peel): rows whose key is one of `n_heavy` <= 128 listed keys leave NO record — gb_scatter looks
    // every key up in an LDS copy of the list, adds such rows to per-workgroup partials in LDS {rows | count, sum, sum2 ...} and folds those
    // into heavy_acc at its end; gb_append_heavy turns the accumulators into ordinary groups in front of the sort.  (Every row of ONE key
    // lands in ONE bucket: a key with a few per cent of the rows overflows its queue, and long before that its single reduce workgroup is
    // the whole pass — src/hash_primitives.hpp:471-479 keeps such keys in one map like any other; the partitioned pass cannot.)
    const long long *heavy_keys;   // [n_heavy] distinct, none of them INT64_MIN
    int32_t n_heavy;
    unsigned long long *heavy_acc; // [n_heavy][1 + 3 nv]: rows, then per value column count, sum bits, sum2 bits
    int32_t abl;            // timing experiments ("gb_abl", the ablation build only — VXH_GB_ABL is a compile-time zero in the product library): 1 = gb_scatter's copy-out computes but does not store, 2 = no copy-out, 4 = no staging and no copy-out
    long long kc_min;
    // results (unsorted)
    unsigned long long *out_count; // groups written so far
    unsigned int *overflow;        // 1: out of spare blocks, 2: a bucket's LDS table too full, 3: result arrays too small
    uint64_t out_cap;
    long long *out_key;
    uint64_t *out_w[GB_MAX_W]; // rows (int64), then per value column: count (int64), sum (f64), sum2 (f64)
};

// ------------------------------------------------------------------------------------------------------------------
// gb_scatter
// ------------------------------------------------------------------------------------------------------------------
// The compact records' mix: a BIJECTION of [0, 2^bits) whose top bits name the bucket.  Round 6: three Feistel rounds over the two halves
// of the number (low bits / 2 bits and the rest; each at most 32 bits wide), every round ONE 32-bit multiply — the top bits of v * C depend
// on every bit of v, so after the rounds the bucket depends on all of the key.  (Rounds 4-5 mixed with two 64-bit multiplies modulo
// 2^bits: ~8 quarter-rate 32-bit multiplies per row — the ablation build showed gb_scatter spending ~2 ms per 1e9 rows in its ALU phases
// with every load, atomic and store taken out, profiles/r06_gb_scatter_ablation.txt.)  Any round function gives a bijection: the inverse
// runs the rounds backwards.
constexpr uint32_t GB_KC_C1 = 0x9e3779b1u, GB_KC_C2 = 0x85ebca77u, GB_KC_C3 = 0xc2b2ae3du;
__device__ __forceinline__ uint32_t gb_kc_f(uint32_t v, uint32_t c, int w) { return (v * c) >> (32 - w); } // the top w bits of the product (1 <= w <= 32)
__device__ __forceinline__ uint64_t gb_kc_mix(uint64_t x, int bits) {
    const int wr = bits >> 1, wl = bits - wr;
    uint32_t r = (uint32_t)(x & ((1ull << wr) - 1ull)), l = (uint32_t)(x >> wr);
    l ^= gb_kc_f(r, GB_KC_C1, wl);
    r ^= gb_kc_f(l, GB_KC_C2, wr);
    l ^= gb_kc_f(r, GB_KC_C3, wl);
    return ((uint64_t)l << wr) | (uint64_t)r;
}
__device__ __forceinline__ uint64_t gb_kc_unmix(uint64_t m, int bits) {
    const int wr = bits >> 1, wl = bits - wr;
    uint32_t r = (uint32_t)(m & ((1ull << wr) - 1ull)), l = (uint32_t)(m >> wr);
    l ^= gb_kc_f(r, GB_KC_C3, wl);
    r ^= gb_kc_f(l, GB_KC_C2, wr);
    l ^= gb_kc_f(r, GB_KC_C1, wl);
    return ((uint64_t)l << wr) | (uint64_t)r;
}
typedef unsigned int gb_u32x3 __attribute__((ext_vector_type(3)));
typedef gb_u32x3 gb_u32x3_a4 __attribute__((aligned(4)));

// K64: 8-byte keys (int64 / uint64: loaded as they are); KEEP: G.keep != null.  Both were run-time branches around every row's loads
// until round 4 — the compiler then waits for each load behind its branch (vmcnt(0) after every global_load in the ISA): the 16-24 loads
// of a thread's tile left one after the other, a memory latency each, and "the next tile requested under the current one" did not exist:
// 8.9 ms per 1e9 rows with the HBM idle most of the time.  Now the loads of a tile are issued back to back and nothing touches what
// they return before the next tile's turn.
// (Tried and removed: taking the next tile's registers over between staging [C] and copy-out [D], so that the wait for its loads is
//  not also a wait for [D]'s stores — 8.30 vs 8.33 ms per 1e9 rows; non-temporal copy-out stores ("gb_abl" bit 3): 8.28 vs 8.31;
//  the rows read with non-temporal loads, so that half-written queue lines might live longer in the L2: 8.74 vs 8.19.)
// HEAVY: G.n_heavy > 0 (RAW records only: W = value columns).
constexpr uint32_t GB_HEAVY_MAX = 128, GB_HEAVY_SLOTS = 256;
constexpr size_t gb_heavy_lds(int w) { return (size_t)GB_HEAVY_SLOTS * 8 + GB_HEAVY_SLOTS + (size_t)GB_HEAVY_MAX * 8 + (size_t)w * GB_HEAVY_MAX * 16 + (size_t)(w > 1 ? w - 1 : 0) * GB_HEAVY_MAX * 4 + 16; }
// KC (round 6): compact 12-byte records (G.kc_bits != 0; W == 1) as a compile-time switch like gb_reduce's — the staged key is then the
// 32-bit remainder (half the LDS bytes of the key staging), and the per-row `if (G.kc_bits)` is gone from [A] and [D].
// D16 (round 6; with a direct table in gb_reduce the remainder has <= 12 bits): the record is split over two streams of the same positions —
// the value's 8 bytes and the remainder's 2 — 10 bytes written and read back per row instead of 12.
template <int W, int R, bool K64, bool KEEP, bool HEAVY, bool KC = false, bool D16 = false>
__global__ void __launch_bounds__(1024) gb_scatter(const GbArgs G) {
    static_assert(!KC || W == 1, "compact records carry one payload word");
    static_assert(!D16 || KC, "the split record is a compact record");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t T = 1024u * R;
    const uint32_t NB = 1u << G.nb_log2; // <= 1024: thread b owns bucket b
    uint64_t *const st_key = (uint64_t *)lds;
    uint64_t *const st_w = st_key + T; // [W][T]
    uint32_t *const cnt = (uint32_t *)(st_w + (size_t)W * T);
    uint32_t *const off = cnt + NB;
    unsigned long long *const gbase = (unsigned long long *)(off + NB); // [NB] first record of this tile's segment of stream b (8-byte aligned: NB is even)
    uint32_t *const s_wave = (uint32_t *)(gbase + NB) + NB; // [16] (behind one spare word per bucket: the layout's size is scatter_lds')
    uint16_t *const st_b = (uint16_t *)(s_wave + 16);
    // HEAVY: the heavy keys' table (open addressing, <= half full) and this workgroup's partials, behind the staging area
    unsigned long long *const hk_key = (unsigned long long *)(((uintptr_t)(st_b + T) + 15) & ~(uintptr_t)15); // [GB_HEAVY_SLOTS]
    unsigned long long *const h_rc = hk_key + GB_HEAVY_SLOTS;                                                   // [GB_HEAVY_MAX] rows (low half) | count of value column 0 (high half)
    double *const h_sum = (double *)(h_rc + GB_HEAVY_MAX);                                                      // [W][GB_HEAVY_MAX]
    double *const h_sum2 = h_sum + (size_t)W * GB_HEAVY_MAX;                                                    // [W][GB_HEAVY_MAX]
    uint32_t *const h_cnt = (uint32_t *)(h_sum2 + (size_t)W * GB_HEAVY_MAX);                                    // [W - 1][GB_HEAVY_MAX] counts of value columns >= 1
    uint8_t *const hk_ord = (uint8_t *)(h_cnt + (size_t)(W > 1 ? W - 1 : 0) * GB_HEAVY_MAX);                    // [GB_HEAVY_SLOTS] position of the slot's key in G.heavy_keys
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t n = G.n;
    if (tid < NB) cnt[tid] = 0u;
    auto heavy_home = [](unsigned long long k) -> uint32_t { return (uint32_t)((k * 0x9e3779b97f4a7c15ULL) >> 56) & (GB_HEAVY_SLOTS - 1u); };
    if (HEAVY) {
        if (tid < GB_HEAVY_SLOTS) hk_key[tid] = (unsigned long long)GB_EMPTY;
        if (tid < GB_HEAVY_MAX) {
            h_rc[tid] = 0ull;
#pragma unroll
            for (int w = 0; w < W; ++w) { h_sum[(size_t)w * GB_HEAVY_MAX + tid] = 0.0; h_sum2[(size_t)w * GB_HEAVY_MAX + tid] = 0.0; if (w > 0) h_cnt[(size_t)(w - 1) * GB_HEAVY_MAX + tid] = 0u; }
        }
        __syncthreads();
        if (tid < (uint32_t)G.n_heavy) { // (distinct keys, a table twice their number: every insert finds a free slot)
            const unsigned long long k = (unsigned long long)G.heavy_keys[tid];
            for (uint32_t sl = heavy_home(k);; sl = (sl + 1u) & (GB_HEAVY_SLOTS - 1u)) {
                if (atomicCAS(&hk_key[sl], (unsigned long long)GB_EMPTY, k) == (unsigned long long)GB_EMPTY) { hk_ord[sl] = (uint8_t)tid; break; }
            }
        }
    }
    const uint32_t set = blockIdx.x % G.ng; // this workgroup's set of streams
    // thread b: where stream (set, b) starts and how many records it has room for
    unsigned long long s_first = 0ull, s_room = 0ull;
    if (tid < NB) {
        const size_t sid = (size_t)set * NB + tid;
        s_first = G.gstart ? G.gstart[sid] : (unsigned long long)sid * G.cap;
        s_room = G.gstart ? G.gstart[sid + 1] - s_first : G.cap;
    }
    __syncthreads();

    // the next tile's rows are requested while the current one is staged and copied out (two register sets; the copy
    // at the end of the iteration is where their loads are waited for, a barrier and a copy-out later)
    long long key[R], key_n[R];
    uint64_t pay[W][R], pay_n[W][R];
    bool ok[R], ok_n[R];
    uint32_t kb[R], kb_n[R]; // KEEP: the rows' keep bytes as loaded
    auto request = [&](uint64_t tile, long long (&k)[R], uint64_t (&p)[W][R], bool (&v)[R], uint32_t (&b)[R]) {
        uint64_t ic[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = tile * T + (uint64_t)r * 1024u + tid;
            v[r] = i < n;
            ic[r] = v[r] ? i : n - 1;
            b[r] = 1u;
        }
        if (VXH_GB_ABL(G, 32)) { // (timing: no row loads at all — synthetic keys and payloads)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                k[r] = (long long)(gb_mix(ic[r]) & ((1ull << 40) - 1ull));
#pragma unroll
                for (int w = 0; w < W; ++w) p[w][r] = ic[r];
            }
            return;
        }
        if (KEEP && G.keep) { // (KEEP without a mask: the filter is G.pred alone)
#pragma unroll
            for (int r = 0; r < R; ++r) b[r] = G.keep[ic[r]]; // (a row outside the filter leaves no record: a group without a row inside does not exist)
        }
        if (K64) {
#pragma unroll
            for (int r = 0; r < R; ++r) k[r] = ((const long long *)G.keys)[ic[r]];
        } else { // (wave-uniform switch around the R loads of a case, not inside every row)
            switch (G.key_dtype) {
            case VXH_I32: case VXH_U32:
#pragma unroll
                for (int r = 0; r < R; ++r) k[r] = (long long)(uint64_t)((const uint32_t *)G.keys)[ic[r]];
                break;
            case VXH_I16: case VXH_U16:
#pragma unroll
                for (int r = 0; r < R; ++r) k[r] = (long long)(uint64_t)((const uint16_t *)G.keys)[ic[r]];
                break;
            case VXH_I64: case VXH_U64:
#pragma unroll
                for (int r = 0; r < R; ++r) k[r] = ((const long long *)G.keys)[ic[r]];
                break;
            default:
#pragma unroll
                for (int r = 0; r < R; ++r) k[r] = (long long)(uint64_t)((const uint8_t *)G.keys)[ic[r]];
                break;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int w = 0; w < W; ++w) p[w][r] = G.payload[w][ic[r]];
        }
    };
    // what the loads returned -> the tile's keys and validity (run at the tile's turn, not at the request)
    auto settle = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!K64) key[r] = gb_fix_key(key[r], G.key_dtype);
            if (KEEP) ok[r] = ok[r] && kb[r] == 1u;
            if (KEEP && G.pred.on) ok[r] = ok[r] && gb_pred_keep(G.pred, __longlong_as_double((long long)pay[0][r]), __longlong_as_double((long long)pay[W > 1 ? 1 : 0][r]));
            if (HEAVY) { // a heavy key's row goes to the workgroup's partials and leaves no record
                const unsigned long long k = (unsigned long long)key[r];
                uint32_t sl = heavy_home(k), h = 0xffffffffu;
                for (;;) {
                    const unsigned long long t = hk_key[sl];
                    if (t == k) { h = hk_ord[sl]; break; }
                    if (t == (unsigned long long)GB_EMPTY) break;
                    sl = (sl + 1u) & (GB_HEAVY_SLOTS - 1u);
                }
                if (ok[r] && h != 0xffffffffu && key[r] != GB_EMPTY) {
                    const double d0 = __longlong_as_double((long long)pay[0][r]);
                    __hip_atomic_fetch_add(&h_rc[h], d0 == d0 ? 0x100000001ull : 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        const double d = __longlong_as_double((long long)pay[w][r]);
                        if (d == d) { // NaN values are skipped by count / sum / sum-moment alike (src/agg_sum.cpp:113, agg_count.cpp:56)
                            if (w > 0) __hip_atomic_fetch_add(&h_cnt[(size_t)(w - 1) * GB_HEAVY_MAX + h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(&h_sum[(size_t)w * GB_HEAVY_MAX + h], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(&h_sum2[(size_t)w * GB_HEAVY_MAX + h], d * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                    }
                    ok[r] = false;
                }
            }
        }
    };
    auto take_over = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            key[r] = key_n[r];
            ok[r] = ok_n[r];
            kb[r] = kb_n[r];
#pragma unroll
            for (int w = 0; w < W; ++w) pay[w][r] = pay_n[w][r];
        }
    };
    // (the first tile goes through the same two steps as every other one — requested into the `_n` set, taken over: loaded straight into
    //  `key` it left [A] with two histories of outstanding loads, and the compiler's wait counts in [A] then waited for the NEXT tile's keys)
    if ((uint64_t)blockIdx.x * T < n) { request(blockIdx.x, key_n, pay_n, ok_n, kb_n); take_over(); }
    for (uint64_t tile = blockIdx.x; tile * T < n; tile += gridDim.x) {
        // the next tile's rows are requested first: their loads fly under [A] .. [D]
        const uint64_t next = tile + gridDim.x;
        const bool has_next = next * T < n;
        if (has_next) request(next, key_n, pay_n, ok_n, kb_n);
        settle();
        // [A] bucket of every row of the tile, position inside the bucket
        uint32_t bucket[R], pos[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (KC) { // compact records: the key becomes its remainder, the bucket what the mix puts on top of it
                const uint64_t m = gb_kc_mix((uint64_t)key[r] - (uint64_t)G.kc_min, G.kc_bits);
                bucket[r] = (uint32_t)(m >> (G.kc_bits - G.nb_log2));
                key[r] = (long long)(m & ((1ull << (G.kc_bits - G.nb_log2)) - 1ull));
            } else {
                bucket[r] = (uint32_t)(gb_mix((uint64_t)key[r]) >> (64 - G.nb_log2));
            }
            pos[r] = 0;
            if (ok[r] && !VXH_GB_ABL(G, 16)) pos[r] = __hip_atomic_fetch_add(&cnt[bucket[r]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (bit 16, timing: no position atomics — the counts stay 0, nothing is staged or copied out)
        }
        __syncthreads();
        // [B] thread b: where bucket b's records of this tile go; exclusive scan of the bucket counts
        uint32_t c = 0;
        if (tid < NB) {
            c = cnt[tid];
            cnt[tid] = 0u;
            // this tile's segment of stream (set, b): ONE returning device atomic (a stream that passes its capacity drops the segment
            // and raises the flag: the host retries with more room)
            unsigned long long at = 0ull;
            if (c) at = atomicAdd(&G.gcount[(size_t)set * NB + tid], (unsigned long long)c);
            const bool fits = at + c <= s_room;
            if (c && !fits) atomicExch(G.overflow, 1u);
            gbase[tid] = fits ? s_first + at : ~0ull;
        }
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
            if ((int)lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        {
            uint32_t before = 0;
            for (uint32_t w2 = 0; w2 < wave; ++w2) before += s_wave[w2];
            if (tid < NB) {
                const uint32_t o = before + inc - c;
                off[tid] = o;
                // [D] wants `segment start - offset of the bucket inside the staged tile`: folded here, once per bucket (a segment starts at
                // >= one stream's room past 0 for every bucket but the first, whose offset is 0: the difference never wraps, never is ~0)
                if (gbase[tid] != ~0ull) gbase[tid] -= o;
            }
        }
        uint32_t total = 0;
        for (uint32_t w2 = 0; w2 < 16; ++w2) total += s_wave[w2];
        __syncthreads();
        // [C] stage sorted by bucket
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (ok[r] && !VXH_GB_ABL(G, 4)) {
                const uint32_t j = off[bucket[r]] + pos[r];
                if (KC) ((uint32_t *)st_key)[j] = (uint32_t)key[r]; else st_key[j] = (uint64_t)key[r];
#pragma unroll
                for (int w = 0; w < W; ++w) st_w[(size_t)w * T + j] = pay[w][r];
                st_b[j] = (uint16_t)bucket[r];
            }
        }
        __syncthreads();
        // [D] copy out: consecutive threads -> consecutive records of a bucket's segment
        for (uint32_t j = tid; j < (VXH_GB_ABL(G, 6) ? 0u : total); j += 1024u) {
            const uint32_t b = st_b[j];
            const unsigned long long base = gbase[b]; // (segment start - the bucket's offset in the staged tile)
            if (base == ~0ull) continue; // (stream full: flagged, the host retries with more room)
            uint64_t dst = base + j;
            if (VXH_GB_ABL(G, 1)) { if ((KC ? (uint64_t)((uint32_t *)st_key)[j] : st_key[j]) + st_w[j] != 0x123456789abcdefull) continue; dst = 0; } // (the staged words are read, nothing is stored)
            if (D16) { // 10 bytes: the value into the stream's 8-byte half, the remainder into its 2-byte half, same position
                ((uint64_t *)G.qkey)[dst] = st_w[j];
                ((uint16_t *)G.qw[0])[dst] = (uint16_t)((uint32_t *)st_key)[j];
            } else if (KC) { // 12-byte record {remainder, payload}
                const uint64_t c2 = st_w[j];
                const gb_u32x3_a4 rec = gb_u32x3_a4{((uint32_t *)st_key)[j], (uint32_t)c2, (uint32_t)(c2 >> 32)};
                if (VXH_GB_ABL(G, 8)) __builtin_nontemporal_store(rec, (gb_u32x3_a4 *)((uint32_t *)G.qrec + dst * 3)); // (experiment: non-temporal)
                else *(gb_u32x3_a4 *)((uint32_t *)G.qrec + dst * 3) = rec;
            } else if (W == 1) { // one 16-byte record {key, payload}: a tile's segment of a bucket is 16 B x its records, contiguous
                const uint64_t a = st_key[j], c2 = st_w[j];
                G.qrec[dst] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)c2, (uint32_t)(c2 >> 32));
            } else {
                G.qkey[dst] = (long long)st_key[j];
#pragma unroll
                for (int w = 0; w < W; ++w) G.qw[w][dst] = st_w[(size_t)w * T + j];
            }
        }
        // (the next tile's [C] comes after two more barriers: nobody overwrites what [D] still reads)
        if (has_next) take_over();
    }
    if (HEAVY) { // this workgroup's partials into the call's accumulators (a few hundred device atomics per workgroup)
        __syncthreads();
        if (tid < (uint32_t)G.n_heavy) {
            const unsigned long long rc = h_rc[tid];
            if (rc & 0xffffffffull) {
                unsigned long long *acc = G.heavy_acc + (size_t)tid * (1 + 3 * W);
                atomicAdd(&acc[0], rc & 0xffffffffull);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const unsigned long long c = w == 0 ? rc >> 32 : (unsigned long long)h_cnt[(size_t)(w - 1) * GB_HEAVY_MAX + tid];
                    if (c) {
                        atomicAdd(&acc[1 + 3 * w], c);
                        atomicAdd((double *)&acc[2 + 3 * w], h_sum[(size_t)w * GB_HEAVY_MAX + tid]);
                        atomicAdd((double *)&acc[3 + 3 * w], h_sum2[(size_t)w * GB_HEAVY_MAX + tid]);
                    }
                }
            }
        }
    }
}

// heavy_acc -> groups: one thread per heavy key with at least one row (behind gb_reduce, in front of the sort)
__global__ void gb_append_heavy(const GbArgs G) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= (uint32_t)G.n_heavy) return;
    const int wout = 1 + 3 * G.nv;
    const unsigned long long *acc = G.heavy_acc + (size_t)h * wout;
    if (acc[0] == 0ull) return;
    const unsigned long long o = atomicAdd(G.out_count, 1ull);
    if (o >= G.out_cap) { atomicExch(G.overflow, 3u); return; }
    G.out_key[o] = G.heavy_keys[h];
    for (int k = 0; k < wout; ++k) G.out_w[k][o] = acc[k];
}

// ------------------------------------------------------------------------------------------------------------------
// gb_reduce
// ------------------------------------------------------------------------------------------------------------------
// LDS table of one bucket: `lines` lines of four keys (SLOTS = 4 * lines) + one entry for the key INT64_MIN, which doubles as EMPTY
// KC: the queues hold compact 12-byte records (GbArgs::kc_bits != 0) — a compile-time switch: as a run-time branch around every
// record load it made the compiler wait for each load on its own (vmcnt(0) behind every global_load: 4.0 -> 5.2 ms per 1e9 records)
// (records per lane per trip with one payload word: 8 instead of 4 — twice the loads in flight — changes nothing, 4.03 vs 4.03 ms per 1e9
//  records in two libraries on one box: after the software pipeline the kernel is bound by its LDS work, ~2 ms of atomics + probes per CU)
#ifndef GB_REDUCE_U
#define GB_REDUCE_U 4
#endif
// DIRECT (round 6; KC records only): the remainder of a compact record IS the slot — when the key range is narrow enough for
// 2^(kc_bits - nb_log2) accumulators to fit the LDS (<= 4096: every key range the dense BinnerOrdinal path takes), the bijective mix
// makes (bucket, remainder) a PERFECT hash: no keys in the table, no probe, no claim, no overflow — a record costs its three atomics.
// This is what the slab-partitioned dense groupby (part_scatter_f64 + part_reduce_fast) does behind a pass 1 that moves 27 GB per 1e9
// rows at 4.0 TB/s; here it sits behind gb_scatter's shared streams (28 GB at 5.0 TB/s).
// K32 (round 6; KC records whose remainder has < 32 bits): the table's keys are 32-bit words — a line of four keys is ONE ds_read_b128
// instead of two, the claim a 32-bit ds_cmpst (the probe's LDS reads were ~40 % of the kernel's LDS time: DESIGN section 3).
// TAG (round 6, late; KC records whose remainder has < 32 bits and no direct table — the scattered 1e6-key groupby): the probe as
// STRAIGHT-LINE code.  The counters of the 4-key-line table (profiles/r06_gb_reduce_pmc.txt) showed gb_reduce bound by instruction issue,
// not by the LDS: 67 scalar + 62 vector instructions per 64 records (the divergent insert-or-get loop: exec-mask bookkeeping, and ~85 % of
// the waves took a second trip for the one lane in 64 whose key lived in an overflow line) against 8 + 17 behind a direct table.  Here the
// bijective mix's remainder supplies the hash: its top 12 bits name a LINE of four 4-byte entries {tag = the remainder's other bits (<= 19),
// group id (12 bits)} — 4096 lines = 64 KiB at ~12 % load — and the accumulators are dense arrays indexed by the group id, handed out in
// order of first appearance.  A record: one ds_read_b128, four xors and a min3/min — an entry of the same tag xors to its id (< 4094), anything
// else to >= 4096 — and the three atomics under `hit`.  Records that miss (the first of a key; ~0.03 % of the keys live in an overflow list
// because their line was full) are collected in a per-lane mask and settled in ONE slow pass per trip of 4 x 64 records.
template <int NV, bool MERGE, bool KC = false, bool DIRECT = false, bool K32 = false, bool TAG = false>
__global__ void __launch_bounds__(1024) gb_reduce(const GbArgs G) {
    static_assert(!DIRECT || (KC && !MERGE), "a direct table needs compact records");
    static_assert(!K32 || (KC && !MERGE && !DIRECT), "32-bit table keys need compact records");
    static_assert(!TAG || (K32 && NV == 1), "the tag table holds 32-bit entries of compact records");
    using KT = typename std::conditional<K32, uint32_t, unsigned long long>::type;
    constexpr KT KEMPTY = K32 ? (KT)0xffffffffu : (KT)GB_EMPTY; // (a remainder of < 32 bits is never all ones; a tag entry has its top two bits clear)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // row counters: 32 bits while counting rows (a workgroup sees < 2^32 of them), 64 bits when merging partial counts
    using CT = typename std::conditional<MERGE, unsigned long long, uint32_t>::type;
    const uint32_t LINES = G.lines, SLOTS = TAG ? G.ids : (DIRECT ? 1u << (G.kc_bits - G.nb_log2) : 4u * LINES), E = DIRECT ? SLOTS : SLOTS + 1, EP = (E + 3) & ~3u; // (TAG: entry SLOTS is the spare accumulator set of the lanes without a hit) // (EP: keeps the arrays 16-byte aligned)
    const uint32_t TS = TAG ? 4u << G.tag_idx_bits : 0u;           // TAG: entries of the tag table (four per line)
    KT *const t_key = (KT *)lds;                                    // [EP]: line l = t_key[4 l .. 4 l + 3]; TAG: [TS] entries
    double *const t_sum = (double *)(t_key + (TAG ? TS : EP));      // [NV][EP] (EP is a multiple of 4: 16-byte aligned behind 32-bit keys too)
    double *const t_sum2 = t_sum + (size_t)NV * EP;                 // [NV][EP]
    // counting rows: rows[] and the NaN values of column 0 nan0[] as two dense 32-bit arrays (count of column 0 = rows - nan0: ONE 4-byte
    // atomic per record instead of an 8-byte one, a second only for a NaN), further columns' counts follow as 32-bit arrays;
    // merging: 64-bit rows[] and counts[][]
    unsigned long long *const t_rc = (unsigned long long *)(t_sum2 + (size_t)NV * EP);                           // RAW: [EP] x 8 bytes = rows32[EP] | nan32[EP]
    CT *const t_rows = (CT *)t_rc;                                                                                // MERGE: [EP]
    uint32_t *const t_rows32 = (uint32_t *)t_rc, *const t_nan32 = t_rows32 + EP;                                  // RAW
    CT *const t_cnt = MERGE ? t_rows + EP : (CT *)(t_rc + EP) - EP;                                               // MERGE: [NV][EP]; RAW: [v >= 1][EP] behind t_rc
    uint32_t *const s_misc = (uint32_t *)((char *)t_rc + (MERGE ? (size_t)8 * EP * (1 + NV) : (size_t)8 * EP + (size_t)4 * EP * (NV - 1))); // [0] claimed slots, [1] output base, [2..17] wave totals
    const uint32_t tid = threadIdx.x, lane = tid & 63u, nwave = blockDim.x >> 6;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)); // (scalar: the block iterator's table loads are s_loads)
    const uint32_t bucket = blockIdx.x;
    // a stream passed its room in gb_scatter (segments were dropped, the counters point at records nobody wrote): nothing to reduce —
    // the host lays the streams out from the counters and runs the pass again
    if (__hip_atomic_load(G.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) return;
    if (TAG) for (uint32_t s = tid; s < TS; s += blockDim.x) t_key[s] = KEMPTY;
    for (uint32_t s = tid; s < EP; s += blockDim.x) {
        if (!DIRECT && !TAG) t_key[s] = KEMPTY;
        if (MERGE) t_rows[s] = (CT)0; else t_rc[s] = 0ull;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            t_sum[(size_t)v * EP + s] = 0.0;
            t_sum2[(size_t)v * EP + s] = 0.0;
            if (MERGE || v > 0) t_cnt[(size_t)v * EP + s] = (CT)0;
        }
    }
    if (tid < 18 + (TAG ? 2 : 0)) s_misc[tid] = 0u; // (TAG: the overflow list's count and lock sit right behind)
    __syncthreads();

    const uint32_t limit = SLOTS - SLOTS / 5; // more distinct keys than this in one bucket (80 % of the slots): the overflow chains get long (a wave pays for the longest of its 64) — flag and let the host retry with more buckets
    constexpr int PW = MERGE ? 1 + 3 * NV : NV;
    constexpr int U = PW >= 7 ? 1 : (PW >= 4 ? 2 : (PW == 1 ? GB_REDUCE_U : 4)); // records per lane per trip (their loads run interleaved), two trips in registers: <= 128 VGPRs at 16 waves
    bool failed = false;
    auto accumulate = [&](uint32_t s, const uint64_t (&p)[PW]) {
        if (MERGE) {
            __hip_atomic_fetch_add(&t_rows[s], (CT)p[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                __hip_atomic_fetch_add(&t_cnt[(size_t)v * EP + s], (CT)p[1 + 3 * v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&t_sum[(size_t)v * EP + s], __longlong_as_double((long long)p[2 + 3 * v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&t_sum2[(size_t)v * EP + s], __longlong_as_double((long long)p[3 + 3 * v]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            // rows: one 4-byte atomic; the count of value column 0 is rows minus its NaNs, counted apart (rare)
            const double d0 = __longlong_as_double((long long)p[0]);
            __hip_atomic_fetch_add(&t_rows32[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (d0 != d0) __hip_atomic_fetch_add(&t_nan32[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const double d = __longlong_as_double((long long)p[v]);
                if (d == d) { // NaN values are skipped by count / sum / sum-moment alike (src/agg_sum.cpp:113, agg_count.cpp:56)
                    if (v > 0) __hip_atomic_fetch_add(&t_cnt[(size_t)v * EP + s], (CT)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&t_sum[(size_t)v * EP + s], d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(&t_sum2[(size_t)v * EP + s], d * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    };
    // home line of a key inside the bucket's table: the high word of a Fibonacci multiply, range-reduced by a multiply-high —
    // independent of the bucket (top bits of splitmix64)
    auto home = [&](long long key) -> uint32_t {
        if (K32) return __umulhi((uint32_t)key * 0x9e3779b1u, LINES); // (a 32-bit remainder: one 32-bit multiply instead of a 64-bit one)
        return __umulhi((uint32_t)(((uint64_t)key * 0x9e3779b97f4a7c15ULL) >> 32), LINES);
    };
    // insert-or-get: the four keys of a line are read and compared at once; 0xffffffff when the table is too full
    const int tag_bits = TAG ? G.kc_bits - G.nb_log2 - G.tag_idx_bits : 0; // (tag_idx_bits: log2 of the table's lines)
    const uint32_t tag_mask = (1u << tag_bits) - 1u;
    constexpr uint32_t TAG_PENDING = 0xffeu, TAG_OV = 256u; // ids <= 0xffd; an empty entry is all ones (a valid one has bit 31 clear)
    uint32_t *const s_ov = s_misc + 18; // TAG: [0] keys in the overflow list, [1] its lock, [2 .. 2 + TAG_OV) their remainders, then their ids
    // the slow road of a record whose line did not show its key: look again (atomically), claim an empty entry of the line, or — line
    // full of other keys — find / append the key in the overflow list.  0xffffffff: out of accumulator sets or of list entries.
    auto slot_of_tag = [&](uint32_t rem) -> uint32_t {
        uint32_t *const line = (uint32_t *)t_key + 4u * (rem >> tag_bits);
        const uint32_t want = (rem & tag_mask) << 12;
        auto new_id = [&]() -> uint32_t {
            const uint32_t id = __hip_atomic_fetch_add(&s_misc[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return id >= SLOTS ? 0xffffffffu : id;
        };
        for (;;) {
            int empty = -1;
            bool pending = false;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                const uint32_t e = __hip_atomic_load(line + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t y = e ^ want;
                if (y < TAG_PENDING) return y;
                if (y == TAG_PENDING) pending = true; // its owner publishes the id within its own trip
                if (e == 0xffffffffu) empty = j;
            }
            if (pending) continue;
            if (empty >= 0) {
                if (atomicCAS(line + empty, 0xffffffffu, want | TAG_PENDING) != 0xffffffffu) continue; // somebody took the entry: look again
                const uint32_t id = new_id(); // (out of ids: the entry is published all the same — nobody spins on it — and the call retried with more buckets)
                __hip_atomic_store(line + empty, want | (id == 0xffffffffu ? 0u : id), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                return id;
            }
            // the line is full of other keys: the overflow list (append-only; read without the lock, appended under it)
            uint32_t n = __hip_atomic_load(&s_ov[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (uint32_t i = 0; i < n; ++i) if (__hip_atomic_load(&s_ov[2 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == rem) return __hip_atomic_load(&s_ov[2 + TAG_OV + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (atomicCAS(&s_ov[1], 0u, 1u) != 0u) continue; // (the holder finishes inside its own trip of this loop)
            uint32_t id = 0xffffffffu;
            const uint32_t n2 = __hip_atomic_load(&s_ov[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (uint32_t i = n; i < n2; ++i) if (__hip_atomic_load(&s_ov[2 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == rem) id = __hip_atomic_load(&s_ov[2 + TAG_OV + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (id == 0xffffffffu && n2 < TAG_OV) {
                id = new_id();
                if (id != 0xffffffffu) {
                    __hip_atomic_store(&s_ov[2 + n2], rem, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&s_ov[2 + TAG_OV + n2], id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(&s_ov[0], n2 + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __hip_atomic_store(&s_ov[1], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            return id;
        }
    };
    auto slot_of = [&](long long key) -> uint32_t {
        if (DIRECT) return (uint32_t)key; // (the remainder: < 2^(kc_bits - nb_log2) = SLOTS by construction)
        if (key == GB_EMPTY) return SLOTS;
        uint32_t l = home(key);
        const KT k = (KT)key, empty = KEMPTY;
        for (uint32_t trips = 0; trips < 2u * LINES; ++trips) {
            KT k0, k1, k2, k3;
            if (K32) {
                const uint4 a = *(const uint4 *)(t_key + 4u * l);
                k0 = (KT)a.x; k1 = (KT)a.y; k2 = (KT)a.z; k3 = (KT)a.w;
            } else {
                const ulonglong2 a = *(const ulonglong2 *)(t_key + 4u * l), b = *(const ulonglong2 *)(t_key + 4u * l + 2);
                k0 = (KT)a.x; k1 = (KT)a.y; k2 = (KT)b.x; k3 = (KT)b.y;
            }
            if (k0 == k) return 4u * l;
            if (k1 == k) return 4u * l + 1;
            if (k2 == k) return 4u * l + 2;
            if (k3 == k) return 4u * l + 3;
            const int e = k0 == empty ? 0 : (k1 == empty ? 1 : (k2 == empty ? 2 : (k3 == empty ? 3 : -1)));
            if (e < 0) { l = l + 1 == LINES ? 0u : l + 1; continue; } // a full line without the key: it can only be further on
            if (s_misc[0] >= limit) return 0xffffffffu;
            const KT old = atomicCAS(&t_key[4u * l + e], empty, k);
            if (old == empty) {
                __hip_atomic_fetch_add(&s_misc[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return 4u * l + e;
            }
            if (old == k) return 4u * l + e;
            // somebody else's key took the word: look at the same line again
        }
        return 0xffffffffu;
    };
    // A wave streams whole blocks of this bucket — the primary block of scatter workgroup w, w + waves, ..., then the spare blocks
    // this bucket owns — U records per lane per trip.  Round 4: the trips are software-pipelined ACROSS the blocks: the next trip's
    // records (of this block or of the wave's next non-empty one) are requested before the current trip's are probed and added, so a
    // wave's loads are in flight under its own LDS work (before: load, wait, work — 12-16 GB read at 2.3-3 TB/s with the LDS idle
    // during the waits).
    const uint32_t NB = 1u << G.nb_log2;
    // the bucket's records: its stream of every set, walked in blocks of RB records dealt to the waves round robin (wave w: block w,
    // w + waves, ... of the concatenated streams)
    constexpr uint32_t RB_LOG2 = 11, RB = 1u << RB_LOG2;
    uint32_t it_s = 0;      // (wave-uniform) stream = set the iterator is in
    uint64_t it_k = wave;   // ... and the wave's next block, counted from the start of that stream
    uint64_t cur_lo = 0;
    uint32_t cur_fill = 0;
    auto next_block = [&]() -> bool {
        for (;;) {
            if (it_s >= G.ng) return false;
            const size_t sid = (size_t)it_s * NB + bucket;
            const unsigned long long reserved = G.gcount[sid];
            const unsigned long long s_first = G.gstart ? G.gstart[sid] : (unsigned long long)sid * G.cap;
            const unsigned long long s_room = G.gstart ? G.gstart[sid + 1] - s_first : G.cap;
            const uint64_t fill = reserved < s_room ? reserved : s_room;
            const uint64_t nblk = (fill + RB - 1) >> RB_LOG2;
            if (it_k < nblk) {
                const uint64_t first = it_k << RB_LOG2;
                cur_lo = s_first + first;
                cur_fill = (uint32_t)(fill - first < RB ? fill - first : RB);
                it_k += nwave;
                return true;
            }
            it_k -= nblk; // (the wave's stride carries over into the next stream)
            ++it_s;
        }
    };
    struct Trip { // what a trip's loads return, untouched: any arithmetic on it here would make the wave wait for the loads at once
        uint64_t dv[U];       // split records behind a direct table: the value ...
        uint16_t di[U];       // ... and the remainder
        gb_u32x3 c[U];        // compact 12-byte records
        uint4 q[U];           // 16-byte records
        long long kk[U];      // separate arrays (several payload words)
        uint64_t pp[U][PW];
        uint32_t fill, j0;
    };
    auto request = [&](Trip &t, uint64_t lo, uint32_t fill, uint32_t j0) {
        t.fill = fill;
        t.j0 = j0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t j = j0 + 64u * u + lane;
            const uint64_t at = lo + (j < fill ? j : 0u);
            if (DIRECT) { t.dv[u] = ((const uint64_t *)G.qkey)[at]; t.di[u] = ((const uint16_t *)G.qw[0])[at]; }
            else if (PW == 1 && KC) t.c[u] = *(const gb_u32x3_a4 *)((const uint32_t *)G.qrec + at * 3);
            else if (PW == 1) t.q[u] = G.qrec[at];
            else {
                t.kk[u] = G.qkey[at];
#pragma unroll
                for (int w = 0; w < PW; ++w) t.pp[u][w] = G.qw[w][at];
            }
        }
    };
    // (Round 6, tried and removed: the U records of a trip probed TOGETHER — all home lines read first, then compared, then accumulated, the
    //  rare miss through slot_of — so that a lane's probe reads do not queue behind the previous record's atomics: 3.62 -> 4.07 ms per 1e9
    //  records.  The kernel is bound by the LDS's throughput, not its latency: three 64-bit atomics per record are ~1.6 ms per CU at the
    //  rates of profiles/r01_microbench_v3_lds_atomics.txt — the direct table's 1.9 ms — and the probe's 16-byte read and compares about as much again.)
    auto work = [&](const Trip &t) {
        if (TAG) { // straight-line probes; the misses of the trip settled together afterwards
            uint32_t miss = 0u;
            uint4 ee[U]; // the U lines are read up front: a read behind the previous record's atomics would wait for them (one lgkm counter, in order)
#pragma unroll
            for (int u = 0; u < U; ++u) ee[u] = *(const uint4 *)((const uint32_t *)t_key + 4u * (t.c[u][0] >> tag_bits));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t rem = t.c[u][0];
                const uint64_t p0 = (uint64_t)t.c[u][1] | ((uint64_t)t.c[u][2] << 32);
                const uint32_t want = (rem & tag_mask) << 12;
                const uint4 e = ee[u];
                const uint32_t a = e.x ^ want, b = e.y ^ want, c = e.z ^ want, d = e.w ^ want;
                const uint32_t m = min(min(a, b), min(c, d)); // the key's entry xors to its id, every other one (and an empty one) to >= 2^12
                const bool valid = t.j0 + 64u * u + lane < t.fill;
                const double v = __longlong_as_double((long long)p0);
                // the three atomics are issued by every lane, branch-free (the compiler then counts them: the next record's line is waited
                // for with lgkmcnt(5), not behind these) — a lane without a hit adds nothing to a spare accumulator set.  A NaN value takes
                // the slow road too (its row counts, its value does not).
                const bool hit = valid && m < TAG_PENDING && v == v;
                const uint32_t id = hit ? m : SLOTS;
                __hip_atomic_fetch_add(&t_rows32[id], hit ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&t_sum[id], hit ? v : -0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // (-0.0: x + -0.0 == x for every x)
                __hip_atomic_fetch_add(&t_sum2[id], hit ? v * v : -0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                miss |= (valid && !hit ? 1u : 0u) << u;
            }
            while (miss) {
                const int u = __builtin_ctz(miss);
                miss &= miss - 1u;
                uint32_t rem = 0u;
                uint64_t p[PW];
#pragma unroll
                for (int q = 0; q < U; ++q) if (q == u) { rem = t.c[q][0]; p[0] = (uint64_t)t.c[q][1] | ((uint64_t)t.c[q][2] << 32); }
                const uint32_t sl = slot_of_tag(rem);
                if (sl == 0xffffffffu) failed = true;
                else accumulate(sl, p);
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t.j0 + 64u * u + lane >= t.fill) continue;
            long long key;
            uint64_t p[PW];
            if (DIRECT) {
                key = (long long)t.di[u];
                p[0] = t.dv[u];
            } else if (PW == 1 && KC) { // {remainder, payload}: the table is keyed by the remainder
                key = (long long)(uint64_t)t.c[u][0];
                p[0] = (uint64_t)t.c[u][1] | ((uint64_t)t.c[u][2] << 32);
            } else if (PW == 1) {
                key = (long long)((uint64_t)t.q[u].x | ((uint64_t)t.q[u].y << 32));
                p[0] = (uint64_t)t.q[u].z | ((uint64_t)t.q[u].w << 32);
            } else {
                key = t.kk[u];
#pragma unroll
                for (int w = 0; w < PW; ++w) p[w] = t.pp[u][w];
            }
            const uint32_t sl = slot_of(key);
            if (sl == 0xffffffffu) failed = true;
            else accumulate(sl, p);
        }
    };
    { // (the request behind the last trip is issued all the same, with fill = 0 at a valid address: a conditional request would leave
      //  two paths with different numbers of loads outstanding in front of work(), and the compiler's wait-count pass then waits for
      //  the NEW trip's loads too — seen in the ISA as vmcnt(3..0) instead of vmcnt(7..4))
        Trip a, b;
        bool have = next_block();
        uint32_t j0 = 0;
        request(a, cur_lo, have ? cur_fill : 0u, 0u);
        while (have) {
            j0 += 64u * U;
            bool more = true;
            if (j0 >= cur_fill) { more = next_block(); j0 = 0; }
            request(b, cur_lo, more ? cur_fill : 0u, j0);
            work(a);
            if (!more) break;
            j0 += 64u * U;
            if (j0 >= cur_fill) { have = next_block(); j0 = 0; }
            request(a, cur_lo, have ? cur_fill : 0u, j0);
            work(b);
        }
    }
    if (failed) atomicExch(G.overflow, 2u);
    __syncthreads();

    // compact the occupied slots into the result arrays: ONE device atomic per bucket reserves the range
    uint32_t mine = 0;
    auto rows_of = [&](uint32_t s) -> unsigned long long { return MERGE ? (unsigned long long)t_rows[s] : (unsigned long long)t_rows32[s]; };
    auto cnt_of = [&](int v, uint32_t s) -> unsigned long long { return (MERGE || v > 0) ? (unsigned long long)t_cnt[(size_t)v * EP + s] : (unsigned long long)(t_rows32[s] - t_nan32[s]); };
    // TAG: the groups are the table's entries and the overflow list's (an entry's place and tag are its key's remainder, its id the accumulators)
    const uint32_t EO = TAG ? TS + s_ov[0] : E;
    auto tag_entry = [&](uint32_t s, uint32_t &rem) -> uint32_t { // -> id, or 0xffffffff for an empty entry
        if (s >= TS) { rem = s_ov[2 + (s - TS)]; return s_ov[2 + TAG_OV + (s - TS)]; }
        const uint32_t e = ((const uint32_t *)t_key)[s];
        rem = ((s >> 2) << tag_bits) | (e >> 12);
        return e == 0xffffffffu ? 0xffffffffu : (e & 0xfffu);
    };
    for (uint32_t s = tid; s < EO; s += blockDim.x) {
        if (TAG) { uint32_t rem; mine += tag_entry(s, rem) != 0xffffffffu ? 1u : 0u; }
        else mine += rows_of(s) != 0ull ? 1u : 0u;
    }
    uint32_t inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)inc, o, 64);
        if ((int)lane >= o) inc += t;
    }
    if (lane == 63) s_misc[2 + wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w2 = 0; w2 < nwave; ++w2) {
        if (w2 < wave) before += s_misc[2 + w2];
        total += s_misc[2 + w2];
    }
    if (tid == 0) {
        const unsigned long long base = atomicAdd(G.out_count, (unsigned long long)total);
        if (base + total > G.out_cap) { atomicExch(G.overflow, 3u); s_misc[1] = 0xffffffffu; }
        else s_misc[1] = (uint32_t)base;
    }
    __syncthreads();
    if (s_misc[1] == 0xffffffffu) return;
    uint64_t o = (uint64_t)s_misc[1] + before + inc - mine;
    for (uint32_t s0 = tid; s0 < EO; s0 += blockDim.x) {
        uint32_t s = s0, rem = 0u;
        if (TAG) { s = tag_entry(s0, rem); if (s == 0xffffffffu) continue; }
        else if (rows_of(s) == 0ull) continue;
        if (!MERGE && NV == 1 && KC) // the group's key from its bucket and remainder, mixed back
            G.out_key[o] = (long long)(gb_kc_unmix(((uint64_t)bucket << (G.kc_bits - G.nb_log2)) | (DIRECT ? (uint64_t)s : (TAG ? (uint64_t)rem : (uint64_t)t_key[s])), G.kc_bits) + (uint64_t)G.kc_min);
        else
            G.out_key[o] = s == SLOTS ? GB_EMPTY : (long long)t_key[s];
        G.out_w[0][o] = (uint64_t)rows_of(s);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            G.out_w[1 + 3 * v][o] = (uint64_t)cnt_of(v, s);
            G.out_w[2 + 3 * v][o] = (uint64_t)__double_as_longlong(t_sum[(size_t)v * EP + s]);
            G.out_w[3 + 3 * v][o] = (uint64_t)__double_as_longlong(t_sum2[(size_t)v * EP + s]);
        }
        ++o;
    }
}

// every `step`-th key of the column as the int64 the reference's ordered_set<T> would hold (vxh_sample_heavy_keys)
__global__ void gb_sample_keys(const void *keys, int dt, uint64_t step, uint32_t m, long long *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = gb_fix_key(gb_load_key(keys, (uint64_t)i * step, dt), dt);
}

__global__ void gb_iota(unsigned int *p, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = (unsigned int)i;
}

__global__ void gb_gather(const uint64_t *src, const unsigned int *perm, uint64_t *dst, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[perm[i]];
}

// derived columns on the sorted results: the finishers of vaex/agg.py:403-416, :440-455
__global__ void gb_derive(const uint64_t *cnt, const uint64_t *sum, const uint64_t *sum2, int which, uint64_t *dst, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const double c = (double)(long long)cnt[i], s = __longlong_as_double((long long)sum[i]), s2 = __longlong_as_double((long long)sum2[i]);
        const double mean = s / c;
        const double raw2 = s2 / c;
        const double var = raw2 - mean * mean;
        const double r = which == VXH_GB_MEAN ? mean : (which == VXH_GB_VAR ? var : sqrt(var));
        dst[i] = (uint64_t)__double_as_longlong(r);
    }
}

unsigned gb_grid(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(b, 4096));
}

struct Dev {
    void *p = nullptr;
    size_t cap = 0;
    // (from the library's block pool: a groupby result's columns are 40 MB per 1e6 groups, allocated per call — hipMalloc /
    //  hipFree there were host-blocking gaps in the middle of the pipeline)
    void need(size_t bytes) {
        if (bytes <= cap) return;
        if (p) vxh_pool_free(p);
        p = nullptr;
        cap = 0;
        p = vxh_pool_alloc(bytes);
        cap = bytes;
    }
    ~Dev() { if (p) vxh_pool_free(p); }
};

} // namespace

// result of one groupby: sorted by key, columns on the device until fetched
struct vxh_groupby {
    int nv = 1;
    uint64_t n_groups = 0;
    Dev cols; // [key | rows | (count, sum, sum2) x nv] x n_groups, 8-byte elements, sorted by key
    Dev tmp;
    uint64_t stride = 0; // elements between columns
    int buckets = 0, slots = 0, retries = 0, compact = 0, heavy = 0, direct = 0, tag = 0;
    float ms_scatter = 0, ms_reduce = 0, ms_sort = 0;
};

namespace {

// process-wide scratch of the pipeline (grow-only; one groupby at a time per process: guarded by the mutex)
struct GbScratch {
    std::mutex mutex;
    Dev queues, small, out, sort_tmp, stage;
};
GbScratch &gb_scratch() {
    static GbScratch *s = new GbScratch();
    return *s;
}

constexpr size_t GB_LDS_MAX = 160 * 1024;
constexpr int GB_DIRECT_BITS = 12; // direct tables: 4096 slots x (rc 8 + sum 8 + sum2 8 + the unused key word 8) = 128 KiB

template <int W, int R>
size_t scatter_lds(int nb_log2) {
    const size_t T = 1024u * R;
    return T * 8 * (1 + W) + ((size_t)1 << nb_log2) * 4 * 5 + 64 + T * 2 + 16;
}
template <int W, int R, bool K64, bool KEEP, bool HEAVY>
void launch_scatter_as(const GbArgs &G, int blocks, size_t lds, hipStream_t st) {
    if (W == 1 && G.kc_bits && G.direct) { // compact records behind a direct table: 8 + 2 bytes in two streams
        constexpr bool KC = W == 1;
        HIP_CHECK(hipFuncSetAttribute((const void *)gb_scatter<W, R, K64, KEEP, HEAVY, KC, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gb_scatter<W, R, K64, KEEP, HEAVY, KC, KC>), dim3(blocks), dim3(1024), lds, st, G);
        return;
    }
    if (W == 1 && G.kc_bits) { // compact records: the KC instantiation (constexpr-guarded: the template only exists for W == 1)
        constexpr bool KC = W == 1;
        HIP_CHECK(hipFuncSetAttribute((const void *)gb_scatter<W, R, K64, KEEP, HEAVY, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gb_scatter<W, R, K64, KEEP, HEAVY, KC>), dim3(blocks), dim3(1024), lds, st, G);
        return;
    }
    HIP_CHECK(hipFuncSetAttribute((const void *)gb_scatter<W, R, K64, KEEP, HEAVY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gb_scatter<W, R, K64, KEEP, HEAVY>), dim3(blocks), dim3(1024), lds, st, G);
}
template <int W, int R>
void launch_scatter(const GbArgs &G, int blocks, hipStream_t st) {
    constexpr bool CAN_PEEL = W <= 2; // (RAW records of one or two value columns; merges carry partial results, not rows)
    const bool heavy = CAN_PEEL && G.n_heavy > 0 && !G.merge;
    const size_t lds = scatter_lds<W, R>(G.nb_log2) + (heavy ? gb_heavy_lds(W) : 0);
    if (lds > GB_LDS_MAX) throw std::runtime_error("groupby: internal: gb_scatter staging exceeds the LDS");
    const bool k64 = G.key_dtype == VXH_I64 || G.key_dtype == VXH_U64, keep = G.keep != nullptr || G.pred.on;
    if (heavy) {
        if (k64) { if (keep) launch_scatter_as<W, R, true, true, CAN_PEEL>(G, blocks, lds, st); else launch_scatter_as<W, R, true, false, CAN_PEEL>(G, blocks, lds, st); }
        else { if (keep) launch_scatter_as<W, R, false, true, CAN_PEEL>(G, blocks, lds, st); else launch_scatter_as<W, R, false, false, CAN_PEEL>(G, blocks, lds, st); }
        return;
    }
    if (k64) { if (keep) launch_scatter_as<W, R, true, true, false>(G, blocks, lds, st); else launch_scatter_as<W, R, true, false, false>(G, blocks, lds, st); }
    else { if (keep) launch_scatter_as<W, R, false, true, false>(G, blocks, lds, st); else launch_scatter_as<W, R, false, false, false>(G, blocks, lds, st); }
}

// bytes of LDS per table slot, and the number of 4-key lines that fit
inline size_t reduce_slot_bytes(int nv, bool merge) { return 8 + 16 * (size_t)nv + (merge ? (size_t)8 * (1 + nv) : (size_t)8 + (size_t)4 * (nv - 1)); }
inline uint32_t reduce_lines(int nv, bool merge) { return (uint32_t)((GB_LDS_MAX - 18 * 4 - 64) / reduce_slot_bytes(nv, merge) / 4 - 1); }

template <int NV, bool MERGE>
void launch_reduce(const GbArgs &G, hipStream_t st) {
    const size_t EP = (4 * (size_t)G.lines + 1 + 3) & ~(size_t)3;
    const size_t lds = EP * reduce_slot_bytes(NV, MERGE) + 18 * 4 + 16;
    if (lds > GB_LDS_MAX) throw std::runtime_error("groupby: internal: gb_reduce table exceeds the LDS");
    if (NV == 1 && !MERGE && G.kc_bits && G.direct) { // compact records whose remainder indexes the table (run_pipeline decides)
        const size_t slots = (size_t)1 << (G.kc_bits - G.nb_log2), ep = (slots + 3) & ~(size_t)3;
        const size_t dlds = ep * reduce_slot_bytes(1, false) + 18 * 4 + 16;
        if (dlds > GB_LDS_MAX) throw std::runtime_error("groupby: internal: gb_reduce direct table exceeds the LDS");
        HIP_CHECK(hipFuncSetAttribute((const void *)gb_reduce<1, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds));
        hipLaunchKernelGGL((gb_reduce<1, false, true, true>), dim3(1u << G.nb_log2), dim3(1024), dlds, st, G);
        return;
    }
    if (NV == 1 && !MERGE && G.kc_bits && G.tag) { // compact records with a remainder of < 32 bits: the tag table (straight-line probe of one 16-byte line)
        const size_t tlds = ((size_t)16 << G.tag_idx_bits) + ((size_t)G.ids + 4) * 24 + (18 + 2 + 2 * 256) * 4 + 16;
        if (tlds > GB_LDS_MAX) throw std::runtime_error("groupby: internal: gb_reduce tag table exceeds the LDS");
        HIP_CHECK(hipFuncSetAttribute((const void *)gb_reduce<1, false, true, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tlds));
        hipLaunchKernelGGL((gb_reduce<1, false, true, false, true, true>), dim3(1u << G.nb_log2), dim3(1024), tlds, st, G);
        return;
    }
    if (NV == 1 && !MERGE && G.kc_bits && G.key32) { // compact records with a remainder of < 32 bits: 32-bit table keys (4 bytes less per slot)
        const size_t klds = EP * (reduce_slot_bytes(1, false) - 4) + 18 * 4 + 16;
        HIP_CHECK(hipFuncSetAttribute((const void *)gb_reduce<1, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)klds));
        hipLaunchKernelGGL((gb_reduce<1, false, true, false, true>), dim3(1u << G.nb_log2), dim3(1024), klds, st, G);
        return;
    }
    if (NV == 1 && !MERGE && G.kc_bits) { // compact records (run_pipeline sets kc_bits for this form only)
        HIP_CHECK(hipFuncSetAttribute((const void *)gb_reduce<1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gb_reduce<1, false, true>), dim3(1u << G.nb_log2), dim3(1024), lds, st, G);
        return;
    }
    HIP_CHECK(hipFuncSetAttribute((const void *)gb_reduce<NV, MERGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((gb_reduce<NV, MERGE>), dim3(1u << G.nb_log2), dim3(1024), lds, st, G);
}

// one pipeline run over device-resident records; results appended (unsorted) to the arrays in G.out_*; returns the
// overflow code (0 = fine)

// key_bits > 0: every key lies in [key_min, key_min + 2^key_bits) (the caller measured the range): compact 12-byte records where the
// remainder below the bucket bits fits 32 bits (GbArgs::kc_*)
unsigned run_pipeline(GbArgs &G, int nv, bool merge, uint64_t n, uint64_t groups_hint, int cus, hipStream_t st, vxh_groupby *res, bool hint_is_a_count = false,
                      long long key_min = 0, int key_bits = 0, const long long *heavy_host = nullptr, int n_heavy = 0) {
    GbScratch &S = gb_scratch();
    const int w = merge ? 1 + 3 * nv : nv;
    // rows per thread per tile.  One payload word: 8 (128 KiB of staging, one workgroup per CU).  Measured per 1e9 rows
    // (profiles/r02_groupby.txt): 8 rows / 1 workgroup per CU 9.8 ms; 3 rows / 2 workgroups per CU 13.6 ms — the three
    // barriers per tile cost more than the overlap of two workgroups' phases buys when a thread has only 3 rows between them.
    const int R = w == 1 ? 8 : (w == 2 ? 4 : (w <= 4 ? 2 : 1));
    const uint64_t T = 1024ull * R;
    const int per_cu = 1;
    // LDS table of a bucket: as many 4-key lines as the 160 KiB hold — 5112 slots x (key 8 + sum 8 + sum2 8 + rows 4 + count 4)
    // with one value column, 2916 with two, 4092 / 2556 when merging (64-bit counters)
    const uint32_t lines = reduce_lines(nv, merge);
    // one payload word: 8 rows per thread per tile (128 KiB of staging) as long as the bucket tables fit beside it
    const int nb_max = 10;
    int nb_log2 = 6;
    // target load of a bucket's table ("gb_load_pct", default 50 % of the slots: 512 buckets for the 2^20 guess or for 1e6 known groups);
    // a bucket that overflows anyway (80 % of its slots taken) costs a retry with four times the buckets
    const int64_t load_pct = ctx().cfg_gb_load_pct;
    const uint64_t per_bucket = std::max<uint64_t>(64, (uint64_t)lines * 4 * (uint64_t)std::min<int64_t>(95, std::max<int64_t>(10, load_pct)) / 100);
    while (nb_log2 < nb_max && ((uint64_t)1 << nb_log2) * per_bucket < std::max<uint64_t>(groups_hint, 1)) nb_log2++;
    // (Round 4 sized the buckets for a KNOWN group count at 76 % mean load — 256 buckets instead of 512 for 1e6 groups: the scatter gained
    //  what microbench5 promised, 8.25 -> 7.2 ms per 1e9 rows, gb_reduce went from 3.8 to 20 ms — a wave's probe costs what its LONGEST
    //  chain of 64 costs.  profiles/r04_groupby_compact.txt; removed in round 5: tables stay at <= 50 % of their slots.)
    struct Events { // (destroyed on every way out, a throwing launch included)
        hipEvent_t e[3] = {nullptr, nullptr, nullptr};
        Events() { for (auto &x : e) HIP_CHECK(hipEventCreate(&x)); }
        ~Events() { for (auto &x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    hipEvent_t e0 = ev.e[0], e1 = ev.e[1], e2 = ev.e[2];
    unsigned code = 0;
    uint64_t slack = 1;
    std::vector<unsigned long long> exact; // stream starts measured by a first attempt that found a stream too small (then: one exact second attempt)
    for (int attempt = 0; attempt < 6; ++attempt) {
        // a key range of at most 2^(10 + GB_DIRECT_BITS) cells: as many buckets as leave a remainder the LDS can index directly ("gb_direct", default on)
        // (a direct table has no load factor: the bucket count follows from the range alone — "gb_direct_nb" buckets at least, as few as the
        //  4096-slot tables allow: fewer streams grow longer segments per tile in gb_scatter)
        if (ctx().cfg_gb_compact && ctx().cfg_gb_direct && w == 1 && !merge && key_bits > 6 && key_bits <= nb_max + GB_DIRECT_BITS) {
            const int floor_nb = (int)std::min<int64_t>(nb_max, std::max<int64_t>(6, ctx().cfg_gb_direct_nb));
            nb_log2 = std::min(std::max(floor_nb, key_bits - GB_DIRECT_BITS), key_bits - 1);
            nb_log2 = std::max(nb_log2, std::min(nb_max, key_bits - GB_DIRECT_BITS));
        }
        const uint64_t NB = (uint64_t)1 << nb_log2;
        const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((n + T - 1) / T, (uint64_t)cus * per_cu));
        // sets of streams: one per XCD ("gb_sets", 8: blockIdx % 8 is the XCD a workgroup lands on), fewer when the launch has fewer
        // workgroups.  A stream's room: its expected share of the rows + 1/4 + a few tiles' worth; four times as much on retry
        const uint64_t NG = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::max<int64_t>(1, ctx().cfg_gb_sets), (uint64_t)blocks));
        uint64_t cap = (uint64_t)((double)n / (double)(NG * NB) * 1.25) + 8 * (T / NB + 1) + 1024;
        cap = std::min<uint64_t>(((cap + 3) & ~(uint64_t)3) * slack, ((n + 3) & ~(uint64_t)3) + 64); // (a stream never needs more room than every row)
        const uint64_t total_records = exact.empty() ? NG * NB * cap : (uint64_t)exact.back();
        if (total_records * 8 * (uint64_t)(1 + w) > (96ull << 30)) { code = 9; break; }
        G.nv = nv; G.w = w; G.merge = merge ? 1 : 0; G.n = n;
        G.nb_log2 = nb_log2; G.slots_log2 = 0; G.lines = lines;
        const bool compact = ctx().cfg_gb_compact && w == 1 && !merge && key_bits > nb_log2 && key_bits - nb_log2 <= 32;
        G.kc_bits = compact ? key_bits : 0;
        G.direct = compact && ctx().cfg_gb_direct && key_bits - nb_log2 <= GB_DIRECT_BITS ? 1 : 0;
        G.key32 = compact && !G.direct && ctx().cfg_gb_key32 && key_bits - nb_log2 < 32 ? 1 : 0;
        // the tag table ("gb_tag", default on): 2^12 lines of four entries (fewer when the remainder is shorter), a 256-key overflow list, and as
        // many accumulator sets (24 bytes) as the rest of the LDS holds (4000 behind 2^12 lines: a bucket is sized for <= 2554 distinct keys)
        G.tag = G.key32 && ctx().cfg_gb_tag ? 1 : 0;
        G.tag_idx_bits = std::min(12, key_bits - nb_log2);
        G.ids = (uint32_t)std::min<size_t>(4092, (((GB_LDS_MAX - (18 + 2 + 2 * 256) * 4 - 64 - ((size_t)16 << G.tag_idx_bits)) / 24) & ~(size_t)3) - 4);
#ifdef VXH_ABLATE
        G.abl = (int32_t)ctx().cfg_gb_abl;
#endif
        G.kc_min = key_min;
        if (res) { res->compact = compact ? 1 : 0; res->direct = G.direct; }
        G.ng = (uint32_t)NG; G.cap = cap;
        // (round 6, late: the streams are sized for the records this attempt writes — 12 bytes for compact records (10 behind a direct table: the same room, so that
        //  a dense and a scattered call of one size share it), 8 (1 + w) otherwise — plus 1/16, rounded up to 2 GB: 18 GB instead of 24.5 per 1e9 rows, and calls
        //  whose stream counts differ a little do not make the grow-only scratch grow again — every growth is a hipFree + hipMalloc of gigabytes)
        {
            size_t qbytes = total_records * (compact ? (size_t)12 : 8 * (size_t)(1 + w)) + 64;
            qbytes += qbytes / 16;
            if (qbytes > ((size_t)1 << 30)) qbytes = (qbytes + ((size_t)2 << 30) - 1) & ~(((size_t)2 << 30) - 1);
            S.queues.need(qbytes);
        }
        const size_t starts_bytes = (NG * NB + 1) * 8;
        const size_t tabs_bytes = (NG * NB * 8 + 64 + starts_bytes + 15) & ~(size_t)15; // stream counters | flags | stream starts (exact layout)
        const bool peel = n_heavy > 0 && !merge && nv <= 2;
        const size_t heavy_bytes = peel ? (size_t)n_heavy * 8 * (size_t)(1 + 1 + 3 * nv) : 0; // keys | accumulators
        const size_t small_bytes = tabs_bytes + heavy_bytes;
        S.small.need(small_bytes);
        char *q = (char *)S.queues.p;
        G.qkey = (long long *)q;
        G.qrec = (uint4 *)q; // (w == 1: 16-byte / 12-byte records in the same space)
        for (int k = 0; k < w; k++) G.qw[k] = (uint64_t *)(q + total_records * 8 * (size_t)(1 + k));
        G.gcount = (unsigned long long *)S.small.p;
        G.overflow = (unsigned int *)(G.gcount + NG * NB); // [0] code
        G.out_count = (unsigned long long *)(G.overflow + 2); // [2..3]
        HIP_CHECK(hipMemsetAsync(S.small.p, 0, small_bytes, st));
        G.gstart = nullptr;
        if (!exact.empty()) {
            G.gstart = (const unsigned long long *)((char *)S.small.p + NG * NB * 8 + 64);
            HIP_CHECK(hipMemcpyAsync((void *)G.gstart, exact.data(), exact.size() * 8, hipMemcpyHostToDevice, st));
        }
        G.n_heavy = 0; G.heavy_keys = nullptr; G.heavy_acc = nullptr;
        if (peel) {
            G.n_heavy = n_heavy;
            G.heavy_keys = (const long long *)((char *)S.small.p + tabs_bytes);
            G.heavy_acc = (unsigned long long *)((char *)S.small.p + tabs_bytes + (size_t)n_heavy * 8);
            HIP_CHECK(hipMemcpyAsync((void *)G.heavy_keys, heavy_host, (size_t)n_heavy * 8, hipMemcpyHostToDevice, st)); // (the caller's array outlives the call's stream wait)
        }
        HIP_CHECK(hipEventRecord(e0, st));
        if (w == 1 && scatter_lds<1, 8>(nb_log2) <= GB_LDS_MAX) launch_scatter<1, 8>(G, blocks, st);
        else if (w == 1) launch_scatter<1, 4>(G, blocks, st); // (1024 buckets: their tables leave room for 4096-row tiles)
        else if (w == 2) launch_scatter<2, 4>(G, blocks, st);
        else if (w == 4) launch_scatter<4, 2>(G, blocks, st);
        else launch_scatter<7, 1>(G, blocks, st);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipEventRecord(e1, st));
        if (nv == 1) { if (merge) launch_reduce<1, true>(G, st); else launch_reduce<1, false>(G, st); }
        else { if (merge) launch_reduce<2, true>(G, st); else launch_reduce<2, false>(G, st); }
        HIP_CHECK(hipGetLastError());
        if (peel) { // the heavy keys' accumulators become groups like any other
            hipLaunchKernelGGL(gb_append_heavy, dim3(1), dim3(GB_HEAVY_MAX), 0, st, G);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipEventRecord(e2, st));
        unsigned int flags[4] = {0, 0, 0, 0};
        HIP_CHECK(hipMemcpyAsync(flags, G.overflow, 16, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        code = flags[0];
        if (res) {
            (void)hipEventElapsedTime(&res->ms_scatter, e0, e1);
            (void)hipEventElapsedTime(&res->ms_reduce, e1, e2);
            res->buckets = (int)NB; res->slots = G.tag ? (int)G.ids : (int)(4 * lines); res->retries = attempt; res->tag = G.tag;
        }
        if (code == 0) {
            unsigned long long cnt = 0;
            memcpy(&cnt, &flags[2], 8);
            if (res) res->n_groups = cnt;
            break;
        }
        if (code == 2) { // a bucket holds too many distinct keys: more buckets
            if (nb_log2 >= 10) { code = 8; break; }
            nb_log2 = std::min(10, nb_log2 + 2);
            exact.clear(); // (measured for the old bucket count)
        } else if (code == 3) { // result arrays too small (the caller sized them from a hint): report
            break;
        }
        if (code == 1) {
            // a stream passed its room (few or skewed keys).  The counters kept counting: they hold every stream's exact length, and
            // the second attempt deals the same tiles to the same workgroups — lay the streams out from them (4-record aligned)
            if (!exact.empty()) { code = 7; break; } // (cannot happen: the layout is exact)
            std::vector<unsigned long long> counts(NG * NB);
            HIP_CHECK(hipMemcpy(counts.data(), G.gcount, counts.size() * 8, hipMemcpyDeviceToHost));
            exact.assign(counts.size() + 1, 0ull);
            for (size_t i = 0; i < counts.size(); i++) exact[i + 1] = exact[i] + ((counts[i] + 3) & ~3ull);
            (void)slack;
        }
    }
    return code;
}

} // namespace

extern "C" {

#define GB_BEGIN try {
#define GB_END                                                                                                         \
    }                                                                                                                  \
    catch (const std::exception &e) {                                                                                  \
        vxh_set_error(e.what());                                                                                       \
        return 1;                                                                                                      \
    }                                                                                                                  \
    return 0;

int vxh_groupby_run(int key_dtype, const void *keys, int n_values, const void *const *values, uint64_t n, int mem, uint64_t groups_hint, uint64_t max_groups, vxh_groupby **out) {
    return vxh_groupby_run_kept(key_dtype, keys, n_values, values, nullptr, n, mem, groups_hint, max_groups, out);
}

int vxh_groupby_run_kept(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem, uint64_t groups_hint, uint64_t max_groups, vxh_groupby **out) {
    return vxh_groupby_run_ranged(key_dtype, keys, n_values, values, keep, n, mem, groups_hint, max_groups, 1, 0, out);
}

int vxh_groupby_run_ranged(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem, uint64_t groups_hint, uint64_t max_groups,
                           int64_t key_min, int64_t key_max, vxh_groupby **out) {
    return vxh_groupby_run_peeled(key_dtype, keys, n_values, values, keep, n, mem, groups_hint, max_groups, key_min, key_max, nullptr, 0, out);
}

int vxh_groupby_run_peeled(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem, uint64_t groups_hint, uint64_t max_groups,
                           int64_t key_min, int64_t key_max, const int64_t *heavy_keys, int n_heavy, vxh_groupby **out) {
    return vxh_groupby_run_selected(key_dtype, keys, n_values, values, keep, n, mem, groups_hint, max_groups, key_min, key_max, heavy_keys, n_heavy, 0, nullptr, 0, out);
}

int vxh_groupby_run_selected(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem, uint64_t groups_hint, uint64_t max_groups,
                             int64_t key_min, int64_t key_max, const int64_t *heavy_keys, int n_heavy, int n_terms, const vxh_groupby_term *terms, uint32_t truth, vxh_groupby **out) {
    GB_BEGIN
    // the filter's terms over the call's own value columns (gb_scatter evaluates them on the payload words of the rows it loads)
    PredDesc pred{};
    if (n_terms < 0 || n_terms > 4 || (n_terms > 0 && !terms)) throw std::runtime_error("groupby: 0 to 4 selection terms");
    for (int t = 0; t < n_terms; t++) {
        static const uint32_t rel_code[6] = {/*LT*/ 1u, /*LE*/ 3u, /*GT*/ 4u, /*GE*/ 6u, /*EQ*/ 2u, /*NE*/ 13u}; // bits: 0 less, 1 equal, 2 greater, 3 unordered (as vxh_grid_bin's fused selections)
        if (terms[t].op < VXH_CMP_LT || terms[t].op > VXH_CMP_NE) throw std::runtime_error("groupby: unknown comparison in a selection term");
        if (terms[t].value_index < 0 || terms[t].value_index >= n_values || terms[t].value_index > 1) throw std::runtime_error("groupby: a selection term reads one of the call's (first two) value columns");
        pred.code[t] = rel_code[terms[t].op - VXH_CMP_LT];
        pred.op[t] = terms[t].op;
        pred.c[t] = terms[t].constant;
        pred.tcol[t] = (uint8_t)terms[t].value_index;
    }
    pred.on = n_terms > 0;
    pred.nterms = n_terms;
    pred.truth = truth;
    // the heavy keys as the pass wants them: distinct, none the table's EMPTY marker, at most GB_HEAVY_MAX (more: the first ones — a
    // heavy key that is not peeled only costs time)
    std::vector<long long> heavy;
    for (int i = 0; i < n_heavy && heavy.size() < GB_HEAVY_MAX; i++) {
        const long long k = (long long)heavy_keys[i];
        if (k != GB_EMPTY && std::find(heavy.begin(), heavy.end(), k) == heavy.end()) heavy.push_back(k);
    }
    int key_bits = 0; // bits of key_max - key_min (0: range unknown)
    if (key_min <= key_max) {
        const uint64_t span = (uint64_t)key_max - (uint64_t)key_min;
        key_bits = 1;
        while (key_bits < 64 && (span >> key_bits)) key_bits++;
    }
    if (key_dtype == VXH_F64 || key_dtype == VXH_F32 || key_dtype < 0 || key_dtype >= VXH_DTYPE_COUNT) throw std::runtime_error("groupby: integer key dtypes only");
    if (n_values < 1 || n_values > GB_MAX_NV) throw std::runtime_error("groupby: 1 or 2 float64 value columns");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); throw std::runtime_error("vaex_hip: no HIP device available (libvaexhip has no CPU fallback)"); }
    (void)hipSetDevice(ctx().device);
    int64_t cus = 256;
    (void)vxh_config_get("cus", &cus);
    Slot &slot = get_slot(0);
    GbScratch &S = gb_scratch();
    std::lock_guard<std::mutex> lock(S.mutex);
    std::unique_ptr<vxh_groupby> res(new vxh_groupby());
    res->nv = n_values;
    if (n == 0) { *out = res.release(); return 0; }
    const size_t ks = (size_t)vxh_dtype_size(key_dtype);
    GbArgs G{};
    G.key_dtype = key_dtype;
    G.pred = pred;
    if (mem == VXH_MEM_DEVICE) {
        order_after_producers(slot);
        G.keys = keys;
        G.keep = keep;
        for (int v = 0; v < n_values; v++) G.payload[v] = (const uint64_t *)values[v];
    } else {
        S.stage.need(n * (ks + 8 * (size_t)n_values + (keep ? 1 : 0)) + 256 * 4);
        char *p = (char *)S.stage.p;
        HIP_CHECK(hipMemcpyAsync(p, keys, n * ks, hipMemcpyHostToDevice, slot.stream));
        G.keys = p;
        p += (n * ks + 255) & ~(size_t)255;
        for (int v = 0; v < n_values; v++) {
            HIP_CHECK(hipMemcpyAsync(p, values[v], n * 8, hipMemcpyHostToDevice, slot.stream));
            G.payload[v] = (const uint64_t *)p;
            p += (n * 8 + 255) & ~(size_t)255;
        }
        if (keep) {
            HIP_CHECK(hipMemcpyAsync(p, keep, n, hipMemcpyHostToDevice, slot.stream));
            G.keep = (const uint8_t *)p;
        }
    }
    if (max_groups == 0) max_groups = std::min<uint64_t>(n, 1ull << 26);
    const uint64_t out_cap = std::min<uint64_t>(n, max_groups) + 1 + heavy.size();
    const int wout = 1 + 3 * n_values;
    S.out.need(out_cap * 8 * (size_t)(1 + wout));
    G.out_cap = out_cap;
    G.out_key = (long long *)S.out.p;
    for (int k = 0; k < wout; k++) G.out_w[k] = (uint64_t *)((char *)S.out.p + out_cap * 8 * (size_t)(1 + k));
    const unsigned code = run_pipeline(G, n_values, false, n, groups_hint ? groups_hint : (1u << 20), (int)cus, slot.stream, res.get(), groups_hint != 0, (long long)key_min, key_bits, heavy.data(), (int)heavy.size());
    res->heavy = (int)heavy.size();
    if (code == 3) throw std::runtime_error("groupby: more groups than max_groups");
    if (code == 8 || code == 9) throw std::runtime_error("groupby: too many distinct keys for the LDS-partitioned path");
    if (code != 0) throw std::runtime_error("groupby: the key distribution is too skewed for the partitioned path");
    // sort by key: radix sort of (key, position), then gather every column
    const uint64_t ng = res->n_groups;
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipEventRecord(e0, slot.stream));
    res->stride = (ng + 31) & ~(uint64_t)31;
    res->cols.need(std::max<uint64_t>(res->stride, 32) * 8 * (size_t)(1 + wout));
    if (ng) {
        res->tmp.need(ng * 4 * 2 + ng * 8 + 256);
        unsigned int *iota = (unsigned int *)res->tmp.p, *perm = iota + ng;
        long long *keys_sorted = (long long *)res->cols.p;
        hipLaunchKernelGGL(gb_iota, dim3(gb_grid(ng)), dim3(256), 0, slot.stream, iota, ng);
        size_t tmp_bytes = 0;
        // (a known range of non-negative keys: the bits above the largest key are zero in every key — the radix sort takes the digits below only.  1e6 groups
        //  of 40-bit keys: 5 passes instead of 8)
        int end_bit = 64;
        if (key_min <= key_max && key_min >= 0) {
            end_bit = 1;
            while (end_bit < 64 && ((uint64_t)key_max >> end_bit)) end_bit++;
        }
        HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, G.out_key, keys_sorted, iota, perm, ng, 0, end_bit, slot.stream));
        S.sort_tmp.need(tmp_bytes + 16);
        HIP_CHECK(rocprim::radix_sort_pairs(S.sort_tmp.p, tmp_bytes, G.out_key, keys_sorted, iota, perm, ng, 0, end_bit, slot.stream));
        for (int k = 0; k < wout; k++)
            hipLaunchKernelGGL(gb_gather, dim3(gb_grid(ng)), dim3(256), 0, slot.stream, G.out_w[k], perm, (uint64_t *)res->cols.p + res->stride * (size_t)(1 + k), ng);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipEventRecord(e1, slot.stream));
    vxh_timer_lap(slot);
    HIP_CHECK(hipStreamSynchronize(slot.stream));
    (void)hipEventElapsedTime(&res->ms_sort, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *out = res.release();
    GB_END
}

// The heavy hitters of a key column from a strided sample (Frame._heavy_keys: every key holding >= min_count of `sample` rows taken every
// n / sample rows — the `max_keys` most frequent of them, ties by key, ascending in the output): gather + rocPRIM radix sort + run-length encode on
// the device, the few runs that reach the threshold picked on the host.  (Until round 6 this was torch.unique over the same sample: torch's sort
// kernels are loaded on first use, ~100 ms of a process's first groupby, profiles/r06_process_first.txt — and the one step of a groupby that
// ran outside this library.)
int vxh_sample_heavy_keys(int key_dtype, const void *keys, uint64_t n, uint32_t sample, uint32_t min_count, int max_keys, int64_t *out_keys, int *n_out) {
    GB_BEGIN
    *n_out = 0;
    if (key_dtype == VXH_F64 || key_dtype == VXH_F32 || key_dtype < 0 || key_dtype >= VXH_DTYPE_COUNT) throw std::runtime_error("groupby: integer key dtypes only");
    if (!n || !sample || max_keys <= 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); throw std::runtime_error("vaex_hip: no HIP device available (libvaexhip has no CPU fallback)"); }
    (void)hipSetDevice(ctx().device);
    Slot &slot = get_slot(0);
    GbScratch &S = gb_scratch();
    std::lock_guard<std::mutex> lock(S.mutex);
    order_after_producers(slot);
    const uint64_t step = std::max<uint64_t>(1, n / sample);
    const uint32_t m = (uint32_t)std::min<uint64_t>(sample, (n + step - 1) / step); // = len(key[::step][:sample])
    size_t sort_bytes = 0, rle_bytes = 0;
    long long *raw = nullptr, *sorted = nullptr, *uniq = nullptr;
    unsigned int *counts = nullptr, *runs = nullptr;
    HIP_CHECK(rocprim::radix_sort_keys(nullptr, sort_bytes, raw, sorted, m, 0, 64, slot.stream));
    HIP_CHECK(rocprim::run_length_encode(nullptr, rle_bytes, sorted, m, uniq, counts, runs, slot.stream));
    const size_t tmp_bytes = (std::max(sort_bytes, rle_bytes) + 255) & ~(size_t)255;
    S.small.need((size_t)m * (8 * 3 + 4) + 256 + tmp_bytes + 256);
    char *p = (char *)S.small.p;
    raw = (long long *)p; p += (size_t)m * 8;
    sorted = (long long *)p; p += (size_t)m * 8;
    uniq = (long long *)p; p += (size_t)m * 8;
    counts = (unsigned int *)p; p += ((size_t)m * 4 + 255) & ~(size_t)255;
    runs = (unsigned int *)p; p += 256;
    void *tmp = p;
    hipLaunchKernelGGL(gb_sample_keys, dim3((m + 255) / 256), dim3(256), 0, slot.stream, keys, key_dtype, step, m, raw);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(rocprim::radix_sort_keys(tmp, sort_bytes, raw, sorted, m, 0, 64, slot.stream));
    HIP_CHECK(rocprim::run_length_encode(tmp, rle_bytes, sorted, m, uniq, counts, runs, slot.stream));
    unsigned int n_runs = 0;
    HIP_CHECK(hipMemcpyAsync(&n_runs, runs, 4, hipMemcpyDeviceToHost, slot.stream));
    HIP_CHECK(hipStreamSynchronize(slot.stream));
    std::vector<long long> hk(n_runs);
    std::vector<unsigned int> hc(n_runs);
    if (n_runs) {
        HIP_CHECK(hipMemcpyAsync(hk.data(), uniq, (size_t)n_runs * 8, hipMemcpyDeviceToHost, slot.stream));
        HIP_CHECK(hipMemcpyAsync(hc.data(), counts, (size_t)n_runs * 4, hipMemcpyDeviceToHost, slot.stream));
        HIP_CHECK(hipStreamSynchronize(slot.stream));
    }
    // (rocPRIM sorts int64 as signed: the runs come in ascending key order — the order of numpy.unique / torch.unique)
    std::vector<uint32_t> pick;
    for (uint32_t i = 0; i < n_runs; i++) if (hc[i] >= min_count) pick.push_back(i);
    std::stable_sort(pick.begin(), pick.end(), [&](uint32_t a, uint32_t b) { return hc[a] > hc[b]; }); // the most frequent first, ties by key
    if (pick.size() > (size_t)max_keys) pick.resize((size_t)max_keys);
    std::sort(pick.begin(), pick.end());
    for (size_t i = 0; i < pick.size(); i++) out_keys[i] = (int64_t)hk[pick[i]];
    *n_out = (int)pick.size();
    GB_END
}

int vxh_groupby_merge(int n_values, const int64_t *keys, const int64_t *rows, const int64_t *const *counts, const double *const *sums, const double *const *sums2, uint64_t n, uint64_t groups_hint, vxh_groupby **out) {
    GB_BEGIN
    if (n_values < 1 || n_values > GB_MAX_NV) throw std::runtime_error("groupby: 1 or 2 value columns");
    (void)hipSetDevice(ctx().device);
    int64_t cus = 256;
    (void)vxh_config_get("cus", &cus);
    Slot &slot = get_slot(0);
    GbScratch &S = gb_scratch();
    std::lock_guard<std::mutex> lock(S.mutex);
    std::unique_ptr<vxh_groupby> res(new vxh_groupby());
    res->nv = n_values;
    if (n == 0) { *out = res.release(); return 0; }
    const int w = 1 + 3 * n_values;
    const size_t col = (n * 8 + 255) & ~(size_t)255;
    S.stage.need(col * (size_t)(1 + w));
    char *p = (char *)S.stage.p;
    GbArgs G{};
    G.key_dtype = VXH_I64;
    HIP_CHECK(hipMemcpyAsync(p, keys, n * 8, hipMemcpyHostToDevice, slot.stream));
    G.keys = p;
    const void *src[GB_MAX_W];
    src[0] = rows;
    for (int v = 0; v < n_values; v++) { src[1 + 3 * v] = counts[v]; src[2 + 3 * v] = sums[v]; src[3 + 3 * v] = sums2[v]; }
    for (int k = 0; k < w; k++) {
        char *d = p + col * (size_t)(1 + k);
        HIP_CHECK(hipMemcpyAsync(d, src[k], n * 8, hipMemcpyHostToDevice, slot.stream));
        G.payload[k] = (const uint64_t *)d;
    }
    const uint64_t out_cap = n + 1;
    S.out.need(out_cap * 8 * (size_t)(1 + w));
    G.out_cap = out_cap;
    G.out_key = (long long *)S.out.p;
    for (int k = 0; k < w; k++) G.out_w[k] = (uint64_t *)((char *)S.out.p + out_cap * 8 * (size_t)(1 + k));
    const unsigned code = run_pipeline(G, n_values, true, n, groups_hint ? groups_hint : n, (int)cus, slot.stream, res.get());
    if (code != 0) throw std::runtime_error("groupby merge: too many distinct keys / too skewed for the LDS-partitioned path");
    const uint64_t ng = res->n_groups;
    res->stride = (ng + 31) & ~(uint64_t)31;
    res->cols.need(std::max<uint64_t>(res->stride, 32) * 8 * (size_t)(1 + w));
    if (ng) {
        res->tmp.need(ng * 4 * 2 + ng * 8 + 256);
        unsigned int *iota = (unsigned int *)res->tmp.p, *perm = iota + ng;
        long long *keys_sorted = (long long *)res->cols.p;
        hipLaunchKernelGGL(gb_iota, dim3(gb_grid(ng)), dim3(256), 0, slot.stream, iota, ng);
        size_t tmp_bytes = 0;
        HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, G.out_key, keys_sorted, iota, perm, ng, 0, 64, slot.stream));
        S.sort_tmp.need(tmp_bytes + 16);
        HIP_CHECK(rocprim::radix_sort_pairs(S.sort_tmp.p, tmp_bytes, G.out_key, keys_sorted, iota, perm, ng, 0, 64, slot.stream));
        for (int k = 0; k < w; k++)
            hipLaunchKernelGGL(gb_gather, dim3(gb_grid(ng)), dim3(256), 0, slot.stream, G.out_w[k], perm, (uint64_t *)res->cols.p + res->stride * (size_t)(1 + k), ng);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipStreamSynchronize(slot.stream));
    *out = res.release();
    GB_END
}

void vxh_groupby_destroy(vxh_groupby *g) {
    if (!g) return;
    (void)hipDeviceSynchronize();
    delete g;
}

uint64_t vxh_groupby_size(const vxh_groupby *g) { return g->n_groups; }

int vxh_groupby_info(const vxh_groupby *g, int what, double *value_out) {
    GB_BEGIN
    switch (what) {
    case 0: *value_out = g->buckets; break;
    case 1: *value_out = g->slots; break;
    case 2: *value_out = g->retries; break;
    case 3: *value_out = g->ms_scatter; break;
    case 4: *value_out = g->ms_reduce; break;
    case 5: *value_out = g->ms_sort; break;
    case 6: *value_out = g->compact; break;
    case 7: *value_out = g->heavy; break;
    case 8: *value_out = g->direct; break;
    case 9: *value_out = g->tag; break;
    default: throw std::runtime_error("groupby info: unknown item");
    }
    GB_END
}

int vxh_groupby_column(vxh_groupby *g, int value_index, int which, void *out_host) {
    GB_BEGIN
    (void)hipSetDevice(ctx().device);
    if (g->n_groups == 0) return 0;
    Slot &slot = get_slot(0);
    const uint64_t ng = g->n_groups;
    const uint64_t *base = (const uint64_t *)g->cols.p;
    const uint64_t *col = nullptr;
    if (which == VXH_GB_KEYS) col = base;
    else if (which == VXH_GB_ROWS) col = base + g->stride;
    else {
        if (value_index < 0 || value_index >= g->nv) throw std::runtime_error("groupby column: value index out of range");
        const uint64_t *cnt = base + g->stride * (size_t)(2 + 3 * value_index), *sum = cnt + g->stride, *sum2 = sum + g->stride;
        if (which == VXH_GB_COUNT) col = cnt;
        else if (which == VXH_GB_SUM) col = sum;
        else if (which == VXH_GB_SUM2) col = sum2;
        else if (which == VXH_GB_MEAN || which == VXH_GB_VAR || which == VXH_GB_STD) {
            g->tmp.need(ng * 8 + ng * 8 + 256 + ng * 8);
            uint64_t *dst = (uint64_t *)g->tmp.p;
            hipLaunchKernelGGL(gb_derive, dim3(gb_grid(ng)), dim3(256), 0, slot.stream, cnt, sum, sum2, which, dst, ng);
            HIP_CHECK(hipGetLastError());
            col = dst; // (no vxh_timer_lap: a few microseconds of kernel between result columns that are already crossing PCIe)
        } else throw std::runtime_error("groupby column: unknown column");
    }
    HIP_CHECK(hipMemcpyAsync(out_host, col, ng * 8, hipMemcpyDeviceToHost, slot.stream));
    HIP_CHECK(hipStreamSynchronize(slot.stream));
    GB_END
}

} // extern "C"

void vxh_preload_groupby(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)gb_append_heavy);
    (void)hipGetLastError();
}

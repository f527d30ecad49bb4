// HIP kernels (gfx950 / CDNA4) for vaex's binned-statistics hot path.
//
//   K1  bin_kernel<STRAT, FAST>  fused  [bin index of every dim] -> [every aggregator's scatter op]
//                                restates Grid::bin_ (src/agg.hpp:106-137) + BinnerScalar::to_bins
//                                (src/binners.cpp:13-57) + BinnerOrdinal::to_bins
//                                (src/binner_ordinal.cpp:20-176) + the aggregate() loops of
//                                src/agg_count.cpp:43-67, src/agg_sum.cpp:98-127, src/agg_minmax.cpp:44-74
//   K4  fold_kernel              replica fold = get_result()'s fold over thread grids
//                                (src/agg_count.cpp:24-41, agg_sum.cpp:80-97, agg_minmax.cpp:27-43)
//   K5  minmax_kernel            legacy statisticNd OP_MIN_MAX on a 0-d grid (src/vaexfast.cpp:1090-1101)
//
// The work is a bandwidth-bound gather/scatter: no MFMA.  Rows are read coalesced (lane i ->
// row base+i), the flat cell index is computed in fp64 with exactly the reference's operation
// order (sub, mul, compare, mul, cvt; compiled with -ffp-contract=off), and the scatter-add goes
//   LDS    : to a workgroup-private copy of the grids in LDS (ds_add_u32 / ds_add_f64 ...),
//            flushed once per workgroup with device-scope atomics         (grids that fit LDS)
//   XCC    : straight to a per-XCD replica in HBM with workgroup-scope (L2-resident) atomics:
//            every workgroup only ever touches the replica of the XCD it runs on, so the RMW
//            stays in that XCD's 4 MiB L2 instead of going to the memory-side atomic unit
//   GLOBAL : straight to replica blockIdx % R with device-scope atomics
// and the replicas are folded by K4 when the result is asked for.
#include "vxh_kernels.hpp"

#include <string.h>

#include <algorithm>

namespace {

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u; // 8 XCDs on MI355X
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) { // src/hash.hpp:40-45
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

// raw little-endian element widened to 64 bits, byte-swapped when the column is non-native
__device__ __forceinline__ uint64_t load_raw(const void *p, uint64_t i, int dt, int flip) {
    switch (dt) {
    case VXH_F64: case VXH_I64: case VXH_U64: {
        uint64_t u = ((const uint64_t *)p)[i];
        return flip ? __builtin_bswap64(u) : u;
    }
    case VXH_F32: case VXH_I32: case VXH_U32: {
        uint32_t u = ((const uint32_t *)p)[i];
        return flip ? __builtin_bswap32(u) : u;
    }
    case VXH_I16: case VXH_U16: {
        uint16_t u = ((const uint16_t *)p)[i];
        return flip ? __builtin_bswap16(u) : u;
    }
    default:
        return ((const uint8_t *)p)[i];
    }
}

// `double value_double = value;` (src/binners.cpp:24)
__device__ __forceinline__ double raw_as_f64(uint64_t u, int dt) {
    switch (dt) {
    case VXH_F64: return __longlong_as_double((long long)u);
    case VXH_F32: return (double)__uint_as_float((uint32_t)u);
    case VXH_I64: return (double)(int64_t)u;
    case VXH_I32: return (double)(int32_t)(uint32_t)u;
    case VXH_I16: return (double)(int16_t)(uint16_t)u;
    case VXH_I8: return (double)(int8_t)(uint8_t)u;
    case VXH_U64: return (double)u;
    case VXH_U32: return (double)(uint32_t)u;
    case VXH_U16: return (double)(uint16_t)u;
    case VXH_U8: return (double)(uint8_t)u;
    default: return u ? 1.0 : 0.0; // bool
    }
}

__device__ __forceinline__ int64_t raw_as_i64(uint64_t u, int dt) {
    switch (dt) {
    case VXH_I32: return (int32_t)(uint32_t)u;
    case VXH_I16: return (int16_t)(uint16_t)u;
    case VXH_I8: return (int8_t)(uint8_t)u;
    case VXH_BOOL: return u ? 1 : 0;
    default: return (int64_t)u; // i64/u64 as bits, u32/u16/u8 zero-extended by load_raw
    }
}

__device__ __forceinline__ bool dt_is_float(int dt) { return dt == VXH_F64 || dt == VXH_F32; }

// x86-64 cvttsd2si semantics ("integer indefinite" for NaN / out of range): what the reference's
// `int64_t value = data_ptr[i] - min_value` does for floating T on the machines it runs on
__device__ __forceinline__ int64_t f64_to_i64_x86(double d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)d;
}

// ------------------------------------------------------------------------------------------
// bin index of one dimension
// ------------------------------------------------------------------------------------------
// BinnerScalar — src/binners.cpp:16-35 (exact operation order; no FMA contraction possible/allowed)
__device__ __forceinline__ uint64_t scalar_sub_index(double v, bool masked, double vmin, double scale, double binsd, uint64_t bins) {
    double scaled = (v - vmin) * scale;
    uint64_t index = 0;
    if (scaled != scaled || masked) {
    } else if (scaled < 0) {
        index = 1;
    } else if (scaled >= 1) {
        index = bins + 2;
    } else {
        index = (uint64_t)(int64_t)((int)(scaled * binsd) + 2);
    }
    return index;
}

__device__ __forceinline__ uint64_t dim_sub_index(const BinnerDesc &b, uint64_t i) {
    bool masked = b.mask != nullptr && b.mask[i] == 1;
    if (b.kind == VXH_BIN_SCALAR) {
        double v = raw_as_f64(load_raw(b.data, i, b.dtype, b.flip), b.dtype);
        return scalar_sub_index(v, masked, b.vmin, b.scale, b.binsd, b.bins);
    } else if (b.kind == VXH_BIN_ORDINAL) {
        // src/binner_ordinal.cpp:138-175 (and the invert / allow_other variants :22-137).  The element is
        // NOT byte-swapped before the subtraction; the int64 difference is (reference behaviour, :25-28).
        uint64_t u = load_raw(b.data, i, b.dtype, 0);
        int64_t value;
        if (b.dtype == VXH_F64) value = f64_to_i64_x86(__longlong_as_double((long long)u) - (double)b.min_value);
        else if (b.dtype == VXH_F32) value = f64_to_i64_x86((double)(__uint_as_float((uint32_t)u) - (float)b.min_value));
        else value = (int64_t)((uint64_t)raw_as_i64(u, b.dtype) - (uint64_t)b.min_value);
        if (b.flip) value = (int64_t)__builtin_bswap64((uint64_t)value);
        int64_t N = (int64_t)b.bins;
        bool oob = value < 0 || value >= N;
        if (b.allow_other) {
            if (masked) return (uint64_t)N + 1;
            if (oob) return (uint64_t)N;
        } else {
            if (masked || oob) return (uint64_t)N;
        }
        return (uint64_t)(b.invert ? N - 1 - value : value);
    } else {
        // hash binner: cells [unknown, bin0..binN-1, null]
        if (masked) return (uint64_t)b.null_bin;
        uint64_t u = load_raw(b.data, i, b.dtype, b.flip);
        int64_t key = dt_is_float(b.dtype) ? (int64_t)u : raw_as_i64(u, b.dtype);
        uint64_t p = splitmix64((uint64_t)key) & b.hmask;
        for (;;) {
            int64_t ord = b.hvals[p];
            if (ord < 0) return 0;
            if (b.hkeys[p] == key) return (uint64_t)ord + 1;
            p = (p + 1) & b.hmask;
        }
    }
}

template <bool FAST>
__device__ __forceinline__ uint64_t flat_index(const BinArgs &A, uint64_t i) {
    uint64_t idx = 0;
    for (int d = 0; d < A.ndim; ++d) {
        const BinnerDesc &b = A.b[d];
        uint64_t sub;
        if (FAST) {
            double v = ((const double *)b.data)[i];
            sub = scalar_sub_index(v, false, b.vmin, b.scale, b.binsd, b.bins);
        } else {
            sub = dim_sub_index(b, i);
        }
        idx += sub * b.stride;
    }
    return idx;
}

// ------------------------------------------------------------------------------------------
// scatter ops.  SCOPE: __HIP_MEMORY_SCOPE_AGENT (device) or __HIP_MEMORY_SCOPE_WORKGROUP
// (for LDS, and for the XCC strategy where the RMW is performed by the XCD-local L2).
// ------------------------------------------------------------------------------------------
template <int SCOPE, typename T>
__device__ __forceinline__ void at_add(T *p, T v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE, typename T>
__device__ __forceinline__ void at_max(T *p, T v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE, typename T>
__device__ __forceinline__ void at_min(T *p, T v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, SCOPE); }

// pow(b, moment) for the small unsigned moments vaex uses (var/std: 2, skew: 3, kurtosis: 4).  The
// reference calls libm pow (src/agg_sum.cpp:159); repeated multiplication differs from it by < 1 ulp
// per term, far inside the 1e-12 fp64 tolerance of the sums.
__device__ __forceinline__ double pow_u(double b, uint32_t m) {
    double r = 1.0;
    for (uint32_t k = 0; k < m; ++k) r *= b;
    return r;
}

// one aggregator, one row.  `cellp` = address of the cell in the chosen grid copy.
// CT: "global" copy uses the device cell type; LDS copy uses u32 for counts.
template <int SCOPE, bool LDS>
__device__ __forceinline__ void agg_apply(const AggDesc &a, void *base, uint64_t idx, uint64_t raw, bool has_data) {
    switch (a.kind) {
    case VXH_AGG_COUNT:
        if (LDS) at_add<SCOPE, uint32_t>((uint32_t *)base + idx, 1u);
        else at_add<SCOPE, unsigned long long>((unsigned long long *)base + idx, 1ull);
        break;
    case VXH_AGG_SUM:
    case VXH_AGG_SUM_MOMENT:
        if (a.cell == VXH_CELL_F64) {
            double b = raw_as_f64(raw, a.dtype);
            if (a.kind == VXH_AGG_SUM_MOMENT) b = pow_u(b, a.moment);
            at_add<SCOPE, double>((double *)base + idx, b);
        } else {
            int64_t b = raw_as_i64(raw, a.dtype);
            if (a.kind == VXH_AGG_SUM_MOMENT) {
                double bd = (a.cell == VXH_CELL_U64) ? (double)(uint64_t)b : (double)b;
                double pw = pow_u(bd, a.moment);
                b = (a.cell == VXH_CELL_U64) ? (int64_t)(uint64_t)pw : (int64_t)pw;
            }
            at_add<SCOPE, unsigned long long>((unsigned long long *)base + idx, (unsigned long long)b);
        }
        break;
    case VXH_AGG_MIN:
    case VXH_AGG_MAX: {
        const bool mx = a.kind == VXH_AGG_MAX;
        switch (a.cell) {
        case VXH_CELL_F64: {
            double v = raw_as_f64(raw, a.dtype);
            if (mx) at_max<SCOPE, double>((double *)base + idx, v); else at_min<SCOPE, double>((double *)base + idx, v);
            break;
        }
        case VXH_CELL_F32: {
            float v = __uint_as_float((uint32_t)raw);
            if (mx) at_max<SCOPE, float>((float *)base + idx, v); else at_min<SCOPE, float>((float *)base + idx, v);
            break;
        }
        case VXH_CELL_I64: {
            long long v = (long long)raw_as_i64(raw, a.dtype);
            if (mx) at_max<SCOPE, long long>((long long *)base + idx, v); else at_min<SCOPE, long long>((long long *)base + idx, v);
            break;
        }
        case VXH_CELL_U64: {
            unsigned long long v = (unsigned long long)raw;
            if (mx) at_max<SCOPE, unsigned long long>((unsigned long long *)base + idx, v); else at_min<SCOPE, unsigned long long>((unsigned long long *)base + idx, v);
            break;
        }
        case VXH_CELL_I32: {
            int v = (int)raw_as_i64(raw, a.dtype);
            if (mx) at_max<SCOPE, int>((int *)base + idx, v); else at_min<SCOPE, int>((int *)base + idx, v);
            break;
        }
        default: {
            unsigned v = (unsigned)raw_as_i64(raw, a.dtype);
            if (mx) at_max<SCOPE, unsigned>((unsigned *)base + idx, v); else at_min<SCOPE, unsigned>((unsigned *)base + idx, v);
            break;
        }
        }
        break;
    }
    default: break;
    }
}

__device__ __forceinline__ size_t cell_size_dev(int cell) { return cell >= VXH_CELL_F32 ? 4 : 8; }
__device__ __forceinline__ size_t lds_cell_size_dev(int kind, int cell) { return kind == VXH_AGG_COUNT ? 4 : cell_size_dev(cell); }

__device__ __forceinline__ uint64_t identity_bits(int kind, int cell) {
    if (kind != VXH_AGG_MIN && kind != VXH_AGG_MAX) return 0;
    const bool mx = kind == VXH_AGG_MAX;
    switch (cell) {
    case VXH_CELL_F64: return mx ? 0xfff0000000000000ull : 0x7ff0000000000000ull;
    case VXH_CELL_F32: return mx ? 0xff800000u : 0x7f800000u;
    case VXH_CELL_I64: return mx ? 0x8000000000000000ull : 0x7fffffffffffffffull;
    case VXH_CELL_U64: return mx ? 0ull : ~0ull;
    case VXH_CELL_I32: return mx ? 0x80000000u : 0x7fffffffu;
    default: return mx ? 0u : 0xffffffffu;
    }
}

template <typename T>
__device__ __forceinline__ void flush_minmax(T *g, T v, bool mx, bool plain) {
    if (plain) {
        T cur = *g;
        *g = mx ? (v > cur ? v : cur) : (v < cur ? v : cur);
    } else if (mx) {
        at_max<__HIP_MEMORY_SCOPE_AGENT, T>(g, v);
    } else {
        at_min<__HIP_MEMORY_SCOPE_AGENT, T>(g, v);
    }
}

// LDS cell c of aggregator a -> HBM cell gc of the chosen replica
__device__ __forceinline__ void flush_cell(const AggDesc &a, char *lds_base, uint64_t c, char *g, uint64_t gc, bool plain) {
    switch (a.kind) {
    case VXH_AGG_COUNT: {
        uint32_t v = ((uint32_t *)lds_base)[c];
        if (v) {
            if (plain) ((unsigned long long *)g)[gc] += v;
            else at_add<__HIP_MEMORY_SCOPE_AGENT, unsigned long long>((unsigned long long *)g + gc, (unsigned long long)v);
        }
        break;
    }
    case VXH_AGG_SUM:
    case VXH_AGG_SUM_MOMENT:
        if (a.cell == VXH_CELL_F64) {
            double v = ((double *)lds_base)[c];
            if (v != 0.0) {
                if (plain) ((double *)g)[gc] += v;
                else at_add<__HIP_MEMORY_SCOPE_AGENT, double>((double *)g + gc, v);
            }
        } else {
            unsigned long long v = ((unsigned long long *)lds_base)[c];
            if (v) {
                if (plain) ((unsigned long long *)g)[gc] += v;
                else at_add<__HIP_MEMORY_SCOPE_AGENT, unsigned long long>((unsigned long long *)g + gc, v);
            }
        }
        break;
    default: {
        const bool mx = a.kind == VXH_AGG_MAX;
        switch (a.cell) {
        case VXH_CELL_F64: flush_minmax<double>((double *)g + gc, ((double *)lds_base)[c], mx, plain); break;
        case VXH_CELL_F32: flush_minmax<float>((float *)g + gc, ((float *)lds_base)[c], mx, plain); break;
        case VXH_CELL_I64: flush_minmax<long long>((long long *)g + gc, ((long long *)lds_base)[c], mx, plain); break;
        case VXH_CELL_U64: flush_minmax<unsigned long long>((unsigned long long *)g + gc, ((unsigned long long *)lds_base)[c], mx, plain); break;
        case VXH_CELL_I32: flush_minmax<int>((int *)g + gc, ((int *)lds_base)[c], mx, plain); break;
        default: flush_minmax<unsigned>((unsigned *)g + gc, ((unsigned *)lds_base)[c], mx, plain); break;
        }
    }
    }
}

// all aggregators of one row
template <int STRAT, bool FAST>
__device__ __forceinline__ void row_aggregate(const BinArgs &A, uint64_t i, uint64_t idx, uint64_t replica, char *lds, uint32_t slab) {
    constexpr bool LDS = STRAT == VXH_STRAT_LDS;
    constexpr int SCOPE = (STRAT == VXH_STRAT_GLOBAL) ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP;
    if (LDS) {
        // this workgroup owns the cells with (cell mod S) == slab; they live at LDS index cell / S
        if (((uint32_t)idx & ((1u << A.slab_log2) - 1u)) != slab) return;
        idx >>= A.slab_log2;
    }
    for (int k = 0; k < A.nagg; ++k) {
        const AggDesc &a = A.a[k];
        if (a.mask != nullptr && a.mask[i] != 1) continue; // aggregator mask: 1 = keep (src/agg_count.cpp:50)
        uint64_t raw = 0;
        if (a.data != nullptr) {
            if (FAST) {
                raw = ((const uint64_t *)a.data)[i];
                double v = __longlong_as_double((long long)raw);
                if (v != v) continue; // NaN rows are skipped (src/agg_sum.cpp:113, agg_count.cpp:56)
            } else {
                raw = load_raw(a.data, i, a.dtype, a.flip);
                if (dt_is_float(a.dtype)) {
                    double v = raw_as_f64(raw, a.dtype);
                    if (v != v) continue;
                }
            }
        }
        void *base;
        if (LDS) base = lds + a.lds_offset;
        else base = (char *)a.grid + replica * A.cells * cell_size_dev(a.cell);
        agg_apply<SCOPE, LDS>(a, base, idx, raw, a.data != nullptr);
    }
}

// ------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------
// LDS strategy geometry: gridDim.x = ngroups * S workgroups, S = 2^slab_log2 interleaved slabs.  The S
// workgroups of a group walk the SAME rows (each keeps only the cells of its slab), and are laid out so
// that — with the dispatcher's observed round-robin of workgroups over the 8 XCDs — they share an XCD
// and therefore its L2 (performance only; correctness never depends on placement):
//   xcd = b & 7, local = b >> 3, slab = local & (S-1), group = (local >> slab_log2) * 8 + xcd.
template <int STRAT, bool FAST>
__global__ void __launch_bounds__(1024) bin_kernel(const BinArgs A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool LDS = STRAT == VXH_STRAT_LDS;

    uint64_t replica;
    uint32_t slab = 0, group = blockIdx.x, ngroups = gridDim.x;
    uint64_t slab_cells = A.cells;
    if (LDS) {
        const uint32_t local = blockIdx.x >> 3;
        slab = local & ((1u << A.slab_log2) - 1u);
        group = (local >> A.slab_log2) * 8u + (blockIdx.x & 7u);
        ngroups = (uint32_t)A.ngroups;
        slab_cells = (A.cells + (1ull << A.slab_log2) - 1) >> A.slab_log2;
        replica = A.flush_plain ? group : group % (uint32_t)A.replicas;
    } else if (STRAT == VXH_STRAT_XCC) {
        replica = (uint64_t)xcc_id() * A.replicas_per_xcc + (blockIdx.x >> 3) % A.replicas_per_xcc;
    } else {
        replica = blockIdx.x % A.replicas;
    }

    if (LDS) {
        // identity-fill the private grids: 0 for counts/sums, the type's limit for min/max
        for (int k = 0; k < A.nagg; ++k) {
            const AggDesc &a = A.a[k];
            const size_t cs = lds_cell_size_dev(a.kind, a.cell);
            char *base = lds + a.lds_offset;
            const uint64_t ident = identity_bits(a.kind, a.cell);
            if (cs == 4) for (uint64_t c = threadIdx.x; c < slab_cells; c += blockDim.x) ((uint32_t *)base)[c] = (uint32_t)ident;
            else for (uint64_t c = threadIdx.x; c < slab_cells; c += blockDim.x) ((uint64_t *)base)[c] = ident;
        }
        __syncthreads();
    }

    const uint64_t stride = (uint64_t)ngroups * blockDim.x;
    uint64_t i = (uint64_t)group * blockDim.x + threadIdx.x;
    // 4 independent rows per thread per trip: 4x the loads in flight before the first dependent op
    for (; i + 3 * stride < A.n; i += 4 * stride) {
        uint64_t idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) idx[u] = flat_index<FAST>(A, i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) row_aggregate<STRAT, FAST>(A, i + u * stride, idx[u], replica, lds, slab);
    }
    for (; i < A.n; i += stride) {
        uint64_t idx = flat_index<FAST>(A, i);
        row_aggregate<STRAT, FAST>(A, i, idx, replica, lds, slab);
    }

    if (LDS) {
        __syncthreads();
        // flush the private slab into replica `replica` of the HBM grid: plain read-modify-write when this
        // workgroup is the replica's only writer (flush_plain), device-scope atomics otherwise
        for (int k = 0; k < A.nagg; ++k) {
            const AggDesc &a = A.a[k];
            char *base = lds + a.lds_offset;
            char *g = (char *)a.grid + replica * A.cells * cell_size_dev(a.cell);
            for (uint64_t c = threadIdx.x; c < slab_cells; c += blockDim.x) {
                const uint64_t gc = (c << A.slab_log2) + slab;
                if (gc >= A.cells) continue;
                flush_cell(a, base, c, g, gc, A.flush_plain != 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// K1b / K1c — partition strategy (see PartArgs)
// ------------------------------------------------------------------------------------------
// one aggregator, one record, given the record's mask flags and input values (already native-endian)
template <int SCOPE, bool LDS>
__device__ __forceinline__ void record_apply(const PartArgs &P, int k, void *base, uint64_t idx, uint32_t flags, const uint64_t *vals) {
    const AggDesc &a = P.A.a[k];
    const uint32_t mb = P.agg_mbit[k];
    if (mb != 0xffu && !((flags >> mb) & 1u)) return;
    uint64_t raw = 0;
    const uint32_t vs = P.agg_vslot[k];
    if (vs != 0xffu) {
        raw = vals[vs];
        if (dt_is_float(a.dtype)) {
            double v = raw_as_f64(raw, a.dtype);
            if (v != v) return; // NaN rows are skipped
        }
    }
    agg_apply<SCOPE, LDS>(a, base, idx, raw, vs != 0xffu);
}

// pass 1: rows -> per-slab record queues.  512 threads, R rows per thread per tile.
template <bool FAST, int R>
__global__ void __launch_bounds__(512) part_scatter(const PartArgs P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr uint32_t NONE = 0xffffffffu;
    constexpr unsigned long long OVERFLOW = ~0ull;
    const uint32_t S = 1u << P.slab_log2;
    const uint32_t T = 512u * R;
    // LDS carve (all offsets multiples of 16)
    uint32_t *s_cnt = (uint32_t *)lds;                                // [S]
    uint32_t *s_off = s_cnt + S;                                      // [S+1] (+pad)
    unsigned long long *s_gbase = (unsigned long long *)(s_off + S + 4); // [S]
    uint64_t *st_val = (uint64_t *)(s_gbase + S);                      // [nvals][T]
    uint32_t *st_idx = (uint32_t *)(st_val + (size_t)P.nvals * T);     // [T]
    uint16_t *st_slab = (uint16_t *)(st_idx + T);                      // [T]
    uint8_t *st_flags = (uint8_t *)(st_slab + T);                      // [T]
    const uint64_t n = P.A.n;

    for (uint64_t tile = blockIdx.x; tile * T < n; tile += gridDim.x) {
        const uint64_t base = tile * T;
        for (uint32_t s = threadIdx.x; s < S; s += 512) s_cnt[s] = 0;
        __syncthreads();

        uint32_t slab[R], loc[R], pos[R], fl[R];
        uint64_t val[R][VXH_PART_MAX_VALS];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t i = base + (uint64_t)r * 512 + threadIdx.x;
            pos[r] = NONE;
            slab[r] = 0; loc[r] = 0; fl[r] = 0;
            if (i < n) {
                uint32_t flags = 0;
                for (int m = 0; m < P.nmasks; ++m) flags |= (P.mdata[m][i] == 1 ? 1u : 0u) << m;
                const bool keep = !(P.all_masked && flags == 0);
                if (keep) {
                    const uint64_t idx = flat_index<FAST>(P.A, i);
                    slab[r] = (uint32_t)idx & (S - 1);
                    loc[r] = (uint32_t)(idx >> P.slab_log2);
                    fl[r] = flags;
#pragma unroll
                    for (int k = 0; k < VXH_PART_MAX_VALS; ++k)
                        if (k < P.nvals) val[r][k] = FAST ? ((const uint64_t *)P.vdata[k])[i] : load_raw(P.vdata[k], i, P.vdtype[k], P.vflip[k]);
                    pos[r] = __hip_atomic_fetch_add(&s_cnt[slab[r]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        // exclusive prefix over the S bucket counts + reservation of queue space (one HBM atomic per slab per tile)
        for (uint32_t s = threadIdx.x; s < S; s += 512) {
            uint32_t off = 0;
            for (uint32_t j = 0; j < s; ++j) off += s_cnt[j];
            s_off[s] = off;
            const uint32_t c = s_cnt[s];
            if (s == S - 1) s_off[S] = off + c;
            unsigned long long gb = 0;
            if (c) {
                gb = atomicAdd(&P.qcount[s], (unsigned long long)c);
                if (gb + c > P.cap) { // does not fit: remember where the valid prefix of the queue ends
                    atomicMin(&P.qlimit[s], gb);
                    gb = OVERFLOW;
                }
            }
            s_gbase[s] = gb;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (pos[r] != NONE) {
                const uint32_t j = s_off[slab[r]] + pos[r];
                st_idx[j] = loc[r];
                st_slab[j] = (uint16_t)slab[r];
                st_flags[j] = (uint8_t)fl[r];
#pragma unroll
                for (int k = 0; k < VXH_PART_MAX_VALS; ++k)
                    if (k < P.nvals) st_val[(size_t)k * T + j] = val[r][k];
            }
        }
        __syncthreads();
        const uint32_t total = s_off[S];
        for (uint32_t j = threadIdx.x; j < total; j += 512) {
            const uint32_t s = st_slab[j];
            const unsigned long long gb = s_gbase[s];
            if (gb != OVERFLOW) {
                const uint64_t dst = (uint64_t)s * P.cap + gb + (j - s_off[s]);
                if (P.idx16) ((uint16_t *)P.qidx)[dst] = (uint16_t)st_idx[j];
                else ((uint32_t *)P.qidx)[dst] = st_idx[j];
                if (P.use_flags) P.qflags[dst] = st_flags[j];
#pragma unroll
                for (int k = 0; k < VXH_PART_MAX_VALS; ++k)
                    if (k < P.nvals) P.qval[k][dst] = st_val[(size_t)k * T + j];
            } else {
                // queue full (pathologically skewed data): scatter this record straight to HBM
                const uint64_t gidx = ((uint64_t)st_idx[j] << P.slab_log2) + s;
                uint64_t vals[VXH_PART_MAX_VALS];
#pragma unroll
                for (int k = 0; k < VXH_PART_MAX_VALS; ++k) vals[k] = k < P.nvals ? st_val[(size_t)k * T + j] : 0;
                for (int k = 0; k < P.A.nagg; ++k) record_apply<__HIP_MEMORY_SCOPE_AGENT, false>(P, k, P.A.a[k].grid, gidx, st_flags[j], vals);
            }
        }
        __syncthreads();
    }
}

// pass 2: slab queues -> LDS-private slab -> HBM replica.
// The queue is streamed with 4 records per lane per load batch and two batches in flight (the LDS atomics
// retire at >1 record/clk/CU — profiles/r01_microbench_v3_lds_atomics.txt — so this pass is a pure stream
// and needs the memory-level parallelism of one).
struct RecBatch {
    uint32_t loc[4];
    uint32_t flags[4];
    uint64_t vals[VXH_PART_MAX_VALS][4];
};

__device__ __forceinline__ void rec_load4(const PartArgs &P, uint64_t at, RecBatch &b) {
    if (P.idx16) {
        const ushort4 q = *(const ushort4 *)((const uint16_t *)P.qidx + at);
        b.loc[0] = q.x; b.loc[1] = q.y; b.loc[2] = q.z; b.loc[3] = q.w;
    } else {
        const uint4 q = *(const uint4 *)((const uint32_t *)P.qidx + at);
        b.loc[0] = q.x; b.loc[1] = q.y; b.loc[2] = q.z; b.loc[3] = q.w;
    }
    if (P.use_flags) {
        const uchar4 f = *(const uchar4 *)(P.qflags + at);
        b.flags[0] = f.x; b.flags[1] = f.y; b.flags[2] = f.z; b.flags[3] = f.w;
    } else {
        b.flags[0] = b.flags[1] = b.flags[2] = b.flags[3] = 0xffu;
    }
#pragma unroll
    for (int k = 0; k < VXH_PART_MAX_VALS; ++k) {
        if (k < P.nvals) {
            const ulonglong2 a = *(const ulonglong2 *)(P.qval[k] + at), c = *(const ulonglong2 *)(P.qval[k] + at + 2);
            b.vals[k][0] = a.x; b.vals[k][1] = a.y; b.vals[k][2] = c.x; b.vals[k][3] = c.y;
        }
    }
}

__device__ __forceinline__ void rec_apply4(const PartArgs &P, const RecBatch &b, char *lds) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        uint64_t vals[VXH_PART_MAX_VALS];
#pragma unroll
        for (int k = 0; k < VXH_PART_MAX_VALS; ++k) vals[k] = k < P.nvals ? b.vals[k][u] : 0;
        for (int k = 0; k < P.A.nagg; ++k) record_apply<__HIP_MEMORY_SCOPE_WORKGROUP, true>(P, k, lds + P.A.a[k].lds_offset, b.loc[u], b.flags[u], vals);
    }
}

__global__ void __launch_bounds__(1024) part_reduce(const PartArgs P) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t S = 1u << P.slab_log2;
    const uint32_t slab = blockIdx.x % S, part = blockIdx.x / S;
    const uint64_t slab_cells = (P.A.cells + S - 1) >> P.slab_log2;
    for (int k = 0; k < P.A.nagg; ++k) {
        const AggDesc &a = P.A.a[k];
        const size_t cs = lds_cell_size_dev(a.kind, a.cell);
        char *base = lds + a.lds_offset;
        const uint64_t ident = identity_bits(a.kind, a.cell);
        if (cs == 4) for (uint64_t c = threadIdx.x; c < slab_cells; c += blockDim.x) ((uint32_t *)base)[c] = (uint32_t)ident;
        else for (uint64_t c = threadIdx.x; c < slab_cells; c += blockDim.x) ((uint64_t *)base)[c] = ident;
    }
    __syncthreads();
    unsigned long long len = P.qcount[slab];
    const unsigned long long lim = P.qlimit[slab];
    if (lim < len) len = lim;
    // this workgroup's share, cut at multiples of 4 records so the vector loads stay aligned
    const uint64_t quads = (len + 3) / 4;
    const uint64_t lo = quads * part / P.parts * 4, hi = std::min<uint64_t>(len, quads * (part + 1) / P.parts * 4);
    const uint64_t qb = (uint64_t)slab * P.cap;
    const uint64_t hi4 = lo + ((hi - lo) & ~(uint64_t)3);
    const uint64_t step = 4ull * blockDim.x;
    uint64_t j = lo + 4ull * threadIdx.x;
    for (; j + step < hi4; j += 2 * step) {
        RecBatch b0, b1;
        rec_load4(P, qb + j, b0);
        rec_load4(P, qb + j + step, b1);
        rec_apply4(P, b0, lds);
        rec_apply4(P, b1, lds);
    }
    for (; j < hi4; j += step) {
        RecBatch b0;
        rec_load4(P, qb + j, b0);
        rec_apply4(P, b0, lds);
    }
    // tail (< 4 records)
    for (uint64_t t = hi4 + threadIdx.x; t < hi; t += blockDim.x) {
        const uint32_t loc = P.idx16 ? ((const uint16_t *)P.qidx)[qb + t] : ((const uint32_t *)P.qidx)[qb + t];
        const uint32_t flags = P.use_flags ? P.qflags[qb + t] : 0xffu;
        uint64_t vals[VXH_PART_MAX_VALS];
#pragma unroll
        for (int k = 0; k < VXH_PART_MAX_VALS; ++k) vals[k] = k < P.nvals ? P.qval[k][qb + t] : 0;
        for (int k = 0; k < P.A.nagg; ++k) record_apply<__HIP_MEMORY_SCOPE_WORKGROUP, true>(P, k, lds + P.A.a[k].lds_offset, loc, flags, vals);
    }
    __syncthreads();
    const uint64_t replica = P.A.flush_plain ? part : part % (uint32_t)P.A.replicas;
    for (int k = 0; k < P.A.nagg; ++k) {
        const AggDesc &a = P.A.a[k];
        char *base = lds + a.lds_offset;
        char *g = (char *)a.grid + replica * P.A.cells * cell_size_dev(a.cell);
        for (uint64_t c = threadIdx.x; c < slab_cells; c += blockDim.x) {
            const uint64_t gc = (c << P.slab_log2) + slab;
            if (gc >= P.A.cells) continue;
            flush_cell(a, base, c, g, gc, P.A.flush_plain != 0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// fill / fold
// ------------------------------------------------------------------------------------------
__global__ void fill_kernel(void *dst, uint64_t n, int cs, uint64_t value) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    if (cs == 4) for (; i < n; i += stride) ((uint32_t *)dst)[i] = (uint32_t)value;
    else for (; i < n; i += stride) ((uint64_t *)dst)[i] = value;
}

template <typename T, int OP> // OP 0 add, 1 min, 2 max
__global__ void fold_kernel(T *grid, uint64_t cells, int replicas, T identity) {
    uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; c < cells; c += stride) {
        T acc = grid[c];
        for (int r = 1; r < replicas; ++r) {
            T v = grid[(uint64_t)r * cells + c];
            if (OP == 0) acc += v;
            else if (OP == 1) acc = v < acc ? v : acc;
            else acc = v > acc ? v : acc;
            grid[(uint64_t)r * cells + c] = identity;
        }
        grid[c] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// K5: min/max of one column (legacy statisticNd OP_MIN_MAX, 0-d grid): plain < / > so NaN never wins
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) minmax_kernel(int dtype, int flip, const void *data, const uint8_t *mask, uint64_t n, double *out2) {
    double mn = __longlong_as_double(0x7ff0000000000000ll), mx = __longlong_as_double((long long)0xfff0000000000000ull);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (mask != nullptr && mask[i] != 1) continue;
        double v = raw_as_f64(load_raw(data, i, dtype, flip), dtype);
        if (v < mn) mn = v;
        if (v > mx) mx = v;
    }
    // wave reduce (64 lanes), then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        double omn = __shfl_down(mn, off, 64), omx = __shfl_down(mx, off, 64);
        if (omn < mn) mn = omn;
        if (omx > mx) mx = omx;
    }
    if ((threadIdx.x & 63) == 0) {
        at_min<__HIP_MEMORY_SCOPE_AGENT, double>(out2, mn);
        at_max<__HIP_MEMORY_SCOPE_AGENT, double>(out2 + 1, mx);
    }
}

} // namespace

size_t vxh_cell_size(int cell) { return cell >= VXH_CELL_F32 ? 4 : 8; }
size_t vxh_lds_cell_size(int kind, int cell) { return kind == VXH_AGG_COUNT ? 4 : vxh_cell_size(cell); }

void vxh_launch_part(const PartArgs &args, const LaunchPlan &plan, int scatter_blocks, size_t scatter_lds, hipStream_t stream) {
    const int R = args.rows_per_thread;
#define VXH_SC(F, RR)                                                                                                  \
    {                                                                                                                  \
        if (scatter_lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)part_scatter<F, RR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)scatter_lds); \
        hipLaunchKernelGGL((part_scatter<F, RR>), dim3(scatter_blocks), dim3(512), scatter_lds, stream, args);         \
    }
    if (plan.fast_f64) { if (R == 8) VXH_SC(true, 8) else if (R == 4) VXH_SC(true, 4) else VXH_SC(true, 2) }
    else { if (R == 8) VXH_SC(false, 8) else if (R == 4) VXH_SC(false, 4) else VXH_SC(false, 2) }
#undef VXH_SC
    if (plan.lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void *)part_reduce, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes);
    hipLaunchKernelGGL(part_reduce, dim3(plan.blocks), dim3(plan.block), plan.lds_bytes, stream, args);
}

void vxh_launch_bin(const BinArgs &args, const LaunchPlan &plan, hipStream_t stream) {
    dim3 g(plan.blocks), b(plan.block);
#define VXH_LAUNCH(S, F)                                                                                               \
    do {                                                                                                               \
        if (plan.lds_bytes > 48 * 1024) (void)hipFuncSetAttribute((const void *)bin_kernel<S, F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds_bytes); \
        hipLaunchKernelGGL((bin_kernel<S, F>), g, b, plan.lds_bytes, stream, args);                                    \
    } while (0)
    if (plan.strategy == VXH_STRAT_LDS) { if (plan.fast_f64) VXH_LAUNCH(VXH_STRAT_LDS, true); else VXH_LAUNCH(VXH_STRAT_LDS, false); }
    else if (plan.strategy == VXH_STRAT_XCC) { if (plan.fast_f64) VXH_LAUNCH(VXH_STRAT_XCC, true); else VXH_LAUNCH(VXH_STRAT_XCC, false); }
    else { if (plan.fast_f64) VXH_LAUNCH(VXH_STRAT_GLOBAL, true); else VXH_LAUNCH(VXH_STRAT_GLOBAL, false); }
#undef VXH_LAUNCH
}

void vxh_launch_fill(void *dst, uint64_t ncells, int cell, const void *value8, hipStream_t stream) {
    uint64_t v;
    memcpy(&v, value8, 8);
    int cs = (int)vxh_cell_size(cell);
    uint64_t blocks = (ncells + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dst, ncells, cs, v);
}

void vxh_launch_fold(void *grid, uint64_t cells, int replicas, int cell, int kind, const void *identity8, hipStream_t stream) {
    if (replicas <= 1) return;
    uint64_t blocks = (cells + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    dim3 g((unsigned)blocks), b(256);
    const int op = kind == VXH_AGG_MIN ? 1 : (kind == VXH_AGG_MAX ? 2 : 0);
#define VXH_FOLD(T)                                                                                                    \
    {                                                                                                                  \
        T ident;                                                                                                       \
        memcpy(&ident, identity8, sizeof(T));                                                                          \
        if (op == 0) hipLaunchKernelGGL((fold_kernel<T, 0>), g, b, 0, stream, (T *)grid, cells, replicas, ident);       \
        else if (op == 1) hipLaunchKernelGGL((fold_kernel<T, 1>), g, b, 0, stream, (T *)grid, cells, replicas, ident);  \
        else hipLaunchKernelGGL((fold_kernel<T, 2>), g, b, 0, stream, (T *)grid, cells, replicas, ident);               \
    }
    switch (cell) {
    case VXH_CELL_I64: VXH_FOLD(long long) break;
    case VXH_CELL_F64: VXH_FOLD(double) break;
    case VXH_CELL_U64: VXH_FOLD(unsigned long long) break;
    case VXH_CELL_F32: VXH_FOLD(float) break;
    case VXH_CELL_I32: VXH_FOLD(int) break;
    default: VXH_FOLD(unsigned) break;
    }
#undef VXH_FOLD
}

void vxh_launch_minmax(int dtype, int flip, const void *data, const uint8_t *mask, uint64_t n, double *out2_dev, hipStream_t stream) {
    uint64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(minmax_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, dtype, flip, data, mask, n, out2_dev);
}

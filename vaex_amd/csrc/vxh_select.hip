// Row-wise helper kernels: device-side selections, packed multi-key group keys, column products.
//
// Device-side selections: a predicate over up to four columns evaluated on the GPU into the aggregators' keep-mask.
//
// The reference evaluates a selection like "(x > 0) & (v < 3.5)" with numpy on the host, once per chunk, into a boolean
// array (vaex/execution.py:530-549 via vaex/scopes.py:138-177) that TaskPartAggregation.process hands to every aggregator
// as its data mask (vaex/cpu.py:740-784; polarity 1 = keep: src/agg_count.cpp:50, src/agg_sum.cpp:108).  Here the predicate
// itself crosses the C-ABI (include/vaex_hip.h "device-side selections") as comparison terms `column <op> constant` plus a
// truth table over the terms' outcomes; this kernel turns it into the same byte mask in HBM — no numpy pass over the chunk,
// no mask bytes over PCIe.  Comparisons follow numpy: every comparison with NaN is false except !=; an integer column is
// compared exactly with an integer constant and as float64 with a float constant; a float32 column is compared in float32
// (the constant is rounded first: `f4 <= 0.3` holds for float32(0.3), as in numpy).
#include "vxh_internal.hpp"
#include "vxh_kernels.hpp"

namespace {

__device__ __forceinline__ bool cmp_f64(double x, int op, double c) {
    switch (op) {
    case VXH_CMP_LT: return x < c;
    case VXH_CMP_LE: return x <= c;
    case VXH_CMP_GT: return x > c;
    case VXH_CMP_GE: return x >= c;
    case VXH_CMP_EQ: return x == c;
    default: return x != c;
    }
}
template <typename I>
__device__ __forceinline__ bool cmp_int(I x, int op, I c) {
    switch (op) {
    case VXH_CMP_LT: return x < c;
    case VXH_CMP_LE: return x <= c;
    case VXH_CMP_GT: return x > c;
    case VXH_CMP_GE: return x >= c;
    case VXH_CMP_EQ: return x == c;
    default: return x != c;
    }
}

// the left side of an expression term: a postfix program over float64 columns, a stack of four doubles kept in registers (pushes
// and pops are moves — no indexed array, nothing in scratch memory); every step is a wave-uniform switch
__device__ __forceinline__ double eval_program(const SelArgs &A, int t, uint64_t i) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    const int n = A.nsteps[t];
    for (int k = 0; k < n; ++k) {
        const vxh_sel_step &st = A.prog[t][k];
        switch (st.op) {
        case VXH_SEL_COL: s3 = s2; s2 = s1; s1 = s0; s0 = ((const double *)A.col[st.column])[i]; break;
        case VXH_SEL_CONST: s3 = s2; s2 = s1; s1 = s0; s0 = st.value; break;
        case VXH_SEL_ADD: s0 = s1 + s0; s1 = s2; s2 = s3; break;
        case VXH_SEL_SUB: s0 = s1 - s0; s1 = s2; s2 = s3; break;
        case VXH_SEL_MUL: s0 = s1 * s0; s1 = s2; s2 = s3; break;
        case VXH_SEL_DIV: s0 = s1 / s0; s1 = s2; s2 = s3; break;
        case VXH_SEL_NEG: s0 = -s0; break;
        case VXH_SEL_SQUARE: s0 = s0 * s0; break;
        case VXH_SEL_SQRT: s0 = sqrt(s0); break;
        case VXH_SEL_LT: s0 = s1 < s0 ? 1.0 : 0.0; s1 = s2; s2 = s3; break;   // (IEEE comparisons: false next to a NaN, != true — numpy's)
        case VXH_SEL_LE: s0 = s1 <= s0 ? 1.0 : 0.0; s1 = s2; s2 = s3; break;
        case VXH_SEL_GT: s0 = s1 > s0 ? 1.0 : 0.0; s1 = s2; s2 = s3; break;
        case VXH_SEL_GE: s0 = s1 >= s0 ? 1.0 : 0.0; s1 = s2; s2 = s3; break;
        case VXH_SEL_EQ: s0 = s1 == s0 ? 1.0 : 0.0; s1 = s2; s2 = s3; break;
        case VXH_SEL_NE: s0 = s1 != s0 ? 1.0 : 0.0; s1 = s2; s2 = s3; break;
        default: s0 = fabs(s0); break;
        }
    }
    return s0;
}

__device__ __forceinline__ bool term_at(const SelArgs &A, int t, uint64_t i) {
    const SelTerm &T = A.t[t];
    if (A.nsteps[t] > 0) return cmp_f64(eval_program(A, t, i), T.op, T.value);
    const void *p = A.col[T.column];
    const int op = T.op;
    switch (A.dtype[T.column]) {
    case VXH_F64: return cmp_f64(((const double *)p)[i], op, T.value);
    case VXH_F32: return cmp_f64((double)((const float *)p)[i], op, (double)(float)T.value); // numpy compares a float32 column with the constant ROUNDED to float32
    case VXH_I64: {
        const int64_t x = ((const int64_t *)p)[i];
        return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value);
    }
    case VXH_U64: {
        const uint64_t x = ((const uint64_t *)p)[i];
        if (!T.is_int) return cmp_f64((double)x, op, T.value);
        if (T.ivalue < 0) return op == VXH_CMP_GT || op == VXH_CMP_GE || op == VXH_CMP_NE; // every uint64 is above a negative constant
        return cmp_int<uint64_t>(x, op, (uint64_t)T.ivalue);
    }
    case VXH_I32: { const int64_t x = ((const int32_t *)p)[i]; return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value); }
    case VXH_I16: { const int64_t x = ((const int16_t *)p)[i]; return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value); }
    case VXH_I8: { const int64_t x = ((const int8_t *)p)[i]; return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value); }
    case VXH_U32: { const int64_t x = ((const uint32_t *)p)[i]; return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value); }
    case VXH_U16: { const int64_t x = ((const uint16_t *)p)[i]; return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value); }
    default: { const int64_t x = ((const uint8_t *)p)[i]; return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value); } // U8 / BOOL
    }
}

// four consecutive rows per thread: one 32-bit store of four mask bytes (the tail rows one by one)
__global__ __launch_bounds__(256) void sel_eval(SelArgs A) {
    const uint64_t quads = (A.n + 3) / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i0 = q * 4;
        uint32_t packed = 0;
        const int rows = (int)(A.n - i0 < 4 ? A.n - i0 : 4);
        for (int r = 0; r < rows; r++) {
            uint32_t bits = 0;
            for (int t = 0; t < A.nterms; t++) bits |= (term_at(A, t, i0 + r) ? 1u : 0u) << t;
            uint32_t keep = (A.truth >> bits) & 1u;
            if (A.and_mask) keep &= A.and_mask[i0 + r] != 0 ? 1u : 0u;
            packed |= keep << (8 * r);
        }
        if (rows == 4) {
            *(uint32_t *)(A.out + i0) = packed; // (out is 256-byte aligned scratch)
        } else {
            for (int r = 0; r < rows; r++) A.out[i0 + r] = (uint8_t)(packed >> (8 * r));
        }
    }
}

// The same mask where every term compares a float64 column as it is with a constant (no program) — what filters and selections mostly are
// (round 6, late: the generic kernel above loads one row at a time behind a dtype switch, every load waited for by its comparison:
// 2.95 ms per 1e9 rows x one column = 3.0 TB/s of its 9 bytes a row, profiles/r06_groupby_predicate.txt).  A lane takes U quads of four
// consecutive rows a round; the 16-byte loads of ALL of them (NC columns, clamped to the last quad so that none is conditional) are issued
// before the first comparison; one 4-byte store of mask bytes per quad.  The rows behind the last full quad go one by one.
template <int NC, int U>
__global__ __launch_bounds__(256) void sel_eval_f64(SelArgs A) {
    const uint64_t quads = A.n / 4, stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t q0 = (uint64_t)blockIdx.x * 256u + threadIdx.x; q0 < quads; q0 += stride * U) {
        double x[NC][U][4];
        uint32_t am[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t q = q0 + (uint64_t)u * stride, qc = q < quads ? q : quads - 1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double2 *p = (const double2 *)((const double *)A.col[c] + qc * 4);
                const double2 lo = p[0], hi = p[1];
                x[c][u][0] = lo.x; x[c][u][1] = lo.y; x[c][u][2] = hi.x; x[c][u][3] = hi.y;
            }
            am[u] = A.and_mask ? *(const uint32_t *)(A.and_mask + qc * 4) : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t q = q0 + (uint64_t)u * stride;
            if (q < quads) {
                uint32_t packed = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint32_t bits = 0;
#pragma unroll
                    for (int t = 0; t < VXH_SEL_MAX_TERMS; ++t) {
                        if (t < A.nterms) {
                            double v = x[0][u][r];
#pragma unroll
                            for (int c = 1; c < NC; ++c) v = A.t[t].column == c ? x[c][u][r] : v; // (wave-uniform)
                            bits |= (cmp_f64(v, A.t[t].op, A.t[t].value) ? 1u : 0u) << t;
                        }
                    }
                    uint32_t keep = (A.truth >> bits) & 1u;
                    if (A.and_mask) keep &= ((am[u] >> (8 * r)) & 0xffu) != 0u ? 1u : 0u;
                    packed |= keep << (8 * r);
                }
                *(uint32_t *)(A.out + q * 4) = packed; // (out is 256-byte aligned scratch)
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(A.n & 3u)) {
        const uint64_t i = quads * 4 + threadIdx.x;
        uint32_t bits = 0;
        for (int t = 0; t < A.nterms; t++) bits |= (term_at(A, t, i) ? 1u : 0u) << t;
        uint32_t keep = (A.truth >> bits) & 1u;
        if (A.and_mask) keep &= A.and_mask[i] != 0 ? 1u : 0u;
        A.out[i] = (uint8_t)keep;
    }
}
// number of columns the fast form would read (1..4), or 0 when the selection is not of its kind
static int sel_eval_f64_columns(const SelArgs &A) {
    if (A.nterms < 1 || A.nterms > VXH_SEL_MAX_TERMS || A.n < 4) return 0;
    int nc = 0;
    for (int t = 0; t < A.nterms; t++) {
        const int c = A.t[t].column;
        if (A.nsteps[t] > 0 || c < 0 || c >= VXH_SEL_MAX_COLUMNS || A.dtype[c] != VXH_F64) return 0;
        nc = std::max(nc, c + 1);
    }
    for (int c = 0; c < nc; c++) // (a column no term reads is loaded all the same: it must be one)
        if (!A.col[c] || A.dtype[c] != VXH_F64 || ((uintptr_t)A.col[c] & 15u)) return 0;
    if (((uintptr_t)A.and_mask & 3u) || ((uintptr_t)A.out & 3u)) return 0;
    return nc;
}

// packed group key of a multi-key groupby: sum_i (key_i - min_i) * multiplier_i as int64 — the expression vaex's
// GrouperCombined builds out of its parents' ordinals (vaex/groupby.py:526-584 `_combine`)
__device__ __forceinline__ int64_t load_i64(const void *p, int dtype, uint64_t i) {
    switch (dtype) {
    case VXH_I64: case VXH_U64: return ((const int64_t *)p)[i];
    case VXH_I32: return ((const int32_t *)p)[i];
    case VXH_I16: return ((const int16_t *)p)[i];
    case VXH_I8: return ((const int8_t *)p)[i];
    case VXH_U32: return ((const uint32_t *)p)[i];
    case VXH_U16: return ((const uint16_t *)p)[i];
    default: return ((const uint8_t *)p)[i];
    }
}
__global__ __launch_bounds__(256) void pack_keys(PackArgs A) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += (uint64_t)gridDim.x * blockDim.x) {
        int64_t packed = 0;
        for (int k = 0; k < A.nkeys; k++) packed += (int64_t)((uint64_t)(load_i64(A.col[k], A.dtype[k], i) - A.min_value[k]) * (uint64_t)A.multiplier[k]);
        A.out[i] = packed;
    }
}

// row-wise product of two float64 columns (NaN where either is NaN): the off-diagonal inputs of OP_COV (src/vaexfast.cpp:1117-1153)
__global__ __launch_bounds__(256) void product_f64(const double *a, const double *b, double *out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = a[i] * b[i];
}

// A scalar binner column of ANY dtype / byte order / with a missing-value mask as the float64 column the fast kernels read:
// BinnerScalar<T>::to_bins converts the element to double before anything else (src/binners.cpp:16-35; byte-swapped first for
// `_non_native`, src/agg.hpp:18-26), and a masked row lands in cell 0 like a NaN does (:26-29) — so double(value), NaN where masked, is
// the same column to every binning kernel.  Two rows per thread and trip.
__device__ __forceinline__ double element_as_f64(const void *p, int dtype, int flip, uint64_t i) {
    switch (dtype) {
    case VXH_F64: { uint64_t x = ((const uint64_t *)p)[i]; if (flip) x = __builtin_bswap64(x); return __longlong_as_double((long long)x); }
    case VXH_F32: { uint32_t x = ((const uint32_t *)p)[i]; if (flip) x = __builtin_bswap32(x); return (double)__uint_as_float(x); }
    case VXH_I64: { uint64_t x = ((const uint64_t *)p)[i]; if (flip) x = __builtin_bswap64(x); return (double)(int64_t)x; }
    case VXH_U64: { uint64_t x = ((const uint64_t *)p)[i]; if (flip) x = __builtin_bswap64(x); return (double)x; }
    case VXH_I32: { uint32_t x = ((const uint32_t *)p)[i]; if (flip) x = __builtin_bswap32(x); return (double)(int32_t)x; }
    case VXH_U32: { uint32_t x = ((const uint32_t *)p)[i]; if (flip) x = __builtin_bswap32(x); return (double)x; }
    case VXH_I16: { uint16_t x = ((const uint16_t *)p)[i]; if (flip) x = __builtin_bswap16(x); return (double)(int16_t)x; }
    case VXH_U16: { uint16_t x = ((const uint16_t *)p)[i]; if (flip) x = __builtin_bswap16(x); return (double)x; }
    case VXH_I8: return (double)((const int8_t *)p)[i];
    case VXH_U8: return (double)((const uint8_t *)p)[i];
    default: return ((const uint8_t *)p)[i] ? 1.0 : 0.0; // bool
    }
}
// OUT = double, or float for the dtypes float32 holds exactly (8- / 16-bit integers, bool, float32 itself): half the bytes written and
// read back, and BinnerScalar<float>'s `double(value)` of the converted column is the same double.  Four consecutive rows per thread:
// ONE load of 4 x itemsize bytes, one or two 16-byte stores (VEC: the column is 16-byte aligned; otherwise element by element).
template <typename OUT, bool VEC>
__global__ __launch_bounds__(256) void column_convert(const void *data, const uint8_t *mask, int dtype, int flip, uint64_t n, OUT *out) {
    const OUT nan = (OUT)__longlong_as_double(0x7ff8000000000000ll);
    const uint64_t quads = n >> 2;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i0 = q * 4;
        OUT r[4];
        if (VEC) {
            // the four elements' bytes in registers, then converted one by one from there
            union { uint4 v[2]; uint8_t b[32]; } raw;
            const int isz = dtype == VXH_F64 || dtype == VXH_I64 || dtype == VXH_U64 ? 8 : (dtype == VXH_F32 || dtype == VXH_I32 || dtype == VXH_U32 ? 4 : (dtype == VXH_I16 || dtype == VXH_U16 ? 2 : 1));
            const char *p = (const char *)data + i0 * (uint64_t)isz;
            if (isz == 8) { raw.v[0] = ((const uint4 *)p)[0]; raw.v[1] = ((const uint4 *)p)[1]; }
            else if (isz == 4) raw.v[0] = *(const uint4 *)p;
            else if (isz == 2) *(uint2 *)raw.b = *(const uint2 *)p;
            else *(uint32_t *)raw.b = *(const uint32_t *)p;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = (OUT)element_as_f64(raw.b, dtype, flip, (uint64_t)k);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = (OUT)element_as_f64(data, dtype, flip, i0 + (uint64_t)k);
        }
        if (mask) {
            const uint32_t m = VEC ? *(const uint32_t *)(mask + i0) : ((uint32_t)mask[i0] | ((uint32_t)mask[i0 + 1] << 8) | ((uint32_t)mask[i0 + 2] << 16) | ((uint32_t)mask[i0 + 3] << 24));
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((m >> (8 * k)) & 0xffu) == 1u) r[k] = nan; // (binner masks: 1 = masked, src/binners.cpp:26)
        }
        if (sizeof(OUT) == 4) {
            *(float4 *)(out + i0) = make_float4((float)r[0], (float)r[1], (float)r[2], (float)r[3]);
        } else {
            ((double2 *)(out + i0))[0] = make_double2((double)r[0], (double)r[1]);
            ((double2 *)(out + i0))[1] = make_double2((double)r[2], (double)r[3]);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { // the last 1-3 rows
        const uint64_t i = (n & ~(uint64_t)3) + threadIdx.x;
        const OUT v = (OUT)element_as_f64(data, dtype, flip, i);
        out[i] = (mask && mask[i] == 1) ? nan : v;
    }
}

// An integer VALUE column of any width / byte order as int64 (round 6): count / sum over int8 / int16 / unsigned / byte-swapped columns ride the
// int64 fast paths (upcast<T> of src/agg_sum.cpp:6-62 gives them the int64 / uint64 grids those paths fill; an unsigned value zero-extends, and a
// uint64 sum is the same 64 bits as the int64 sum of its bit patterns).  Four rows per thread and trip like column_convert.
__device__ __forceinline__ long long element_as_i64(const void *p, int dtype, int flip, uint64_t i) {
    switch (dtype) {
    case VXH_I64: case VXH_U64: { uint64_t x = ((const uint64_t *)p)[i]; if (flip) x = __builtin_bswap64(x); return (long long)x; }
    case VXH_I32: { uint32_t x = ((const uint32_t *)p)[i]; if (flip) x = __builtin_bswap32(x); return (long long)(int32_t)x; }
    case VXH_U32: { uint32_t x = ((const uint32_t *)p)[i]; if (flip) x = __builtin_bswap32(x); return (long long)x; }
    case VXH_I16: { uint16_t x = ((const uint16_t *)p)[i]; if (flip) x = __builtin_bswap16(x); return (long long)(int16_t)x; }
    case VXH_U16: { uint16_t x = ((const uint16_t *)p)[i]; if (flip) x = __builtin_bswap16(x); return (long long)x; }
    case VXH_I8: return (long long)((const int8_t *)p)[i];
    case VXH_U8: return (long long)((const uint8_t *)p)[i];
    default: return ((const uint8_t *)p)[i] ? 1ll : 0ll; // bool
    }
}
template <bool VEC>
__global__ __launch_bounds__(256) void column_convert_i64(const void *data, int dtype, int flip, uint64_t n, long long *out) {
    const uint64_t quads = n >> 2;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i0 = q * 4;
        long long r[4];
        if (VEC) { // ONE load of the four elements' bytes (the column is 16-byte aligned), converted from registers
            union { uint4 v[2]; uint8_t b[32]; } raw;
            const int isz = dtype == VXH_I64 || dtype == VXH_U64 ? 8 : (dtype == VXH_I32 || dtype == VXH_U32 ? 4 : (dtype == VXH_I16 || dtype == VXH_U16 ? 2 : 1));
            const char *p = (const char *)data + i0 * (uint64_t)isz;
            if (isz == 8) { raw.v[0] = ((const uint4 *)p)[0]; raw.v[1] = ((const uint4 *)p)[1]; }
            else if (isz == 4) raw.v[0] = *(const uint4 *)p;
            else if (isz == 2) *(uint2 *)raw.b = *(const uint2 *)p;
            else *(uint32_t *)raw.b = *(const uint32_t *)p;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = element_as_i64(raw.b, dtype, flip, (uint64_t)k);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = element_as_i64(data, dtype, flip, i0 + (uint64_t)k);
        }
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        ((ll2 *)(out + i0))[0] = ll2{r[0], r[1]};
        ((ll2 *)(out + i0))[1] = ll2{r[2], r[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const uint64_t i = (n & ~(uint64_t)3) + threadIdx.x;
        out[i] = element_as_i64(data, dtype, flip, i);
    }
}

// Codes of a group key (round 6, late): what the fused groupby groups instead of the column itself when the key has missing values or is a float —
// the value as int64 (integers sign- / zero-extended, bool 0 / 1; float kinds: the bit pattern of the value as a double, every NaN under `nan_code`),
// `null_code` where the mask says missing (numpy's convention: 1).  vaex groups such keys through ordered_set<T>'s null / NaN slots
// (src/hash_primitives.hpp:455-470); here they are ordinary keys of the partitioned pass.  One pass, 8 + 8 (+ 1) bytes per row.
__global__ __launch_bounds__(256) void key_codes(const void *data, const uint8_t *mask, int dtype, int flip, uint64_t n, long long null_code, long long nan_code, long long *out) {
    const bool is_float = dtype == VXH_F64 || dtype == VXH_F32;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        long long c;
        if (is_float) {
            const double x = element_as_f64(data, dtype, flip, i);
            c = x != x ? nan_code : __double_as_longlong(x);
        } else {
            c = element_as_i64(data, dtype, flip, i);
        }
        if (mask && mask[i] == 1) c = null_code;
        out[i] = c;
    }
}

} // namespace

void vxh_launch_key_codes(const void *data, const uint8_t *mask, int dtype, int flip, uint64_t n, long long null_code, long long nan_code, long long *out, hipStream_t stream) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>((n + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(key_codes, dim3(blocks), dim3(256), 0, stream, data, mask, dtype, flip, n, null_code, nan_code, out);
}

void vxh_launch_column_convert_i64(const void *data, int dtype, int flip, uint64_t n, void *out, hipStream_t stream) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>((n / 4 + 255) / 256 + 1, 256 * 32);
    if (((uintptr_t)data & 15) == 0) hipLaunchKernelGGL(column_convert_i64<true>, dim3(blocks), dim3(256), 0, stream, data, dtype, flip, n, (long long *)out);
    else hipLaunchKernelGGL(column_convert_i64<false>, dim3(blocks), dim3(256), 0, stream, data, dtype, flip, n, (long long *)out);
}

void vxh_launch_column_convert(const void *data, const uint8_t *mask, int dtype, int flip, uint64_t n, void *out, int out_f32, hipStream_t stream) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>((n / 4 + 255) / 256 + 1, 256 * 32);
    const bool vec = (((uintptr_t)data | (uintptr_t)mask) & 15) == 0;
    if (out_f32) {
        if (vec) hipLaunchKernelGGL((column_convert<float, true>), dim3(blocks), dim3(256), 0, stream, data, mask, dtype, flip, n, (float *)out);
        else hipLaunchKernelGGL((column_convert<float, false>), dim3(blocks), dim3(256), 0, stream, data, mask, dtype, flip, n, (float *)out);
    } else {
        if (vec) hipLaunchKernelGGL((column_convert<double, true>), dim3(blocks), dim3(256), 0, stream, data, mask, dtype, flip, n, (double *)out);
        else hipLaunchKernelGGL((column_convert<double, false>), dim3(blocks), dim3(256), 0, stream, data, mask, dtype, flip, n, (double *)out);
    }
}

void vxh_launch_pack_keys(const PackArgs &A, hipStream_t stream) {
    if (!A.n) return;
    const int blocks = (int)std::min<uint64_t>((A.n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(pack_keys, dim3(blocks), dim3(256), 0, stream, A);
}
void vxh_launch_product_f64(const double *a, const double *b, double *out, uint64_t n, hipStream_t stream) {
    if (!n) return;
    const int blocks = (int)std::min<uint64_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(product_f64, dim3(blocks), dim3(256), 0, stream, a, b, out, n);
}

void vxh_launch_sel_eval(const SelArgs &A, hipStream_t stream) {
    if (!A.n) return;
    const uint64_t quads = (A.n + 3) / 4;
    const int blocks = (int)std::min<uint64_t>((quads + 255) / 256, 256 * 16);
    switch (sel_eval_f64_columns(A)) {
    case 1: hipLaunchKernelGGL((sel_eval_f64<1, 4>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 2: hipLaunchKernelGGL((sel_eval_f64<2, 2>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 3: hipLaunchKernelGGL((sel_eval_f64<3, 1>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 4: hipLaunchKernelGGL((sel_eval_f64<4, 1>), dim3(blocks), dim3(256), 0, stream, A); return;
    default: break;
    }
    hipLaunchKernelGGL(sel_eval, dim3(blocks), dim3(256), 0, stream, A);
}

void vxh_preload_select(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)product_f64);
    (void)hipGetLastError();
}

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
#include <type_traits>

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

__device__ __forceinline__ bool term_cmp(const SelTerm &T, int dtype, const void *p, uint64_t i);
__device__ __forceinline__ bool term_at(const SelArgs &A, int t, uint64_t i) {
    const SelTerm &T = A.t[t];
    if (A.nsteps[t] > 0) return cmp_f64(eval_program(A, t, i), T.op, T.value);
    return term_cmp(T, A.dtype[T.column], A.col[T.column], i);
}
// an element of a column of `dtype`, given as its bytes (zero-extended), against the term's constant
__device__ __forceinline__ bool term_cmp_bits(const SelTerm &T, int dtype, uint64_t b) {
    const int op = T.op;
    switch (dtype) {
    case VXH_F64: return cmp_f64(__longlong_as_double((long long)b), op, T.value);
    case VXH_F32: return cmp_f64((double)__uint_as_float((uint32_t)b), op, (double)(float)T.value); // numpy compares a float32 column with the constant ROUNDED to float32
    case VXH_U64: {
        const uint64_t x = b;
        if (!T.is_int) return cmp_f64((double)x, op, T.value);
        if (T.ivalue < 0) return op == VXH_CMP_GT || op == VXH_CMP_GE || op == VXH_CMP_NE; // every uint64 is above a negative constant
        return cmp_int<uint64_t>(x, op, (uint64_t)T.ivalue);
    }
    default: {
        int64_t x;
        switch (dtype) {
        case VXH_I64: x = (int64_t)b; break;
        case VXH_I32: x = (int32_t)(uint32_t)b; break;
        case VXH_I16: x = (int16_t)(uint16_t)b; break;
        case VXH_I8: x = (int8_t)(uint8_t)b; break;
        case VXH_U32: x = (int64_t)(uint32_t)b; break;
        case VXH_U16: x = (int64_t)(uint16_t)b; break;
        default: x = (int64_t)(uint8_t)b; break; // U8 / BOOL
        }
        return T.is_int ? cmp_int<int64_t>(x, op, T.ivalue) : cmp_f64((double)x, op, T.value);
    }
    }
}
__device__ __forceinline__ int sel_itemsize(int dtype) {
    return dtype == VXH_F64 || dtype == VXH_I64 || dtype == VXH_U64 ? 8 : (dtype == VXH_F32 || dtype == VXH_I32 || dtype == VXH_U32 ? 4 : (dtype == VXH_I16 || dtype == VXH_U16 ? 2 : 1));
}
// element i of a column of `dtype` at p against the term's constant
__device__ __forceinline__ bool term_cmp(const SelTerm &T, int dtype, const void *p, uint64_t i) {
    uint64_t b;
    switch (sel_itemsize(dtype)) {
    case 8: b = ((const uint64_t *)p)[i]; break;
    case 4: b = ((const uint32_t *)p)[i]; break;
    case 2: b = ((const uint16_t *)p)[i]; break;
    default: b = ((const uint8_t *)p)[i]; break;
    }
    return term_cmp_bits(T, dtype, b);
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

// The same mask where every term compares a COLUMN as it is with a constant (no program) — what filters and selections mostly are
// (round 6, late: the generic kernel above loads one row at a time behind a dtype switch, every load waited for by its comparison:
// 2.95 ms per 1e9 rows x one float64 column = 3.0 TB/s of its 9 bytes a row, profiles/r06_groupby_predicate.txt).  A lane takes U quads
// of four consecutive rows a round; ONE load of 4 x itemsize bytes per column and quad (two for 8-byte columns), ALL of them — NC columns,
// clamped to the last quad so that none is conditional — issued before the first comparison, the elements compared from registers by
// the generic kernel's own rule (term_cmp); one 4-byte store of mask bytes per quad.  The rows behind the last full quad go one by one.
// element r (0..3, a compile-time number where it is called) of a quad held as its dwords
__device__ __forceinline__ uint64_t sel_quad_element(const uint32_t (&w)[8], int isz, int r) {
    if (isz == 8) return (uint64_t)w[2 * r] | ((uint64_t)w[2 * r + 1] << 32);
    if (isz == 4) return w[r];
    if (isz == 2) return (w[r >> 1] >> (16 * (r & 1))) & 0xffffu;
    return (w[0] >> (8 * r)) & 0xffu;
}
// (first form of this kernel: term_cmp_bits per row and term — a dtype switch, an is-integer branch and a comparison switch, all wave-uniform,
//  all taken per row: 2-3.3 ms per 1e9 rows WHATEVER the column's width, +1.2 ms per further term — bound by its scalar branches, not by
//  memory, profiles/r06_sel_eval.txt.  Now the dtype switch runs once per quad and column and leaves every element in two forms — a double,
//  and its integer value with the sign bit flipped (an unsigned order) — and a term is straight-line code: the relation (less / equal /
//  greater / unordered) of the form the term compares in, one bit of the comparison's 4-bit code; nothing per term is decided per row.)
struct SelTermReg {   // what a term needs, per thread in scalar registers
    double cd;        // the constant as the double the column's values are compared with
    uint64_t cb;      // the constant as an integer, sign bit flipped
    uint32_t code;    // bit `relation` set: the term holds (0 for a term that does not exist: never)
    bool use_int;     // compare the integer forms (integer column and integer constant)
    bool all_greater; // ... an unsigned 64-bit column against a negative constant: every value is greater
    int column;
};
template <int NC, int U>
__global__ __launch_bounds__(256) void sel_eval_vec(SelArgs A) {
    const uint64_t quads = A.n / 4, stride = (uint64_t)gridDim.x * 256u;
    SelTermReg T[VXH_SEL_MAX_TERMS];
#pragma unroll
    for (int t = 0; t < VXH_SEL_MAX_TERMS; ++t) {
        const SelTerm &S = A.t[t];
        const int c = t < A.nterms ? S.column : 0;
        int d = A.dtype[0];
#pragma unroll
        for (int k = 1; k < NC; ++k) d = c == k ? A.dtype[k] : d;
        const int op = S.op;
        const uint32_t code = op == VXH_CMP_LT ? 1u : op == VXH_CMP_LE ? 3u : op == VXH_CMP_GT ? 4u : op == VXH_CMP_GE ? 6u : op == VXH_CMP_EQ ? 2u : 13u; // bits: 0 less, 1 equal, 2 greater, 3 unordered
        T[t].code = t < A.nterms ? code : 0u;
        T[t].column = c;
        T[t].cd = d == VXH_F32 ? (double)(float)S.value : S.value; // numpy compares a float32 column with the constant ROUNDED to float32
        T[t].use_int = S.is_int && d != VXH_F64 && d != VXH_F32;
        T[t].all_greater = d == VXH_U64 && S.ivalue < 0;
        T[t].cb = d == VXH_U64 ? (uint64_t)S.ivalue : (uint64_t)S.ivalue ^ 0x8000000000000000ull;
    }
    bool need_d = false, need_i = false; // does any term compare doubles / integers (a form nobody compares is not made)
#pragma unroll
    for (int t = 0; t < VXH_SEL_MAX_TERMS; ++t) {
        if (t < A.nterms) { need_d = need_d || !T[t].use_int; need_i = need_i || T[t].use_int; }
    }
    for (uint64_t q0 = (uint64_t)blockIdx.x * 256u + threadIdx.x; q0 < quads; q0 += stride * U) {
        uint32_t raw[NC][U][8]; // a quad's bytes as dwords (only ever indexed by compile-time numbers: registers)
        uint32_t am[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t q = q0 + (uint64_t)u * stride, qc = q < quads ? q : quads - 1;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int isz = sel_itemsize(A.dtype[c]); // (wave-uniform)
                const char *p = (const char *)A.col[c] + qc * 4 * (uint64_t)isz;
                uint32_t (&w)[8] = raw[c][u];
#pragma unroll
                for (int k = 0; k < 8; ++k) w[k] = 0u;
                if (isz == 8) {
                    const uint4 lo = ((const uint4 *)p)[0], hi = ((const uint4 *)p)[1];
                    w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
                } else if (isz == 4) {
                    const uint4 lo = *(const uint4 *)p;
                    w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w;
                } else if (isz == 2) {
                    const uint2 lo = *(const uint2 *)p;
                    w[0] = lo.x; w[1] = lo.y;
                } else {
                    w[0] = *(const uint32_t *)p;
                }
            }
            am[u] = A.and_mask ? *(const uint32_t *)(A.and_mask + qc * 4) : 0u;
        }
        // (the quads one by one through a generic lambda with a compile-time number: as a `#pragma unroll` loop over u the compiler left the
        //  loop rolled — "unable to perform the requested transformation" — and `raw` went to scratch memory, 144 bytes a lane)
        auto emit = [&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const uint64_t q = q0 + (uint64_t)u * stride;
            if (q < quads) {
                double xd[NC][4];   // the elements as the doubles a float constant is compared with
                uint64_t xb[NC][4]; // ... as integers in an unsigned order (integer columns)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const int d = A.dtype[c], isz = sel_itemsize(d);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint64_t b = sel_quad_element(raw[c][u], isz, r);
                        int64_t x;
                        switch (d) { // (once per quad and column; wave-uniform)
                        case VXH_F64: xd[c][r] = __longlong_as_double((long long)b); xb[c][r] = 0; break;
                        case VXH_F32: xd[c][r] = (double)__uint_as_float((uint32_t)b); xb[c][r] = 0; break;
                        case VXH_U64: xd[c][r] = need_d ? (double)b : 0.0; xb[c][r] = b; break;
                        default:
                            switch (d) {
                            case VXH_I64: x = (int64_t)b; break;
                            case VXH_I32: x = (int32_t)(uint32_t)b; break;
                            case VXH_I16: x = (int16_t)(uint16_t)b; break;
                            case VXH_I8: x = (int8_t)(uint8_t)b; break;
                            case VXH_U32: x = (int64_t)(uint32_t)b; break;
                            case VXH_U16: x = (int64_t)(uint16_t)b; break;
                            default: x = (int64_t)(uint8_t)b; break; // U8 / BOOL
                            }
                            xd[c][r] = need_d ? (double)x : 0.0; xb[c][r] = (uint64_t)x ^ 0x8000000000000000ull;
                            break;
                        }
                    }
                }
                uint32_t bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int t = 0; t < VXH_SEL_MAX_TERMS; ++t) {
                    if (t < A.nterms) { // (wave-uniform, like the two branches below: decided per term and quad, not per row)
                        if (T[t].use_int) {
                            if (T[t].all_greater) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) bits[r] |= ((T[t].code >> 2) & 1u) << t;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    uint64_t vb = xb[0][r];
#pragma unroll
                                    for (int c = 1; c < NC; ++c) vb = T[t].column == c ? xb[c][r] : vb;
                                    const uint32_t rel = vb < T[t].cb ? 0u : (vb == T[t].cb ? 1u : 2u);
                                    bits[r] |= ((T[t].code >> rel) & 1u) << t;
                                }
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                double vd = xd[0][r];
#pragma unroll
                                for (int c = 1; c < NC; ++c) vd = T[t].column == c ? xd[c][r] : vd;
                                const uint32_t rel = vd < T[t].cd ? 0u : (vd == T[t].cd ? 1u : (vd > T[t].cd ? 2u : 3u));
                                bits[r] |= ((T[t].code >> rel) & 1u) << t;
                            }
                        }
                    }
                }
                uint32_t packed = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uint32_t keep = (A.truth >> bits[r]) & 1u;
                    if (A.and_mask) keep &= ((am[u] >> (8 * r)) & 0xffu) != 0u ? 1u : 0u;
                    packed |= keep << (8 * r);
                }
                *(uint32_t *)(A.out + q * 4) = packed; // (out is 256-byte aligned scratch)
            }
        };
        emit(std::integral_constant<int, 0>{});
        if constexpr (U > 1) emit(std::integral_constant<int, 1>{});
        if constexpr (U > 2) emit(std::integral_constant<int, 2>{});
        if constexpr (U > 3) emit(std::integral_constant<int, 3>{});
        static_assert(U <= 4, "emit() is spelled out for four quads");
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(A.n & 3u)) {
        const uint64_t i = quads * 4 + threadIdx.x;
        uint32_t bits = 0;
#pragma unroll
        for (int t = 0; t < VXH_SEL_MAX_TERMS; ++t) { // (compile-time term and column numbers: the descriptor stays where the kernel arguments are)
            if (t < A.nterms) {
                bool hit = false;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (A.t[t].column == c) hit = term_cmp(A.t[t], A.dtype[c], A.col[c], i);
                bits |= (hit ? 1u : 0u) << t;
            }
        }
        uint32_t keep = (A.truth >> bits) & 1u;
        if (A.and_mask) keep &= A.and_mask[i] != 0 ? 1u : 0u;
        A.out[i] = (uint8_t)keep;
    }
}
// number of columns the quad form would read (1..4), or 0 when the selection is not of its kind
static int sel_eval_vec_columns(const SelArgs &A) {
    if (A.nterms < 1 || A.nterms > VXH_SEL_MAX_TERMS || A.n < 4) return 0;
    int nc = 0;
    for (int t = 0; t < A.nterms; t++) {
        const int c = A.t[t].column;
        if (A.nsteps[t] > 0 || c < 0 || c >= VXH_SEL_MAX_COLUMNS) return 0;
        nc = std::max(nc, c + 1);
    }
    for (int c = 0; c < nc; c++) { // (a column no term reads is loaded all the same: it must be one; a quad of it must be one aligned load)
        if (!A.col[c] || A.dtype[c] >= VXH_DTYPE_COUNT) return 0;
        const int d = A.dtype[c];
        const int isz = d == VXH_F64 || d == VXH_I64 || d == VXH_U64 ? 8 : (d == VXH_F32 || d == VXH_I32 || d == VXH_U32 ? 4 : (d == VXH_I16 || d == VXH_U16 ? 2 : 1));
        if ((uintptr_t)A.col[c] & (uintptr_t)(isz == 8 ? 15 : 4 * isz - 1)) return 0;
    }
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

// ... for the usual handful of keys (round 6, late): the kernel above loads one element per key at a time — every load behind a dtype
// switch, waited for by the multiply-add that uses it.  Here a lane takes U rows a round (a wave's rows of a round are consecutive: every
// load is one coalesced line per key), the dtype switch runs once per key and round AROUND the U loads of that key, and all NK x U loads are
// in flight before the first multiply (profiles/r06_pack_keys.txt).
template <int NK, int U>
__global__ __launch_bounds__(256) void pack_keys_n(PackArgs A) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i0 = (uint64_t)blockIdx.x * 256u + threadIdx.x; i0 < A.n; i0 += stride * U) {
        int64_t x[NK][U];
        uint64_t ic[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const uint64_t i = i0 + (uint64_t)u * stride; ic[u] = i < A.n ? i : A.n - 1; } // (clamped: no load is conditional)
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const void *p = A.col[k];
            switch (A.dtype[k]) { // (wave-uniform; the U loads of a case back to back)
            case VXH_I64: case VXH_U64:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const int64_t *)p)[ic[u]];
                break;
            case VXH_I32:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const int32_t *)p)[ic[u]];
                break;
            case VXH_U32:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const uint32_t *)p)[ic[u]];
                break;
            case VXH_I16:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const int16_t *)p)[ic[u]];
                break;
            case VXH_U16:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const uint16_t *)p)[ic[u]];
                break;
            case VXH_I8:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const int8_t *)p)[ic[u]];
                break;
            default:
#pragma unroll
                for (int u = 0; u < U; ++u) x[k][u] = ((const uint8_t *)p)[ic[u]];
                break;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * stride;
            if (i < A.n) {
                int64_t packed = 0;
#pragma unroll
                for (int k = 0; k < NK; ++k) packed += (int64_t)((uint64_t)(x[k][u] - A.min_value[k]) * (uint64_t)A.multiplier[k]);
                A.out[i] = packed;
            }
        }
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
    switch (A.nkeys) { // (the row-at-a-time kernel below: five to eight keys; what it cost for fewer is in profiles/r06_pack_keys.txt)
    case 1: hipLaunchKernelGGL((pack_keys_n<1, 4>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 2: hipLaunchKernelGGL((pack_keys_n<2, 4>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 3: hipLaunchKernelGGL((pack_keys_n<3, 4>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 4: hipLaunchKernelGGL((pack_keys_n<4, 2>), dim3(blocks), dim3(256), 0, stream, A); return;
    default: break;
    }
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
    switch (sel_eval_vec_columns(A)) {
    case 1: hipLaunchKernelGGL((sel_eval_vec<1, 4>), dim3(blocks), dim3(256), 0, stream, A); return;
    case 2: hipLaunchKernelGGL((sel_eval_vec<2, 2>), dim3(blocks), dim3(256), 0, stream, A); return;
    default: break; // (three or four columns: the compiler turns the choice of a term's registers into an indexed access and the quads go to scratch memory — the generic kernel)
    }
    hipLaunchKernelGGL(sel_eval, dim3(blocks), dim3(256), 0, stream, A);
}

void vxh_preload_select(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)product_f64);
    (void)hipGetLastError();
}

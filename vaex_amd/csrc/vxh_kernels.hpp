// Kernel-side descriptors shared by the launch code (vxh_api.hip) and the kernels
// (vxh_kernels.hip).  Everything here is passed BY VALUE in the kernarg segment so that the
// per-dimension / per-aggregator parameters are read with scalar loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vaex_hip.h"

#define VXH_MAX_DIM 16 // reference: MAX_DIM = 16 (src/agg.hpp:29)
#define VXH_MAX_AGG 12 // aggregators fused per launch; longer lists are split over several launches

enum vxh_binner_kind : uint8_t { VXH_BIN_SCALAR = 0, VXH_BIN_ORDINAL = 1, VXH_BIN_HASH = 2 };

// device cell type of an aggregator grid
enum vxh_cell : uint8_t { VXH_CELL_I64 = 0, VXH_CELL_F64 = 1, VXH_CELL_U64 = 2, VXH_CELL_F32 = 3, VXH_CELL_I32 = 4, VXH_CELL_U32 = 5 };

struct BinnerDesc {
    const void *data;    // n elements of dtype
    const uint8_t *mask; // 1 = masked, or null
    double vmin;         // scalar
    double scale;        // scalar: 1/(vmax-vmin), computed on the host in double like binners.cpp:16
    double binsd;        // scalar: (double)bins
    uint64_t bins;       // scalar: bins; ordinal/hash: ordinal_count
    int64_t min_value;   // ordinal
    uint64_t stride;     // cells
    const int64_t *hkeys; // hash: table of packed {key, ordinal} slots (16 bytes each; ordinal -1 = empty)
    const int64_t *hvals; // (unused: nullptr)
    uint64_t hmask;       // hash: capacity-1
    int64_t null_bin;     // hash: cell for masked rows
    int64_t hmin_ord;     // hash: ordinal of the key INT64_MIN (the table's EMPTY sentinel: kept in the map's side words), or -1
    uint8_t kind, dtype, flip, allow_other, invert;
    uint8_t f32mode;      // scalar: the legacy statisticNd<float> arithmetic (src/vaexfast.cpp:1185-1262): (value - min) * scale in float32;
                          // 1: the product with the bin count in double, 2: in float32 (its two-dimensional loop)
    float vmin_f, scale_f;
};

struct AggDesc {
    const void *data;    // n elements of dtype, or null (count(*))
    const uint8_t *mask; // 1 = keep, or null
    void *grid;          // replica 0 of the device grid; replica r at grid + r*cells
    uint32_t moment;
    uint32_t lds_offset; // byte offset of this aggregator's private grid in LDS (LDS variants)
    uint8_t kind, dtype, flip, cell;
};

// Fused selection (round 4): every aggregator of the launch shares ONE device-side selection whose terms all read the same float64
// column; the fast kernels evaluate it on the rows they bin instead of reading a keep-mask that a separate pass (sel_eval) wrote.
// A term `x <op> c` holds iff bit (relation of x to c: 0 less, 1 equal, 2 greater, 3 unordered = NaN) of `code` is set — numpy's rules:
// every comparison with NaN is false except != (vxh_select.hip cmp_f64).  keep = bit (outcomes of the terms) of `truth`.
struct PredDesc {
    const void *col;   // float64, one element per row of the launch
    const void *col2;  // round 5: a SECOND float64 column (null: every term reads `col`) — "(v > 3) & (w < 1)"; tcol[t] says which one term t reads
    uint8_t tcol[4];
    int32_t on;        // 0: no fused selection (aggregator masks, if any, are byte masks)
    int32_t nterms;    // 1..4
    uint32_t truth;
    uint32_t code[4];
    int32_t op[4];     // the terms as they came (vxh_cmp): what sel_eval needs when a launch cannot take the fused form
    double c[4];
};

struct BinArgs {
    PredDesc pred;
    uint64_t n;     // rows in this launch
    uint64_t cells; // length1d
    int32_t ndim;
    int32_t nagg;
    int32_t replicas;        // device grid replicas (>=1)
    int32_t replicas_per_xcc; // XCC strategy: replicas = 8 * replicas_per_xcc
    int32_t slab_log2;       // LDS strategy: the grid is cut into S = 2^slab_log2 interleaved slabs (cell & (S-1))
    int32_t ngroups;         // LDS strategy: gridDim.x = ngroups * S; the S workgroups of a group read the same rows
    int32_t flush_plain;     // LDS strategy: flush with plain read-add-write into replica `group` (exclusive owner)
    int32_t count16;         // LDS copies of count grids are packed 16-bit halves (every aggregator is a count)
    BinnerDesc b[VXH_MAX_DIM];
    AggDesc a[VXH_MAX_AGG];
};

enum vxh_strategy : int {
    VXH_STRAT_AUTO = 0,
    VXH_STRAT_GLOBAL = 1, // device-scope atomics straight into replica (blockIdx % replicas)
    VXH_STRAT_XCC = 2,    // L2-local (workgroup-scope) atomics into the replica set of the block's own XCD
    VXH_STRAT_LDS = 3,    // workgroup-private grids (or interleaved slabs of them) in LDS, flushed once per workgroup
    VXH_STRAT_PART = 4,   // two passes: partition rows into per-slab record queues, then aggregate each slab in LDS
};

// Timing experiments that make results WRONG on purpose (cold rows dropped, records kept out of HBM, ...) exist only in the ablation build
// (`make ablate`: -DVXH_ABLATE, a second library under tools/); in the product library the tests below are compile-time zeros.
#ifdef VXH_ABLATE
#define VXH_ABL(P, mask) ((P).no_pipeline & (mask))
#else
#define VXH_ABL(P, mask) 0
#endif

#define VXH_PART_MAX_VALS 4
#define VXH_PART_MAX_MASKS 8

// Partition strategy (grids too large for one workgroup's LDS).  Pass 1 (part_scatter) reads the rows once,
// computes the flat cell index, and appends a compact record {local index, [mask flags], aggregator inputs}
// to the queue of the slab that owns the cell (slab = cell & (S-1), local = cell >> log2 S), bucketing each
// tile in LDS so the queue writes are coalesced.  Pass 2 (part_reduce) gives every slab to `parts`
// workgroups that aggregate their share of the queue into an LDS-private copy of the slab and flush it.
struct PartArgs {
    BinArgs A;
    int32_t slab_log2;
    int32_t nvals;      // distinct aggregator input columns carried in a record
    int32_t nmasks;     // distinct aggregator masks
    int32_t all_masked; // every aggregator has a mask: rows with no mask bit set emit no record
    int32_t use_flags;  // records carry a flags byte (bit m = mask m keeps the row)
    int32_t idx16;      // local index stored as uint16
    int32_t parts;      // pass-2 workgroups per slab
    int32_t rows_per_thread;
    int32_t no_pipeline; // debugging knob: use the non-pipelined pass-1 kernel
    int32_t scatter_lds_one; // bytes of ONE pass-1 LDS carve (the pipelined kernel uses two)
    uint64_t cap;       // queue capacity per slab (records)
    const void *vdata[VXH_PART_MAX_VALS];
    const uint8_t *mdata[VXH_PART_MAX_MASKS];
    uint8_t vdtype[VXH_PART_MAX_VALS], vflip[VXH_PART_MAX_VALS];
    uint8_t agg_vslot[VXH_MAX_AGG]; // 0xff: no input column
    uint8_t agg_mbit[VXH_MAX_AGG];  // 0xff: no mask
    unsigned long long *qcount;     // [S] records reserved
    unsigned long long *qlimit;     // [S] first reservation that did not fit (or ~0)
    void *qidx;                     // [S][cap] uint16 / uint32
    uint8_t *qflags;                // [S][cap]
    uint64_t *qval[VXH_PART_MAX_VALS]; // [S][cap]
    // slot-private partition accumulators, one per aggregator, laid out [part][slab][local] in the HBM cell type:
    // pass 2 flushes its LDS slab into acc[k] + (part*S + slab)*slab_cells with contiguous, exclusive
    // read-modify-writes; part_merge folds the parts into the aggregator's grid once per vxh_grid_bin call.
    // (Flushing straight into grid replicas touches one 64-128 B line per 8 B cell — the slabs interleave — which
    //  was ~130 us of every part_reduce launch: profiles/r01_chunk_fit.txt.)
    void *acc[VXH_MAX_AGG];
    int32_t val_i64; // the value column(s) are int64 and every aggregator counts or sums them into int64 cells: the records' payloads, the
                     // box's and pass 2's LDS sums are two's-complement integers (part_scatter_wv, part_reduce_fast, part_hot_merge)
    int32_t val_ct;  // part_scatter_wv: element type of the value column — 0: 8 bytes (float64, or int64 with val_i64), 1: float32 (widened to
                     // float64 on load: the records carry doubles), 2: int32 (sign-extended to int64 on load; val_i64 is set)
    int32_t bin_ct;  // part_scatter_wv: element type of the binner columns — 0: float64, 1: float32 (every one of them; widened on load like BinnerScalar<float>), 2: int64, 3: int32
    int32_t blk; // pass 1 = part_scatter_blk (block-reserved queues, 4096-row tiles)
    int32_t f32; // ... its float instantiation: every binner column and the value column float32 (the records carry float64 all the same)
    // pass 1 = part_scatter_wv (barrier-free, wave-private staging rings): wv = waves per workgroup (0: not this
    // kernel); each wave's LDS area of wv_wave_bytes starts at wv_base + wave * wv_wave_bytes
    int32_t wv, wv_base, wv_wave_bytes;
    // ... how its tiles are dealt: a workgroup takes SUPER-BLOCKS of wv x wv_span consecutive tiles (its waves side by side, wv_span trips
    // deep), the super-blocks go round robin over the workgroups.  1 = tiles dealt one by one over the whole launch (every trip of a wave is
    // 8 MiB further on in every column: a page the CU has not touched before); large = every workgroup walks its own contiguous range
    int32_t wv_span;
    int32_t qrec12;    // the queue holds 12-byte records {value bits, local index} in qidx (ring-less variant with one value column); qval unused
    uint64_t qsink;    // ring-less variant: record index of the first sink record (one per wave, 16 records apart) behind the sub-queues
    // wv_direct == 2 (one record stream per (workgroup, slab)): HBM copy of the block entries, [workgroup][slab][qbtab_stride] x 16 bytes
    // {record index base, block number | (epoch | slow << 31) << 32}; entries of earlier launches carry another epoch
    unsigned long long *qbtab;
    int32_t qbtab_stride, epoch;
    int32_t wv_direct; // part_scatter_wv without rings: cold records go from the registers straight to the queue blocks
    int32_t wv_phase;  // wv_direct == 4: the bit of the 100 MHz wall clock (s_memrealtime) whose flips are the chip's write bursts (13: every 82 us)
    // part_scatter_wv's queue layout: every (wave, slab) fills BLOCKS of qblk records that it reserves from the
    // sub-queue's counter one at a time (normally a single one per launch: qblk is sized for the wave's expected share),
    // and writes how many records each block really holds into qtab[sub * qtab_stride + block].  Pass 2 walks the
    // blocks of its sub-queue.  (qblk == 0: the older kernels' layout — one contiguous run of records per sub-queue.)
    int32_t qblk, qtab_stride;
    uint32_t *qtab;
    // wv_direct == 3: the queue is `parts` REGIONS of cap records (= cap / 64 groups) in qidx (uint16 local indices) and qval[0];
    // a pass-1 wave of workgroup b reserves blocks of qblk GROUPS from qcount[b % parts] (counted in groups), writes the groups a
    // block really holds to qtab[region * qtab_stride + block] and every group's header — byte s = records of slabs 0..s, byte
    // S - 1 = records of the group — to qhdr[region * (cap / 64) + group]
    unsigned long long *qhdr;
    // "hot box" (part_scatter_f64<2,1,4,0,HOT=true>): a w x h rectangle of cells — chosen from a sample of the
    // call's rows as the densest one that fits — is aggregated in LDS by pass 1 itself (fp64 sum + uint32 count per
    // cell); only rows outside it (and rows whose value is NaN) are emitted as records.  Each pass-1 workgroup
    // flushes its private box into hot_sum/hot_cnt[blockIdx][cell]; part_hot_merge folds them into the grids.
    struct HotBox {
        int32_t on;
        uint32_t x0, y0, w, h;     // in sub-index units of dims 0 and 1 (edge cells included)
        uint32_t lds_offset;       // of the box inside pass 1's dynamic LDS
        uint32_t mom2;             // the box also keeps the sum of squares (AggSumMoment with moment 2: var / std), LDS order: sum | sum2 | count
        double *sum_acc;           // [pass-1 workgroups][w*h]
        double *sum2_acc;          // ... (mom2)
        unsigned long long *cnt_acc;
        uint32_t cnt16;            // packed box counters (part_scatter_wv DIRECT = 1, one value column): 1 = uint16, two per LDS word (10-byte
                                   // cells), 2 = uint8, four per word (9-byte cells); a workgroup checks sum(counters) == hot rows it saw
                                   // whenever it flushes them and raises *overflow otherwise
        uint32_t flush_trips;      // uint8 counters: the workgroup flushes and clears them every `flush_trips` trips of its waves' tile loop
                                   // (2 tiles per wave and trip), so that no cell sees 256 rows in between
        unsigned int *overflow;
    } hot;
};

// LDS of part_scatter_blk ahead of the box: bucket counters, segment table, block tails, 4096-record staging
#define VXH_BLK_FIXED_LDS(NVAL, S) (((S) <= 64 ? 64 : 256) * 56 + 16 + 4096 * (8 * (NVAL) + 2 + 1))

// part_scatter_wv: ring depth per (wave, slab) and the flush granule (= records per reserved queue segment)
#define VXH_WV_D 128
#define VXH_WV_G 64
// LDS of ONE wave of part_scatter_wv: [value ring][index ring][S counters]
#define VXH_WV_SHARED_QB 1024u
#define VXH_WV_SHARED_NB 16u
#define VXH_WV_SHARED_LDS(S) ((((size_t)(S) * 4 + 15) & ~(size_t)15) + (size_t)(S) * VXH_WV_SHARED_NB * 16)
#define VXH_WV_WAVE_LDS_DIRECT(S) ((((size_t)(S) * 20) + 15) & ~(size_t)15)
// wv_direct == 3 (round 4, "grouped"): ONE record stream per wave.  Cold records are compacted (ballot / mbcnt) into a wave-private
// LDS ring of 2 x 64 records {value 8 B | flat cell index 4 B}; every 64 of them leave as one GROUP: sorted by slab inside the
// group, written as whole aligned lines (512 B of values + 128 B of uint16 local indices, non-temporal) plus an 8-byte header of
// the slabs' end offsets.  Pass 2 (part_reduce_grp) reads, of every group, the segment of its own slab.
#define VXH_WV_GROUP 64u
#define VXH_WV_HELD 16 /* wv_direct == 4: groups (of 64 records = 3 VGPRs) a wave may hold back between two write bursts */
#define VXH_WV_WAVE_LDS_GROUPED ((size_t)(2 * VXH_WV_GROUP) * 12 + 128) /* ring of 128 records + 16 group headers waiting for their line */
#define VXH_WV_WAVE_LDS(NVAL, S) ((((size_t)(S) * VXH_WV_D * (2 + 8 * (size_t)(NVAL)) + (size_t)(S) * 4) + 15) & ~(size_t)15)

// hot-box sample over COARSE cells (round 6): nseg evenly spaced segments of seg_rows rows of two float64 binner columns, counted into
// 2^cf x 2^cf blocks of grid cells — few enough to privatise in LDS (65 x 65 counters for a 259 x 259 grid)
struct HotSampleArgs {
    const double *x, *y;
    double vmin[2], scale[2], binsd[2];
    uint64_t bins[2];
    uint64_t length, seg_rows;
    uint32_t nseg, wgs_per_seg, csx, csy, cf;
    unsigned long long *out; // [csy][csx]
};
void vxh_launch_hot_sample(const HotSampleArgs &args, hipStream_t stream);

struct HotMergeArgs {
    uint32_t x0, y0, w, h, blocks, nagg;
    uint64_t stride_y;             // cells per step of dim 1
    int32_t atomic, val_i64; // val_i64: the box sums are int64 (PartArgs::val_i64)
    double *sum_acc, *sum2_acc;
    unsigned long long *cnt_acc;
    void *grid[VXH_MAX_AGG];       // replica 0 of every aggregator
    uint8_t takes_sum[VXH_MAX_AGG]; // 1: fp64 sum grid (+= box sum), 2: fp64 sum-of-squares grid (+= box sum2), 0: int64 count grid (+= box count)
};

struct PartMergeArgs {
    uint64_t cells, slab_cells;
    int32_t slab_log2, parts, nagg, atomic; // atomic: other slots may be adding into the same grids
    void *acc[VXH_MAX_AGG];
    void *grid[VXH_MAX_AGG];
    uint64_t ident[VXH_MAX_AGG]; // identity bit pattern of the HBM cell
    uint8_t kind[VXH_MAX_AGG], cell[VXH_MAX_AGG];
};

struct LaunchPlan {
    int strategy;
    int block;     // threads per workgroup
    int blocks;    // workgroups
    size_t lds_bytes;
    int use_replicas; // replicas [0, use_replicas) are written by this launch
    bool fast_vals;  // every aggregator input is float64 native (or absent)
    bool vals_i64;   // every aggregator input is int64 native (or absent; at least one), aggregators count / sum into int64 cells
    bool bin_f64;    // all binners scalar f64 native unmasked
    bool bin_f32;    // all binners scalar f32 native unmasked (and not in float32-scaling mode)
    bool bin_i64, bin_i32; // all binners scalar int64 / int32 native unmasked (part_scatter_wv converts them on load: PartArgs::bin_ct 2 / 3)
    bool vals_f32;   // every aggregator input is float32 native (or absent; at least one), aggregators count / sum / sum-moment into float64 cells
    bool vals_i32;   // every aggregator input is int32 native (or absent; at least one), aggregators count / sum into int64 cells
    bool key_i64;    // ONE ordinal binner over a native unmasked int64 column (groupby on an integer key)
    bool count_fast; // launch K1d (count_lds_f64) instead of bin_kernel<LDS>
    bool fast_f64; // all binners scalar f64 native unmasked, all aggregator inputs f64 native / absent
    bool fast_f32; // the same with float32 everywhere (part_scatter_blk<..., float>)
    int count_ct;  // dtype every (scalar, native, unmasked) binner column has, when that is one the count kernel K1d is instantiated for (VXH_F64 / F32 / I64 / I32), else -1
    const char *name;
};

// K1d (count_lds_f64) serves: LDS strategy, one slab, float64 fast path, 1..3 dims, ONE count(*) aggregator
inline bool vxh_count_fast(const BinArgs &a, const LaunchPlan &p) {
    return p.strategy == VXH_STRAT_LDS && p.count_ct >= 0 && a.slab_log2 == 0 && a.ndim >= 1 && a.ndim <= 3 && a.nagg == 1 &&
           a.a[0].kind == VXH_AGG_COUNT && a.a[0].data == nullptr;
}

// implemented in vxh_kernels.hip
// device-side selections (vxh_select.hip): terms `column <op> constant`, keep = bit (outcomes of the terms) of `truth`
#define VXH_SEL_MAX_TERMS 4
#define VXH_SEL_MAX_COLUMNS 4
struct SelTerm {
    int32_t column, op; // vxh_cmp
    int32_t is_int, pad;
    double value;
    int64_t ivalue;
};
struct SelArgs {
    const void *col[VXH_SEL_MAX_COLUMNS];
    uint8_t dtype[VXH_SEL_MAX_COLUMNS];
    int32_t nterms;
    uint32_t truth;
    SelTerm t[VXH_SEL_MAX_TERMS];
    const uint8_t *and_mask; // optional: keep only where this is non-zero too (missing values: vaex/cpu.py:770-784)
    uint8_t *out;
    uint64_t n;
    int32_t nsteps[VXH_SEL_MAX_TERMS];                       // > 0: term t compares the result of prog[t] (float64 columns only) instead of a column
    vxh_sel_step prog[VXH_SEL_MAX_TERMS][VXH_SEL_MAX_STEPS];
};
// AggFirst (first / last value per cell by an order column): vxh_kernels.hip first_pass<1..3>
struct FirstArgs {
    BinArgs A; // binners + n (the aggregator descriptors are unused)
    const void *val, *ord; // ord == nullptr: the row's index inside the call is its order (src/agg_first.cpp:136)
    const uint8_t *mask;   // keep-mask (1 = keep) or nullptr
    uint8_t val_dtype, ord_dtype, flip, invert;
    uint32_t mask_block;   // 0: mask[row]; 1024: mask[row % 1024], the reference's block-local mask index ("first_mask_block" knob)
    uint64_t stamp0;       // stamp of row 0: rows of earlier calls win ties
    uint64_t *key, *row, *value;   // per cell: sortable order key of the winner, its stamp (~0 = empty cell), its value (canonical bits)
    uint64_t *tmp_key, *tmp_row;   // per cell, this call only
};
void vxh_launch_first(const FirstArgs &args, hipStream_t stream);
// AggNUnique / AggList (vxh_api.hip: vxh_collect_*): every row of a call becomes a pair {canonical value bits, flat cell
// index}; rows that do not take part get cell 0xffffffff (sorted behind everything else and dropped)
struct CollectArgs {
    BinArgs A;               // binners + n
    const void *val;
    const uint8_t *data_mask;      // nullptr or per row: 0 = missing value (src/agg_nunique.cpp:72)  [list: != 1 ... see mode]
    const uint8_t *selection_mask; // nullptr or per row: 0 = the row is not looked at at all (:70)
    uint8_t val_dtype, flip;
    uint8_t mode;            // 0 nunique: NaN / missing rows only counted per cell; 1 list: keep-mask semantics of src/agg_list.cpp:98-118
    uint8_t drop_nan, drop_null; // list: NaN / missing rows are not even counted
    uint32_t mask_block;     // list only: 0 = data_mask[row]; 1024 = data_mask[row % 1024] (src/agg_list.cpp:103, the "first_mask_block" knob)
    uint64_t *out_val;       // [n]
    uint32_t *out_cell;      // [n]
    unsigned long long *null_rows, *nan_rows; // per cell
};
void vxh_launch_collect(const CollectArgs &args, hipStream_t stream);
void vxh_launch_pair_flags(const uint64_t *val, const uint32_t *cell, uint8_t *flags, uint64_t n, int distinct, hipStream_t stream);
void vxh_launch_cell_counts(const uint32_t *cell, uint64_t n, unsigned long long *counts, hipStream_t stream);
void vxh_launch_sel_eval(const SelArgs &args, hipStream_t stream);
#define VXH_PACK_MAX_KEYS 8
struct PackArgs {
    const void *col[VXH_PACK_MAX_KEYS];
    uint8_t dtype[VXH_PACK_MAX_KEYS];
    int32_t nkeys;
    int64_t min_value[VXH_PACK_MAX_KEYS], multiplier[VXH_PACK_MAX_KEYS];
    int64_t *out;
    uint64_t n;
};
void vxh_launch_pack_keys(const PackArgs &args, hipStream_t stream);
// a scalar binner column of any dtype / byte order / with a missing-value mask (1 = masked) as float64 — or float32 (out_f32: 8- / 16-bit
// integers, bool, float32: exactly representable) — NaN where masked
void vxh_launch_column_convert(const void *data, const uint8_t *mask, int dtype, int flip, uint64_t n, void *out, int out_f32, hipStream_t stream);
void vxh_launch_column_convert_i64(const void *data, int dtype, int flip, uint64_t n, void *out, hipStream_t stream);
// int64 codes of a group key with missing values / of a float key (vxh_code_column)
void vxh_launch_key_codes(const void *data, const uint8_t *mask, int dtype, int flip, uint64_t n, long long null_code, long long nan_code, long long *out, hipStream_t stream);
void vxh_launch_product_f64(const double *a, const double *b, double *out, uint64_t n, hipStream_t stream);

void vxh_launch_part_scatter(const PartArgs &args, const LaunchPlan &plan, int scatter_blocks, size_t scatter_lds, hipStream_t stream);
void vxh_launch_part_reduce(const PartArgs &args, const LaunchPlan &plan, hipStream_t stream);
bool vxh_part_reduce_is_fast(const PartArgs &args, const LaunchPlan &plan);
void vxh_launch_bin(const BinArgs &args, const LaunchPlan &plan, hipStream_t stream);
void vxh_launch_part_merge(const PartMergeArgs &args, hipStream_t stream);
void vxh_launch_hot_merge(const HotMergeArgs &args, hipStream_t stream);
void vxh_launch_fill(void *dst, uint64_t ncells, int cell, const void *value8, hipStream_t stream);
// dst[c] = fold(replica_0[c] .. replica_{R-1}[c]); replicas 1.. are reset to the identity
void vxh_launch_fold(void *grid, uint64_t cells, int replicas, int cell, int kind, const void *identity8, hipStream_t stream);
void vxh_launch_minmax_int(int dtype, int flip, const void *data, const uint8_t *mask, uint64_t n, long long *out2_dev, hipStream_t stream);
void vxh_launch_minmax(int dtype, int flip, const void *data, const uint8_t *mask, uint64_t n, double *out2_dev, hipStream_t stream);
size_t vxh_cell_size(int cell);
size_t vxh_lds_cell_size(int kind, int cell, int count16 = 0);

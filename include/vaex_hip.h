/*
 * vaex_hip.h — C-ABI of libvaexhip.so: MI355X (gfx950) kernels for vaex's N-d binned
 * statistics / groupby-aggregation hot path.
 *
 * This is the drop-in boundary.  Each entry point replaces one member of the reference's
 * native `vaex.superagg` (and, for the hash map, `vaex.superutils`) pybind11 surface; the
 * citation after each declaration is the reference code it stands in for (paths relative to
 * /root/reference/packages/vaex-core/).  A pybind11 shim (vaex_amd/csrc/superagg_module.cpp)
 * re-exposes these as the Python classes vaex looks up by name (vaex/utils.py:754-791), so
 * `vaex.superagg` can be swapped for `vaex_amd.superagg` with no change to vaex's Python
 * (INTEGRATION.md).
 *
 * Conventions
 *   - Plain C: opaque handles, pointers and sizes.  No torch / pybind types.
 *   - Every function returning `int` returns 0 on success, non-zero on failure; the message
 *     (same wording as the reference's std::runtime_error texts where one exists) is
 *     available from vxh_last_error() on the calling thread.
 *   - `thread` is the reference's per-thread slot index (agg_base.hpp:97-98): distinct slots
 *     may be driven concurrently from different host threads; each slot owns a HIP stream.
 *     ONE host thread at a time per slot.  Slots 0 .. VXH_AUX_SLOT - 1 are the indices of the host's
 *     worker pool (vaex: the executor's thread_index); VXH_AUX_SLOT .. VXH_AUX_SLOT + 7 are auxiliary
 *     slots for callers that run NEXT to that pool (the legacy statisticNd entry, called from pool
 *     threads in the same pass as aggregation task parts, serialises itself on the first of them).
 *   - Data pointers are BORROWED (src/agg_base.hpp:166-179: raw pointer, no incref).  `mem` says
 *     where the pointer lives, and that decides for how long it is borrowed:
 *       VXH_MEM_HOST   - host memory (numpy chunk).  vxh_grid_bin stages it through pinned
 *                        buffers + hipMemcpyAsync on the slot's stream; the host pointer is no
 *                        longer read once vxh_grid_bin returns (the reference's contract: the
 *                        caller keeps the array alive for the duration of the call, vaex/cpu.py:708-710).
 *       VXH_MEM_DEVICE - HBM-resident column (device pointer); used in place, nothing copied.
 *                        vxh_grid_bin returns as soon as its kernels are ENQUEUED on the slot's
 *                        stream, so the column is read AFTER the call has returned: it must stay
 *                        allocated and unmodified until the slot has drained — vxh_slot_wait(thread),
 *                        vxh_synchronize(), or a call that hands a result to the host (vxh_agg_result /
 *                        vxh_agg_host_view) has returned; vxh_slot_busy(thread) polls.  Handing the
 *                        block back to a stream-ordered caching allocator before that is a use
 *                        after free.  (The pybind11 shim does this bookkeeping for Python callers:
 *                        it keeps a reference to every device array it was given until the slot
 *                        that read it is idle.)
 *                        Ordering against the PRODUCER of a device column: the slot's work is
 *                        ordered after everything enqueued on the legacy default stream before the
 *                        call (and after every blocking stream through it).  A column written on a
 *                        NON-BLOCKING stream is ordered by the caller: vxh_slot_wait_stream(thread,
 *                        producer) before vxh_grid_bin, or vxh_slot_set_stream to run the slot on
 *                        that stream.
 *   - Grid cell layout: dim 0 fastest (src/agg.hpp:63-73); scalar binner = bins+3 cells
 *     [nan/masked, underflow, bin0..binN-1, overflow] (src/binners.cpp:13-59); ordinal binner
 *     = N+2 (+1) cells [0..N-1, (other), null, nan] (src/binner_ordinal.cpp:11-13, :178).
 */
#ifndef VAEX_HIP_H
#define VAEX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VXH_AUX_SLOT 256 /* first auxiliary thread slot (see `thread` above) */
#define VXH_ABI_VERSION 1

/* element types, in the order of src/create_alltypes.hpp; names as src/utils.hpp:32-90 */
typedef enum vxh_dtype {
    VXH_F64 = 0, VXH_F32 = 1, VXH_I64 = 2, VXH_I32 = 3, VXH_I16 = 4, VXH_I8 = 5,
    VXH_U64 = 6, VXH_U32 = 7, VXH_U16 = 8, VXH_U8 = 9, VXH_BOOL = 10, VXH_DTYPE_COUNT = 11
} vxh_dtype;

typedef enum vxh_agg_kind {
    VXH_AGG_COUNT = 0,      /* AggCount_<T>      src/agg_count.cpp:7-68    grid int64            */
    VXH_AGG_SUM = 1,        /* AggSum_<T>        src/agg_sum.cpp:131-147   grid upcast<T>        */
    VXH_AGG_SUM_MOMENT = 2, /* AggSumMoment_<T>  src/agg_sum.cpp:149-166   grid upcast<T>        */
    VXH_AGG_MIN = 3,        /* AggMin_<T>        src/agg_minmax.cpp:77-140 grid T, fill +inf/max */
    VXH_AGG_MAX = 4         /* AggMax_<T>        src/agg_minmax.cpp:7-75   grid T, fill -inf/min */
} vxh_agg_kind;

typedef enum vxh_mem { VXH_MEM_HOST = 0, VXH_MEM_DEVICE = 1 } vxh_mem;

typedef struct vxh_binner vxh_binner;
typedef struct vxh_grid vxh_grid;
typedef struct vxh_agg vxh_agg;
typedef struct vxh_hashmap vxh_hashmap;
typedef struct vxh_selection vxh_selection;

/* ---- library ------------------------------------------------------------------------- */
int vxh_abi_version(void);
/* message of the last failure on this thread ("" if none) */
const char *vxh_last_error(void);
/* number of visible HIP devices (0 and success when there is no GPU / no driver) */
int vxh_device_count(int *count);
/* select the device all subsequently created objects live on (one process per GPU) */
int vxh_set_device(int device);
/* block until all work enqueued by this library has finished */
int vxh_synchronize(void);
/* load the library's code objects and create thread slot 0 now instead of inside the first compute call (the reference has no
 * counterpart: a host that imports vaex.superagg has its machine code mapped by the dynamic loader) */
int vxh_warmup(void);
/* run slot `thread`'s work on a caller-owned hipStream_t (NULL = library-owned stream) */
int vxh_slot_set_stream(int thread, void *hip_stream);
/* lifetime of VXH_MEM_DEVICE pointers (see "Data pointers" above; the reference has no counterpart — its bin() is synchronous,
 * src/agg.hpp:84-137): *busy = 1 while work enqueued for slot `thread` may still read the caller's device columns */
int vxh_slot_busy(int thread, int *busy);
/* block until slot `thread` has drained: from here on the device columns it was given may be freed or overwritten */
int vxh_slot_wait(int thread);
/* order slot `thread`'s next work after everything enqueued so far on `producer_stream` (a hipStream_t: the stream that writes the
 * caller's device columns; only needed for non-blocking streams — the legacy default stream is ordered implicitly) */
int vxh_slot_wait_stream(int thread, void *producer_stream);
/* tuning knobs for experiments and tests ("strategy", "part_chunk", "parts", "wv", "wv_waves", "wv_waves_direct", "blk", "hot",
 * "hot_cache", "hot_min_pct", "hot_direct_pct", "count16", "stage_bytes", "feeder", "cache_bytes", ...) and the two switches that
 * switch two reference quirks OFF ("first_mask_block", "nunique_row_counts": see AggFirst / AggNUnique below; the defaults are
 * the reference's behaviour): the full list is the
 * if-chain of vxh_config_set in vaex_amd/csrc/vxh_api.hip; DESIGN.md §3 */
int vxh_config_set(const char *key, int64_t value);
int vxh_config_get(const char *key, int64_t *value);
/* name of the kernel variant the last vxh_grid_bin on `thread` launched (for tests / bench) */
const char *vxh_last_kernel(int thread);

/* ---- chunk feeder and device column cache ---------------------------------------------- */
/* vaex's executor hands the kernels one chunk (chunk_size rows, default 1 Mi) of every needed column at a time, from
 * memory-mapped / numpy memory, once per pass and again on every later pass over the same columns
 * (vaex/dataset.py:506-531 chunk_iterator; vaex/execution.py:283-292 one pass, :432-435 the pool's chunk loop).
 * Host arrays handed to vxh_grid_bin (VXH_MEM_HOST) go through a per-slot feeder: a ring of 3 device arenas (knob
 * "stage_bytes" each, grown on demand), the DMA on a copy stream of its own, event-chained to the slot's compute stream —
 * the copy of chunk i+1 overlaps the kernels of chunk i, vxh_grid_bin never waits for kernels, and the caller's arrays are
 * not read any more once it returns.  Knob "feeder": 1 (default) the engine reads the caller's memory; 2 the calling thread
 * first copies into page-locked buffers of the ring, so the call returns before the DMA ran (a CPU pass per chunk: slower
 * per thread, for callers whose threads must not wait for PCIe); 0 copies on the compute stream (no overlap inside a slot).
 *
 * vxh_cache_register declares [host, host+bytes) immutable until vxh_cache_unregister (a memory-mapped column, a numpy
 * column of a DataFrame): chunks inside such ranges are kept in HBM, keyed by (pointer, bytes), least-recently-used within
 * the knob "cache_bytes" (default 64 GiB of the 288), and the next pass over them runs at HBM speed.  With VXH_CACHE_PIN the
 * range is also page-locked (hipHostRegister), so the DMA engine reads it in place without the CPU copy into the ring.
 * *pinned (may be NULL) tells whether the range got page-locked. */
enum { VXH_CACHE_PIN = 1 };
int vxh_cache_register(const void *host, uint64_t bytes, int flags, int *pinned);
/* forget the range and free its chunks (waits for device work that may be reading them) */
int vxh_cache_unregister(const void *host);
/* free every cached chunk, keep the registrations */
int vxh_cache_clear(void);
/* out = {bytes cached, chunks cached, hits, misses, evictions, registered ranges} */
int vxh_cache_stats(uint64_t out[6]);

/* ---- binners ------------------------------------------------------------------------- */
/* BinnerScalar<T,…,FlipEndian>(threads, expression, vmin, vmax, bins) — src/binners.cpp:9-12, :97 */
int vxh_binner_scalar_create(int threads, int dtype, int flip_endian, double vmin, double vmax, uint64_t bins, vxh_binner **out);
/* the arithmetic of the legacy statisticNd<float> instead of BinnerScalar's (src/vaexfast.cpp:1185-1262: `T scales[]`, `(value -
 * minima[d]) * scales[d]` in float32): mode 1 = the product with the bin count in double (its 1-, 3+-dimensional and use_edges loops),
 * mode 2 = in float32 (its two-dimensional loop, `T scaled`, :1240-1246), 0 = off.  Used by vaex_amd.vaexfast.statisticNd_f4. */
int vxh_binner_scalar_set_f32_scaling(vxh_binner *binner, int mode);
/* BinnerOrdinal<T,…>(threads, expression, ordinal_count, min_value, allow_other, invert) — src/binner_ordinal.cpp:15-17, :217 */
int vxh_binner_ordinal_create(int threads, int dtype, int flip_endian, int64_t ordinal_count, int64_t min_value, int allow_other, int invert, vxh_binner **out);
/* BinnerHash<T>(threads, expression, hash_map) — src/binner_hash.cpp:13-20, :152.  Cells
 * [unknown key, bin0..binN-1, null]; the map must stay alive as long as the binner. */
int vxh_binner_hash_create(int threads, int dtype, vxh_hashmap *map, vxh_binner **out);
/* BinnerHash<T, ..., FlipEndian> with the reference's cells — src/binner_hash.cpp:13-20 (constructor: hash_bins = hashmap->size(),
 * missing_bin = null_index() + 1, nan_bin = nan_index() + 1), :23-69 (to_bins), :71 (shape() = hash_bins + 2); map_many:
 * src/hash_primitives.hpp:567-590.  Cells [invalid, ordinal 0 .. ordinal size-1, (never written)]: `size` counts the null key and NaN
 * (they are ordinals of the set), a masked row goes to null_index + 1 (cell 0 when the set holds no null), a NaN to nan_index + 1 when
 * the set saw one.  Keys the set does not hold (and NaN next to a set without one) go to cell 0 — the reference reads their -1 through
 * its unsigned index type and writes one cell past the grid (:36-40, :56-60); that is the one deliberate difference.  dtype: the 11
 * numeric dtypes (float keys are looked up by bit pattern, the way vaex_amd.hashset stores them); the map's ordinals are the public
 * ones (vxh_hashmap_set_public_ordinals). */
int vxh_binner_hash_create_ref(int threads, int dtype, int flip_endian, vxh_hashmap *map, uint64_t size, int64_t null_index, int64_t nan_index, vxh_binner **out);
/* BinnerScalar::copy / BinnerOrdinal::copy — src/binners.cpp:11, binner_ordinal.cpp:18 */
int vxh_binner_copy(const vxh_binner *binner, vxh_binner **out);
void vxh_binner_destroy(vxh_binner *binner);
/* shape(): bins+3 / N+2 / N+3 — src/binners.cpp:59, binner_ordinal.cpp:178 */
uint64_t vxh_binner_shape(const vxh_binner *binner);
/* set_data(thread, ar): n elements of the binner's dtype — src/binners.cpp:60-70 */
int vxh_binner_set_data(vxh_binner *binner, int thread, const void *data, uint64_t n, int mem);
/* set_data_mask(thread, ar): uint8, 1 = masked (numpy convention) — src/binners.cpp:75-82, :26 */
int vxh_binner_set_data_mask(vxh_binner *binner, int thread, const uint8_t *mask, uint64_t n, int mem);
/* clear_data_mask(thread) — src/binners.cpp:71-74 */
int vxh_binner_clear_data_mask(vxh_binner *binner, int thread);
/* data_length(thread) — src/binners.cpp:58 */
uint64_t vxh_binner_data_length(const vxh_binner *binner, int thread);

/* ---- grid ---------------------------------------------------------------------------- */
/* Grid(binners): shapes, strides (dim 0 stride 1), length1d — src/agg.hpp:57-74.
 * The grid keeps the binner handles (not copies); they must outlive it. */
int vxh_grid_create(vxh_binner *const *binners, int dimensions, vxh_grid **out);
void vxh_grid_destroy(vxh_grid *grid);
uint64_t vxh_grid_length1d(const vxh_grid *grid);
int vxh_grid_dimensions(const vxh_grid *grid);
int vxh_grid_shapes(const vxh_grid *grid, uint64_t *shapes_out);
int vxh_grid_strides(const vxh_grid *grid, uint64_t *strides_out);
/* Grid::bin(thread, aggregators, length) — src/agg.hpp:84-137: for `length` rows compute the
 * flat cell index from every binner's slot-`thread` data and feed every aggregator.  One
 * fused kernel pass for all aggregators (they share the index, like the 1024-row
 * indices1d block of the reference). */
int vxh_grid_bin(vxh_grid *grid, int thread, vxh_agg *const *aggs, int n_aggs, uint64_t length);

/* ---- aggregators --------------------------------------------------------------------- */
/* Agg{Count,Sum,SumMoment,Min,Max}_<T>(grid, grids, threads[, moment]) —
 * src/agg_count.cpp:201, agg_sum.cpp:218/:227, agg_minmax.cpp:146/:161.
 * `grids` only sizes the host-visible buffer ((grids, *shapes), src/agg_base.hpp:106-125);
 * on the device there is one logical grid plus internal replicas. */
int vxh_agg_create(int kind, int dtype, int flip_endian, vxh_grid *grid, int grids, int threads, uint32_t moment, vxh_agg **out);
void vxh_agg_destroy(vxh_agg *agg);
/* set_data(thread, ar, index) — src/agg_base.hpp:166-179 (index is ignored there too) */
int vxh_agg_set_data(vxh_agg *agg, int thread, const void *data, uint64_t n, int mem);
/* set_data_mask(thread, ar): uint8, 1 = KEEP — src/agg_base.hpp:134-147, agg_count.cpp:50 */
int vxh_agg_set_data_mask(vxh_agg *agg, int thread, const uint8_t *mask, uint64_t n, int mem);
/* clear_data_mask(thread) — src/agg_base.hpp:148-151 */
int vxh_agg_clear_data_mask(vxh_agg *agg, int thread);

/* ---- device-side selections -------------------------------------------------------------- */
/* The reference evaluates a selection ("(x > 0) & (v < 3.5)") with numpy on every chunk (vaex/execution.py:530-549,
 * vaex/scopes.py:138-177) and hands the boolean array to the aggregators as their data mask (vaex/cpu.py:740-784).
 * A vxh_selection carries the predicate instead: up to 4 terms `column <op> constant` over up to 4 columns, and a truth
 * table — bit b of `truth` says whether a row is kept when the terms' outcomes, term t in bit t, spell b (so any
 * combination of & | ~ of the terms is one 16-bit number).  The library evaluates it on the GPU into the keep-mask the
 * kernels read: no host pass over the chunk and no mask bytes over PCIe.  Comparisons follow numpy: NaN makes every
 * comparison false except !=; an integer column is compared exactly with an integer constant (is_int, ivalue) and as
 * float64 with a float constant.  Columns are native-endian. */
typedef enum vxh_cmp { VXH_CMP_LT = 0, VXH_CMP_LE = 1, VXH_CMP_GT = 2, VXH_CMP_GE = 3, VXH_CMP_EQ = 4, VXH_CMP_NE = 5 } vxh_cmp;
typedef struct vxh_sel_term {
    int32_t column; /* index into the selection's columns */
    int32_t op;     /* vxh_cmp */
    int32_t is_int; /* the constant is an integer: use ivalue for integer columns */
    int32_t reserved;
    double value;
    int64_t ivalue;
} vxh_sel_term;
int vxh_selection_create(int threads, int n_columns, const int *dtypes, int n_terms, const vxh_sel_term *terms, uint32_t truth, vxh_selection **out);
void vxh_selection_destroy(vxh_selection *selection);
/* Round 5 — arithmetic and virtual columns (`a*x + b > c`, `x**2 + y**2 < r`, a selection over `r = sqrt(x**2 + y**2)`): the LEFT side of
 * term `term` becomes an expression over the selection's float64 columns instead of one column — a postfix program of up to
 * VXH_SEL_MAX_STEPS steps evaluated per row in float64 (stack depth <= 4), its result compared with the term's constant.  Only operations
 * whose IEEE results are correctly rounded are offered (add, subtract, multiply, divide, negate, square = x*x as numpy's `x**2`, square
 * root, absolute value), so the mask equals what vaex's numpy evaluation of the same expression gives (vaex/scopes.py:138-177) bit for
 * bit; every column a program reads must be float64 (other dtypes follow numpy's promotion rules on the host: not offered).  Such a
 * selection is evaluated into its keep-mask by a pass of its own (never inside the binning kernels). */
typedef enum vxh_sel_op { VXH_SEL_COL = 0, VXH_SEL_CONST = 1, VXH_SEL_ADD = 2, VXH_SEL_SUB = 3, VXH_SEL_MUL = 4, VXH_SEL_DIV = 5, VXH_SEL_NEG = 6,
                          VXH_SEL_SQUARE = 7, VXH_SEL_SQRT = 8, VXH_SEL_ABS = 9,
                          /* round 6: comparisons of the two top entries (1.0 / 0.0): `x > y`, `x + y <= 2 * z` as `<program> != 0` */
                          VXH_SEL_LT = 10, VXH_SEL_LE = 11, VXH_SEL_GT = 12, VXH_SEL_GE = 13, VXH_SEL_EQ = 14, VXH_SEL_NE = 15 } vxh_sel_op;
#define VXH_SEL_MAX_STEPS 16
typedef struct vxh_sel_step {
    int32_t op;     /* vxh_sel_op */
    int32_t column; /* VXH_SEL_COL: index into the selection's columns */
    double value;   /* VXH_SEL_CONST */
} vxh_sel_step;
int vxh_selection_set_program(vxh_selection *selection, int term, int n_steps, const vxh_sel_step *steps);
/* the chunk of column `column` slot `thread` works on (borrowed like the aggregators' data: set_data, src/agg_base.hpp:166-179) */
int vxh_selection_set_data(vxh_selection *selection, int thread, int column, const void *data, uint64_t n, int mem);
/* the selection's keep-mask (1 = keep) of slot `thread`'s first n rows as bytes in caller-owned DEVICE memory (4-byte aligned, n bytes
 * rounded up to 4): one pass on the slot's stream over device-resident columns, enqueued — not waited for; what the slot is given
 * afterwards runs behind it.  For the passes that take a ready-made mask (vxh_groupby_run_kept, vxh_minmax): what vaex's numpy evaluation
 * hands them per chunk (vaex/execution.py:530-549) */
int vxh_selection_evaluate(vxh_selection *selection, int thread, uint64_t n, uint8_t *out_device);
/* attach (NULL: detach) a selection to an aggregator: rows are kept where the predicate holds AND the data mask, if one is
 * set, is non-zero.  The selection is borrowed and must outlive its use in vxh_grid_bin. */
int vxh_agg_set_selection(vxh_agg *agg, vxh_selection *selection);

/* ---- AggFirst ------------------------------------------------------------------------------ */
/* AggFirst_<T>_<T2>(grid, grids, threads, invert) — src/agg_first.cpp:7-163: per cell the value of the row whose order
 * value is smallest (invert: largest — vaex.agg.last), rows with a NaN value or order skipped, ties kept by the earlier
 * row.  set_data index 0 is the value column (dtype), index 1 the order column (dtype_order); without an order column a
 * row's order is its index inside the call (:136).  flip_endian applies to both.  merge() does not exist (the reference
 * throws, :42).  vxh_first_bin bins slot `thread` of the grid's binners like vxh_grid_bin.  vxh_first_result: values_out
 * (cells of dtype, empty cells read 99 like the reference's fill :22-28), masked_out (1 = empty cell), order_out (cells of
 * dtype_order; may be NULL).
 * Reference quirk, reproduced by default: the keep-mask (1 = keep) is read at `data_mask_ptr[j]` with j counted inside the
 * current 1024-row block of Grid::bin_ (src/agg_first.cpp:131 — every other aggregator reads `[j + offset]`), i.e. at
 * mask[row % 1024] of the call's mask: what the call means only for calls of <= 1024 rows.  vxh_config_set("first_mask_block",
 * 0) reads mask[row] instead (tests/test_gpu_first.py pins both against the reference's compiled class). */
typedef struct vxh_first vxh_first;
int vxh_first_create(int dtype, int dtype_order, int flip_endian, vxh_grid *grid, int grids, int threads, int invert, vxh_first **out);
void vxh_first_destroy(vxh_first *first);
int vxh_first_set_data(vxh_first *first, int thread, int index, const void *data, uint64_t n, int mem);
int vxh_first_set_data_mask(vxh_first *first, int thread, const uint8_t *mask, uint64_t n, int mem); /* NULL clears */
int vxh_first_bin(vxh_first *first, int thread, uint64_t length);
int vxh_first_result(vxh_first *first, void *values_out, uint8_t *masked_out, void *order_out);
/* dtype size * grids * cells: what vaex's memory check expects of the reference class (vaex/agg.py:300-318) */
size_t vxh_first_bytes_used(const vxh_first *first);

/* ---- AggNUnique / AggList -------------------------------------------------------------------- */
/* One collector object behind both: every vxh_collect_bin call appends its rows' {value, flat cell} pairs to a device array
 * (sorted and, for nunique, reduced to the distinct pairs whenever it has doubled, and when a result is asked for).
 *   mode 0 = AggNUnique_<T>(grid, grids, threads, dropmissing, dropnan) — src/agg_nunique.cpp:7-95: per cell the number of
 *     distinct values (+1 if a missing value was seen, +1 if a NaN was seen, unless dropped).  data mask: 0 = missing value;
 *     selection mask: 0 = the row is skipped (:70-73).  drop_a = dropmissing, drop_b = dropnan.
 *     Reference quirk, reproduced by default: dropmissing / dropnan subtract the number of missing / NaN ROWS of a cell
 *     (`count -= counter->null_count`, :31-34) — right only for cells with at most one such row;
 *     vxh_config_set("nunique_row_counts", 0) takes the ONE entry they occupy away instead.  Values are told apart by their
 *     bits (-0.0 and +0.0 are two, as for the reference's hash of the bits).
 *   mode 1 = AggList_<T>_<T2>(grid, grids, threads, dropnan, dropnull) — src/agg_list.cpp:7-128: per cell the values in row
 *     order, then its NaNs, then one slot per counted missing value (the reference leaves those slots uninitialised; here
 *     they read 0).  data mask: 1 = value present, 0 = missing (other bytes: the row is ignored, :103,:116).
 *     drop_a = dropnan, drop_b = dropnull.  The mask is read like AggFirst's: `data_mask_ptr[j]` inside the 1024-row block
 *     (:103) by default, mask[row] with "first_mask_block" = 0.
 * grids must be 1 (the reference's own restriction).  merge() is not offered on the class surface (the reference's are empty /
 * throw); ranks exchange their state through vxh_collect_pairs / vxh_collect_merge_pairs. */
typedef struct vxh_collect vxh_collect;
int vxh_collect_create(int mode, int dtype, int flip_endian, vxh_grid *grid, int grids, int threads, int drop_a, int drop_b, vxh_collect **out);
void vxh_collect_destroy(vxh_collect *collect);
int vxh_collect_set_data(vxh_collect *collect, int thread, const void *data, uint64_t n, int mem);
int vxh_collect_set_data_mask(vxh_collect *collect, int thread, const uint8_t *mask, uint64_t n, int mem);      /* NULL clears */
int vxh_collect_set_selection_mask(vxh_collect *collect, int thread, const uint8_t *mask, uint64_t n, int mem); /* NULL clears */
int vxh_collect_bin(vxh_collect *collect, int thread, uint64_t length);
/* int64 per cell, dim 0 fastest (get_result, src/agg_nunique.cpp:17-45) */
int vxh_collect_nunique_result(vxh_collect *collect, int64_t *out_cells);
/* offsets_out[cells + 1]; values_out NULL: only the offsets and *flat_length_out; else flat_length elements of dtype */
int vxh_collect_list_result(vxh_collect *collect, int64_t *offsets_out, void *values_out, uint64_t *flat_length_out);
/* The collector's state for an exchange between ranks (the cross-rank form of TaskPartAggregation.reduce, vaex/cpu.py:788-796 —
 * the reference's own merge of these two aggregators is empty / throws, src/agg_nunique.cpp:46-59, src/agg_list.cpp:45): the
 * compacted {canonical value bits, flat cell} pairs — nunique: the distinct ones; list: every kept value, cells ascending, row
 * order inside a cell — and the per-cell counts of missing / NaN rows (cells int64 each).  Call with values_out == NULL for *n_out. */
int vxh_collect_pairs(vxh_collect *collect, uint64_t *n_out, uint64_t *values_out, uint32_t *cells_out, int64_t *null_rows_out, int64_t *nan_rows_out);
/* appends pairs exported by a collector over the same grid (host arrays) and adds its missing / NaN row counts; list: the
 * appended values come BEHIND the present ones in every cell (merge in rank order = row order) */
int vxh_collect_merge_pairs(vxh_collect *collect, uint64_t n, const uint64_t *values, const uint32_t *cells, const int64_t *null_rows, const int64_t *nan_rows);

/* ---- row-wise helpers ---------------------------------------------------------------------- */
/* device memory for the helpers' results (a plain hipMalloc / hipFree) */
int vxh_device_alloc(size_t bytes, void **out);
void vxh_device_free(void *p);
/* packed key of a multi-key groupby: out[i] = sum_k (column_k[i] - min_value_k) * multiplier_k, int64 — what vaex's
 * GrouperCombined evaluates with numpy from its parents' ordinals (vaex/groupby.py:526-584 `_combine`: multipliers are the
 * cumulative products of the parents' group counts).  Integer columns (host or device), result on the device. */
int vxh_pack_keys(int n_keys, const void *const *columns, const int *dtypes, const int *mems, const int64_t *min_values, const int64_t *multipliers, uint64_t n, int64_t *out_device);
/* the column a groupby reads INSTEAD of a key with missing values / a float key / a value column with missing entries (round 6):
 *   VXH_CODE_KEY    out int64:   `null_code` where mask[i] == 1 (numpy's convention), else the value — integers sign- / zero-extended, bool 0 / 1,
 *                                float kinds: the bit pattern of the value as a double, `nan_code` for every NaN.  What the reference keeps in
 *                                ordered_set<T>'s null / NaN slots (src/hash_primitives.hpp:455-470, vaex/hash.py:179-190) becomes ordinary keys.
 *   VXH_CODE_VALUE  out float64: NaN where mask[i] == 1, else (double) value — count / sum / the moments skip NaN exactly as they skip a
 *                                masked row (src/agg_count.cpp:50-56, agg_sum.cpp:108-115).
 * data / mask: host or device (mask may be null); flip: byte-swapped elements; out: device, n elements. */
enum vxh_code_mode { VXH_CODE_KEY = 0, VXH_CODE_VALUE = 1 };
int vxh_code_column(int mode, int dtype, const void *data, int mem_data, const uint8_t *mask, int mem_mask, int flip, uint64_t n, int64_t null_code, int64_t nan_code, void *out_device);
/* out[i] = a[i] * b[i] (NaN where either is NaN): the pair columns of the legacy OP_COV statistic (src/vaexfast.cpp:1117-1153) */
int vxh_product_f64(const double *a, int mem_a, const double *b, int mem_b, uint64_t n, double *out_device);
/* bytes_used() = sizeof(grid_type) * grids * length1d — src/agg_base.hpp:29 (vaex/agg.py:311-318 checks it) */
size_t vxh_agg_bytes_used(const vxh_agg *agg);
/* dtype of one grid cell as exposed to the host (int64 for count, upcast<T> for sums, T for min/max) */
int vxh_agg_grid_dtype(const vxh_agg *agg);
int vxh_agg_grids(const vxh_agg *agg);
/* buffer protocol — src/agg_base.hpp:106-125: pointer to a host array of shape
 * (grids, *shapes), dim-0-fastest strides.  Brings the host copy up to date (device result
 * in grid 0, identity elsewhere); writes through the pointer are picked up by the next
 * vxh_grid_bin (vaex/cpu.py:658 seeds initial values this way). */
int vxh_agg_host_view(vxh_agg *agg, void **ptr_out);
/* merge(others) — src/agg_count.cpp:15-23, agg_sum.cpp:72-79, agg_minmax.cpp:19-26 */
int vxh_agg_merge(vxh_agg *agg, vxh_agg *const *others, int n_others);
/* get_result(): fold and copy the length1d cells of the result into out (host) —
 * src/agg_count.cpp:24-41, agg_sum.cpp:80-97, agg_minmax.cpp:27-43 */
int vxh_agg_result(vxh_agg *agg, void *out);
/* device pointer of the folded grid (length1d cells of the DEVICE cell type: int64 for count
 * and integer sums, double for float sums; min/max of <4-byte types are widened to 4 bytes)
 * — for an in-place RCCL all-reduce across ranks; call vxh_agg_device_touch afterwards. */
int vxh_agg_device_grid(vxh_agg *agg, void **dev_ptr_out, int *device_dtype_out);
int vxh_agg_device_touch(vxh_agg *agg);
/* reset to the initial fill (initial_fill — src/agg_count.cpp:11, agg_minmax.cpp:13-18) */
int vxh_agg_reset(vxh_agg *agg);

/* ---- hash map (group keys -> dense ordinals) ----------------------------------------- */
/* ordered_set<T>: src/hash_primitives.hpp:436-730 (update :98-295, add_new :471-479,
 * map_ordinal :611-691, key_array :303-328).  GPU open-addressing table, splitmix64 hash
 * (src/hash.hpp:40-45).  Ordinals are dense 0..count-1 in (nondeterministic) insertion order;
 * parity with the reference is per key.  dtype: any integer dtype (keys widened to int64). */
int vxh_hashmap_create(int dtype, uint64_t capacity_hint, vxh_hashmap **out);
void vxh_hashmap_destroy(vxh_hashmap *map);
/* update(keys): insert unseen keys (mask: 1 = null, counted once as the null ordinal) */
int vxh_hashmap_update(vxh_hashmap *map, const void *keys, const uint8_t *mask, uint64_t n, int mem);
/* ordered_set::create(keys): fill an EMPTY map so that keys[i] (distinct, host int64) gets ordinal i —
 * src/hash_primitives.hpp:486-537.  Deterministic ordinals: what ranks agree on before a grid all-reduce. */
int vxh_hashmap_set_keys(vxh_hashmap *map, const int64_t *keys, uint64_t n);
/* number of distinct keys (excluding null) — size() */
int vxh_hashmap_count(vxh_hashmap *map, int64_t *count_out);
/* whether a null was seen, and its ordinal (= count) — null_index() src/hash.hpp:337-353 */
int vxh_hashmap_null_index(vxh_hashmap *map, int64_t *index_out);
/* map_ordinal(keys): int64 ordinals, -1 for unknown keys; out lives where `mem` says */
int vxh_hashmap_map_ordinal(vxh_hashmap *map, const void *keys, uint64_t n, int mem, int64_t *out);
/* key_array(): the distinct keys ordered by ordinal, as int64, into host memory */
int vxh_hashmap_keys(vxh_hashmap *map, int64_t *keys_out);
/* the ordinals a BinnerHash on this map sees: public[i] (host, n entries) replaces the table's ordinal i; n = 0: the table's own.
 * The reference's ordered_set keeps the null key and NaN among its ordinals (add_null / add_nan, src/hash_primitives.hpp:455-470)
 * and fixes the ordinals of a set made by `create` (:486-537); vaex_amd.hashset keeps those on the host and hands them over here. */
int vxh_hashmap_set_public_ordinals(vxh_hashmap *map, const int64_t *public_ordinals, uint64_t n);

/* ---- legacy fused statistic (vaexfast.statisticNd) ------------------------------------ */
/* OP_MIN_MAX on a 0-d grid — src/vaexfast.cpp:1090-1101, :1198-1203 (used by df.minmax /
 * limits=None, vaex/dataframe.py:1520): out2 = {min, max} over non-NaN values (and, with
 * mask, rows whose mask byte is 1); {+inf, -inf} when empty.  Any dtype, computed in double. */
int vxh_minmax(int dtype, int flip_endian, const void *data, const uint8_t *mask, uint64_t n, int mem, double *out2);

/* exact int64 {min, max} of an integer column ({INT64_MAX, INT64_MIN} when empty): the range test of the
 * groupby "simplify to BinnerInteger" rule, vaex/groupby.py:263-272 */
int vxh_minmax_int(int dtype, int flip_endian, const void *data, const uint8_t *mask, uint64_t n, int mem, int64_t *out2);

/* the two scans a FIRST df.groupby(key).agg(value) over fresh device columns pays, in one pass over the 16 bytes of a row: the exact
 * {min, max} of an int64 key column (vxh_minmax_int: the "simplify to BinnerInteger" test, vaex/groupby.py:263-272) and the number of
 * NaN values of a float64 value column (whether count(value) can stand in for a group's presence, vaex/groupby.py:955-972).
 * Both columns in device memory, 16-byte aligned.  out3 = {key min, key max, NaN values}; {INT64_MAX, INT64_MIN, 0} when empty. */
int vxh_scan_key_value(const int64_t *keys, const double *values, uint64_t n, int64_t *out3);

/* ---- finishers on the device -------------------------------------------------------------- */
/* What vaex computes with numpy on the result grids, on the device grids instead: vaex/agg.py:403-416 (mean =
 * sum / count), :440-455 (variance = m2 / count - mean^2; std = sqrt), and the drop of empty groups of a groupby
 * (vaex/groupby.py:955-972: rows of the result = cells whose count is > 0, in cell order). */
typedef enum vxh_finish_op {
    VXH_FIN_COPY = 0, /* in0's cell as it is: float cells as double, signed integer cells as int64, unsigned as uint64 */
    VXH_FIN_MEAN = 1, /* in0 = sum, in1 = count */
    VXH_FIN_VAR = 2,  /* in0 = sum of squares, in1 = sum, in2 = count */
    VXH_FIN_STD = 3
} vxh_finish_op;
/* n_out result columns over the cells [first_cell, first_cell + n_cells) of 1-d... N-d grids taken flat:
 * out[j][r] = op[j](in0[j], in1[j], in2[j]) at the r-th kept cell; `present` (a count aggregator, or NULL = keep every
 * cell) selects the cells with present > 0, in cell order; index_out[r] (optional) = that cell's index relative to
 * first_cell.  out[j] / index_out: host arrays of n_cells 8-byte elements (pinned memory from vxh_host_alloc makes
 * the copy a plain DMA); *n_kept = rows written. */
int vxh_finish(int n_out, const int *ops, vxh_agg *const *in0, vxh_agg *const *in1, vxh_agg *const *in2, vxh_agg *present,
               uint64_t first_cell, uint64_t n_cells, void *const *out, int64_t *index_out, uint64_t *n_kept);
/* page-locked host memory (result columns, staging) */
int vxh_host_alloc(size_t bytes, void **out);
/* a whole pageable host array to device memory the caller owns, pushed by `threads` host threads (0 = 8) on streams of their own;
 * returns when the bytes are there.  For hosts that hold a whole column and want it in HBM for one call (the wrapped df.groupby,
 * vaex/dataframe.py:7133-7205, over plain numpy columns): the chunk passes of vaex/cpu.py:678-786 get their PCIe rate from the pool
 * threads copying chunks side by side, a single hipMemcpy of a pageable column does not.  No counterpart in the reference. */
int vxh_upload(const void *host, void *device, uint64_t bytes, int threads);
void vxh_host_free(void *ptr);

/* ---- hash groupby in one partitioned pass -------------------------------------------------- */
/* df.groupby(<integer key>).agg({v: count / sum / mean / var / std}) — what vaex computes in two passes
 * (ordered_set.update, vaex/hash.py:152-171 + src/hash_primitives.hpp:98-295; then map_ordinal :611-691 +
 * BinnerOrdinal + AggCount / AggSum / AggSumMoment, vaex/cpu.py:678-786) — as ONE radix-partitioned aggregation whose
 * hash table is probed in LDS (vaex_amd/csrc/vxh_groupby.hip).  Per distinct key: rows, and per value column the count,
 * sum and sum of squares of its non-NaN values; groups come back ascending by key (vaex/groupby.py sorts them). */
typedef struct vxh_groupby vxh_groupby;
typedef enum vxh_groupby_column_kind {
    VXH_GB_KEYS = 0,  /* int64 */
    VXH_GB_ROWS = 1,  /* int64: rows of the group (agg.count()) */
    VXH_GB_COUNT = 2, /* int64: non-NaN values (agg.count(v)) */
    VXH_GB_SUM = 3,   /* float64 */
    VXH_GB_SUM2 = 4,  /* float64: sum of squares (AggSumMoment, moment 2) */
    VXH_GB_MEAN = 5,  /* float64: sum / count                     (vaex/agg.py:403-416) */
    VXH_GB_VAR = 6,   /* float64: sum2 / count - mean^2           (vaex/agg.py:440-455) */
    VXH_GB_STD = 7
} vxh_groupby_column_kind;
/* keys: n elements of an integer dtype; values: n_values (1 or 2) float64 columns; mem as everywhere.  groups_hint:
 * expected number of distinct keys (0 = 2^20; only sizes the partition — a wrong hint costs a retry); max_groups: upper
 * bound for the result (0 = min(n, 2^26)).  Fails (non-zero, message in vxh_last_error) when the keys are too many
 * (> ~3.5e6) or too skewed for the LDS-partitioned path: the caller then takes ordered_set + BinnerHash. */
int vxh_groupby_run(int key_dtype, const void *keys, int n_values, const void *const *values, uint64_t n, int mem,
                    uint64_t groups_hint, uint64_t max_groups, vxh_groupby **out);
/* the same with a keep-mask over the call (one byte per row where the keys live, 1 = the row takes part; null = every row): a filtered
 * frame / a selection shared by every aggregation — vaex compacts the chunks of a filtered frame before its passes see them
 * (vaex/execution.py:515-523); here rows outside the mask leave no record and groups without a row inside it do not exist. */
int vxh_groupby_run_kept(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem,
                         uint64_t groups_hint, uint64_t max_groups, vxh_groupby **out);
/* ... with the key RANGE the caller measured (vxh_minmax_int; key_min > key_max: unknown = vxh_groupby_run_kept): when key_max - key_min
 * leaves, below the bucket bits of an invertible mix, a remainder of at most 32 bits, the pass moves 12-byte records {remainder,
 * value} instead of 16-byte {key, value} ones — a quarter off what it writes and reads back.  Keys outside the range: undefined groups. */
int vxh_groupby_run_ranged(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem,
                           uint64_t groups_hint, uint64_t max_groups, int64_t key_min, int64_t key_max, vxh_groupby **out);
/* ... with HEAVY keys named by the caller (n_heavy <= 128 distinct int64 values — the head of a Zipf law, a default / missing-value
 * key — found in a sample of the column; host array): every row of ONE key lands in ONE bucket of the partitioned pass, so a key with
 * a few per cent of the rows overflows its queue, and long before that its single reduce workgroup is the whole pass.  The reference
 * has no such cliff — ordered_set<T>::update keeps every key in its map (src/hash_primitives.hpp:471-479) and BinnerOrdinal adds a
 * heavy key's rows to one cell like any other (src/binner_ordinal.cpp:138-175) — so the listed keys are peeled off INSIDE the pass:
 * gb_scatter looks every row's key up in an LDS copy of the list, adds heavy rows to per-workgroup partials in LDS and leaves no
 * record for them; their totals join the result as ordinary groups.  A key that is listed but absent yields no group; a heavy key
 * that is not listed only costs time.  n_heavy = 0: vxh_groupby_run_ranged. */
int vxh_groupby_run_peeled(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem,
                           uint64_t groups_hint, uint64_t max_groups, int64_t key_min, int64_t key_max, const int64_t *heavy_keys, int n_heavy,
                           vxh_groupby **out);
/* ... with the filter given as TERMS over the call's own value columns — `df[df.v > 3].groupby(key, agg={'s': vaex.agg.sum('v')})`: vaex
 * evaluates the filter into a mask per chunk and compacts the chunk before its passes see it (vaex/dataframe.py _filter / vaex/execution.py:515-523,
 * expression evaluation vaex/scopes.py:138-177); here gb_scatter evaluates the terms on the payload words it loads anyway — no mask pass
 * in front, no keep byte per row written and read back.  Term t holds when `values[value_index] <op> constant` does (vxh_cmp; float64
 * comparison with numpy's NaN rules, as the binning kernels' fused selections); the row takes part when bit (outcomes of the terms,
 * term 0 = bit 0) of `truth` is set — and, if `keep` is given too, its byte is 1.  n_terms = 0: vxh_groupby_run_peeled. */
typedef struct vxh_groupby_term {
    int32_t value_index; /* 0 or 1: which of the call's value columns the term reads */
    int32_t op;          /* vxh_cmp */
    double constant;
} vxh_groupby_term;
int vxh_groupby_run_selected(int key_dtype, const void *keys, int n_values, const void *const *values, const uint8_t *keep, uint64_t n, int mem,
                             uint64_t groups_hint, uint64_t max_groups, int64_t key_min, int64_t key_max, const int64_t *heavy_keys, int n_heavy,
                             int n_terms, const vxh_groupby_term *terms, uint32_t truth, vxh_groupby **out);
/* the same aggregation over PARTIAL results (other chunks', other ranks'): host arrays of n partial groups */
int vxh_groupby_merge(int n_values, const int64_t *keys, const int64_t *rows, const int64_t *const *counts,
                      const double *const *sums, const double *const *sums2, uint64_t n, uint64_t groups_hint, vxh_groupby **out);
void vxh_groupby_destroy(vxh_groupby *g);
/* the heavy hitters of a device-resident integer key column, from a strided sample of `sample` rows (every (n / sample)-th row): the keys
 * holding at least `min_count` sampled rows, at most `max_keys` of them (the most frequent; ties by key), ascending, as int64 (a uint64
 * key by its bit pattern) — the list vxh_groupby_run_peeled takes.  The reference has no counterpart: its hash map keeps a heavy key like
 * any other (src/hash_primitives.hpp:471-479); the partitioned pass must know them beforehand.  out_keys: room for max_keys entries. */
int vxh_sample_heavy_keys(int key_dtype, const void *keys, uint64_t n, uint32_t sample, uint32_t min_count, int max_keys, int64_t *out_keys, int *n_out);
/* number of groups */
uint64_t vxh_groupby_size(const vxh_groupby *g);
/* one result column (vxh_groupby_column_kind; value_index selects the value column for COUNT..STD) into a host array of
 * vxh_groupby_size elements of 8 bytes */
int vxh_groupby_column(vxh_groupby *g, int value_index, int which, void *out_host);
/* diagnostics: 0 buckets, 1 LDS slots per bucket, 2 retries, 3 / 4 / 5 = ms of the scatter / reduce / sort kernels, 6 = 1 when the pass moved 12-byte records, 7 = heavy keys peeled inside the pass,
 * 8 = 1 when gb_reduce indexed its LDS table with the record's remainder (key ranges of <= 2^22 cells: no keys in the table, no probe),
 * 9 = 1 when gb_reduce probed its tag table (compact records with a remainder of < 32 bits: lines of four {tag, group id} entries, straight-line probe) */
int vxh_groupby_info(const vxh_groupby *g, int what, double *value_out);

/* ---- multi-GPU reduce ------------------------------------------------------------------ */
/* One process per GPU; rows are sharded, every rank bins its rows into private grids, and the ranks' grids are combined with
 * ONE RCCL all-reduce per grid over xGMI (ncclSum on int64 / uint64 / fp64 cells, ncclMin / ncclMax for AggMin / AggMax) —
 * the cross-rank form of Aggregator::merge (src/agg_count.cpp:15-23, agg_sum.cpp:72-79, agg_minmax.cpp:19-26), which the
 * reference drives over its thread grids in TaskPartAggregation.reduce (vaex/cpu.py:788-796); SURVEY section 8b's
 * `vxh_allreduce(aggs[], comm)`.  The host distributes rank 0's id by its own means (MPI, a socket, torch.distributed's store):
 *   rank 0: vxh_comm_unique_id(id); everybody: vxh_comm_init(n_ranks, rank, id, &comm) after vxh_set_device(local GPU).
 * vxh_allreduce folds the replicas and runs the collective on the library's slot-0 stream: stream-ordered with the binning
 * before it and with everything enqueued after it, no host-side stop; afterwards every rank's aggregators hold the global grids
 * (vxh_agg_result / get_result returns them). */
typedef struct vxh_comm vxh_comm;
#define VXH_COMM_ID_BYTES 128
int vxh_comm_unique_id(char *id_out /* [VXH_COMM_ID_BYTES] */);
int vxh_comm_init(int n_ranks, int rank, const char *id /* [VXH_COMM_ID_BYTES] */, vxh_comm **out);
void vxh_comm_destroy(vxh_comm *comm);
int vxh_comm_size(const vxh_comm *comm);
int vxh_comm_rank(const vxh_comm *comm);
int vxh_allreduce(vxh_agg *const *aggs, int n_aggs, vxh_comm *comm);

/* ---- profiling helpers ---------------------------------------------------------------- */
/* HIP events on slot `thread`'s stream: record start/stop around vxh_grid_bin calls, read ms */
int vxh_timer_start(int thread);
int vxh_timer_stop(int thread, float *elapsed_ms_out);
/* Stream time from vxh_timer_start to the end of the last KERNEL the calls in between enqueued on the slot's stream; what follows
 * on the stream (result columns crossing PCIe) is in vxh_timer_stop's figure only.  Call after vxh_timer_stop.  (Measurement only:
 * bench.py's `kernel_ms`; no counterpart in the reference.) */
int vxh_timer_kernels_ms(int thread, float *elapsed_ms_out);

#ifdef __cplusplus
}
#endif
#endif /* VAEX_HIP_H */

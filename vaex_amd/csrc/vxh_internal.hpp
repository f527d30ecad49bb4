// Internal host-side types of libvaexhip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "vxh_kernels.hpp"

#define VXH_MAX_SLOTS (VXH_AUX_SLOT + 8) // 0 .. VXH_AUX_SLOT - 1: the host pool's thread indices; the rest: auxiliary slots (include/vaex_hip.h)
#define VXH_STAGE_RING 3

void vxh_hip_check(hipError_t e, const char *what, const char *file, int line);
#define HIP_CHECK(expr) vxh_hip_check((expr), #expr, __FILE__, __LINE__)
void vxh_set_error(const std::string &msg);
int vxh_dtype_size(int dt);

// one registered array of a per-thread slot (src/agg_base.hpp:97-98: data_ptr[thread], data_size[thread])
struct SlotData {
    const void *ptr = nullptr;
    uint64_t n = 0;
    int mem = VXH_MEM_HOST;
};

struct vxh_binner {
    int kind = 0, dtype = 0, flip = 0, threads = 1;
    // scalar
    double vmin = 0, vmax = 1;
    uint64_t bins = 0;
    int f32mode = 0; // vxh_binner_scalar_set_f32_scaling
    // ordinal
    int64_t ordinal_count = 0, min_value = 0;
    bool allow_other = false, invert = false;
    // hash
    vxh_hashmap *map = nullptr;
    bool ref_cells = false;   // vxh_binner_hash_create_ref: the reference's cell layout
    uint64_t ref_size = 0;    // ... hashmap->size(): keys incl. the null key and NaN
    int64_t ref_null_bin = 0, ref_nan_bin = 0;
    std::vector<SlotData> data, mask;
};

struct vxh_grid {
    std::vector<vxh_binner *> binners;
    std::vector<uint64_t> shapes, strides;
    uint64_t length1d = 1;
};

enum { AUTH_NONE = 0, AUTH_DEVICE = 1, AUTH_HOST = 2 };

struct vxh_selection {
    int threads = 1, n_columns = 0, n_terms = 0;
    int dtype[4] = {0, 0, 0, 0};
    struct Term { int column, op, is_int; double value; int64_t ivalue; } term[4];
    uint32_t truth = 0;
    std::vector<SlotData> data[4]; // per column, per thread slot
    int nsteps[4] = {0, 0, 0, 0};  // > 0: the term's left side is this postfix program (vxh_selection_set_program)
    vxh_sel_step prog[4][VXH_SEL_MAX_STEPS];
};

struct vxh_agg {
    vxh_selection *selection = nullptr; // borrowed: vxh_agg_set_selection
    int kind = 0, dtype = 0, flip = 0;
    uint32_t moment = 0;
    vxh_grid *grid = nullptr;
    int grids = 1, threads = 1;
    int host_dtype = 0; // dtype of a cell as the host sees it (reference grid_type)
    int cell = 0;       // vxh_cell: device cell type
    uint64_t identity = 0;
    // device state
    void *dev = nullptr; // replicas x length1d cells
    int replicas = 1;    // allocated
    int used = 1;        // replicas [0, used) may hold data; the rest are identity
    bool folded = true; // replicas 1.. hold only the identity
    int init = 0;        // replicas [0, init) have been filled with the identity (the rest: on first use, vxh_grid_bin)
    int auth = AUTH_NONE;
    std::vector<unsigned char> mirror; // lazily allocated (grids, *shapes) host buffer
    std::mutex mutex;
    std::vector<SlotData> data, mask;
};

// AggFirst / "last" (src/agg_first.cpp): ONE device state for all thread slots — calls are serialised (`mutex`, and every
// call ends with a wait on its stream), so the `grids` of the reference only set the size vaex's memory check expects
struct vxh_first {
    int dtype = 0, dtype_order = 0, flip = 0, invert = 0;
    vxh_grid *grid = nullptr;
    int grids = 1, threads = 1;
    uint64_t *state = nullptr; // 5 planes of length1d words: key, row, value, tmp_key, tmp_row
    uint64_t stamp = 0;
    std::mutex mutex;
    std::vector<SlotData> data, order, mask;
};

// AggNUnique / AggList: the rows' {value, cell} pairs, kept on the device
struct vxh_collect {
    int mode = 0; // 0 nunique, 1 list
    int dtype = 0, flip = 0, drop_a = 0, drop_b = 0; // nunique: dropmissing, dropnan; list: dropnan, dropnull
    vxh_grid *grid = nullptr;
    int threads = 1;
    uint64_t *val = nullptr;   // [cap]
    uint32_t *cell = nullptr;  // [cap]
    uint64_t n = 0, cap = 0;   // pairs held / capacity
    uint64_t compact = 0;      // pairs [0, compact) are sorted by (cell, value) and, for nunique, distinct
    unsigned long long *null_rows = nullptr, *nan_rows = nullptr; // per cell
    std::mutex mutex;
    std::vector<SlotData> data, mask, selection;
};

struct Slot {
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // binner columns converted to float64 for the fast kernels (vxh_grid_bin, "convert_binners"): one grow-only buffer per dimension
    void *conv_buf[3] = {nullptr, nullptr, nullptr};
    size_t conv_cap[3] = {0, 0, 0};
    uint64_t conv_calls = 0; // calls that converted at least one column
    void *vconv_buf[2] = {nullptr, nullptr}; // VALUE columns converted to int64 / float64 for the fast paths (round 6)
    size_t vconv_cap[2] = {0, 0};
    uint64_t vconv_calls = 0;
    // Chunk feeder (host chunks of a vxh_grid_bin call): a ring of VXH_STAGE_RING device arenas per slot.  The DMA into
    // the arena runs on `copy_stream`, the kernels on `stream` wait for the `copied` event, and `done` (recorded behind the
    // kernels) frees the entry for re-use — so the copy of chunk i+1 overlaps the binning of chunk i within ONE slot and
    // vxh_grid_bin never waits for kernels.  cfg_feeder 1: the engine reads the caller's memory (pageable: pinned / staged
    // by the runtime, consumed when the copy call returns; page-locked: vxh_grid_bin waits for the copies before it
    // returns).  cfg_feeder 2: the calling thread copies the arrays into the entry's page-locked buffer first, so the call
    // returns before the DMA has run — for callers whose threads must not wait for PCIe (costs a CPU pass over the chunk).
    struct Stage {
        void *dev = nullptr;
        void *pinned = nullptr;
        size_t cap = 0;
        hipEvent_t done = nullptr;   // the kernels that read the arena have finished
        hipEvent_t copied = nullptr; // the DMA into the arena has finished
    } stage[VXH_STAGE_RING];
    hipStream_t copy_stream = nullptr;
    int cur = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    hipEvent_t t_lap = nullptr; // recorded behind the last KERNEL of a call (before its results cross PCIe): vxh_timer_kernels_ms
    bool lap_set = false;
    hipEvent_t after_null = nullptr; // order_after_producers()
    // partition strategy: two queue buffers so that pass 1 of chunk i+1 (on `stream`) overlaps pass 2 of chunk i (on `stream2`)
    struct PartBuf {
        void *scratch = nullptr;
        size_t cap = 0;
        hipEvent_t scattered = nullptr, reduced = nullptr;
        bool busy = false; // `reduced` has been recorded and not yet waited for by `stream`
    } part[2];
    hipStream_t stream2 = nullptr;
    unsigned part_next = 0;
    unsigned redo_count = 0; // vxh_grid_bin attempts that were thrown away and run again (packed box counters: a wrap, or a record that needed the slow path)
    void *qbtab = nullptr; // PartArgs::qbtab (shared-stream pass 1): entries are tagged with the launch's epoch, never cleared
    size_t qbtab_cap = 0;
    int32_t epoch = 0;
    // partition accumulators (PartArgs::acc): identity-filled when the layout signature changes, put back to the
    // identity by part_merge at the end of every vxh_grid_bin call
    void *acc = nullptr;
    size_t acc_cap = 0;
    uint64_t acc_sig = 0;
    void *acc_ptr[VXH_MAX_AGG] = {};
    // hot box of the current vxh_grid_bin call (PartArgs::hot) and its per-workgroup accumulators
    struct Hot {
        int nval = 1;      // value columns the box aggregates (1: fp64 sum + count per cell, 0: count only)
        bool cnt16 = false;    // packed box counters this call (exact: checked per workgroup, the call is redone with wider ones if one wrapped)
        int cnt_shift = 0;     // ... log2(counters per LDS word): 1 = uint16 (10-byte cells), 2 = uint8 (9-byte cells, flushed every flush_trips trips)
        uint32_t flush_trips = 0;
        int max_shift = 2;     // widest packing allowed for this call (the redo after a wrap lowers it)
        int key_max_shift = 2; // ... remembered with the sample (key_*): columns whose counters wrapped once do not try again
        unsigned int *flag = nullptr; // device word the workgroups raise
        bool mom2 = false; // ... and the fp64 sum of squares (an AggSumMoment with moment 2 among the aggregators: 20-byte cells)
        bool gen2 = false; // part_scatter_hot (vs the HOT instantiation of part_scatter_f64)
        bool wv = false;   // the box lives in part_scatter_wv
        int wv_waves = 0;
        bool on = false, last_on = false; // last_on: the most recent call used the box (reporting)
        uint32_t x0 = 0, y0 = 0, w = 0, h = 0;
        int blocks = 0;          // pass-1 workgroups (= accumulator blocks)
        void *acc = nullptr;     // [blocks][w*h] double sums, then [blocks][w*h] u64 counts
        size_t acc_cap = 0;
        int64_t wv_mode = 5;       // the "wv" mode of this call
        uint64_t acc_zero_sig = 0, acc_layout_sig = 0; // layout the accumulators are known to be all-zero for (0: not known) / layout of the current call
        void *sample = nullptr;  // cells x int64: count grid of the sample
        size_t sample_cap = 0;
        // the box of the previous sampled call, reused when the same columns are binned with the same limits again
        const void *key_ptr[2] = {nullptr, nullptr};
        double key_lim[6] = {0, 0, 0, 0, 0, 0};
        uint64_t key_len = 0;
        std::vector<int64_t> key_grid; // the sample's counts per cell — per 2^key_cf x 2^key_cf block of cells, [key_csy][key_csx] (valid while key_fraction >= 0)
        uint32_t key_cf = 0, key_csx = 0, key_csy = 0;
        uint64_t key_fine_cells = 0;   // cells of the grid the sample was taken for
        int64_t key_total = 0;
        // searched boxes for up to two LDS budgets (the ring-less pass 1 and part_scatter_blk leave the box different room)
        uint64_t key_cells[3] = {0, 0, 0};   // the box searches of this sample, by cell budget (uint8 / uint16 counters next to part_scatter_wv, part_scatter_blk)
        uint32_t key_box[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        double key_box_fraction[3] = {-1, -1, -1};
        int key_next = 0;
        double key_fraction = -1;
        double last_fraction = 0; // share of the sample inside the box (vxh_config_get("hot_fraction_ppm"))
    } hot;
    void *fin_buf = nullptr; // vxh_finish scratch (grow-only)
    void *sel_buf = nullptr; // keep-masks of device-side selections (grow-only; stream-ordered re-use)
    size_t sel_cap = 0;
    size_t fin_cap = 0;
    const char *last_kernel = "";
    unsigned pred_fused = 0, pred_materialized = 0; // launches that evaluated the call's shared selection in the binning kernel / got a mask from sel_eval instead
    int last_slabs = 0; // partition strategy, most recent chunk: slabs of pass 2
    int last_pass1 = 0; // partition strategy, most recent chunk: 0 part_scatter / part_scatter_f64, 1 part_scatter_blk, 2 part_scatter_wv, 3 its ring-less variant
};

struct Context {
    std::mutex mutex;
    bool initialised = false;
    int device = 0;
    int cus = 256;
    size_t max_lds = 65536;
    Slot *slots[VXH_MAX_SLOTS] = {};
    hipEvent_t reduced = nullptr; // recorded on slot 0's stream behind the latest vxh_allreduce
    bool reduced_set = false;
    // tuning knobs (vxh_config_set)
    int64_t cfg_strategy = VXH_STRAT_AUTO;
    int64_t cfg_replicas = 0; // 0 = auto
    int64_t cfg_block = 0;
    int64_t cfg_blocks = 0;
    int64_t cfg_wv_blocks = 0;    // part_scatter_wv: workgroups of the launch (0 = one per CU)
    int64_t cfg_count_box_pct = 90; // count(*) on a 2-D grid that needs packed uint16 LDS counters goes through the partition strategy's hot box instead when the box holds
                                    // at least this share of the sampled rows (0: never; vxh_grid_bin)
    int64_t cfg_convert_binners = 1 << 22; // scalar binner columns of dtypes the fast kernels do not read (int8 / int16 / unsigned / byte-swapped / masked ...) are converted
                                           // to float64 in a pass of their own for calls of at least this many device rows (0: never; they then take the generic kernels)
    int64_t cfg_wv_phase = 12;    // "wv" = 6: bit of the 100 MHz wall clock whose flips are the chip's write bursts (12: every 41 us — 11 / 12 / 13 / 14: 4.64 / 4.58 / 4.60 / 4.68 ms on the bench pass)
    int64_t cfg_stage_bytes = 64 << 20;
    int64_t cfg_feeder = 1;       // host chunks: 1 copy stream + arena ring, 2 the same through page-locked buffers (see Slot::Stage), 0 copies on the compute stream
    int64_t cfg_cache_bytes = 64ll << 30; // device column cache budget (only ranges registered with vxh_cache_register are cached)
    int64_t cfg_slab_log2 = -1;   // -1 = auto
    int64_t cfg_lds_replicas = 0; // 0 = auto
    int64_t cfg_nunique_row_counts = 1; // AggNUnique dropmissing / dropnan: 1 (default) = the reference's `count -= null_count` (rows, src/agg_nunique.cpp:31-34); 0 = one entry less for a cell that saw missing values / NaNs
    int64_t cfg_first_mask_block = 1024; // AggFirst / AggList keep-mask index: 1024 (default) = mask[row % 1024], what src/agg_first.cpp:131 / agg_list.cpp:103 do; 0 = mask[row] (what they mean)
    int64_t cfg_part_chunk = 1 << 28; // rows per partition chunk (scratch: ~2 x record bytes x this; larger chunks amortise the launches)
    int64_t cfg_parts = 0;        // pass-2 workgroups per slab (0 = auto)
    int64_t cfg_part_lds = 0;     // LDS bytes per pass-2 slab (0 = auto)
    int64_t cfg_blk = 1;           // second-generation pass 1 (part_scatter_blk) where its signature allows (0: part_scatter_f64)
    int64_t cfg_wv = 6;            // third-generation pass 1 (part_scatter_wv: barrier-free): 0 = never; >= 1: with wave-private rings wherever its
                                   // signature allows and no hot box is on.  Next to a hot box: 1 = part_scatter_blk, 2 = the rings, 3 = no rings, cold records
                                   // straight from the registers into per-(wave, slab) queue blocks, 4 = into per-(workgroup, slab) blocks (<= 16 slabs)
                                   // (profiles/r02_direct_ab.txt: 158 / 157 / 177 / 177 Grows/s on the bench pass); 5 (round 4, default) = compacted into a wave-private ring,
                                   // slab-sorted 64-record groups in ONE stream per wave + part_reduce_grp (<= 8 slabs, 8-byte columns; otherwise as 3);
                                   // 6 (round 5, default) = 5 with the groups' record stores held back in registers and issued in chip-wide bursts
                                   // on the flips of a wall-clock bit ("wv_phase"): 4.96 -> 4.60 ms on the bench pass
    int64_t cfg_wv_block = 0;      // ... records per queue block of a (wave, slab) (0 = sized from the expected share); tests force tiny blocks
    int64_t cfg_fuse_selection = 1; // a selection shared by every aggregator of a call, over one float64 column, is evaluated inside the binning kernels (0: always through sel_eval's byte mask)
    int64_t cfg_gb_compact = 1;    // fused hash groupby: 12-byte records when the measured key range allows (vxh_groupby_run_ranged)
    int64_t cfg_gb_tag = 1;       // fused groupby: gb_reduce's tag table for compact records whose remainder has < 32 bits (0: the 4-key-line probing table, for A/B runs)
    int64_t cfg_gb_key32 = 1;     // fused groupby: 32-bit keys in gb_reduce's probing table when the compact record's remainder has < 32 bits (0: 64-bit keys, for A/B runs)
    int64_t cfg_gb_direct_nb = 8; // ... log2 of the fewest buckets a direct-table pass takes (6 .. 10; 256 buckets: gb_scatter 5.4 -> 5.1 ms per 1e9 rows against 512, profiles/r06_groupby.txt)
    int64_t cfg_gb_direct = 1;    // fused groupby: key ranges of <= 2^22 cells index gb_reduce's LDS table with the record's remainder (0: the probing table, for A/B runs)
    int64_t cfg_gb_sets = 8;       // fused hash groupby: sets of record streams shared by the workgroups w % sets (8: one per XCD)
    int64_t cfg_gb_abl = 0;        // fused hash groupby, timing experiments only (GbArgs::abl)
    int64_t cfg_gb_load_pct = 50;  // fused hash groupby: target load of a bucket's LDS table when the bucket count is chosen
    int64_t cfg_hot_chunk_factor = 4; // rows per partition chunk next to a hot box = this x part_chunk
    int64_t cfg_part_cap = 0;      // ... records per sub-queue (0 = sized from the expected share); tests force tiny queues to reach the slow path
    int64_t cfg_wv_waves_grouped = 8; // ... waves per workgroup of the grouped variant ("wv" = 5): 1.5 KB of ring each
    int64_t cfg_wv_span = 16;      // ... consecutive trips a workgroup's waves take before the workgroup jumps to its next super-block (PartArgs::wv_span; 16: -1.1 ... -1.4 % on the bench pass, profiles/r04_headline_ab.txt)
    int64_t cfg_wv_waves_direct = 16; // ... waves per workgroup of the ring-less variant ("wv" = 3, next to a hot box)
    int64_t cfg_wv_waves = 8;      // ... waves per workgroup (one workgroup per CU); fewer when the rings would not fit
    int64_t cfg_hot = 1;           // hot box in pass 1 of the partition strategy (0 off)
    int64_t cfg_hot_flush_trips = 0; // tests: uint8 counters with this flush interval next to a FORCED box
    int64_t cfg_hot_cnt16 = 2;     // packed counters in the box next to the ring-less pass 1 (one value column): 1 = uint16 (20 % more cells than uint32),
                                   // 2 = uint8 where the sampled share of the fullest cell allows (33 % more), 0 = uint32
    int64_t cfg_hot_min_rows = 1 << 24; // calls shorter than this do not pay for the sample
    int64_t cfg_hot_coarse = 1;    // sample the hot box on 4 x 4 blocks of cells, privatised in LDS, one launch (0: the per-cell sample with device atomics, eight launches)
    int64_t cfg_hot_cache = 1;     // reuse the sampled box when the same columns are binned with the same limits again (0: sample every call)
    int64_t cfg_hot_min_pct = 10;  // use the box only when it catches at least this share of the sample (profiles/r02_box_share.txt: worth it from ~15 %)
    int64_t cfg_hot_direct_pct = 62; // ... and the ring-less pass 1 (scattered record stores) only from this share on; below it part_scatter_blk stages the records
    int64_t cfg_hot_box[4] = {0, 0, 0, 0}; // x0, y0, w, h override (w > 0) — tests / experiments
    int64_t cfg_scatter_wgs = 0;  // pass-1 workgroups per CU (0 = as many as LDS allows, at most 4)
    int64_t cfg_count_fast = 1;   // 0: keep count(*) passes on the generic bin_kernel (for A/B measurements)
    int64_t cfg_count16 = 1;      // packed 16-bit LDS counters for all-count passes: 0 off, 1 LDS strategy, 2 also partition pass 2
    int64_t cfg_part_overlap = 0; // pass 2 of chunk i on a second stream, overlapping pass 1 of chunk i+1
    int64_t cfg_no_pipeline = 0;  // 1: non-pipelined pass-1 kernel
    int64_t cfg_part_rows = 0;    // pass-1 rows per thread per tile: 8, 4 or 2 (0 = auto)
};

Context &ctx();
Slot &get_slot(int thread);
// Device scratch that outlives a call's objects: aggregator grids, groupby results.  hipMalloc / hipFree are host-blocking
// and cost 0.1-0.4 ms for the 100-256 MB a 1e6-cell aggregator used to ask for on EVERY df.groupby / df.count call (the
// stream sat idle meanwhile: 1.2-1.6 ms per 1e9-row call); freed blocks are kept in size classes (bounded)
// and handed out again.  The caller guarantees that nothing in flight still touches a block it frees.
void *vxh_pool_alloc(size_t bytes);
void vxh_pool_free(void *p);
void vxh_pool_trim(void);
// marks "the kernels of this call are all enqueued": what follows on the stream is result traffic (D2H)
void vxh_timer_lap(Slot &slot);
// Device-resident inputs (VXH_MEM_DEVICE) are usually produced by the caller's framework on the legacy default stream
// (torch's default stream on ROCm): make the slot's (non-blocking) stream wait for everything enqueued there so far, so
// that a kernel of this library never reads a column whose producer kernel is still running.  Callers that produce
// their columns on another stream hand that stream over with vxh_slot_set_stream.
void order_after_producers(Slot &slot);

// code-object preload hooks, one per translation unit (vxh_warmup)
void vxh_preload_kernels(void);
void vxh_preload_hashmap(void);
void vxh_preload_select(void);
void vxh_preload_finish(void);
void vxh_preload_groupby(void);

// hash map hooks (vxh_hashmap.hip)
int64_t vxh_hashmap_size_for_binner(vxh_hashmap *map);
void vxh_hashmap_fill_binner_desc(vxh_hashmap *map, BinnerDesc *bd);

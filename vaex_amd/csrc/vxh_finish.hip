// K6 — finishers on the device: the numpy that vaex runs on the result grids of a groupby
//   mean = sum / count, variance = m2 / count - mean^2, std = sqrt(variance)     (vaex/agg.py:403-416, :440-455)
//   groups whose count is 0 are dropped, the others keep their cell order       (vaex/groupby.py:955-972)
// restated as three small kernels over the folded device grids, so that what crosses PCIe after a 1e6-group
// aggregation is the finished columns of the groups that exist (into pinned host memory) instead of every
// primitive grid, and the host does not spend ~13 ms of numpy on 1e6-cell arrays (profiles/r01_configs.txt).
#include "vxh_internal.hpp"
#include <thread>
#include <string>
#include <vector>
#include <algorithm>

#include <algorithm>
#include <map>
#include <mutex>
#include <stdexcept>

namespace {

constexpr int FIN_MAX_OUT = 16;
constexpr int FIN_BLOCK = 256;
constexpr int FIN_PER_THREAD = 4;
constexpr int FIN_TILE = FIN_BLOCK * FIN_PER_THREAD;

struct FinArgs {
    int32_t n_out, has_present;
    uint64_t first, n;
    const void *present; // int64 count grid, or null
    int32_t op[FIN_MAX_OUT];
    const void *in[FIN_MAX_OUT][3];
    uint8_t cell[FIN_MAX_OUT][3];
    void *out[FIN_MAX_OUT]; // device, 8-byte elements
    long long *index_out;   // device
    unsigned int *block_count, *block_offset;
};

__device__ __forceinline__ double cell_as_f64(const void *g, int cell, uint64_t c) {
    switch (cell) {
    case VXH_CELL_F64: return ((const double *)g)[c];
    case VXH_CELL_F32: return (double)((const float *)g)[c];
    case VXH_CELL_I64: return (double)((const long long *)g)[c];
    case VXH_CELL_U64: return (double)((const unsigned long long *)g)[c];
    case VXH_CELL_I32: return (double)((const int *)g)[c];
    default: return (double)((const unsigned *)g)[c];
    }
}

// COPY keeps integers integers: float cells -> double bits, signed cells -> int64, unsigned cells -> uint64
__device__ __forceinline__ uint64_t cell_raw64(const void *g, int cell, uint64_t c) {
    switch (cell) {
    case VXH_CELL_F64: return ((const uint64_t *)g)[c];
    case VXH_CELL_F32: return (uint64_t)__double_as_longlong((double)((const float *)g)[c]);
    case VXH_CELL_I64: case VXH_CELL_U64: return ((const uint64_t *)g)[c];
    case VXH_CELL_I32: return (uint64_t)(long long)((const int *)g)[c];
    default: return (uint64_t)((const unsigned *)g)[c];
    }
}

__device__ __forceinline__ uint64_t finish_one(const FinArgs &F, int j, uint64_t c) {
    const int op = F.op[j];
    if (op == VXH_FIN_COPY) return cell_raw64(F.in[j][0], F.cell[j][0], c);
    if (op == VXH_FIN_MEAN) { // sum / count (numpy: 0/0 -> nan, x/0 -> +-inf)
        const double s = cell_as_f64(F.in[j][0], F.cell[j][0], c), n = cell_as_f64(F.in[j][1], F.cell[j][1], c);
        return (uint64_t)__double_as_longlong(s / n);
    }
    // VAR / STD: in0 = sum of squares, in1 = sum, in2 = count — the reference's operation order (vaex/agg.py:447-449)
    const double m2 = cell_as_f64(F.in[j][0], F.cell[j][0], c), s = cell_as_f64(F.in[j][1], F.cell[j][1], c), n = cell_as_f64(F.in[j][2], F.cell[j][2], c);
    const double mean = s / n;
    const double raw2 = m2 / n;
    const double var = raw2 - mean * mean;
    return (uint64_t)__double_as_longlong(op == VXH_FIN_VAR ? var : sqrt(var));
}

__device__ __forceinline__ bool cell_present(const FinArgs &F, uint64_t c) { return !F.has_present || ((const long long *)F.present)[c] > 0; }

__global__ void __launch_bounds__(FIN_BLOCK) fin_count(const FinArgs F) {
    __shared__ unsigned int s_sum[FIN_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FIN_TILE;
    unsigned int mine = 0;
#pragma unroll
    for (int u = 0; u < FIN_PER_THREAD; ++u) {
        const uint64_t i = base + (uint64_t)threadIdx.x * FIN_PER_THREAD + u;
        if (i < F.n && cell_present(F, F.first + i)) ++mine;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) F.block_count[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// exclusive scan of the block counts (<= a few thousand) by one workgroup; total into block_offset[nb]
__global__ void __launch_bounds__(1024) fin_scan(const unsigned int *count, unsigned int *offset, unsigned int nb) {
    __shared__ unsigned int s_wave[16];
    __shared__ unsigned int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (unsigned int b0 = 0; b0 < nb; b0 += 1024) {
        const unsigned int b = b0 + threadIdx.x;
        const unsigned int v = b < nb ? count[b] : 0u;
        unsigned int inc = v;
        const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int t = (unsigned int)__shfl_up((int)inc, off, 64);
            if ((int)lane >= off) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        unsigned int before = s_carry;
        for (unsigned int w = 0; w < wave; ++w) before += s_wave[w];
        if (b < nb) offset[b] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) offset[nb] = s_carry;
}

__global__ void __launch_bounds__(FIN_BLOCK) fin_emit(const FinArgs F) {
    __shared__ unsigned int s_wave[FIN_BLOCK / 64];
    const uint64_t base = (uint64_t)blockIdx.x * FIN_TILE;
    bool keep[FIN_PER_THREAD];
    unsigned int mine = 0;
#pragma unroll
    for (int u = 0; u < FIN_PER_THREAD; ++u) {
        const uint64_t i = base + (uint64_t)threadIdx.x * FIN_PER_THREAD + u;
        keep[u] = i < F.n && cell_present(F, F.first + i);
        mine += keep[u] ? 1u : 0u;
    }
    // position of this thread's first kept cell: block offset + kept cells of the threads before it (cell order)
    unsigned int inc = mine;
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int t = (unsigned int)__shfl_up((int)inc, off, 64);
        if ((int)lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned int pos = (F.has_present ? F.block_offset[blockIdx.x] : (unsigned int)base) + inc - mine;
    for (unsigned int w = 0; w < wave; ++w) pos += s_wave[w];
#pragma unroll
    for (int u = 0; u < FIN_PER_THREAD; ++u) {
        if (!keep[u]) continue;
        const uint64_t i = base + (uint64_t)threadIdx.x * FIN_PER_THREAD + u;
        for (int j = 0; j < F.n_out; ++j) ((uint64_t *)F.out[j])[pos] = finish_one(F, j, F.first + i);
        if (F.index_out) F.index_out[pos] = (long long)i;
        ++pos;
    }
}

// the two scans a first df.groupby(k).agg(v) over fresh device columns needs — the exact range of the int64 key (dense or scattered?
// vaex/groupby.py:263-272) and whether the float64 value column holds a NaN (then count(v) cannot stand in for the groups' presence) —
// in ONE pass over the 16 bytes of a row: 16-byte loads (two rows per lane and trip), wave reduction, three device atomics per wave
__global__ void __launch_bounds__(256) scan_key_value_kernel(const long long *keys, const double *vals, uint64_t n, long long *out3) {
    long long mn = 0x7fffffffffffffffll, mx = (long long)0x8000000000000000ull;
    unsigned long long nan = 0;
    const uint64_t pairs = n >> 1, stride = (uint64_t)gridDim.x * blockDim.x;
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        const ll2 k = __builtin_nontemporal_load((const ll2 *)keys + i);
        const d2 v = __builtin_nontemporal_load((const d2 *)vals + i);
        mn = k[0] < mn ? k[0] : mn; mx = k[0] > mx ? k[0] : mx;
        mn = k[1] < mn ? k[1] : mn; mx = k[1] > mx ? k[1] : mx;
        nan += (v[0] != v[0]) + (v[1] != v[1]);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long k = keys[n - 1];
        mn = k < mn ? k : mn; mx = k > mx ? k : mx;
        nan += vals[n - 1] != vals[n - 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long a = __shfl_down(mn, o, 64), b = __shfl_down(mx, o, 64);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
        nan += __shfl_down(nan, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(out3, mn);
        atomicMax(out3 + 1, mx);
        if (nan) atomicAdd((unsigned long long *)(out3 + 2), nan);
    }
}

struct HostCache {
    std::mutex mutex;
    std::multimap<size_t, void *> free_blocks; // by block size
    std::map<void *, size_t> size_of;          // every block this allocator handed out
    size_t cached_bytes = 0;
};
HostCache &host_cache() {
    static HostCache *hc = new HostCache(); // (never destroyed: blocks may be returned during interpreter shutdown)
    return *hc;
}

} // namespace

extern "C" {

// Page-locking memory costs milliseconds per call (32 MB of result columns: more than the kernels that fill them), so
// freed blocks are kept: a small size-bucketed cache, bounded at 2 GiB.
// whole-column upload: `threads` host threads (<= 16; 0 = 8) each push a contiguous slice of the pageable array across PCIe on a stream
// of their own, 32 MiB pieces at a time.  (A pageable hipMemcpy is bounded by the ONE host thread that feeds the runtime's staging
// buffers, ~15-25 GB/s here; vaex's chunk passes reach 55 GB/s because its pool threads copy their chunks concurrently — a caller that
// holds a whole column, like the wrapped df.groupby, gets the same rate this way.)
int vxh_upload(const void *host, void *device, uint64_t bytes, int threads) {
    try {
        if (!host || !device) throw std::runtime_error("vxh_upload: null pointer");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { (void)hipGetLastError(); throw std::runtime_error("vaex_hip: no HIP device available (libvaexhip has no CPU fallback)"); }
        int dev = 0;
        (void)hipGetDevice(&dev);
        // whatever the caller enqueued on its own streams for the destination (an allocator's fill, an earlier reader) comes first: the
        // copy threads' streams know nothing of them
        if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error("vxh_upload: the device reports an earlier error");
        const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(threads > 0 ? std::min(threads, 16) : 8), (bytes + (64u << 20) - 1) / (64u << 20)));
        const uint64_t slice = ((bytes + (uint64_t)nt - 1) / (uint64_t)nt + 4095) & ~(uint64_t)4095;
        std::vector<std::string> errors((size_t)nt);
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) {
            pool.emplace_back([&, t]() {
                try {
                    if (hipSetDevice(dev) != hipSuccess) throw std::runtime_error("hipSetDevice");
                    hipStream_t st = nullptr;
                    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) throw std::runtime_error("hipStreamCreate");
                    const uint64_t lo = std::min<uint64_t>(bytes, (uint64_t)t * slice), hi = std::min<uint64_t>(bytes, lo + slice);
                    hipError_t e = hipSuccess;
                    for (uint64_t o = lo; o < hi && e == hipSuccess; o += (32u << 20))
                        e = hipMemcpyAsync((char *)device + o, (const char *)host + o, (size_t)std::min<uint64_t>(32u << 20, hi - o), hipMemcpyHostToDevice, st);
                    if (e == hipSuccess) e = hipStreamSynchronize(st);
                    (void)hipStreamDestroy(st);
                    if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e));
                } catch (const std::exception &ex) { errors[(size_t)t] = ex.what(); }
            });
        }
        for (auto &th : pool) th.join();
        for (auto &er : errors) if (!er.empty()) throw std::runtime_error("vxh_upload: " + er);
    } catch (const std::exception &e) {
        vxh_set_error(e.what());
        return 1;
    }
    return 0;
}

int vxh_scan_key_value(const int64_t *keys, const double *values, uint64_t n, int64_t *out3) {
    try {
        (void)hipSetDevice(ctx().device);
        if (((uintptr_t)keys | (uintptr_t)values) & 15) throw std::runtime_error("vxh_scan_key_value: columns must be 16-byte aligned device arrays");
        Slot &slot = get_slot(0);
        order_after_producers(slot);
        long long init[3] = {INT64_MAX, INT64_MIN, 0};
        long long *dev = (long long *)vxh_pool_alloc(64);
        HIP_CHECK(hipMemcpyAsync(dev, init, sizeof(init), hipMemcpyHostToDevice, slot.stream));
        if (n) {
            const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((n / 2 + 255) / 256, (uint64_t)ctx().cus * 16));
            hipLaunchKernelGGL(scan_key_value_kernel, dim3(blocks), dim3(256), 0, slot.stream, (const long long *)keys, values, n, dev);
            HIP_CHECK(hipGetLastError());
        }
        vxh_timer_lap(slot);
        HIP_CHECK(hipMemcpyAsync(init, dev, sizeof(init), hipMemcpyDeviceToHost, slot.stream));
        HIP_CHECK(hipStreamSynchronize(slot.stream));
        vxh_pool_free(dev);
        out3[0] = init[0]; out3[1] = init[1]; out3[2] = init[2];
    } catch (const std::exception &e) {
        vxh_set_error(e.what());
        return 1;
    }
    return 0;
}

int vxh_host_alloc(size_t bytes, void **out) {
    try {
        (void)hipSetDevice(ctx().device);
        HostCache &hc = host_cache();
        const size_t want = std::max<size_t>(bytes, 1);
        {
            std::lock_guard<std::mutex> lock(hc.mutex);
            auto it = hc.free_blocks.lower_bound(want);
            if (it != hc.free_blocks.end() && it->first <= 2 * want + (1u << 20)) {
                void *p = it->second;
                hc.cached_bytes -= it->first;
                hc.free_blocks.erase(it);
                *out = p;
                return 0;
            }
        }
        const size_t size = (want + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        void *p = nullptr;
        HIP_CHECK(hipHostMalloc(&p, size, hipHostMallocDefault));
        {
            std::lock_guard<std::mutex> lock(hc.mutex);
            hc.size_of[p] = size;
        }
        *out = p;
    } catch (const std::exception &e) {
        vxh_set_error(e.what());
        return 1;
    }
    return 0;
}

void vxh_host_free(void *p) {
    if (!p) return;
    HostCache &hc = host_cache();
    {
        std::lock_guard<std::mutex> lock(hc.mutex);
        auto it = hc.size_of.find(p);
        if (it != hc.size_of.end() && hc.cached_bytes + it->second <= (2ull << 30)) {
            hc.free_blocks.emplace(it->second, p);
            hc.cached_bytes += it->second;
            return;
        }
        if (it != hc.size_of.end()) hc.size_of.erase(it);
    }
    (void)hipHostFree(p);
}

int vxh_finish(int n_out, const int *ops, vxh_agg *const *in0, vxh_agg *const *in1, vxh_agg *const *in2, vxh_agg *present, uint64_t first_cell, uint64_t n_cells,
               void *const *out, int64_t *index_out, uint64_t *n_kept) {
    try {
        if (n_out < 1 || n_out > FIN_MAX_OUT) throw std::runtime_error("vxh_finish: 1..16 result columns");
        (void)hipSetDevice(ctx().device);
        Slot &slot = get_slot(0);
        FinArgs F{};
        F.n_out = n_out;
        F.first = first_cell;
        F.n = n_cells;
        auto grid_of = [&](vxh_agg *a, const void **ptr, uint8_t *cell) {
            if (!a) throw std::runtime_error("vxh_finish: missing input aggregator");
            if (first_cell + n_cells > a->grid->length1d) throw std::runtime_error("vxh_finish: cell range outside the grid");
            void *p = nullptr;
            int dt = 0;
            if (vxh_agg_device_grid(a, &p, &dt) != 0) throw std::runtime_error(vxh_last_error());
            *ptr = p;
            *cell = (uint8_t)a->cell;
        };
        for (int j = 0; j < n_out; j++) {
            F.op[j] = ops[j];
            if (ops[j] < VXH_FIN_COPY || ops[j] > VXH_FIN_STD) throw std::runtime_error("vxh_finish: unknown op");
            grid_of(in0[j], &F.in[j][0], &F.cell[j][0]);
            if (ops[j] >= VXH_FIN_MEAN) grid_of(in1[j], &F.in[j][1], &F.cell[j][1]);
            if (ops[j] >= VXH_FIN_VAR) grid_of(in2[j], &F.in[j][2], &F.cell[j][2]);
        }
        if (present) {
            uint8_t cell = 0;
            grid_of(present, &F.present, &cell);
            if (cell != VXH_CELL_I64) throw std::runtime_error("vxh_finish: `present` must be a count aggregator");
            F.has_present = 1;
        }
        if (n_cells == 0) {
            *n_kept = 0;
            return 0;
        }
        const unsigned nb = (unsigned)((n_cells + FIN_TILE - 1) / FIN_TILE);
        // device scratch: result columns, kept cell indices, block counts / offsets (grow-only, per slot 0)
        const size_t col_bytes = (n_cells * 8 + 255) & ~(size_t)255;
        const size_t need = col_bytes * (size_t)(n_out + 1) + ((size_t)(2 * nb + 2) * 4 + 255);
        if (need > slot.fin_cap) {
            HIP_CHECK(hipStreamSynchronize(slot.stream));
            if (slot.fin_buf) (void)hipFree(slot.fin_buf);
            slot.fin_buf = nullptr;
            HIP_CHECK(hipMalloc(&slot.fin_buf, need));
            slot.fin_cap = need;
        }
        char *buf = (char *)slot.fin_buf;
        for (int j = 0; j < n_out; j++) F.out[j] = buf + col_bytes * (size_t)j;
        F.index_out = index_out ? (long long *)(buf + col_bytes * (size_t)n_out) : nullptr;
        F.block_count = (unsigned int *)(buf + col_bytes * (size_t)(n_out + 1));
        F.block_offset = F.block_count + nb;
        uint64_t kept = n_cells;
        if (F.has_present) {
            hipLaunchKernelGGL(fin_count, dim3(nb), dim3(FIN_BLOCK), 0, slot.stream, F);
            hipLaunchKernelGGL(fin_scan, dim3(1), dim3(1024), 0, slot.stream, F.block_count, F.block_offset, nb);
        }
        hipLaunchKernelGGL(fin_emit, dim3(nb), dim3(FIN_BLOCK), 0, slot.stream, F);
        HIP_CHECK(hipGetLastError());
        vxh_timer_lap(slot);
        if (F.has_present) {
            unsigned int total = 0;
            HIP_CHECK(hipMemcpyAsync(&total, F.block_offset + nb, 4, hipMemcpyDeviceToHost, slot.stream));
            HIP_CHECK(hipStreamSynchronize(slot.stream));
            kept = total;
        }
        for (int j = 0; j < n_out; j++)
            if (kept) HIP_CHECK(hipMemcpyAsync(out[j], F.out[j], kept * 8, hipMemcpyDeviceToHost, slot.stream));
        if (index_out && kept) HIP_CHECK(hipMemcpyAsync(index_out, F.index_out, kept * 8, hipMemcpyDeviceToHost, slot.stream));
        HIP_CHECK(hipStreamSynchronize(slot.stream));
        *n_kept = kept;
    } catch (const std::exception &e) {
        vxh_set_error(e.what());
        return 1;
    }
    return 0;
}

} // extern "C"

void vxh_preload_finish(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)scan_key_value_kernel);
    (void)hipGetLastError();
}

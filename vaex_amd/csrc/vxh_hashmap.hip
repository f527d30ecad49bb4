// K3 — GPU open-addressing hash map: group keys -> dense ordinals.
//
// Stands in for the part of vaex.superutils.ordered_set_<int> the groupby path uses
// (src/hash_primitives.hpp:436-730): update() = insert-or-get, map_ordinal() = lookup (-1 for
// unknown), key_array() = keys ordered by ordinal.  Same hash finaliser as the reference
// (splitmix64, src/hash.hpp:40-45) but ONE flat table in HBM instead of `nmaps` mutex-guarded
// hopscotch shards: slots are claimed with a 64-bit compare-and-swap on the key word and the
// winner draws the ordinal from a device counter, so ordinals are dense 0..count-1 in
// (nondeterministic) claim order — the reference's are shard-offset + shard-local insertion
// order, equally nondeterministic with threads; parity is per key.
//
// Layout: keys[cap] int64 (EMPTY = INT64_MIN), vals[cap] int64 ordinal (-1 = empty), cap a power
// of two, load kept <= 3/4 by host-side batching + 4x growth.  The key INT64_MIN itself and the
// null (masked) key are tracked in four side words next to the counter.
#include "vxh_internal.hpp"

#include <stdexcept>
#include <vector>

namespace {

constexpr long long EMPTY = (long long)0x8000000000000000ull;

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ long long load_key(const void *p, uint64_t i, int dt) {
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

// side words: [0] count  [1] null seen  [2] INT64_MIN key seen  [3] ordinal of INT64_MIN key  [4] table got too full
// Optimistic: rows are inserted until the table holds max_count keys; a row that would need a NEW slot beyond that
// raises side[4] and is skipped — the host then grows the table and re-runs the same (idempotent) call.
// U keys per lane per trip: the U first probes (random 8-byte reads of a table that lives in Infinity Cache / HBM) are
// in flight together; at load <= 1/4 most keys are settled by that probe and the rest finish in the sequential loop.
__global__ void __launch_bounds__(256) hm_insert(const void *data, int dt, const uint8_t *mask, uint64_t n, long long *keys, long long *vals, uint64_t hmask, unsigned long long *side, unsigned long long max_count) {
    constexpr int U = 4;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += U * stride) {
        long long key[U], first[U];
        uint64_t p[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = i0 + (uint64_t)u * stride;
            live[u] = i < n;
            key[u] = EMPTY;
            if (live[u]) {
                if (mask != nullptr && mask[i] == 1) {
                    if (side[1] == 0) atomicExch(&side[1], 1ull);
                    live[u] = false;
                } else {
                    key[u] = load_key(data, i, dt);
                    if (key[u] == EMPTY) {
                        if (side[2] == 0 && atomicCAS(&side[2], 0ull, 1ull) == 0ull) side[3] = atomicAdd(&side[0], 1ull);
                        live[u] = false;
                    }
                }
            }
            p[u] = splitmix64((uint64_t)key[u]) & hmask;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) first[u] = keys[p[u]]; // (dead lanes read a valid slot and ignore it)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u] || first[u] == key[u]) continue;
            uint64_t q = p[u];
            long long cur = first[u]; // may be a stale EMPTY (L1); the CAS below then returns the true owner
            for (;;) {
                if (cur == key[u]) break;
                if (cur == EMPTY) {
                    // agent-scope load: a plain load could be served from this CU's L1 forever (never refreshed by other CUs' atomics)
                    if (__hip_atomic_load(&side[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= max_count) {
                        if (side[4] == 0) atomicExch(&side[4], 1ull);
                        break;
                    }
                    long long old = (long long)atomicCAS((unsigned long long *)&keys[q], (unsigned long long)EMPTY, (unsigned long long)key[u]);
                    if (old == EMPTY) {
                        vals[q] = (long long)atomicAdd(&side[0], 1ull);
                        break;
                    }
                    if (old == key[u]) break;
                }
                q = (q + 1) & hmask;
                cur = keys[q];
            }
        }
    }
}

__global__ void __launch_bounds__(256) hm_lookup(const void *data, int dt, uint64_t n, const long long *keys, const long long *vals, uint64_t hmask, const unsigned long long *side, long long *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long long key = load_key(data, i, dt);
        long long ord = -1;
        if (key == EMPTY) {
            ord = side[2] ? (long long)side[3] : -1;
        } else {
            uint64_t p = splitmix64((uint64_t)key) & hmask;
            for (;;) {
                long long cur = keys[p];
                if (cur == key) { ord = vals[p]; break; }
                if (cur == EMPTY) break;
                p = (p + 1) & hmask;
            }
        }
        out[i] = ord;
    }
}

// insert keys[i] with ordinal i (ordered_set::create — src/hash_primitives.hpp:486-537): keys must be distinct
__global__ void __launch_bounds__(256) hm_insert_ordered(const long long *in_keys, uint64_t n, long long *keys, long long *vals, uint64_t hmask, unsigned long long *side) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long long key = in_keys[i];
        if (key == EMPTY) { side[2] = 1; side[3] = i; continue; }
        uint64_t p = splitmix64((uint64_t)key) & hmask;
        for (;;) {
            long long old = (long long)atomicCAS((unsigned long long *)&keys[p], (unsigned long long)EMPTY, (unsigned long long)key);
            if (old == EMPTY) { vals[p] = (long long)i; break; }
            if (old == key) break; // duplicate: first writer wins
            p = (p + 1) & hmask;
        }
    }
}

// re-insert every occupied slot of the old table into the new one, keeping its ordinal
__global__ void __launch_bounds__(256) hm_rehash(const long long *okeys, const long long *ovals, uint64_t ocap, long long *keys, long long *vals, uint64_t hmask) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < ocap; i += stride) {
        const long long key = okeys[i];
        if (key == EMPTY) continue;
        uint64_t p = splitmix64((uint64_t)key) & hmask;
        for (;;) {
            long long old = (long long)atomicCAS((unsigned long long *)&keys[p], (unsigned long long)EMPTY, (unsigned long long)key);
            if (old == EMPTY) { vals[p] = ovals[i]; break; }
            p = (p + 1) & hmask;
        }
    }
}

__global__ void __launch_bounds__(256) hm_collect(const long long *keys, const long long *vals, uint64_t cap, long long *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < cap; i += stride)
        if (keys[i] != EMPTY) out[vals[i]] = keys[i];
}

__global__ void hm_fill(long long *p, uint64_t n, long long v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

unsigned grid_for(uint64_t n) {
    uint64_t b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b == 0) b = 1;
    return (unsigned)b;
}

} // namespace

struct vxh_hashmap {
    int dtype = VXH_I64;
    uint64_t cap = 0;
    long long *keys = nullptr;
    long long *vals = nullptr;
    unsigned long long *side = nullptr; // 8 words on the device
    unsigned long long host_side[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // {key, ordinal} pairs, one 16-byte slot per table position: what a BinnerHash probes (ONE random line per probe
    // instead of one in keys[] and one in vals[]); rebuilt lazily after the table changed
    long long *packed = nullptr;
    uint64_t packed_cap = 0;
    bool packed_valid = false;
    // device ordinal -> the ordinal the owner of the map hands out (vxh_hashmap_set_public_ordinals: vaex_amd.hashset puts the null key
    // and NaN among the keys and fixes the ordinals of sets made by `create`); applied when the slots are packed.  Empty: identity
    long long *perm = nullptr;
    std::vector<int64_t> perm_host;
    // staging of host key / mask / ordinal chunks: grow-only, re-used by every update / map_ordinal / set_keys / keys call of
    // this map (all of them hold `mutex` and end with a wait on the stream, so one buffer is enough)
    void *scratch = nullptr;
    size_t scratch_cap = 0;
    std::mutex mutex;
};

static char *hm_scratch(vxh_hashmap *m, size_t bytes) {
    if (bytes > m->scratch_cap) {
        if (m->scratch) (void)hipFree(m->scratch);
        m->scratch = nullptr;
        m->scratch_cap = 0;
        const size_t cap = std::max<size_t>(bytes + bytes / 4, 1 << 20);
        HIP_CHECK(hipMalloc(&m->scratch, cap));
        m->scratch_cap = cap;
    }
    return (char *)m->scratch;
}

__global__ void hm_pack(const long long *keys, const long long *vals, uint64_t cap, long long *packed, const long long *perm, uint64_t perm_n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < cap; i += stride) {
        const long long v = vals[i];
        packed[2 * i] = keys[i];
        packed[2 * i + 1] = (perm && v >= 0 && (uint64_t)v < perm_n) ? perm[v] : v;
    }
}

static void hm_alloc_table(uint64_t cap, long long **keys, long long **vals, hipStream_t st) {
    HIP_CHECK(hipMalloc(keys, cap * 8));
    HIP_CHECK(hipMalloc(vals, cap * 8));
    hipLaunchKernelGGL(hm_fill, dim3(grid_for(cap)), dim3(256), 0, st, *keys, cap, EMPTY);
    hipLaunchKernelGGL(hm_fill, dim3(grid_for(cap)), dim3(256), 0, st, *vals, cap, -1ll);
}

static void hm_refresh(vxh_hashmap *m, hipStream_t st) {
    HIP_CHECK(hipMemcpyAsync(m->host_side, m->side, 64, hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
}

static void hm_grow(vxh_hashmap *m, uint64_t new_cap, hipStream_t st) {
    long long *nk, *nv;
    hm_alloc_table(new_cap, &nk, &nv, st);
    hipLaunchKernelGGL(hm_rehash, dim3(grid_for(m->cap)), dim3(256), 0, st, m->keys, m->vals, m->cap, nk, nv, new_cap - 1);
    HIP_CHECK(hipStreamSynchronize(st));
    (void)hipFree(m->keys);
    (void)hipFree(m->vals);
    m->keys = nk;
    m->vals = nv;
    m->cap = new_cap;
}

int64_t vxh_hashmap_size_for_binner(vxh_hashmap *m) {
    std::lock_guard<std::mutex> lock(m->mutex);
    return (int64_t)m->host_side[0];
}

void vxh_hashmap_fill_binner_desc(vxh_hashmap *m, BinnerDesc *bd) {
    std::lock_guard<std::mutex> lock(m->mutex);
    if (!m->packed_valid) {
        Slot &s = get_slot(0);
        if (m->packed_cap != m->cap) {
            HIP_CHECK(hipStreamSynchronize(s.stream));
            if (m->packed) (void)hipFree(m->packed);
            m->packed = nullptr;
            HIP_CHECK(hipMalloc(&m->packed, m->cap * 16));
            m->packed_cap = m->cap;
        }
        hipLaunchKernelGGL(hm_pack, dim3(grid_for(m->cap)), dim3(256), 0, s.stream, m->keys, m->vals, m->cap, m->packed, m->perm, (uint64_t)m->perm_host.size());
        HIP_CHECK(hipStreamSynchronize(s.stream)); // (the binner may be used from another slot's stream)
        m->packed_valid = true;
    }
    bd->hkeys = (const int64_t *)m->packed; // packed {key, ordinal} slots: hvals == nullptr says so
    bd->hvals = nullptr;
    bd->hmask = m->cap - 1;
    bd->bins = m->host_side[0];
    bd->null_bin = (int64_t)m->host_side[0] + 1;
    bd->hmin_ord = m->host_side[2] ? (int64_t)m->host_side[3] : -1;
    if (bd->hmin_ord >= 0 && (uint64_t)bd->hmin_ord < m->perm_host.size()) bd->hmin_ord = m->perm_host[(size_t)bd->hmin_ord];
}

extern "C" {

#define HM_BEGIN try {
#define HM_END                                                                                                         \
    }                                                                                                                  \
    catch (const std::exception &e) {                                                                                  \
        vxh_set_error(e.what());                                                                                       \
        return 1;                                                                                                      \
    }                                                                                                                  \
    return 0;

int vxh_hashmap_create(int dtype, uint64_t capacity_hint, vxh_hashmap **out) {
    HM_BEGIN
    if (dtype == VXH_F64 || dtype == VXH_F32 || dtype < 0 || dtype >= VXH_DTYPE_COUNT) throw std::runtime_error("hash map: only integer key dtypes are supported");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) { (void)hipGetLastError(); throw std::runtime_error("vaex_hip: no HIP device available (libvaexhip has no CPU fallback)"); }
    (void)hipSetDevice(ctx().device);
    Slot &s = get_slot(0);
    vxh_hashmap *m = new vxh_hashmap();
    m->dtype = dtype;
    uint64_t cap = 1ull << 22; // 64 MiB of keys + ordinals: up to ~3 M distinct keys before the first growth
    while (cap < capacity_hint * 2) cap <<= 1;
    m->cap = cap;
    hm_alloc_table(cap, &m->keys, &m->vals, s.stream);
    HIP_CHECK(hipMalloc(&m->side, 64));
    HIP_CHECK(hipMemsetAsync(m->side, 0, 64, s.stream));
    HIP_CHECK(hipStreamSynchronize(s.stream));
    *out = m;
    HM_END
}

void vxh_hashmap_destroy(vxh_hashmap *m) {
    if (!m) return;
    (void)hipDeviceSynchronize();
    (void)hipFree(m->keys);
    (void)hipFree(m->vals);
    if (m->packed) (void)hipFree(m->packed);
    if (m->perm) (void)hipFree(m->perm);
    if (m->scratch) (void)hipFree(m->scratch);
    (void)hipFree(m->side);
    delete m;
}

int vxh_hashmap_update(vxh_hashmap *m, const void *keys, const uint8_t *mask, uint64_t n, int mem) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    (void)hipSetDevice(ctx().device);
    Slot &s = get_slot(0);
    const size_t es = (size_t)vxh_dtype_size(m->dtype);
    const void *dkeys = keys;
    const uint8_t *dmask = mask;
    if (mem == VXH_MEM_DEVICE) order_after_producers(s);
    if (mem == VXH_MEM_HOST && n) {
        const size_t kbytes = (n * es + 255) & ~(size_t)255;
        char *tmp = hm_scratch(m, kbytes + (mask ? n : 0));
        HIP_CHECK(hipMemcpyAsync(tmp, keys, n * es, hipMemcpyHostToDevice, s.stream));
        dkeys = tmp;
        if (mask) {
            HIP_CHECK(hipMemcpyAsync(tmp + kbytes, mask, n, hipMemcpyHostToDevice, s.stream));
            dmask = (const uint8_t *)(tmp + kbytes);
        }
    }
    // whole array per launch; at most half full before, 3/4 full after; on overflow grow 4x and repeat (idempotent)
    for (int attempt = 0; n && attempt < 40; ++attempt) {
        if (m->host_side[0] * 2 > m->cap) hm_grow(m, m->cap * 4, s.stream);
        const unsigned long long max_count = m->cap / 4 * 3;
        HIP_CHECK(hipMemsetAsync(m->side + 4, 0, 8, s.stream));
        m->packed_valid = false;
        hipLaunchKernelGGL(hm_insert, dim3(grid_for(n)), dim3(256), 0, s.stream, dkeys, m->dtype, dmask, n, m->keys, m->vals, m->cap - 1, m->side, max_count);
        HIP_CHECK(hipGetLastError());
        hm_refresh(m, s.stream);
        if (!m->host_side[4]) break;
        hm_grow(m, m->cap * 4, s.stream);
    }
    HM_END
}

int vxh_hashmap_set_keys(vxh_hashmap *m, const int64_t *keys, uint64_t n) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    (void)hipSetDevice(ctx().device);
    if (m->host_side[0] != 0) throw std::runtime_error("hash map is not empty");
    if (n == 0) return 0;
    Slot &s = get_slot(0);
    while (m->cap < 2 * n) hm_grow(m, m->cap * 2, s.stream);
    long long *d = (long long *)hm_scratch(m, n * 8);
    HIP_CHECK(hipMemcpyAsync(d, keys, n * 8, hipMemcpyHostToDevice, s.stream));
    m->packed_valid = false;
    hipLaunchKernelGGL(hm_insert_ordered, dim3(grid_for(n)), dim3(256), 0, s.stream, d, n, m->keys, m->vals, m->cap - 1, m->side);
    HIP_CHECK(hipGetLastError());
    unsigned long long cnt = n;
    HIP_CHECK(hipMemcpyAsync(m->side, &cnt, 8, hipMemcpyHostToDevice, s.stream));
    hm_refresh(m, s.stream);
    HM_END
}

int vxh_hashmap_set_public_ordinals(vxh_hashmap *m, const int64_t *perm, uint64_t n) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    (void)hipSetDevice(ctx().device);
    Slot &s = get_slot(0);
    HIP_CHECK(hipStreamSynchronize(s.stream));
    if (m->perm) (void)hipFree(m->perm);
    m->perm = nullptr;
    m->perm_host.assign(perm, perm + (perm ? n : 0));
    if (!m->perm_host.empty()) {
        HIP_CHECK(hipMalloc(&m->perm, n * 8));
        HIP_CHECK(hipMemcpyAsync(m->perm, m->perm_host.data(), n * 8, hipMemcpyHostToDevice, s.stream));
        HIP_CHECK(hipStreamSynchronize(s.stream));
    }
    m->packed_valid = false;
    HM_END
}

int vxh_hashmap_count(vxh_hashmap *m, int64_t *count_out) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    *count_out = (int64_t)m->host_side[0];
    HM_END
}

int vxh_hashmap_null_index(vxh_hashmap *m, int64_t *index_out) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    *index_out = m->host_side[1] ? (int64_t)m->host_side[0] : -1;
    HM_END
}

int vxh_hashmap_map_ordinal(vxh_hashmap *m, const void *keys, uint64_t n, int mem, int64_t *out) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    (void)hipSetDevice(ctx().device);
    if (n == 0) return 0;
    Slot &s = get_slot(0);
    const size_t es = (size_t)vxh_dtype_size(m->dtype);
    const void *dkeys = keys;
    long long *dout = (long long *)out;
    if (mem == VXH_MEM_DEVICE) order_after_producers(s);
    if (mem == VXH_MEM_HOST) {
        const size_t kbytes = (n * es + 255) & ~(size_t)255;
        char *tmp = hm_scratch(m, kbytes + n * 8);
        HIP_CHECK(hipMemcpyAsync(tmp, keys, n * es, hipMemcpyHostToDevice, s.stream));
        dkeys = tmp;
        dout = (long long *)(tmp + kbytes);
    }
    hipLaunchKernelGGL(hm_lookup, dim3(grid_for(n)), dim3(256), 0, s.stream, dkeys, m->dtype, n, m->keys, m->vals, m->cap - 1, m->side, dout);
    HIP_CHECK(hipGetLastError());
    if (mem == VXH_MEM_HOST) {
        HIP_CHECK(hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, s.stream));
        HIP_CHECK(hipStreamSynchronize(s.stream));
    } else {
        HIP_CHECK(hipStreamSynchronize(s.stream)); // (device ordinals: the caller may read them on any stream — torch's, another slot's)
    }
    HM_END
}

int vxh_hashmap_keys(vxh_hashmap *m, int64_t *keys_out) {
    HM_BEGIN
    std::lock_guard<std::mutex> lock(m->mutex);
    (void)hipSetDevice(ctx().device);
    const uint64_t count = m->host_side[0];
    if (count == 0) return 0;
    Slot &s = get_slot(0);
    long long *d = (long long *)hm_scratch(m, count * 8);
    hipLaunchKernelGGL(hm_collect, dim3(grid_for(m->cap)), dim3(256), 0, s.stream, m->keys, m->vals, m->cap, d);
    HIP_CHECK(hipMemcpyAsync(keys_out, d, count * 8, hipMemcpyDeviceToHost, s.stream));
    HIP_CHECK(hipStreamSynchronize(s.stream));
    if (m->host_side[2]) keys_out[m->host_side[3]] = (int64_t)EMPTY; // the INT64_MIN key lives in the side words
    HM_END
}

} // extern "C"

void vxh_preload_hashmap(void) {
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)hm_fill);
    (void)hipGetLastError();
}

// kreeq_amd.hip -- host side of the C ABI (include/kreeq_amd.h) of the MI355X-native kreeq hot path: handle,
// table sizing, partition planning and kernel launches.  The kernels live in kq_kernels.h (+ kq_device.h,
// kq_partition.h).  gfx950 only; no CPU fallback anywhere in this library.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>       // device radix sort: only for the key order of kq_export (plumbing, not on the hot path)

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/kreeq_amd.h"
#include "kq_device.h"
#include "kq_partition.h"

using namespace kq;

#include "kq_kernels.h"

// ================================================================================================
// host side
// ================================================================================================

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPC(expr)                                                                                   \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(e_ == hipErrorOutOfMemory ? KQ_ERR_NOMEM : KQ_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                  \
    } while (0)

// A device buffer that is contiguous in the virtual address space and SCRAMBLED in the physical one: 2 MiB chunks (hipMemCreate)
// mapped in a permuted order.  Measured: the split level that writes a few thousand sub-bucket streams into a
// multi-GB array runs 15-20 % faster into such a buffer than into one hipMalloc block that happens to be physically contiguous
// (1.97 -> 1.58 ms per slice at 3 Gbp), and a hipMalloc block is or is not contiguous depending on what the process freed before --
// the source of the 1.65 / 2.03 ms cases of that kernel from run to run.  Falls back to hipMalloc if the mapping calls fail.
struct Scrambled {
    void* p = nullptr; size_t bytes = 0, chunk = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    bool vmm = false;
};
static void scr_free(Scrambled& b) {
    if (!b.p) return;
    (void)hipDeviceSynchronize();
    if (b.vmm) {
        (void)hipMemUnmap(b.p, b.hs.size() * b.chunk);
        for (auto& hnd : b.hs) (void)hipMemRelease(hnd);
        (void)hipMemAddressFree(b.p, b.hs.size() * b.chunk);
    } else (void)hipFree(b.p);
    b = Scrambled();
}
static bool scr_try_vmm(Scrambled& b, size_t bytes, int device) {
    static const size_t chunk_mb = getenv("KQ_SCRAMBLE_MB") ? (size_t)atoi(getenv("KQ_SCRAMBLE_MB")) : 2;      // 0: plain hipMalloc (A/B)
    if (!chunk_mb) return false;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) { (void)hipGetLastError(); return false; }
    const size_t chunk = ((chunk_mb << 20) + gran - 1) / gran * gran, n = (bytes + chunk - 1) / chunk;
    void* base = nullptr;
    if (hipMemAddressReserve(&base, n * chunk, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hs.reserve(n);
    bool ok = true;
    for (size_t i = 0; i < n && ok; ++i) { hipMemGenericAllocationHandle_t hnd; ok = hipMemCreate(&hnd, chunk, &prop, 0) == hipSuccess; if (ok) hs.push_back(hnd); }
    size_t mapped = 0;
    if (ok) {
        size_t mul = 7919;                                             // chunk i of the virtual range = physical chunk (i * mul + 13) mod n
        while (std::gcd(mul, n) != 1) ++mul;
        for (; mapped < n && ok; ++mapped) ok = hipMemMap((char*)base + mapped * chunk, chunk, 0, hs[(mapped * mul + 13) % n], 0) == hipSuccess;
        if (!ok) --mapped;
    }
    if (ok) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(base, n * chunk, &acc, 1) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        if (mapped) (void)hipMemUnmap(base, mapped * chunk);
        for (auto& hnd : hs) (void)hipMemRelease(hnd);
        (void)hipMemAddressFree(base, n * chunk);
        return false;
    }
    b.p = base; b.bytes = n * chunk; b.chunk = chunk; b.hs.swap(hs); b.vmm = true;
    return true;
}
static int scr_ensure(Scrambled& b, size_t need, int device) {
    if (b.bytes >= need) return KQ_OK;
    scr_free(b);
    const size_t sz = need + std::min<size_t>(need / 4, (size_t)256 << 20) + 4096;
    if (scr_try_vmm(b, sz, device)) return KQ_OK;
    if (hipMalloc(&b.p, sz) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return fail(KQ_ERR_NOMEM, "partition scratch allocation failed"); }
    b.bytes = sz; b.vmm = false;
    return KQ_OK;
}

struct kq_handle;
static void arena_release(kq_handle* h);
struct kq_handle {
    int device = 0, k = 0, map_count = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int n_cu = 256;
    Slot* slots = nullptr;
    uint64_t n_regions = 0;          // of the table's geometry (the scale of the hash -> region map)
    // window (multi-GPU shards, KQ_OPT_BUCKET_WINDOW): only the regions of the hash-prefix buckets [win_lo, win_hi) are
    // allocated; `slots` is the allocation, view().slots the address region 0 would have
    uint32_t win_lo = 0, win_hi = 256;
    bool windowed = false;
    HcSlot* hc = nullptr;
    uint64_t hc_cap = 0;
    uint32_t* rstart = nullptr;      // device: first top-32 hash value of every region (n_regions + 1 entries)
    DevState* st = nullptr;          // device
    DevState* st_host = nullptr;     // pinned mirror
    // scratch (grown on demand)
    void* scratch = nullptr; size_t scratch_bytes = 0;
    void* stage = nullptr; size_t stage_bytes = 0;     // device staging for host-buffer entry points
    uint64_t kmers_bound = 0;        // upper bound of instances inserted (sizing the side table)
    uint64_t used_bound = 0;         // upper bound of occupied slots (skips the state read-back)
    uint64_t hc_check_at = 0;        // instance count from which the side table's fill is looked at again (reserve)
    bool table_empty = true;         // nothing inserted since kq_create / kq_clear
    uint32_t mid_rps = 2048;         // KQ_OPT_NARROW_MID: regions per hash-prefix bucket from which the split gets a middle level
    int merge_path = 0;              // KQ_OPT_MERGE_PATH (of the destination handle): 0 auto, 1 per-entry atomics, 2 region by region
    int lookup_path = 0;             // KQ_OPT_LOOKUP_PATH: 0 auto, 1 direct (k_lookup), 2 partitioned (k_lookup_regions)
    bool slots_dirty = false;        // the slot array is logically empty but its memory is not initialised yet (lazy clear)
    bool trust_capacity = false;     // KQ_OPT_TRUST_CAPACITY: capacity_hint bounds the distinct k-mers
    int count_path = 0;              // KQ_OPT_COUNT_PATH: 0 auto, 1 direct (global atomics), 2 partitioned
    uint64_t slice_kmers = 1ull << 28;   // KQ_OPT_SLICE_KMERS
    bool slice_user = false;             // set explicitly: no automatic enlargement
    uint32_t filt_lo = 0, filt_hi = 0;   // KQ_OPT_COUNT_MAP_RANGE (set to [0, map_count) at creation)
    bool profile = false;                // KQ_OPT_PROFILE: HIP events around the stages of the partitioned count
    std::vector<std::pair<const char*, hipEvent_t>> marks;
    void* part = nullptr; size_t part_bytes = 0;       // partitioned path: record buffers + offsets
    Scrambled part2, fork_part2;                       // the second record array of large plans (the middle level's output), and the fork's
    // Fork / join of the slices of ONE count call (see count_seq_dev): consecutive slices run their partition stages on two
    // internal streams with a scratch set each, so that slice j+1's P1 fills the issue slots slice j's levels leave idle.
    // `stream` is where launches go (a fork stream while a forked slice is being enqueued), `base` the handle's stream
    // (its own or the caller's): everything that reads the table or the device state runs there, behind a join.
    hipStream_t base = nullptr;
    hipStream_t fork_stream[2] = {nullptr, nullptr};
    void* fork_part[2] = {nullptr, nullptr}; size_t fork_part_bytes[2] = {0, 0};      // [0] aliases part / part_bytes while a forked slice runs
    bool fork_busy[2] = {false, false};              // work enqueued on the fork stream that `base` has not been joined with
    bool fork_stale[2] = {false, false};             // a table pass was enqueued on `base` since the fork stream last waited for one
    static constexpr int EV_RING = 64;
    hipEvent_t ev_ring[EV_RING] = {};                // fork / join / pass events, used round-robin: an event is re-recorded only
    uint64_t ev_next = 0;                            // after the host has seen its previous use complete (ev_get)
    hipEvent_t ev_pass = nullptr;                    // the last table pass on `base` (forked slices write into the arena it read)
    int overlap = 1;                                 // KQ_OPT_OVERLAP
    int kernel_set = 0;                              // KQ_OPT_KERNEL_SET (measurement only)
    // KQ_OPT_COUNT_MAP_PASSES: count matrices of resident batches, made for all map ranges by the first pass that scans a slice
    struct HistKey { const void* ab; const void* pinv; uint64_t lead, len, er_lo, er_hi; uint32_t g1, n_rng; int k;
                     bool operator==(const HistKey& o) const { return ab == o.ab && pinv == o.pinv && lead == o.lead && len == o.len && er_lo == o.er_lo && er_hi == o.er_hi && g1 == o.g1 && n_rng == o.n_rng && k == o.k; } };
    struct HistEntry { HistKey key; unsigned long long* m1_all; size_t bytes; };
    std::vector<HistEntry> hist_cache;
    size_t hist_cache_bytes = 0;
    int map_passes = 1;
    // pending record sets (see "pending sets" below): region-sorted records of earlier slices / batches that have not
    // been applied to the table yet; one k_count_regions pass takes them all
    void* arena = nullptr; size_t arena_bytes = 0, arena_used = 0;
    Scrambled arena_scr;                 // owns `arena` when it is a scrambled buffer (KQ_SCRAMBLE_ARENA)
    P3Set* d_sets = nullptr;             // device array [P3_MAX_SETS]
    int n_pend = 0, pend_fmt = -1, pend_aux_fmt = 0;
    uint64_t pend_records = 0;           // upper bound of the records in the pending sets
    int64_t pend_budget = -1;            // KQ_OPT_PENDING_BYTES: -1 auto, 0 = apply every slice at once
    uint64_t table_passes = 0;           // k_count_regions passes so far
    // pipelined host ingest (kq_count_batch_async): copies on their own stream into a ring of device staging buffers
    static constexpr int IN_SLOTS = 3, IN_TICKETS = 512;
    hipStream_t copy_stream = nullptr;
    void* in_buf[IN_SLOTS] = {nullptr, nullptr, nullptr}; size_t in_bytes[IN_SLOTS] = {0, 0, 0};
    hipEvent_t in_consumed[IN_SLOTS] = {nullptr, nullptr, nullptr};      // the count that read the slot has been enqueued and finished
    std::vector<hipEvent_t> in_copied;   // ring of "copy of ticket t done" events
    uint64_t in_next = 0;
    std::mutex in_m;                     // kq_count_batch_async may be called from several threads: the calls take turns
    bool test_fail_plan = false;         // KQ_OPT_TEST_FAIL_PLAN: the next partition plan fails with KQ_ERR_NOMEM (failure-path tests)
    void* hot = nullptr; size_t hot_bytes = 0;         // k_count_regions' list of skewed regions

    TableView view() const {
        TableView v; v.slots = slots - (reg_lo() << REGION_SHIFT); v.n_regions = n_regions; v.hc = hc; v.hc_mask = hc_cap - 1; v.st = st; v.k = (uint32_t)k;
        v.reg_lo = reg_lo(); v.reg_hi = reg_hi();
        v.rps = k >= HI_K ? (uint32_t)(n_regions >> 8) : 0u;
        v.rstart = rstart;
        return v;
    }
    uint64_t reg_lo() const { return windowed ? (uint64_t)win_lo * (n_regions >> 8) : 0; }
    uint64_t reg_hi() const { return windowed ? (uint64_t)win_hi * (n_regions >> 8) : n_regions; }
    uint64_t n_alloc_regions() const { return reg_hi() - reg_lo(); }
    uint64_t n_slots() const { return n_alloc_regions() << REGION_SHIFT; }      // allocated slots
};

static void marks_reset(kq_handle* h);
static int flush_pending(kq_handle* h);

// ---- fork / join (count_seq_dev) ----------------------------------------------------------------------------------
// One event per use: the ring hands out an event whose previous record has completed (the host waits for it if it has
// not: that bounds how far the host runs ahead to EV_RING forks / joins).  Re-recording an event that a queued
// hipStreamWaitEvent still refers to is exactly what this avoids.
static int ev_get(kq_handle* h, hipEvent_t* out) {
    hipEvent_t& e = h->ev_ring[h->ev_next++ % kq_handle::EV_RING];
    if (!e) HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    else HIPC(hipEventSynchronize(e));
    *out = e;
    return KQ_OK;
}
// `base` waits for everything the fork streams have been given
static int join_forks(kq_handle* h) {
    for (int w = 0; w < 2; ++w) {
        if (!h->fork_busy[w]) continue;
        hipEvent_t e; int rc = ev_get(h, &e); if (rc) return rc;
        HIPC(hipEventRecord(e, h->fork_stream[w]));
        HIPC(hipStreamWaitEvent(h->base, e, 0));
        h->fork_busy[w] = false;
    }
    return KQ_OK;
}

// grid-stride kernels: at most `per_cu` workgroups per CU.  8 = what is resident (kernels that flush per-workgroup
// state at the end); the tile scanners take 32: the dispatcher balances the surplus (-4 % on k_lookup / k_count_direct)
static int grid_for(const kq_handle* h, uint64_t work_items, int per_block, int per_cu = 8) {
    uint64_t blocks = (work_items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)h->n_cu * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static void arena_release(kq_handle* h) {
    if (!h->arena) return;
    if (h->arena_scr.p) scr_free(h->arena_scr); else (void)hipFree(h->arena);
    h->arena = nullptr; h->arena_bytes = 0;
}
static int ensure_buf(void** p, size_t* have, size_t need) {
    if (*have >= need) return KQ_OK;
    if (*p) { HIPC(hipFree(*p)); *p = nullptr; *have = 0; }
    size_t sz = need + std::min<size_t>(need / 4, (size_t)256 << 20) + 4096;      // geometric growth for small buffers, bounded slack for multi-GB ones
    HIPC(hipMalloc(p, sz));
    *have = sz;
    return KQ_OK;
}

static void clear_words(kq_handle* h, void* p, uint32_t words_per_slot, uint64_t n_slots) {
    const uint64_t n_pairs = n_slots * words_per_slot / 2;      // both table sizes make this exact
    hipLaunchKernelGGL(k_clear_slots, dim3(grid_for(h, n_pairs, 256)), dim3(256), 0, h->stream, (ulonglong2*)p, words_per_slot, n_pairs);
}
// an empty main table is all zero bytes (w0 == 0 <=> free slot)
static void clear_main(kq_handle* h, Slot* p, uint64_t n_slots) { (void)hipMemsetAsync(p, 0, (size_t)n_slots * sizeof(Slot), h->stream); }
// lazy kq_clear: give the slot array its empty image now (every table user except the partitioned count)
static void materialize(kq_handle* h) {
    if (!h->slots_dirty) return;
    clear_main(h, h->slots, h->n_slots());
    h->slots_dirty = false;
}
// `alloc_regions` of the n_regions of the geometry get memory (all of them unless the table is a window)
static int alloc_main(kq_handle* h, uint64_t n_regions, uint64_t alloc_regions, Slot** out, uint32_t** rstart_out) {
    Slot* p = nullptr;
    uint32_t* rs = nullptr;
    size_t bytes = (size_t)(alloc_regions << REGION_SHIFT) * sizeof(Slot);
    HIPC(hipMalloc((void**)&p, bytes));
    if (hipMalloc((void**)&rs, (size_t)(n_regions + 1) * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(p); return fail(KQ_ERR_NOMEM, "region index allocation failed"); }
    clear_main(h, p, alloc_regions << REGION_SHIFT);
    hipLaunchKernelGGL(k_region_starts, dim3(grid_for(h, n_regions + 1, 256)), dim3(256), 0, h->stream, rs, n_regions);
    *out = p; *rstart_out = rs;
    return KQ_OK;
}
static int alloc_hc(kq_handle* h, uint64_t cap, HcSlot** out) {
    HcSlot* p = nullptr;
    size_t bytes = (size_t)cap * sizeof(HcSlot);
    HIPC(hipMalloc((void**)&p, bytes));
    clear_words(h, p, sizeof(HcSlot) / 8, cap);
    *out = p;
    return KQ_OK;
}

static int read_state_raw(kq_handle* h) {
    HIPC(hipMemcpyAsync(h->st_host, h->st, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    return KQ_OK;
}
static int read_state(kq_handle* h) {
    int frc = flush_pending(h);          // the device state speaks for everything counted so far
    if (frc) return frc;
    return read_state_raw(h);
}
static int check_errors(kq_handle* h) {
    int rc = read_state(h);
    if (rc) return rc;
    if (h->st_host->err_table_full) return fail(KQ_ERR_TABLE_FULL, "k-mer table region overflow (slots %llu / %llu)",
                                                 (unsigned long long)h->st_host->slots_used, (unsigned long long)h->n_slots());
    if (h->st_host->err_hc_full) return fail(KQ_ERR_TABLE_FULL, "high-copy side table overflow (%llu / %llu)",
                                              (unsigned long long)h->st_host->hc_used, (unsigned long long)h->hc_cap);
    return KQ_OK;
}

// grow (rehash) so that `extra` more distinct k-mers fit at load <= 0.85, and the side table can
// take every k-mer that may reach cov >= 255 after `extra_instances` more instances.
static int grow_main(kq_handle* h, uint64_t need_slots) {
    int frc = flush_pending(h);          // pending records are sorted by the regions of the current geometry
    if (frc) return frc;
    uint64_t want = h->n_slots(), new_regions = h->n_regions;       // doubling keeps every multiple the geometry (and a window) needs
    while (want < need_slots) { want *= 2; new_regions *= 2; }
    if (new_regions >= (1ull << 32)) return fail(KQ_ERR_TABLE_FULL, "cannot grow k-mer table to %llu slots: region index overflow", (unsigned long long)want);
    Slot* fresh = nullptr;
    uint32_t* fresh_rs = nullptr;
    int rc = alloc_main(h, new_regions, want >> REGION_SHIFT, &fresh, &fresh_rs);
    if (rc) return rc == KQ_ERR_NOMEM ? fail(KQ_ERR_TABLE_FULL, "cannot grow k-mer table to %llu slots: out of device memory",
                                             (unsigned long long)want) : rc;
    Slot* old = h->slots; const uint64_t n_old = h->n_slots();
    const TableView old_view = h->view();
    uint32_t* old_rs = h->rstart;
    h->slots = fresh; h->n_regions = new_regions; h->rstart = fresh_rs;
    if (h->slots_dirty) h->slots_dirty = false;     // lazily cleared table: nothing to move, and the fresh array is clean
    else hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, n_old, 256)), dim3(256), 0, h->stream, h->view(), old_view);
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipFree(old));
    if (old_rs) HIPC(hipFree(old_rs));
    return KQ_OK;
}
static int grow_hc(kq_handle* h, uint64_t need) {
    uint64_t want = h->hc_cap;
    while (want < need) want *= 2;
    HcSlot* fresh = nullptr;
    int rc = alloc_hc(h, want, &fresh);
    if (rc) return rc;
    HcSlot* old = h->hc; const uint64_t n_old = h->hc_cap;
    h->hc = fresh; h->hc_cap = want;
    HIPC(hipMemsetAsync(&h->st->hc_used, 0, sizeof(unsigned long long), h->stream));
    hipLaunchKernelGGL(k_rehash_hc, dim3(grid_for(h, n_old, 256)), dim3(256), 0, h->stream, h->view(), old, n_old);
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipFree(old));
    return KQ_OK;
}

constexpr uint64_t HC_CAP_BIG = 1ull << 25;      // side-table floor of jobs beyond 2^31 instances (2.7 GB)
// Make room for a batch that may add `extra` distinct k-mers and `extra_instances` instances.
//  main table : load <= 0.85 even if every k-mer of the batch is new (unless the caller vouched for
//               capacity_hint with KQ_OPT_TRUST_CAPACITY); host-side upper bounds avoid a device round trip.
//  side table : #k-mers with cov >= 255 <= instances / 255; kept at load <= 0.5 up to 2^23 entries
//               (1.2 GB); beyond that growth follows the observed fill.
static int reserve(kq_handle* h, uint64_t extra, uint64_t extra_instances) {
    int rc;
    bool state_read = false;
    if (!h->trust_capacity) {
        uint64_t need = (uint64_t)((double)(h->used_bound + extra) / 0.85) + REGION_SLOTS;
        if (need > h->n_slots()) {
            rc = read_state(h); if (rc) return rc;
            state_read = true;
            h->used_bound = h->st_host->slots_used;
            need = (uint64_t)((double)(h->used_bound + extra) / 0.85) + REGION_SLOTS;
            if (need > h->n_slots()) { rc = grow_main(h, need); if (rc) return rc; }
        }
    }
    h->used_bound += extra;
    h->kmers_bound += extra_instances;
    // side table: provable sizing (one entry per 255 instances, load <= 0.5) up to 2^23 entries; beyond that it
    // follows the observed fill with room for an eightfold growth of the high-copy set (as coverage doubles, the copy
    // number that reaches cov 255 halves): looked at once per doubling of the instance count here, and before every
    // table pass over pending records (flush_pending).  A side table that fills up anyway is reported as
    // KQ_ERR_TABLE_FULL by the next synchronising call
    const uint64_t bound = h->kmers_bound / 255 + 1;
    uint64_t need_hc = 2 * std::min<uint64_t>(bound, 1ull << 23);
    if (bound > (1ull << 23)) need_hc = HC_CAP_BIG;
    if (bound > (1ull << 23) && h->kmers_bound >= h->hc_check_at && h->n_pend == 0) {
        if (!state_read) { rc = read_state_raw(h); if (rc) return rc; }
        need_hc = std::max<uint64_t>(need_hc, 8 * h->st_host->hc_used);
        h->hc_check_at = 2 * h->kmers_bound;
    }
    if (need_hc > h->hc_cap) { rc = grow_hc(h, need_hc); if (rc) return rc; }
    return KQ_OK;
}

// aligned view of a device byte string: 16-byte aligned base + lead
static inline void aligned_view(const char* d, const uint8_t** ab, uint64_t* lead) {
    uintptr_t p = (uintptr_t)d;
    *lead = p & 15;
    *ab = (const uint8_t*)(p - *lead);
}

static int stage_in(kq_handle* h, const void* host, size_t bytes, void** dev) {
    int rc = ensure_buf(&h->stage, &h->stage_bytes, bytes + 64);
    if (rc) return rc;
    if (bytes) HIPC(hipMemcpyAsync(h->stage, host, bytes, hipMemcpyHostToDevice, h->stream));
    *dev = h->stage;
    return KQ_OK;
}

// sort by key with a few host threads: chunk sorts, then pairwise merges
static void parallel_sort_entries(kq_entry* a, uint64_t n) {
    auto less = [](const kq_entry& x, const kq_entry& y) { return x.key < y.key; };
    unsigned hw = std::thread::hardware_concurrency();
    unsigned t = std::max(1u, std::min(16u, hw ? hw : 1u));
    while (t > 1 && n / t < (1u << 16)) t /= 2;
    if (t <= 1) { std::sort(a, a + n, less); return; }
    std::vector<uint64_t> cut(t + 1);
    for (unsigned i = 0; i <= t; ++i) cut[i] = n * i / t;
    {
        std::vector<std::thread> th;
        for (unsigned i = 0; i < t; ++i) th.emplace_back([&, i] { std::sort(a + cut[i], a + cut[i + 1], less); });
        for (auto& x : th) x.join();
    }
    for (unsigned width = 1; width < t; width *= 2) {
        std::vector<std::thread> th;
        for (unsigned i = 0; i + width < t; i += 2 * width) {
            const uint64_t lo = cut[i], mid = cut[i + width], hi = cut[std::min(i + 2 * width, t)];
            th.emplace_back([=] { std::inplace_merge(a + lo, a + mid, a + hi, less); });
        }
        for (auto& x : th) x.join();
    }
}

extern "C" {

const char* kq_last_error(void) { return g_err.c_str(); }
int kq_abi_version(void) { return KQ_ABI_VERSION; }

int kq_device_available(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int kq_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(KQ_ERR_INVALID, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(KQ_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(KQ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIPC(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPC(hipMemGetInfo(&f, &t));
    *free_bytes = f; *total_bytes = t;
    return KQ_OK;
}

// table geometry: region counts the record formats can address
static uint64_t round_regions(uint64_t regions, int k) {
    if (regions < 16) regions = 16;
    // FMT_NARROW / FMT_TOP8 records: 256 hash-prefix buckets of whole regions; k >= HI_K needs it for every table
    // (a slot drops the top 8 hash bits, which its region then implies)
    if (regions >= (uint64_t)NB_MAX || k >= HI_K) regions = (regions + 255) / 256 * 256;
    if (regions >= (1ull << 19)) {                                                  // ... of 2^s sub-buckets of whole regions each, <= 256 regions per sub-bucket
        uint32_t sbits = 3;
        while (sbits < SUB_BITS_MAX && (((regions + 255) / 256) >> sbits) > 256) ++sbits;
        const uint64_t unit = 256ull << sbits;
        regions = (regions + unit - 1) / unit * unit;
    }
    return regions;
}
// KQ_OPT_BUCKET_WINDOW: the (empty) table becomes the window [lo, hi) of the 256 hash-prefix buckets of a geometry that is
// 256 / (hi - lo) times larger: the memory stays what kq_create sized, the handle then holds -- and answers for -- only
// the k-mers of those buckets (a multi-GPU shard; kq_insert_sharded_dev brings them)
static int set_window(kq_handle* h, uint32_t lo, uint32_t hi) {
    if (lo >= hi || hi > 256) return fail(KQ_ERR_INVALID, "bucket window [%u, %u) is not a range of the 256 hash-prefix buckets", lo, hi);
    if (h->k > (int)NARROW_MAX_K) return fail(KQ_ERR_INVALID, "bucket windows need k <= %u (5-byte records)", NARROW_MAX_K);
    if (!h->table_empty || h->n_pend || h->kmers_bound) return fail(KQ_ERR_INVALID, "the bucket window is chosen before anything is counted");
    // an empty windowed table moves to another window of the same width without a new allocation (bucket-range passes of one
    // GPU: count the buckets [0, 128), clear, count [128, 256) into the same memory)
    if (h->windowed && hi - lo == h->win_hi - h->win_lo && !(lo == 0 && hi == 256)) {
        HIPC(hipStreamSynchronize(h->stream));
        h->win_lo = lo; h->win_hi = hi;
        return KQ_OK;
    }
    const uint64_t nb = hi - lo;
    const uint64_t alloc_now = h->n_alloc_regions();
    uint64_t virt = round_regions(std::max<uint64_t>((alloc_now + nb - 1) / nb * 256, (uint64_t)NB_MAX), h->k);
    if (virt >= (1ull << 32)) return fail(KQ_ERR_INVALID, "bucket window too narrow for a table of %llu regions", (unsigned long long)alloc_now);
    HIPC(hipStreamSynchronize(h->stream));
    // the new table first: a failed allocation leaves the handle as it was
    const bool windowed = !(lo == 0 && hi == 256);
    const uint64_t alloc_new = windowed ? (uint64_t)(hi - lo) * (virt >> 8) : virt;
    Slot* fresh = nullptr;
    uint32_t* fresh_rs = nullptr;
    int rc = alloc_main(h, virt, alloc_new, &fresh, &fresh_rs);
    if (rc) return rc;
    HIPC(hipStreamSynchronize(h->stream));
    if (h->slots) (void)hipFree(h->slots);
    if (h->rstart) (void)hipFree(h->rstart);
    h->slots = fresh; h->rstart = fresh_rs;
    h->n_regions = virt; h->win_lo = lo; h->win_hi = hi; h->windowed = windowed;
    h->slots_dirty = false;
    return KQ_OK;
}

int kq_create(kq_handle** out, int device, int k, int map_count, uint64_t capacity_hint) {
    if (!out) return fail(KQ_ERR_INVALID, "out is null");
    *out = nullptr;
    if (k < 2 || k > 32) return fail(KQ_ERR_INVALID, "k must be in 2..32 (got %d)", k);       // src/input.cpp:142
    if (map_count < 1 || map_count > 65535) return fail(KQ_ERR_INVALID, "map_count must be in 1..65535 (got %d)", map_count);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(KQ_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(KQ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIPC(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPC(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KQ_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code only", device, prop.gcnArchName);
    kq_handle* h = new (std::nothrow) kq_handle();
    if (!h) return fail(KQ_ERR_NOMEM, "host allocation failed");
    h->device = device; h->k = k; h->map_count = map_count; h->n_cu = prop.multiProcessorCount;
    h->filt_hi = (uint32_t)map_count;
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete h; return fail(KQ_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    h->stream = h->base = h->own_stream;
    int rc = KQ_OK;
    do {
        if (hipMalloc((void**)&h->st, sizeof(DevState)) != hipSuccess || hipHostMalloc((void**)&h->st_host, sizeof(DevState)) != hipSuccess) {
            rc = fail(KQ_ERR_NOMEM, "state allocation failed"); break;
        }
        (void)hipMemsetAsync(h->st, 0, sizeof(DevState), h->stream);
        uint64_t slots = (uint64_t)((double)(capacity_hint ? capacity_hint : (1u << 20)) / 0.7);   // load <= 0.7 at the hinted size
        const uint64_t regions = round_regions((slots + REGION_SLOTS - 1) >> REGION_SHIFT, k);
        rc = alloc_main(h, regions, regions, &h->slots, &h->rstart); if (rc) break;
        h->n_regions = regions;
        uint64_t hc = 1u << 16;
        rc = alloc_hc(h, hc, &h->hc); if (rc) break;
        h->hc_cap = hc;
        if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = fail(KQ_ERR_HIP, "table initialisation failed"); break; }
    } while (0);
    if (rc) { kq_destroy(h); return rc; }
    *out = h;
    return KQ_OK;
}

void kq_destroy(kq_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->slots) (void)hipFree(h->slots);
    if (h->rstart) (void)hipFree(h->rstart);
    if (h->hc) (void)hipFree(h->hc);
    if (h->st) (void)hipFree(h->st);
    if (h->st_host) (void)hipHostFree(h->st_host);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->stage) (void)hipFree(h->stage);
    if (h->part) (void)hipFree(h->part);
    scr_free(h->part2); scr_free(h->fork_part2);
    for (auto& e : h->hist_cache) (void)hipFree(e.m1_all);
    for (int w = 0; w < 2; ++w) { if (h->fork_stream[w]) { (void)hipStreamSynchronize(h->fork_stream[w]); (void)hipStreamDestroy(h->fork_stream[w]); } }
    if (h->fork_part[1]) (void)hipFree(h->fork_part[1]);
    for (auto e : h->ev_ring) if (e) (void)hipEventDestroy(e);
    if (h->ev_pass) (void)hipEventDestroy(h->ev_pass);
    for (int i = 0; i < kq_handle::IN_SLOTS; ++i) { if (h->in_buf[i]) (void)hipFree(h->in_buf[i]); if (h->in_consumed[i]) (void)hipEventDestroy(h->in_consumed[i]); }
    for (auto e : h->in_copied) (void)hipEventDestroy(e);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    arena_release(h);
    if (h->d_sets) (void)hipFree(h->d_sets);
    if (h->hot) (void)hipFree(h->hot);
    marks_reset(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int kq_clear(kq_handle* h) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    HIPC(hipSetDevice(h->device));
    h->slots_dirty = true;           // the 24 B/slot clear is folded into the next partitioned count (k_count_regions writes every region); anything else materialises it first
    clear_words(h, h->hc, sizeof(HcSlot) / 8, h->hc_cap);
    HIPC(hipMemsetAsync(h->st, 0, sizeof(DevState), h->stream));
    h->kmers_bound = 0;
    h->used_bound = 0;
    h->hc_check_at = 0;
    h->table_empty = true;
    h->n_pend = 0; h->arena_used = 0; h->pend_records = 0;      // records not applied yet are dropped with the rest
    return KQ_OK;
}

int kq_set_stream(kq_handle* h, void* s) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    HIPC(hipStreamSynchronize(h->stream));
    h->stream = h->base = s ? (hipStream_t)s : h->own_stream;
    return KQ_OK;
}
int kq_set_option(kq_handle* h, int option, int64_t value) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    switch (option) {
        case KQ_OPT_TRUST_CAPACITY: h->trust_capacity = value != 0; return KQ_OK;
        case KQ_OPT_COUNT_PATH:
            if (value < 0 || value > 2) return fail(KQ_ERR_INVALID, "KQ_OPT_COUNT_PATH must be 0, 1 or 2");
            h->count_path = (int)value; return KQ_OK;
        case KQ_OPT_COUNT_MAP_RANGE: {
            const int64_t lo = value & 0xFFFF, hi = (value >> 16) & 0xFFFF;
            if (lo >= hi || hi > h->map_count) return fail(KQ_ERR_INVALID, "map range [%lld,%lld) outside [0,%d]", (long long)lo, (long long)hi, h->map_count);
            h->filt_lo = (uint32_t)lo; h->filt_hi = (uint32_t)hi; return KQ_OK;
        }
        case KQ_OPT_NARROW_MID:
            if (value < 2 || value > (1 << 14)) return fail(KQ_ERR_INVALID, "KQ_OPT_NARROW_MID must be in [2, 16384]");
            h->mid_rps = (uint32_t)value; return KQ_OK;
        case KQ_OPT_MERGE_PATH:
            if (value < 0 || value > 2) return fail(KQ_ERR_INVALID, "KQ_OPT_MERGE_PATH must be 0, 1 or 2");
            h->merge_path = (int)value; return KQ_OK;
        case KQ_OPT_LOOKUP_PATH:
            if (value < 0 || value > 2) return fail(KQ_ERR_INVALID, "KQ_OPT_LOOKUP_PATH must be 0, 1 or 2");
            h->lookup_path = (int)value; return KQ_OK;
        case KQ_OPT_PROFILE: h->profile = value != 0; if (!h->profile) marks_reset(h); return KQ_OK;
        case KQ_OPT_TEST_FAIL_PLAN: h->test_fail_plan = value != 0; return KQ_OK;
        case KQ_OPT_COUNT_MAP_PASSES: {
            if (value != 1 && value != 2 && value != 4 && value != 8) return fail(KQ_ERR_INVALID, "KQ_OPT_COUNT_MAP_PASSES must be 1, 2, 4 or 8");
            if ((h->map_count & (h->map_count - 1)) != 0 || value > h->map_count) return fail(KQ_ERR_INVALID, "KQ_OPT_COUNT_MAP_PASSES needs a power-of-two map count");
            HIPC(hipStreamSynchronize(h->stream));
            for (auto& e : h->hist_cache) (void)hipFree(e.m1_all);
            h->hist_cache.clear(); h->hist_cache_bytes = 0;
            h->map_passes = (int)value; return KQ_OK;
        }
        case KQ_OPT_KERNEL_SET:
            if (value < 0 || value > 7) return fail(KQ_ERR_INVALID, "KQ_OPT_KERNEL_SET is a mask of bits 1, 2, 4");
            h->kernel_set = (int)value; return KQ_OK;
        case KQ_OPT_OVERLAP:
            if (value < 0 || value > 2) return fail(KQ_ERR_INVALID, "KQ_OPT_OVERLAP must be 0, 1 or 2");
            h->overlap = (int)value; return KQ_OK;
        case KQ_OPT_PENDING_BYTES: {
            if (value < -1) return fail(KQ_ERR_INVALID, "KQ_OPT_PENDING_BYTES must be -1 (auto), 0 (off) or a byte count");
            int rc = flush_pending(h);
            if (rc) return rc;
            if (h->arena && value != h->pend_budget) { HIPC(hipStreamSynchronize(h->stream)); arena_release(h); }
            h->pend_budget = value; return KQ_OK;
        }
        case KQ_OPT_BUCKET_WINDOW:
            HIPC(hipSetDevice(h->device));
            return set_window(h, (uint32_t)(value & 0xFFFF), (uint32_t)((value >> 16) & 0xFFFF));
        case KQ_OPT_SLICE_KMERS:
            if (value < 1) return fail(KQ_ERR_INVALID, "KQ_OPT_SLICE_KMERS must be positive");
            h->slice_kmers = (uint64_t)value; h->slice_user = true; return KQ_OK;
        default: return fail(KQ_ERR_INVALID, "unknown option %d", option);
    }
}
int kq_get_profile(kq_handle* h, char* buf, uint64_t cap) {
    if (!h || !buf || !cap) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    HIPC(hipStreamSynchronize(h->stream));
    std::string out;
    for (size_t i = 1; i < h->marks.size(); ++i) {
        float ms = 0;
        HIPC(hipEventElapsedTime(&ms, h->marks[i - 1].second, h->marks[i].second));
        char tmp[96];
        snprintf(tmp, sizeof tmp, "%s%s=%.4f", out.empty() ? "" : ";", h->marks[i].first, ms);
        out += tmp;
    }
    if (out.size() + 1 > cap) return fail(KQ_ERR_CAPACITY, "profile buffer too small: need %zu", out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return KQ_OK;
}
void* kq_get_stream(kq_handle* h) { return h ? (void*)h->stream : nullptr; }
int kq_sync(kq_handle* h) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    HIPC(hipSetDevice(h->device));
    HIPC(hipStreamSynchronize(h->stream));
    return check_errors(h);
}
int kq_flush(kq_handle* h) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    HIPC(hipSetDevice(h->device));
    return flush_pending(h);
}
int kq_get_info(kq_handle* h, kq_info* out) {
    if (!h || !out) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    int rc = read_state(h);
    if (rc) return rc;
    out->kmers_counted = h->st_host->kmers_added;
    out->slots_used = h->st_host->slots_used;
    out->slots_total = h->n_slots();
    out->hc_used = h->st_host->hc_used;
    out->hc_total = h->hc_cap;
    out->table_bytes = h->n_slots() * sizeof(Slot) + h->hc_cap * sizeof(HcSlot);
    out->table_passes = h->table_passes;
    return KQ_OK;
}

// ---- count ---------------------------------------------------------------------------------
// ---- partitioned count: host orchestration -------------------------------------------------------
struct PartPlan {
    PartCfg cfg;              // P1 (bases -> coarse buckets)
    bool two_level;
    int fmt;                  // record format between the stages (FMT_*)
    uint64_t n_max, R;
    uint32_t g1;              // P1 scatter workgroups
    uint64_t m1_n, m2_n, sums_n, groups_n;
    // device pointers into h->part
    uint64_t *recs1, *recs2;
    uint8_t *aux1, *aux2;     // WIDE records: edge bytes travelling with recs1 / recs2
    unsigned long long *m1, *seg_off, *unit_base, *group_base, *sums, *total, *hot;
    uint32_t* m2;
};
static void plan_cfg(const kq_handle* h, PartCfg* cfg, bool allow_narrow = false) {
    cfg->n_regions = h->n_regions;
    // fan-outs: the first split (fused with the sequence scan) is insensitive to its fan-out up to
    // ~512 bins, the second is bound by the length of the runs it writes (4096 / fan-out records), so
    // the first level takes as many bins as it can: 256..512 coarse buckets, the rest in level two
    // (measured on configs[1]: 262 x 64 beats the balanced 66 x 256 by 0.2 ms per 130 M records)
    uint32_t g = 1;
    while (((cfg->n_regions + (1ull << g) - 1) >> g) > 512 && g < 10) ++g;
    while (((cfg->n_regions + (1ull << g) - 1) >> g) >= (uint64_t)NB_MAX) ++g;
    cfg->g_shift = cfg->n_regions < (uint64_t)NB_MAX ? 0 : g;
    cfg->n_coarse = (uint32_t)((cfg->n_regions + (1ull << cfg->g_shift) - 1) >> cfg->g_shift);
    cfg->mode = 0; cfg->map_count = (uint32_t)h->map_count;
    cfg->map_mask = (h->map_count & (h->map_count - 1)) == 0 ? (uint32_t)h->map_count - 1 : 0;
    cfg->filt_lo = 0; cfg->filt_hi = (uint32_t)h->map_count;
    cfg->raw_out = 0;
    // 5-byte records (FMT_NARROW): first split on the top 8 hash bits, which needs every bucket to own a
    // whole number of regions (kq_create rounds large tables to a multiple of 256 regions; doubling keeps it)
    cfg->narrow = 0; cfg->sub_bits = 0; cfg->owner_sub = 0; cfg->n_rng = 0; cfg->win_lo = 0; cfg->win_hi = 1u << NARROW_CBITS;
    if (allow_narrow && (h->k <= (int)NARROW_MAX_K || h->k > PART_MAX_K) && cfg->n_regions >= (uint64_t)NB_MAX && cfg->n_regions % (1u << NARROW_CBITS) == 0) {
        const uint64_t rps = cfg->n_regions >> NARROW_CBITS;
        // one level bucket -> regions while a bucket has < mid_rps regions (the multisplit writes runs of 4096 / fan-out
        // records, so the last fan-out should stay near 256), else a middle level of 2^sb sub-buckets (sb <= 8): covers
        // every table that fits the HBM.  kq_create rounds the region count so that sb reaches its target
        uint32_t sb = 0;
        if (rps >= h->mid_rps) {
            const uint64_t target_nb = std::max<uint64_t>(1, std::min<uint64_t>(256, h->mid_rps / 8));
            while ((rps >> sb) > target_nb && sb < SUB_BITS_MAX && rps % (2ull << sb) == 0) ++sb;
        }
        if ((rps >> sb) < (uint64_t)NB_MAX && (sb > 0 || rps < (uint64_t)NB_MAX)) { cfg->narrow = 1; cfg->sub_bits = sb; }
    }
    if (cfg->narrow) { cfg->n_coarse = 1u << NARROW_CBITS; if (cfg->g_shift == 0) cfg->g_shift = 1; }
}
// carve the scratch buffer for a batch of at most n_max records; n_tiles = 0 when the input is records
static int plan_alloc(kq_handle* h, PartPlan* p, uint64_t n_max, uint64_t n_tiles, uint32_t p1_bins, bool allow_narrow = false,
                      uint64_t min_segs = 0 /*segments / output groups an exchange level needs beyond the table's own*/) {
    if (h->test_fail_plan) { h->test_fail_plan = false; return fail(KQ_ERR_NOMEM, "partition scratch allocation failed (injected by KQ_OPT_TEST_FAIL_PLAN)"); }
    if (n_max >= (1ull << 32) - 16) return fail(KQ_ERR_INVALID, "a partition pass handles fewer than 2^32 records (got %llu): slice the input", (unsigned long long)n_max);
    plan_cfg(h, &p->cfg, allow_narrow);
    n_max = (n_max + 7) & ~7ull;                                // every record array starts 16-byte aligned and has slack for vector loads
    p->two_level = p->cfg.g_shift != 0;
    p->fmt = !p->cfg.narrow ? FMT_PACK8 : h->k > PART_MAX_K ? FMT_TOP8 : FMT_NARROW;      // the caller switches to FMT_WIDE where it applies
    p->n_max = n_max; p->R = p->cfg.n_regions;
    // scatter workgroups: six times what is resident (3 or 2 per CU), each with its own cursor column: the hardware
    // dispatcher balances them (a grid of exactly the resident count ran 8 % longer: k_p1_scatter 577 -> 531 us)
    p->g1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)h->n_cu * (p1_bins < 512 ? 18 : 12)));
    p->m1_n = (uint64_t)p1_bins * p->g1 * P1_F;
    const uint64_t nb_max = std::max<uint64_t>(std::max<uint64_t>(1ull << p->cfg.g_shift, p->cfg.n_coarse), p->cfg.narrow ? (p->cfg.n_regions >> NARROW_CBITS) >> p->cfg.sub_bits : 0);
    const uint64_t seg_max = std::max<uint64_t>(min_segs, p->cfg.narrow ? ((uint64_t)1 << NARROW_CBITS) << p->cfg.sub_bits : (uint64_t)NB_MAX);      // segments of the widest level
    p->m2_n = (n_max / P2_UNIT + seg_max + 2) * nb_max;         // u32 entries, enough for either level
    p->groups_n = std::max<uint64_t>(std::max<uint64_t>(p->R, (uint64_t)p->cfg.n_coarse << p->cfg.g_shift), min_segs) + 2;
    p->sums_n = std::max(p->m1_n, p->groups_n) / SCAN_CHUNK + 2;
    const uint64_t aux_words = (n_max + 7) / 8 + 1;
    const uint64_t rec_words = p->fmt == FMT_NARROW ? n_max / 2 + 2 : n_max;          // narrow records: u32 + lockstep byte
    // the second record array of a large plan lives in a physically scrambled buffer of its own (struct Scrambled)
    const bool split2 = (rec_words + aux_words) * 8 >= ((size_t)512 << 20);
    if (split2) { int rc2 = scr_ensure(h->part2, (size_t)(rec_words + aux_words) * 8, h->device); if (rc2) return rc2; }
    const size_t words = (size_t)((split2 ? 1 : 2) * (rec_words + aux_words) + p->m1_n + 2 * (seg_max + 2) + p->groups_n + p->sums_n + 4 + (p->R + 2) + (p->m2_n + 1) / 2);
    int rc = ensure_buf(&h->part, &h->part_bytes, words * 8);
    if (rc) return rc;
    p->recs1 = (uint64_t*)h->part;
    p->aux1 = (uint8_t*)(p->recs1 + rec_words);
    uint64_t* rest = (uint64_t*)(p->aux1 + aux_words * 8);
    if (split2) { p->recs2 = (uint64_t*)h->part2.p; p->aux2 = (uint8_t*)(p->recs2 + rec_words); }
    else { p->recs2 = rest; p->aux2 = (uint8_t*)(p->recs2 + rec_words); rest = (uint64_t*)(p->aux2 + aux_words * 8); }
    p->m1 = (unsigned long long*)rest;
    p->seg_off = p->m1 + p->m1_n;
    p->unit_base = p->seg_off + seg_max + 2;
    p->group_base = p->unit_base + seg_max + 2;
    p->sums = p->group_base + p->groups_n;
    p->total = p->sums + p->sums_n;
    p->hot = p->total + 4;
    p->m2 = (uint32_t*)(p->hot + p->R + 2);
    return KQ_OK;
}
// exclusive scan of n u64 on the device (in place); *total (device) receives the sum
static void scan_u64(kq_handle* h, unsigned long long* a, uint64_t n, unsigned long long* sums_scratch, unsigned long long* total) {
    if (n <= 2 * SCAN_CHUNK) {                     // one workgroup: up to 32 elements per thread
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, h->stream, a, n, total);
        return;
    }
    const uint64_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)chunks), dim3(1024), 0, h->stream, a, n, sums_scratch);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, h->stream, sums_scratch, chunks, total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)chunks), dim3(1024), 0, h->stream, a, n, sums_scratch);
}
// P1 on bases with the given bin function; afterwards p->seg_off[0..bins] are the bucket offsets
// (seg_off[bins] = number of records) and `out` holds the records grouped by bin
// stage marker of the last partitioned count (KQ_OPT_PROFILE): one HIP event per call, on the handle's stream
static void mark(kq_handle* h, const char* name) {
    if (!h->profile) return;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, h->stream);
    h->marks.emplace_back(name, e);
}
static void marks_reset(kq_handle* h) {
    for (auto& m : h->marks) (void)hipEventDestroy(m.second);
    h->marks.clear();
}

static void run_p1(kq_handle* h, PartPlan* p, const PartCfg& cfg, const uint8_t* ab, uint64_t lead, uint64_t len, EmitRange er, uint64_t* out,
                   uint8_t* out_aux, int aux_fmt, const uint16_t* pinv = nullptr /*packed input: ab = the code words (tile_fetch)*/) {
    // plain table split (no owner split, no map-range filter): branch-free bin functions
    const bool plain = cfg.mode == 0 && cfg.filt_lo == 0 && cfg.filt_hi == cfg.map_count;
    const bool owner_plain = cfg.mode == 1 && cfg.map_mask != 0 && cfg.filt_lo == 0 && cfg.filt_hi == cfg.map_count;   // multi-GPU owner split
    const bool narrow_filt = cfg.mode == 0 && cfg.narrow && !plain && cfg.map_mask != 0 && h->k <= (int)NARROW_MAX_K;   // map-range pass on a bucketed table
    const bool narrow_win = plain && cfg.narrow && h->k <= (int)NARROW_MAX_K && (cfg.win_lo != 0 || cfg.win_hi != (1u << NARROW_CBITS));   // windowed table: own buckets only
    const int binmode = owner_plain ? 3 : narrow_filt ? 4 : narrow_win ? 6 : !plain ? 0 : cfg.narrow ? 2 : 1;
    // KQ_OPT_COUNT_MAP_PASSES = n: this map-range pass is one of n over RESIDENT batches (the caller vouches that a batch keeps
    // its address and content between the passes).  The first pass that scans a slice counts for all n ranges in one histogram
    // launch (bin = range * 256 + bucket: the block of a range is contiguous in the bin-major matrix) and keeps the raw counts;
    // every pass takes its block from there -- (n - 1) of the n histogram scans of a slice are never run.
    bool have_hist = false;
    if (narrow_filt && h->map_passes > 1 && P1_F == 1) {
        const uint32_t n = (uint32_t)h->map_passes, per = cfg.map_count / n, rr = cfg.filt_lo / per;
        if (cfg.filt_lo == rr * per && cfg.filt_hi == (rr + 1) * per) {
            const kq_handle::HistKey key{ab, pinv, lead, len, er.lo, er.hi, p->g1, n, h->k};
            const size_t block = (size_t)(1u << NARROW_CBITS) * p->g1 * sizeof(unsigned long long);
            kq_handle::HistEntry* ent = nullptr;
            for (auto& e : h->hist_cache) if (e.key == key) { ent = &e; break; }
            if (!ent && h->hist_cache_bytes + n * block <= ((size_t)4 << 30)) {
                unsigned long long* buf = nullptr;
                if (hipMalloc((void**)&buf, n * block) == hipSuccess) {
                    PartCfg all = cfg;
                    all.n_rng = n; all.n_coarse = n << NARROW_CBITS;
                    if (h->k == 21) hipLaunchKernelGGL((k_p1_hist<5, 21>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, all, er, p->g1, buf, pinv);
                    else hipLaunchKernelGGL((k_p1_hist<5, 0>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, all, er, p->g1, buf, pinv);
                    h->hist_cache.push_back(kq_handle::HistEntry{key, buf, n * block});
                    h->hist_cache_bytes += n * block;
                    ent = &h->hist_cache.back();
                } else (void)hipGetLastError();
            }
            if (ent) {
                (void)hipMemcpyAsync(p->m1, (const char*)ent->m1_all + rr * block, block, hipMemcpyDeviceToDevice, h->stream);
                have_hist = true;
            }
        }
    }
    // the same for bucket-range passes (windowed table): the unfiltered bucket matrix serves every window -- the rows of the
    // other buckets are zeroed before the scan
    if (narrow_win && h->map_passes > 1 && P1_F == 1) {
        const kq_handle::HistKey key{ab, pinv, lead, len, er.lo, er.hi, p->g1, 0xB0C4E7u, h->k};
        const size_t block = (size_t)(1u << NARROW_CBITS) * p->g1 * sizeof(unsigned long long), row = (size_t)p->g1 * sizeof(unsigned long long);
        kq_handle::HistEntry* ent = nullptr;
        for (auto& e : h->hist_cache) if (e.key == key) { ent = &e; break; }
        if (!ent && h->hist_cache_bytes + block <= ((size_t)4 << 30)) {
            unsigned long long* buf = nullptr;
            if (hipMalloc((void**)&buf, block) == hipSuccess) {
                PartCfg all = cfg;
                all.win_lo = 0; all.win_hi = 1u << NARROW_CBITS;
                if (h->k == 21) hipLaunchKernelGGL((k_p1_hist<2, 21>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, all, er, p->g1, buf, pinv);
                else hipLaunchKernelGGL((k_p1_hist<2, 0>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, all, er, p->g1, buf, pinv);
                h->hist_cache.push_back(kq_handle::HistEntry{key, buf, block});
                h->hist_cache_bytes += block;
                ent = &h->hist_cache.back();
            } else (void)hipGetLastError();
        }
        if (ent) {
            (void)hipMemcpyAsync(p->m1, ent->m1_all, block, hipMemcpyDeviceToDevice, h->stream);
            if (cfg.win_lo) (void)hipMemsetAsync(p->m1, 0, (size_t)cfg.win_lo * row, h->stream);
            if (cfg.win_hi < (1u << NARROW_CBITS)) (void)hipMemsetAsync((char*)p->m1 + (size_t)cfg.win_hi * row, 0, (size_t)((1u << NARROW_CBITS) - cfg.win_hi) * row, h->stream);
            have_hist = true;
        }
    }
#define KQ_P1H(B, K) hipLaunchKernelGGL((k_p1_hist<B, K>), dim3(p->g1 * P1_F), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, cfg, er, p->g1, p->m1, pinv)
    if (have_hist) { }
    else if (binmode == 6) { if (h->k == 21) KQ_P1H(6, 21); else KQ_P1H(6, 0); }
    else if (binmode == 2) { if (h->k == 21) KQ_P1H(2, 21); else KQ_P1H(2, 0); }
    else if (binmode == 1) { if (h->k == 31) KQ_P1H(1, 31); else KQ_P1H(1, 0); }
    else if (binmode == 3) { if (h->k == 21) KQ_P1H(3, 21); else KQ_P1H(3, 0); }
    else if (binmode == 4) { if (h->k == 21) KQ_P1H(4, 21); else KQ_P1H(4, 0); }
    else KQ_P1H(0, 0);
#undef KQ_P1H
    scan_u64(h, p->m1, (uint64_t)cfg.n_coarse * p->g1 * P1_F, p->sums, p->total);
    hipLaunchKernelGGL(k_p1_offsets, dim3((cfg.n_coarse + 256) / 256), dim3(256), 0, h->stream, p->m1, p->total, cfg, p->g1, p->seg_off);
    mark(h, "k_p1_hist+scan");
    const bool small = cfg.n_coarse < 512;                     // 48 KiB LDS variant: three workgroups per CU
#define KQ_P1S(W, N, B, K) hipLaunchKernelGGL((k_p1_scatter<W, N, B, K>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, cfg, er, p->m1, out, out_aux, aux_fmt, pinv)
#define KQ_P1T(B, K) do { if (h->kernel_set & 1) KQ_P1S(FMT_TOP8, 512, B, K); else \
        hipLaunchKernelGGL((k_p1_scatter_s<B, K, 1, true>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, cfg, er, p->m1, (uint32_t*)out, out_aux, pinv); } while (0)
    if (cfg.narrow && h->k > PART_MAX_K) { if (plain && h->k == 31) KQ_P1T(2, 31); else if (plain) KQ_P1T(2, 0); else KQ_P1T(0, 0); }
    // narrow records: the streamed scatter (k_p1_scatter_s); a pass that keeps every k-mer splits two tiles per round, a filtered one one
#define KQ_P1N(B, K, T) do { if (h->kernel_set & 1) KQ_P1S(FMT_NARROW, 512, B, K); else \
        hipLaunchKernelGGL((k_p1_scatter_s<B, K, T>), dim3(p->g1), dim3(TILE_THREADS * T), 0, h->stream, ab, lead, len, h->k, cfg, er, p->m1, (uint32_t*)out, out_aux, pinv); } while (0)
    else if (narrow_filt)  { if (h->k == 21) KQ_P1N(4, 21, 1); else KQ_P1N(4, 0, 1); }
    else if (narrow_win)   { if (h->k == 21) KQ_P1N(6, 21, 1); else KQ_P1N(6, 0, 1); }
    else if (cfg.narrow)   { if (plain && h->k == 21) KQ_P1N(2, 21, KQ_P1S_TPR); else if (plain) KQ_P1N(2, 0, KQ_P1S_TPR); else KQ_P1N(0, 0, 1); }     // 256 buckets
    else if (out_aux && plain && h->k == 31) { if (small) KQ_P1S(FMT_WIDE, 512, 1, 31); else KQ_P1S(FMT_WIDE, NB_MAX, 1, 31); }   // the HiFi k
    else if (out_aux) { if (small) KQ_P1S(FMT_WIDE, 512, 0, 0); else KQ_P1S(FMT_WIDE, NB_MAX, 0, 0); }
    else if (owner_plain && !out_aux && small) { if (h->k == 21) KQ_P1S(FMT_PACK8, 512, 3, 21); else KQ_P1S(FMT_PACK8, 512, 3, 0); }
    else if (plain)   { if (small) KQ_P1S(FMT_PACK8, 512, 1, 0); else KQ_P1S(FMT_PACK8, NB_MAX, 1, 0); }
    else              { if (small) KQ_P1S(FMT_PACK8, 512, 0, 0); else KQ_P1S(FMT_PACK8, NB_MAX, 0, 0); }
#undef KQ_P1S
#undef KQ_P1N
#undef KQ_P1T
    mark(h, "k_p1_scatter");
}
// one generic split level: in (grouped by p->seg_off[0..n_seg]) -> out grouped by (segment, bin);
// afterwards p->group_base[0..n_seg*nb] are the output offsets
static void run_level(kq_handle* h, PartPlan* p, const LevelCfg& lv, const uint64_t* in, const uint8_t* in_aux, uint64_t* out, uint8_t* out_aux,
                      unsigned long long* gb = nullptr /*where the output offsets go (default p->group_base)*/,
                      const unsigned long long* seg_hi = nullptr /*end of every input segment (default: the next one's start)*/) {
    if (!gb) gb = p->group_base;
    if (!seg_hi) seg_hi = p->seg_off + 1;
    LevelCfg lvr = lv;                                            // rank replication: as many counters per bin as fit 512 (at most 64, one per lane)
    lvr.rep_shift = 0;
    while (lvr.rep_shift < 6 && (((uint64_t)lv.nb + 1) << (lvr.rep_shift + 1)) <= 512) ++lvr.rep_shift;
    const LevelCfg& lv_ = lvr;
    const int fmt = lv.narrow == 2 ? FMT_TOP8 : lv.narrow ? FMT_NARROW : in_aux != nullptr ? FMT_WIDE : FMT_PACK8;       // input format; lv.top8: packed in, narrow out
    const uint64_t groups = (uint64_t)(lv.n_seg / lv.spb) * lv.nb;
    hipLaunchKernelGGL(k_lv_units, dim3(1), dim3(1024), 0, h->stream, p->seg_off, seg_hi, lv_, p->unit_base);
    // one workgroup per work unit (upper bound of the unit count; surplus workgroups exit at once):
    // the hardware dispatcher balances them, a fixed grid looping over units left a 30 % tail
    const unsigned unit_grid = (unsigned)std::min<uint64_t>(p->n_max / P2_UNIT + lv.n_seg + 1, 1u << 30);
    if (fmt == FMT_TOP8) hipLaunchKernelGGL(k_lv_hist<FMT_TOP8>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, in_aux, lv_, p->seg_off, seg_hi, p->unit_base, p->m2);
    else if (fmt == FMT_NARROW) hipLaunchKernelGGL(k_lv_hist<FMT_NARROW>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, in_aux, lv_, p->seg_off, seg_hi, p->unit_base, p->m2);
    else if (fmt == FMT_WIDE) hipLaunchKernelGGL(k_lv_hist<FMT_WIDE>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, in_aux, lv_, p->seg_off, seg_hi, p->unit_base, p->m2);
    else hipLaunchKernelGGL(k_lv_hist<FMT_PACK8>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, in_aux, lv_, p->seg_off, seg_hi, p->unit_base, p->m2);
    // few units per segment (many segments): one thread per group; else one wave per group (a segment of thousands of units)
    if (p->n_max / P2_UNIT + 1 <= 8 * (uint64_t)(lv.n_seg / lv.spb))
        hipLaunchKernelGGL(k_lv_offsets_thread, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, h->stream, p->m2, lv_, p->unit_base, gb);
    else
        hipLaunchKernelGGL(k_lv_offsets, dim3((unsigned)((groups * 64 + 255) / 256)), dim3(256), 0, h->stream, p->m2, lv_, p->unit_base, gb);
    (void)hipMemsetAsync(gb + groups, 0, 8, h->stream);   // failure surfaces at the caller's hipGetLastError
    scan_u64(h, gb, groups + 1, p->sums, p->total + 1);
    mark(h, "k_lv_hist+offsets+scan");
    const bool small = lv.nb < 512;
#define KQ_LVS(W, N) hipLaunchKernelGGL((k_lv_scatter<W, N>), dim3(unit_grid), dim3(LV_THREADS), 0, h->stream, in, in_aux, lv_, \
                                        p->seg_off, seg_hi, p->unit_base, p->m2, gb, out, out_aux)
    if (lv.top8)              { KQ_LVS(FMT_PACK8_TO_NARROW, 512); }
    else if (fmt == FMT_TOP8) { if (small) KQ_LVS(FMT_TOP8, 512); else KQ_LVS(FMT_TOP8, NB_MAX); }
    // narrow records, at most 512 (bin, replica) counters: the streamed scatter (k_lv_scatter_s) for the level that writes tight records
#define KQ_LVN(T) hipLaunchKernelGGL((k_lv_scatter_s<T>), dim3(unit_grid), dim3(LV_THREADS), 0, h->stream, (const uint32_t*)in, in_aux, lv_, \
                                      p->seg_off, seg_hi, p->unit_base, p->m2, gb, (uint32_t*)out, out_aux)
    else if (fmt == FMT_NARROW && small && lv.rstart && !(h->kernel_set & 4)) { KQ_LVN(true); }
    else if (fmt == FMT_NARROW && small && !lv.rstart && (h->kernel_set & 2)) { KQ_LVN(false); }     // (measured: 0 .. +8 % against k_lv_scatter; not the default)
#undef KQ_LVN
    else if (fmt == FMT_NARROW && lv.rstart) { if (small) KQ_LVS(FMT_NARROW_TO_TIGHT, 512); else KQ_LVS(FMT_NARROW_TO_TIGHT, NB_MAX); }
    else if (fmt == FMT_NARROW) { if (small) KQ_LVS(FMT_NARROW, 512); else KQ_LVS(FMT_NARROW, NB_MAX); }
    else if (fmt == FMT_WIDE) { if (small) KQ_LVS(FMT_WIDE, 512); else KQ_LVS(FMT_WIDE, NB_MAX); }
    else                      { if (small) KQ_LVS(FMT_PACK8, 512); else KQ_LVS(FMT_PACK8, NB_MAX); }
#undef KQ_LVS
    mark(h, "k_lv_scatter");
}
static LevelCfg level_coarse_to_regions(const PartCfg& cfg) {
    LevelCfg lv; lv.n_regions = cfg.n_regions; lv.n_seg = cfg.n_coarse; lv.nb = 1u << cfg.g_shift; lv.seg_shift = cfg.g_shift; lv.out_shift = 0; lv.in_raw = 0; lv.k = 0; lv.narrow = 0; lv.top8 = 0;
    lv.nr_shift = lv.nr_rps = lv.nr_sub = lv.nr_inv = 0; lv.nr_div = 1;
    lv.nr_mid = 0; lv.nr_fshift = 24; lv.nr_fmask = 0xFFFFFFu; lv.nr_f24 = 0;
    lv.spb = 1;
    return lv;
}
// FMT_NARROW: 256 top-bit buckets -> their regions (bucket b owns regions [b * nb, (b + 1) * nb))
// `sub_bits` > 0: a middle level first cuts every bucket into 2^sub_bits sub-buckets (tables of more than
// 2048 regions per bucket: the last level then has 256 << sub_bits segments of rps >> sub_bits regions)
static LevelCfg level_narrow(const PartCfg& cfg, uint32_t sub_bits = 0, bool middle = false, bool top8_records = false) {
    LevelCfg lv; lv.n_regions = cfg.n_regions;
    const uint32_t rps = (uint32_t)(cfg.n_regions >> NARROW_CBITS), subsz = rps >> sub_bits;
    lv.seg_shift = 0; lv.out_shift = 0; lv.in_raw = 0; lv.k = 0; lv.narrow = top8_records ? 2 : 1; lv.top8 = 0;
    lv.nr_rps = rps; lv.nr_sub = subsz;
    if (middle) { lv.n_seg = 1u << NARROW_CBITS; lv.nb = 1u << sub_bits; lv.nr_shift = 0; lv.nr_div = subsz; }
    else        { lv.n_seg = (1u << NARROW_CBITS) << sub_bits; lv.nb = subsz; lv.nr_shift = sub_bits; lv.nr_div = 1; }
    lv.nr_inv = lv.nr_div > 1 ? (uint32_t)(((1ull << 32) + lv.nr_div - 1) / lv.nr_div) : 0;
    lv.nr_mid = middle ? 1u : 0u; lv.nr_fshift = 24 - sub_bits; lv.nr_fmask = (1u << (24 - sub_bits)) - 1u;
    lv.nr_f24 = ((uint64_t)subsz << (24 - sub_bits)) <= (1ull << 32) && subsz < (1u << 24) ? 1u : 0u;
    lv.spb = 1;
    return lv;
}
// bucket -> regions for FMT_NARROW records, in one level or (large tables) two; afterwards `*sorted` holds the
// records grouped by region and p->group_base their offsets
static void run_narrow_levels(kq_handle* h, PartPlan* p, const uint64_t** sorted, const uint8_t** sorted_aux, const P3Set* dst = nullptr, bool tight = false);
// FMT_TIGHT output of the last split level (count path only: the lookup kernels read 5-byte records)
static bool tight_ok(const kq_handle* h, const PartPlan& p) {
#ifdef KQ_NO_TIGHT
    (void)h; (void)p; return false;
#else
    return p.fmt == FMT_NARROW && p.R >= TIGHT_MIN_REGIONS && h->rstart != nullptr;
#endif
}
static LevelCfg level_flat_to_coarse(const PartCfg& cfg) {
    LevelCfg lv; lv.n_regions = cfg.n_regions; lv.n_seg = 1; lv.nb = cfg.n_coarse; lv.seg_shift = 32; lv.out_shift = cfg.g_shift; lv.in_raw = 0; lv.k = 0; lv.narrow = 0; lv.top8 = 0;
    lv.nr_shift = lv.nr_rps = lv.nr_sub = lv.nr_inv = 0; lv.nr_div = 1;
    lv.nr_mid = 0; lv.nr_fshift = 24; lv.nr_fmask = 0xFFFFFFu; lv.nr_f24 = 0;
    lv.spb = 1;
    return lv;
}
// ---- pending sets ---------------------------------------------------------------------------------
// A k_count_regions pass streams every region image of the table in and out once, whatever the number of records it
// applies.  So the region-sorted record set that P1 + the split levels make of a slice is not applied at once: it is
// kept in an arena (5 bytes per record for the default k) and one pass takes all pending sets together -- when the
// arena is full, before the table geometry changes, and before anything reads the table or the device state
// (kq_sync, summary, lookup, export, merge ...).  Counting is commutative, so results do not depend on when a set is
// applied.  This is what lets a human-scale table (tens of GB) be streamed once per ~10^10 records instead of once
// per slice; KQ_OPT_PENDING_BYTES bounds the arena (0 = apply every slice at once, the round-1 behaviour).
__global__ void k_p3set(P3Set* sets, int i, P3Set v) { sets[i] = v; }

static size_t set_bytes(uint64_t n_max, int fmt, uint64_t R) {
    const size_t rec = (fmt == FMT_NARROW || fmt == FMT_TIGHT) ? 4 : 8;
    const bool aux = fmt == FMT_NARROW || fmt == FMT_WIDE;
    return ((((size_t)n_max * rec + 63) & ~(size_t)63) + (aux ? (((size_t)n_max + 63) & ~(size_t)63) : 0) + (size_t)(R + 2) * 8 + 255) & ~(size_t)255;
}
// room for one more set in the arena (flushes / allocates as needed); false: no deferral for this slice
static int arena_take(kq_handle* h, uint64_t n_max, int fmt, uint64_t R, P3Set* out, bool* ok) {
    *ok = false;
    if (h->pend_budget == 0) return KQ_OK;
    const size_t need = set_bytes(n_max, fmt, R);
    bool was_full = false;
    if (h->n_pend && (h->n_pend >= P3_MAX_SETS || h->pend_fmt != fmt || h->arena_used + need > h->arena_bytes)) {
        was_full = h->pend_fmt == fmt && h->n_pend < P3_MAX_SETS;
        int rc = flush_pending(h);
        if (rc) return rc;
    }
    // The arena starts at a few sets and doubles every time it fills up, up to a few times the table and what is free less
    // a reserve (a fixed KQ_OPT_PENDING_BYTES is taken as it is): a short job never pays for allocating tens of GB (hipMalloc
    // costs milliseconds per GB), a long one gets there within its first batches
    size_t want = h->arena_bytes;
    if (need > want) want = h->pend_budget > 0 ? (size_t)h->pend_budget : std::max<size_t>(4 * need, (size_t)64 << 20);
    else if (was_full && h->pend_budget < 0) want = 2 * h->arena_bytes;
    if (want > h->arena_bytes) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return KQ_OK;
        size_t budget = want;
        // ceiling: what is free now (the partition scratch of this slice size is allocated already) less a reserve of 1/8 of
        // the device for whatever comes later (side-table growth, lookup / export buffers); half of it when that is tight
        const size_t avail = free_b + h->arena_bytes, reserve = std::max<size_t>(total_b / 8, (size_t)8 << 30);
        const size_t ceiling = avail > 2 * reserve ? avail - reserve : avail / 2;
        if (h->pend_budget < 0) budget = std::min<size_t>(want, std::min<size_t>(ceiling, std::max<size_t>(4 * need, 4 * (size_t)h->n_slots() * sizeof(Slot))));
        if (budget <= h->arena_bytes && need <= h->arena_bytes) budget = 0;         // at its ceiling already
        if (budget && budget < need) return KQ_OK;
        if (budget) {
            if (h->arena) { HIPC(hipStreamSynchronize(h->stream)); HIPC(hipStreamSynchronize(h->base)); arena_release(h); }
            static const bool scramble_arena = !getenv("KQ_SCRAMBLE_ARENA") || atoi(getenv("KQ_SCRAMBLE_ARENA")) != 0;      // (0: plain hipMalloc, for A/B)
            if (scramble_arena && budget >= ((size_t)1 << 30) && scr_try_vmm(h->arena_scr, budget, h->device)) h->arena = h->arena_scr.p;
            else
            if (hipMalloc(&h->arena, budget) != hipSuccess) { (void)hipGetLastError(); h->arena = nullptr; return KQ_OK; }
            h->arena_bytes = budget; h->arena_used = 0;
        }
    }
    uint8_t* base = (uint8_t*)h->arena + h->arena_used;
    const size_t rec = (fmt == FMT_NARROW || fmt == FMT_TIGHT) ? 4 : 8;
    const bool aux = fmt == FMT_NARROW || fmt == FMT_WIDE;
    out->recs = (const uint64_t*)base;
    base += ((size_t)n_max * rec + 63) & ~(size_t)63;
    out->aux = aux ? base : nullptr;
    if (aux) base += ((size_t)n_max + 63) & ~(size_t)63;
    out->base = (const unsigned long long*)base;
    out->n_max = n_max;
    h->arena_used += need;
    *ok = true;
    return KQ_OK;
}
// make `set` (records of format fmt, sorted by region of the CURRENT geometry) part of the next table pass
static int pend_add(kq_handle* h, const P3Set& set, int fmt, int aux_fmt) {
    if (h->n_pend && (h->n_pend >= P3_MAX_SETS || h->pend_fmt != fmt || h->pend_aux_fmt != aux_fmt)) {
        int rc = flush_pending(h);
        if (rc) return rc;
    }
    if (!h->d_sets) HIPC(hipMalloc((void**)&h->d_sets, sizeof(P3Set) * P3_MAX_SETS));
    hipLaunchKernelGGL(k_p3set, dim3(1), dim3(1), 0, h->stream, h->d_sets, h->n_pend, set);
    h->pend_fmt = fmt; h->pend_aux_fmt = aux_fmt;
    ++h->n_pend;
    h->pend_records += set.n_max;
    return KQ_OK;
}
// P3 over all pending sets.  Always on the handle's own stream (`base`), behind a join with the fork streams that wrote the
// sets; the fork streams in turn wait for the pass before they go on (their next sets reuse the arena it reads).
static int flush_pending_base(kq_handle* h);
static int flush_pending(kq_handle* h) {
    if (!h->n_pend) return KQ_OK;
    hipStream_t work = h->stream;
    h->stream = h->base;
    int rc = join_forks(h);
    if (!rc) rc = flush_pending_base(h);
    if (!rc && h->fork_stream[0]) {
        hipEvent_t e; rc = ev_get(h, &e);
        if (!rc && hipEventRecord(e, h->base) != hipSuccess) rc = fail(KQ_ERR_HIP, "hipEventRecord failed");
        for (int w = 0; w < 2 && !rc; ++w) if (hipStreamWaitEvent(h->fork_stream[w], e, 0) != hipSuccess) rc = fail(KQ_ERR_HIP, "hipStreamWaitEvent failed");
    }
    h->stream = work;
    return rc;
}
static int flush_pending_base(kq_handle* h) {
    const uint64_t R = h->n_regions;
    int rc = ensure_buf(&h->hot, &h->hot_bytes, (size_t)(R + 2) * 8);
    if (rc) return rc;
    if (h->kmers_bound / 255 + 1 > (1ull << 23)) {       // large jobs: the side table follows its observed fill (see reserve)
        rc = read_state_raw(h); if (rc) return rc;
        const uint64_t need_hc = std::max<uint64_t>(HC_CAP_BIG, 8 * h->st_host->hc_used);
        if (need_hc > h->hc_cap) { rc = grow_hc(h, need_hc); if (rc) return rc; }
    }
    unsigned long long* hot = (unsigned long long*)h->hot;       // [0] = count, then up to R region ids
    HIPC(hipMemsetAsync(hot, 0, 8, h->stream));
    const dim3 grid((unsigned)std::min<uint64_t>(h->n_alloc_regions(), 1u << 30)), grid_hot(h->n_cu), block(P3_THREADS);   // one workgroup per (allocated) region: dispatcher-balanced
    // 1: the slot array holds the empty image (skip reading it); 2: logically empty but not initialised (lazy kq_clear):
    // also write the image of regions without records.  A dirty array is never read, whatever table_empty says
    const int empty = h->slots_dirty ? 2 : h->table_empty ? 1 : 0;
    const int fmt = h->pend_fmt;
    const uint32_t rps = (fmt == FMT_NARROW || fmt == FMT_TIGHT || fmt == FMT_TOP8) ? (uint32_t)(R >> NARROW_CBITS) : 1u;
#define KQ_P3(F) do { \
        hipLaunchKernelGGL((k_count_regions<F, false>), grid, block, 0, h->stream, h->view(), h->d_sets, (uint32_t)h->n_pend, h->pend_aux_fmt, empty, hot, rps); \
        hipLaunchKernelGGL((k_count_regions<F, true>), grid_hot, block, 0, h->stream, h->view(), h->d_sets, (uint32_t)h->n_pend, h->pend_aux_fmt, empty, hot, rps); } while (0)
#ifdef KQ_NO_N32
    if (fmt == FMT_NARROW || fmt == FMT_TIGHT) KQ_P3(FMT_NARROW); else      // (pend_aux_fmt == AUX_TIGHT tells the generic kernel about FMT_TIGHT sets)
#endif
    if (fmt == FMT_NARROW || fmt == FMT_TIGHT) {
        // ordinary regions: the compact 32-bit-key kernel; skewed ones: the generic folding kernel
#ifdef KQ_P3_V1       // A/B build: round 2's loop formulation
#define KQ_N32(KC, T) hipLaunchKernelGGL((k_count_regions_n32<KC, T>), grid, block, 0, h->stream, h->view(), h->d_sets, (uint32_t)h->n_pend, empty, hot, rps)
#else
#define KQ_N32(KC, T) hipLaunchKernelGGL((k_count_regions_q4<KC, T>), grid, dim3(Q4_THREADS), 0, h->stream, h->view(), h->d_sets, (uint32_t)h->n_pend, empty, hot, rps)
#endif
        if (fmt == FMT_TIGHT) { if (h->k == 21) KQ_N32(21, true); else KQ_N32(0, true); }
        else                  { if (h->k == 21) KQ_N32(21, false); else KQ_N32(0, false); }
#undef KQ_N32
        hipLaunchKernelGGL((k_count_regions<FMT_NARROW, true>), grid_hot, block, 0, h->stream, h->view(), h->d_sets, (uint32_t)h->n_pend, h->pend_aux_fmt, empty, hot, rps);
    }
    else if (fmt == FMT_TOP8) KQ_P3(FMT_TOP8);
    else if (fmt == FMT_WIDE) KQ_P3(FMT_WIDE);
    else KQ_P3(FMT_PACK8);
#undef KQ_P3
    h->n_pend = 0; h->arena_used = 0; h->pend_records = 0;
    ++h->table_passes;
    // only now (every failure path of the partition stages lies before this point): the table has content, and a lazily
    // cleared slot array has been written in full
    h->slots_dirty = false;
    h->table_empty = false;
    mark(h, "k_count_regions");
    HIPC(hipGetLastError());
    return KQ_OK;
}

// can the record split reach every region of this table?  (5-byte records: up to 256 x 8 x 2047 regions, i.e.
// any table that fits the HBM; 8-byte / wide records: two fan-outs below NB_MAX)
static bool part_table_ok(const kq_handle* h, bool narrow_possible) {
    if (h->n_regions <= (1ull << 20)) return true;
    if (!narrow_possible) return false;
    PartCfg c; plan_cfg(h, &c, true);
    return c.narrow != 0;
}
// `dst` != nullptr: the last level writes records, lockstep bytes and region offsets there (a pending set in the arena)
static void run_narrow_levels(kq_handle* h, PartPlan* p, const uint64_t** sorted, const uint8_t** sorted_aux, const P3Set* dst, bool tight) {
    const uint32_t sb = p->cfg.sub_bits;
    const bool t8 = p->fmt == FMT_TOP8;
    uint8_t* a1 = t8 ? nullptr : p->aux1;
    uint8_t* a2 = t8 ? nullptr : p->aux2;
    uint64_t* fin = dst ? const_cast<uint64_t*>(dst->recs) : (sb == 0 ? p->recs2 : p->recs1);
    uint8_t* fin_aux = (t8 || tight) ? nullptr : dst ? const_cast<uint8_t*>(dst->aux) : (sb == 0 ? a2 : a1);
    unsigned long long* fin_base = dst ? const_cast<unsigned long long*>(dst->base) : p->group_base;
    LevelCfg last = level_narrow(p->cfg, sb, false, t8);
    if (tight) last.rstart = h->rstart;
    if (sb == 0) {
        run_level(h, p, last, p->recs1, a1, fin, fin_aux, fin_base);
    } else {
        run_level(h, p, level_narrow(p->cfg, sb, true, t8), p->recs1, a1, p->recs2, a2);
        (void)hipMemcpyAsync(p->seg_off, p->group_base, (size_t)(((1u << NARROW_CBITS) << sb) + 1) * 8, hipMemcpyDeviceToDevice, h->stream);
        run_level(h, p, last, p->recs2, a2, fin, fin_aux, fin_base);
    }
    *sorted = fin; *sorted_aux = fin_aux;
}

// a set in the scratch buffers is applied at once (the scratch is reused by the next slice)
static int pend_or_apply(kq_handle* h, const P3Set& set, int fmt, int aux_fmt, bool in_arena) {
    int rc = pend_add(h, set, fmt, aux_fmt);
    if (rc) return rc;
    return in_arena ? KQ_OK : flush_pending(h);
}

// partitioned count of one batch of bases: P1 (coarse split) -> P2 (region split) -> a pending set for P3 (LDS regions)
static int count_partitioned(kq_handle* h, const uint8_t* ab, uint64_t lead, uint64_t len, EmitRange er, const uint16_t* pinv = nullptr) {
    PartPlan p;
    PartCfg c0; plan_cfg(h, &c0, true);
    int rc = plan_alloc(h, &p, len, n_tiles_of(lead, len), c0.n_coarse, true);
    if (rc) return rc;
    // 5-byte records up to k = 21 and 8-byte hash-remainder records above k = 28 (tables with hash-prefix buckets),
    // 8-byte packed records up to k = 28, hash + edge byte otherwise
    if (h->k > PART_MAX_K && p.fmt != FMT_TOP8) p.fmt = FMT_WIDE;
    const bool has_aux = p.fmt == FMT_WIDE || p.fmt == FMT_NARROW;
    uint8_t* a1 = has_aux ? p.aux1 : nullptr;
    uint8_t* a2 = has_aux ? p.aux2 : nullptr;
    p.cfg.filt_lo = h->filt_lo; p.cfg.filt_hi = h->filt_hi;      // KQ_OPT_COUNT_MAP_RANGE
    if (h->windowed) { p.cfg.win_lo = h->win_lo; p.cfg.win_hi = h->win_hi; }      // a window drops foreign buckets in P1
    const bool leveled = p.fmt == FMT_NARROW || p.fmt == FMT_TOP8 || p.two_level;
    P3Set set; bool in_arena = false;
    const bool tight = tight_ok(h, p);
    const int set_fmt = tight ? FMT_TIGHT : p.fmt;
    // A map-range pass (KQ_OPT_COUNT_MAP_RANGE: the reference's memory-bounded mode, src/kreeq.cpp:59-74) keeps a fraction of
    // the k-mers it scans: its pending set is sized by the record count P1 found, not by the starts of the slice, so that the
    // arena holds as many RECORDS per table pass as it would without the filter (one small read-back per slice)
    const bool filtered = p.cfg.filt_lo != 0 || p.cfg.filt_hi != p.cfg.map_count || h->windowed;      // (a window keeps its own buckets only)
    if (leveled && !filtered) { rc = arena_take(h, p.n_max, set_fmt, p.R, &set, &in_arena); if (rc) return rc; }
    marks_reset(h);
    mark(h, "start");
    run_p1(h, &p, p.cfg, ab, lead, len, er, p.recs1, a1, AUX_IDX6, pinv);
    if (leveled && filtered) {
        unsigned long long n_recs = 0;
        HIPC(hipMemcpyAsync(&n_recs, p.total, sizeof n_recs, hipMemcpyDeviceToHost, h->stream));
        HIPC(hipStreamSynchronize(h->stream));
        rc = arena_take(h, std::min<uint64_t>(p.n_max, ((uint64_t)n_recs + 15) & ~7ull), set_fmt, p.R, &set, &in_arena); if (rc) return rc;
    }
    if (p.fmt == FMT_NARROW || p.fmt == FMT_TOP8) {
        const uint64_t* sorted; const uint8_t* sorted_aux;
        run_narrow_levels(h, &p, &sorted, &sorted_aux, in_arena ? &set : nullptr, tight);
        if (!in_arena) set = P3Set{sorted, sorted_aux, p.group_base, p.n_max};
    } else if (p.two_level) {
        if (in_arena) run_level(h, &p, level_coarse_to_regions(p.cfg), p.recs1, a1, const_cast<uint64_t*>(set.recs), const_cast<uint8_t*>(set.aux), const_cast<unsigned long long*>(set.base));
        else { run_level(h, &p, level_coarse_to_regions(p.cfg), p.recs1, a1, p.recs2, a2); set = P3Set{p.recs2, a2, p.group_base, p.n_max}; }
    } else {
        set = P3Set{p.recs1, a1, p.seg_off, p.n_max};          // bins were the regions themselves
    }
    HIPC(hipGetLastError());
    return pend_or_apply(h, set, set_fmt, tight ? AUX_TIGHT : AUX_IDX6, in_arena);
}
// partitioned count of n records already on the device (multi-GPU receive side, kq_insert_records_dev).
// d_aux == nullptr: packed 8-byte records; else WIDE records with d_aux in `aux_fmt`.
// `raw`: d_recs holds raw keys (kq_insert_records); otherwise the records of kq_emit_packed_dev (table hashes)
static int count_partitioned_records(kq_handle* h, const uint64_t* d_recs, const uint8_t* d_aux, int aux_fmt, uint64_t n, bool raw) {
    PartPlan p;
    int rc = plan_alloc(h, &p, n, 0, 1, /*allow_narrow=*/!d_aux && !raw);
    if (rc) return rc;
    P3Set set; bool in_arena = false;
    if (p.fmt == FMT_NARROW) {
        // packed records of kq_emit_packed_dev on a narrow-eligible table: the first level splits on the top 8 hash
        // bits and writes 5-byte records, the rest is the narrow path of count_partitioned
        const bool tight = tight_ok(h, p);
        rc = arena_take(h, p.n_max, tight ? FMT_TIGHT : p.fmt, p.R, &set, &in_arena); if (rc) return rc;
        hipLaunchKernelGGL(k_set2, dim3(1), dim3(1), 0, h->stream, p.seg_off, 0ull, (unsigned long long)n);
        LevelCfg first = level_flat_to_coarse(p.cfg);
        first.top8 = 1; first.nb = 1u << NARROW_CBITS;
        run_level(h, &p, first, d_recs, nullptr, p.recs1, p.aux1);
        HIPC(hipMemcpyAsync(p.seg_off, p.group_base, (size_t)((1u << NARROW_CBITS) + 1) * 8, hipMemcpyDeviceToDevice, h->stream));
        const uint64_t* sorted; const uint8_t* sorted_aux;
        run_narrow_levels(h, &p, &sorted, &sorted_aux, in_arena ? &set : nullptr, tight);
        if (!in_arena) set = P3Set{sorted, sorted_aux, p.group_base, p.n_max};
        HIPC(hipGetLastError());
        return pend_or_apply(h, set, tight ? FMT_TIGHT : FMT_NARROW, tight ? AUX_TIGHT : AUX_IDX6, in_arena);
    }
    if (d_aux) p.fmt = FMT_WIDE;
    uint8_t* a1 = d_aux ? p.aux1 : nullptr;
    uint8_t* a2 = d_aux ? p.aux2 : nullptr;
    hipLaunchKernelGGL(k_set2, dim3(1), dim3(1), 0, h->stream, p.seg_off, 0ull, (unsigned long long)n);
    LevelCfg first = level_flat_to_coarse(p.cfg);
    first.in_raw = raw ? 1 : 0; first.k = (uint32_t)h->k;
    run_level(h, &p, first, d_recs, d_aux, p.recs1, a1);        // group_base = coarse offsets
    if (p.two_level) {
        // the coarse offsets become the segment table of the next level
        HIPC(hipMemcpyAsync(p.seg_off, p.group_base, (size_t)(p.cfg.n_coarse + 1) * 8, hipMemcpyDeviceToDevice, h->stream));
        run_level(h, &p, level_coarse_to_regions(p.cfg), p.recs1, a1, p.recs2, a2);
        set = P3Set{p.recs2, a2, p.group_base, p.n_max};
    } else {
        set = P3Set{p.recs1, a1, p.group_base, p.n_max};
    }
    HIPC(hipGetLastError());
    return pend_or_apply(h, set, p.fmt, aux_fmt, false);
}

// count a resident sequence: ASCII bytes (d_inv == nullptr) or the packed form (d_bases = the code words, d_inv = the masks)
static int count_seq_dev(kq_handle* h, const char* d_bases, const uint16_t* d_inv, uint64_t len) {
    if (!h || (!d_bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (len < (uint64_t)h->k) return KQ_OK;                                  // src/graph-builder.cpp:60
    const uint64_t kmers = len - h->k + 1;
    // a resident batch of any size is processed in slices of <= 2^28 k-mer starts (the partition
    // scratch is 16 B per start); a slice scans one extra base on the left and k on the right, so
    // k-mers and edges across a cut are seen exactly once
    // Without pending sets (KQ_OPT_PENDING_BYTES = 0) the partitioned path streams the whole table once per slice, so a
    // slice should bring a few records per slot: with a large table (and memory to spare for 16 B of scratch per start)
    // slices grow up to 2^31 starts.  With pending sets the table pass is shared by many slices and the default holds.
    uint64_t slice = h->slice_kmers;
    if (h->pend_budget != 0 && !h->slice_user && kmers > slice) {
        // large tables split records over up to 65536 segments before the last level: slices of up to 2^31 starts keep
        // a few work units per segment and the number of pending sets per table pass low (one slice instead of two per
        // 1.3e9-k-mer batch: table pass -3 %, step -1.8 % at 1000 Mbp); 10 B of scratch per start for the default k, so
        // only as far as a quarter of the free HBM pays for it
        const uint64_t base_slice = slice;
        for (uint64_t cap : {1ull << 31, 1ull << 30}) {
            slice = std::min<uint64_t>(cap, std::max<uint64_t>(base_slice, h->n_slots() / 4));
            const uint64_t n_slices = (kmers + slice - 1) / slice;
            slice = (kmers + n_slices - 1) / n_slices;           // equal slices
            size_t free_b = 0, total_b = 0;
            if (cap == (1ull << 30) || slice <= (1ull << 30)) break;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + h->part_bytes >= (size_t)11 * slice + ((size_t)2 << 30)) break;   // the plan takes ~10.2 B per start
        }
    }
    if (h->pend_budget == 0 && !h->slice_user && kmers > slice && 2 * h->n_slots() > slice) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t by_mem = (uint64_t)((free_b + h->part_bytes) / 24);          // 16 B scratch + margin per start
            slice = std::max(slice, std::min<uint64_t>(std::min<uint64_t>(2 * h->n_slots(), 1ull << 31), by_mem));
        }
    }
    // Fork / join: with two or more slices in this call, consecutive slices alternate between two internal streams (and two
    // scratch sets), both behind the caller's position on `base` (the input is ready there), and `base` is joined with them
    // before the call returns -- the caller's stream-ordered view of the handle does not change, and no fork outlives a call.
    struct ForkGuard {
        kq_handle* h; bool swapped = false;
        void leave() { if (swapped) { std::swap(h->part, h->fork_part[1]); std::swap(h->part_bytes, h->fork_part_bytes[1]); std::swap(h->part2, h->fork_part2); swapped = false; } h->stream = h->base; }
        ~ForkGuard() { leave(); (void)join_forks(h); }
    } guard{h};
    // (not in a map-range pass: its slices read their record count back, and measured at full size the kernels of two such
    // slices only time-slice the GPU -- no gain for a second scratch set taken from the arena)
    const bool forked = h->overlap && !h->profile && h->count_path != 1 && (kmers + slice - 1) / slice >= 2 && slice >= (1u << 20) &&
                        (h->overlap == 2 || (h->filt_lo == 0 && h->filt_hi == (uint32_t)h->map_count && !h->windowed));
    if (forked) {
        for (int w = 0; w < 2; ++w) if (!h->fork_stream[w]) HIPC(hipStreamCreateWithFlags(&h->fork_stream[w], hipStreamNonBlocking));
        hipEvent_t e_in; int erc = ev_get(h, &e_in); if (erc) return erc;
        HIPC(hipEventRecord(e_in, h->base));
        for (int w = 0; w < 2; ++w) HIPC(hipStreamWaitEvent(h->fork_stream[w], e_in, 0));
    }
    uint64_t slice_no = 0;
    for (uint64_t a = 0; a < kmers; a += slice, ++slice_no) {
        const uint64_t b = std::min(kmers, a + slice);
        int rc = reserve(h, b - a, b - a);
        if (rc) return rc;
        const uint64_t sub_off = a ? a - 1 : 0;
        const uint64_t sub_len = std::min(len, b + h->k) - sub_off;
        const EmitRange er{a - sub_off, b - sub_off};
        const uint8_t* ab; uint64_t lead;
        const uint16_t* pinv = nullptr;
        if (d_inv) { ab = (const uint8_t*)(reinterpret_cast<const uint32_t*>(d_bases) + sub_off / 16); pinv = d_inv + sub_off / 16; lead = sub_off % 16; }
        else aligned_view(d_bases + sub_off, &ab, &lead);
        // a table pass streams the whole table (2 x 16 B per slot) on top of ~37 B per record; the atomic path costs
        // ~95 ps per record whatever the table size (~480 B at the part's streaming rate): partition unless the table
        // is more than ~200 B per record the pass will apply -- this slice, plus what is pending already, times the
        // slices of this size the arena can still take (at most 8: the caller may read the table any time)
        double pass_records = (double)(b - a);
        if (h->pend_budget != 0) {
            const double per_set = (double)set_bytes(b - a, FMT_PACK8, h->n_regions);
            const double room = h->arena ? (double)h->arena_bytes : 4.0 * (double)h->n_slots() * sizeof(Slot);
            pass_records = (double)h->pend_records + (double)(b - a) * std::max(1.0, std::min(8.0, room / per_set));
        }
        bool part = (b - a) >= (1u << 20) && part_table_ok(h, true) &&
                    (double)h->n_slots() * sizeof(Slot) <= 200.0 * pass_records;
        if (h->count_path == 1) part = false;
        if (h->count_path == 2) {
            if (!part_table_ok(h, true)) return fail(KQ_ERR_INVALID, "table too large for the partitioned path");
            part = true;
        }
        if (part) {
            if (forked) {
                const int w = (int)(slice_no & 1);
                if (w == 1) { std::swap(h->part, h->fork_part[1]); std::swap(h->part_bytes, h->fork_part_bytes[1]); std::swap(h->part2, h->fork_part2); guard.swapped = true; }
                h->stream = h->fork_stream[w];
                h->fork_busy[w] = true;
            }
            rc = count_partitioned(h, ab, lead, sub_len, er, pinv);
            if (forked) h->fork_busy[slice_no & 1] = true;            // (a table pass inside joined the stream; what followed has not been joined)
            guard.leave();
            if (rc) return rc;
            continue;
        }
        if (forked) { rc = join_forks(h); if (rc) return rc; }       // the direct kernel updates the table: behind everything in flight
        PartCfg filt; plan_cfg(h, &filt);
        filt.filt_lo = h->filt_lo; filt.filt_hi = h->filt_hi;
        materialize(h);
        h->table_empty = false;
        hipLaunchKernelGGL(k_count_direct, dim3(grid_for(h, n_tiles_of(lead, sub_len), 1, 32)), dim3(TILE_THREADS), 0, h->stream,
                           h->view(), ab, lead, sub_len, h->k, er, filt, pinv);
        HIPC(hipGetLastError());
    }
    return KQ_OK;
}
int kq_count_batch_dev(kq_handle* h, const char* d_bases, uint64_t len) { return count_seq_dev(h, d_bases, nullptr, len); }
int kq_count_packed_dev(kq_handle* h, const uint32_t* d_codes, const uint16_t* d_inv, uint64_t n_bases) {
    if (!d_inv && n_bases) return fail(KQ_ERR_INVALID, "null argument");
    return count_seq_dev(h, (const char*)d_codes, d_inv, n_bases);
}
// 16 bases -> one u32 of 2-bit codes (base i at bits 2i: A C G T = 0 1 2 3, case-blind) + one u16 of invalid-base bits
// (anything but ACGT/acgt, and the positions behind `len` in the last unit): the tile scanner's own LDS format
void kq_pack_bases(const char* bases, uint64_t len, uint32_t* codes, uint16_t* inv) {
    static const struct Lut { uint8_t v[256]; Lut() { for (int i = 0; i < 256; ++i) v[i] = 4; v['A'] = v['a'] = 0; v['C'] = v['c'] = 1; v['G'] = v['g'] = 2; v['T'] = v['t'] = 3; } } lut;
    const uint64_t full = len / 16;
    const uint8_t* b = (const uint8_t*)bases;
    for (uint64_t u = 0; u < full; ++u, b += 16) {
        uint32_t c = 0, m = 0;
        for (int i = 0; i < 16; ++i) { const uint32_t x = lut.v[b[i]]; c |= (x & 3u) << (2 * i); m |= (x >> 2) << i; }
        codes[u] = c; inv[u] = (uint16_t)m;
    }
    if (len % 16) {
        uint32_t c = 0, m = 0xFFFFu;
        for (uint64_t i = 0; i < len % 16; ++i) { const uint32_t x = lut.v[b[i]]; c |= (x & 3u) << (2 * i); if (!(x >> 2)) m &= ~(1u << i); }
        codes[full] = c; inv[full] = (uint16_t)m;
    }
}
int kq_count_batch(kq_handle* h, const char* bases, uint64_t len) {
    if (!h || (!bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    void* d = nullptr;
    int rc = stage_in(h, bases, len, &d);
    if (rc) return rc;
    rc = kq_count_batch_dev(h, (const char*)d, len);
    if (rc) return rc;
    // the staging buffer is free again once the stream has drained; records may stay pending (errors of a pending
    // pass surface at the next kq_sync / read of the table)
    HIPC(hipStreamSynchronize(h->stream));
    return KQ_OK;
}

// page-locked host memory = ordinary pages registered with the runtime: measured on this box, hipHostRegister of 32 MiB
// takes 0.2-0.5 ms and the copies then run at the same 50+ GB/s as from hipHostMalloc memory, whose allocation costs
// 0.14 ms per MiB (and its release half of that again)
void* kq_host_alloc(uint64_t bytes) {
    const size_t n = ((size_t)(bytes ? bytes : 1) + 4095) & ~(size_t)4095;
    void* p = aligned_alloc(4096, n);
    if (!p) return nullptr;
    if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); free(p); return nullptr; }
    return p;
}
void kq_host_free(void* p) { if (p) { (void)hipHostUnregister(p); free(p); } }

// shared by the ASCII and the packed entry point: copy (a [+ b]) into a staging slot, count behind the copy
static int ingest_async(kq_handle* h, const void* a, size_t a_bytes, const void* b, size_t b_bytes, uint64_t n_bases, uint64_t* ticket) {
    HIPC(hipSetDevice(h->device));
    // One caller at a time, copy included: concurrent copies from pageable memory on several streams were measured and are
    // 3 x SLOWER than one after the other (HIP stages them through per-stream buffers it first has to set up)
    std::lock_guard<std::mutex> lock(h->in_m);
    if (!h->copy_stream) {
        HIPC(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < kq_handle::IN_SLOTS; ++i) HIPC(hipEventCreateWithFlags(&h->in_consumed[i], hipEventDisableTiming));
        h->in_copied.resize(kq_handle::IN_TICKETS);
        for (auto& e : h->in_copied) HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const uint64_t t = h->in_next++;
    const int s = (int)(t % kq_handle::IN_SLOTS);
    hipEvent_t copied = h->in_copied[t % kq_handle::IN_TICKETS];
    *ticket = t;
    // the slot's previous reader must be done before it is overwritten (copy stream waits; the host does not)
    if (t >= (uint64_t)kq_handle::IN_SLOTS) HIPC(hipStreamWaitEvent(h->copy_stream, h->in_consumed[s], 0));
    const size_t b_off = (a_bytes + 63) & ~(size_t)63, bytes = b_off + b_bytes;
    if (h->in_bytes[s] < bytes + 64) {
        // growing a slot: nothing may still read the old buffer
        if (h->in_buf[s]) { HIPC(hipStreamSynchronize(h->copy_stream)); HIPC(hipStreamSynchronize(h->stream)); HIPC(hipFree(h->in_buf[s])); h->in_buf[s] = nullptr; h->in_bytes[s] = 0; }
        const size_t want = bytes + bytes / 4 + 4096;
        HIPC(hipMalloc(&h->in_buf[s], want));
        h->in_bytes[s] = want;
    }
    char* d = (char*)h->in_buf[s];
    if (a_bytes) HIPC(hipMemcpyAsync(d, a, a_bytes, hipMemcpyHostToDevice, h->copy_stream));
    if (b_bytes) HIPC(hipMemcpyAsync(d + b_off, b, b_bytes, hipMemcpyHostToDevice, h->copy_stream));
    HIPC(hipEventRecord(copied, h->copy_stream));
    HIPC(hipStreamWaitEvent(h->stream, copied, 0));
    int rc = count_seq_dev(h, d, b ? (const uint16_t*)(d + b_off) : nullptr, n_bases);
    HIPC(hipEventRecord(h->in_consumed[s], h->stream));
    return rc;
}
int kq_count_batch_async(kq_handle* h, const char* bases, uint64_t len, uint64_t* ticket) {
    if (!h || !ticket || (!bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    return ingest_async(h, bases, len, nullptr, 0, len, ticket);
}
int kq_count_packed_async(kq_handle* h, const uint32_t* codes, const uint16_t* inv, uint64_t n_bases, uint64_t* ticket) {
    if (!h || !ticket || ((!codes || !inv) && n_bases)) return fail(KQ_ERR_INVALID, "null argument");
    const uint64_t units = (n_bases + 15) / 16;
    if (!n_bases) return ingest_async(h, nullptr, 0, nullptr, 0, 0, ticket);
    return ingest_async(h, codes, units * 4, inv, units * 2, n_bases, ticket);
}
int kq_host_wait(kq_handle* h, uint64_t ticket) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    if (ticket >= h->in_next) return fail(KQ_ERR_INVALID, "unknown ticket %llu", (unsigned long long)ticket);
    if (h->in_next - ticket >= (uint64_t)kq_handle::IN_TICKETS) return KQ_OK;      // long recycled: that copy finished many batches ago
    HIPC(hipEventSynchronize(h->in_copied[ticket % kq_handle::IN_TICKETS]));
    return KQ_OK;
}

static int emit_ordered(kq_handle* h, const char* d_bases, uint64_t len, uint64_t* d_keys, uint8_t* d_edges, uint64_t cap,
                        uint64_t* n_out) {
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    const uint64_t nt = n_tiles_of(lead, len);
    int rc = ensure_buf(&h->scratch, &h->scratch_bytes, (nt + 2) * sizeof(unsigned long long));
    if (rc) return rc;
    unsigned long long* tile_counts = (unsigned long long*)h->scratch;
    unsigned long long* total = tile_counts + nt;
    int grid = grid_for(h, nt, 1);
    hipLaunchKernelGGL(k_emit_count, dim3(grid), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, tile_counts);
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, h->stream, tile_counts, nt, total);
    unsigned long long n = 0;
    HIPC(hipMemcpyAsync(&n, total, sizeof n, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    *n_out = n;
    if (!d_keys || !d_edges) return KQ_OK;
    if (n > cap) return fail(KQ_ERR_CAPACITY, "record buffer too small: need %llu, have %llu", n, (unsigned long long)cap);
    hipLaunchKernelGGL(k_emit_write, dim3(grid), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, tile_counts, d_keys, d_edges, cap);
    HIPC(hipGetLastError());
    return KQ_OK;
}

int kq_emit_records(kq_handle* h, const char* bases, uint64_t len, uint64_t* keys, uint8_t* edges, uint64_t cap, uint64_t* n_out) {
    if (!h || !n_out || (!bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    *n_out = 0;
    if (len < (uint64_t)h->k) return KQ_OK;
    void* d = nullptr;
    int rc = stage_in(h, bases, len, &d);
    if (rc) return rc;
    if (!keys || !edges) return emit_ordered(h, (const char*)d, len, nullptr, nullptr, 0, n_out);
    rc = emit_ordered(h, (const char*)d, len, nullptr, nullptr, 0, n_out);
    if (rc) return rc;
    if (*n_out > cap) return fail(KQ_ERR_CAPACITY, "record buffer too small: need %llu, have %llu", (unsigned long long)*n_out, (unsigned long long)cap);
    uint64_t n = *n_out;
    uint64_t* dk = nullptr; uint8_t* de = nullptr;
    if (n) {
        HIPC(hipMalloc((void**)&dk, n * sizeof(uint64_t)));
        if (hipMalloc((void**)&de, n) != hipSuccess) { (void)hipFree(dk); return fail(KQ_ERR_NOMEM, "record buffer allocation failed"); }
        rc = emit_ordered(h, (const char*)d, len, dk, de, n, n_out);
        if (!rc) {
            hipError_t e1 = hipMemcpyAsync(keys, dk, n * sizeof(uint64_t), hipMemcpyDeviceToHost, h->stream);
            hipError_t e2 = hipMemcpyAsync(edges, de, n, hipMemcpyDeviceToHost, h->stream);
            hipError_t e3 = hipStreamSynchronize(h->stream);
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(KQ_ERR_HIP, "copying records back failed");
        }
        (void)hipFree(dk); (void)hipFree(de);
    }
    return rc;
}

int kq_emit_partitioned_dev(kq_handle* h, const char* d_bases, uint64_t len, int n_parts, uint64_t* d_keys, uint8_t* d_edges,
                            uint64_t cap, uint64_t* part_counts) {
    if (!h || !part_counts || n_parts < 1 || n_parts > h->map_count || n_parts >= NB_MAX || (!d_bases && len))
        return fail(KQ_ERR_INVALID, "bad argument");
    HIPC(hipSetDevice(h->device));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = 0;
    if (len < (uint64_t)h->k) return KQ_OK;
    if (len - h->k + 1 >= (1ull << 32) - 16) return fail(KQ_ERR_INVALID, "an owner split handles fewer than 2^32 k-mer starts per call (got %llu): cut the batch", (unsigned long long)(len - h->k + 1));
    if (cap < len - h->k + 1 || !d_keys || !d_edges) return fail(KQ_ERR_CAPACITY, "record buffer too small: need room for %llu records",
                                                                  (unsigned long long)(len - h->k + 1));
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    PartPlan p;
    uint32_t sb = 0;                                             // owner part x sub-bin by lane, see kq_emit_packed_dev
    while (((uint32_t)n_parts << (sb + 1)) <= 256u && sb < 5) ++sb;
    const uint32_t bins = (uint32_t)n_parts << sb;
    int rc = plan_alloc(h, &p, 0, n_tiles_of(lead, len), bins);
    if (rc) return rc;
    PartCfg cfg = p.cfg;
    cfg.mode = 1; cfg.n_coarse = bins; cfg.owner_sub = sb;
    cfg.raw_out = 1;
    run_p1(h, &p, cfg, ab, lead, len, EmitRange{0, ~0ull}, d_keys, d_edges, AUX_EDGE_BYTE);      // WIDE records: key + reference edge byte
    std::vector<unsigned long long> off((size_t)bins + 1);
    HIPC(hipMemcpyAsync(off.data(), p.seg_off, off.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = off[((size_t)i + 1) << sb] - off[(size_t)i << sb];
    return KQ_OK;
}

int kq_emit_packed_dev(kq_handle* h, const char* d_bases, uint64_t len, int n_parts, uint64_t* d_recs, uint64_t cap,
                       uint64_t* part_counts) {
    if (!h || !part_counts || n_parts < 1 || n_parts > h->map_count || n_parts >= NB_MAX || (!d_bases && len))
        return fail(KQ_ERR_INVALID, "bad argument");
    if (h->k > PART_MAX_K) return fail(KQ_ERR_INVALID, "packed 8-byte records need k <= %d (use kq_emit_partitioned_dev)", PART_MAX_K);
    HIPC(hipSetDevice(h->device));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = 0;
    if (len < (uint64_t)h->k) return KQ_OK;
    if (len - h->k + 1 >= (1ull << 32) - 16) return fail(KQ_ERR_INVALID, "an owner split handles fewer than 2^32 k-mer starts per call (got %llu): cut the batch", (unsigned long long)(len - h->k + 1));
    if (cap < len - h->k + 1 || !d_recs) return fail(KQ_ERR_CAPACITY, "record buffer too small: need room for %llu records",
                                                       (unsigned long long)(len - h->k + 1));
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    PartPlan p;
    // up to 256 bins in all: owner part x sub-bin by lane (the order inside a part is unspecified anyway)
    uint32_t sb = 0;
    while (((uint32_t)n_parts << (sb + 1)) <= 256u && sb < 5) ++sb;
    const uint32_t bins = (uint32_t)n_parts << sb;
    int rc = plan_alloc(h, &p, 0, n_tiles_of(lead, len), bins);
    if (rc) return rc;
    PartCfg cfg = p.cfg;
    cfg.mode = 1; cfg.n_coarse = bins; cfg.owner_sub = sb;
    run_p1(h, &p, cfg, ab, lead, len, EmitRange{0, ~0ull}, d_recs, nullptr, AUX_IDX6);
    std::vector<unsigned long long> off((size_t)bins + 1);
    HIPC(hipMemcpyAsync(off.data(), p.seg_off, off.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = off[((size_t)i + 1) << sb] - off[(size_t)i << sb];
    return KQ_OK;
}

// ---- multi-GPU exchange of 5-byte records (k <= 21) -------------------------------------------------
// Ownership is by hash-prefix bucket range, so the sender's work is the first split of the single-GPU count and nothing
// else: P1 writes the bucket-sorted 5-byte records straight into the send buffers, the part of every destination rank is
// one contiguous run.  Receiver (a table that is the window of its buckets, KQ_OPT_BUCKET_WINDOW): the runs of all peers
// are (peer, bucket) segments of its bucket -> region levels, so the received records enter the ordinary narrow path.
// 5 bytes per record cross xGMI.  kq_emit_packed_dev / kq_insert_packed_dev remain for k = 22..28 and small tables.
// (peer q, bucket b) run j = q * 256 + b of the received array -> input segment b * n_peers + q of the receive levels
__global__ void k_sharded_segments(const unsigned long long* __restrict__ start /*exclusive scan of the counts, peer-major*/,
                                   const unsigned long long* __restrict__ counts, uint32_t n_peers,
                                   unsigned long long* __restrict__ seg_lo, unsigned long long* __restrict__ seg_hi) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_peers << NARROW_CBITS) return;
    const uint32_t q = j >> NARROW_CBITS, b = j & ((1u << NARROW_CBITS) - 1u);
    seg_lo[b * n_peers + q] = start[j];
    seg_hi[b * n_peers + q] = start[j] + counts[j];
}

// first bucket of part p of n_parts: the parts are contiguous ranges of the 256 hash-prefix buckets (kreeq_amd/dist.py: bucket_range)
static inline uint32_t part_first_bucket(int p, int n_parts) { return (uint32_t)(((uint64_t)p * 256 + n_parts - 1) / n_parts); }
__global__ void k_bucket_meta(const unsigned long long* __restrict__ seg_off, uint32_t n_parts, unsigned long long* __restrict__ counts) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_parts << NARROW_CBITS) return;
    const uint32_t p = j >> NARROW_CBITS, b = j & ((1u << NARROW_CBITS) - 1u);
    const uint32_t lo = (uint32_t)(((uint64_t)p * 256 + n_parts - 1) / n_parts), hi = (uint32_t)(((uint64_t)(p + 1) * 256 + n_parts - 1) / n_parts);
    counts[j] = (b >= lo && b < hi) ? seg_off[b + 1] - seg_off[b] : 0ull;
}
int kq_emit_sharded_dev(kq_handle* h, const char* d_bases, uint64_t len, int n_parts, uint32_t* d_recs, uint8_t* d_aux, uint64_t cap,
                        uint64_t* d_bucket_counts, uint64_t* part_counts) {
    if (!h || !d_bucket_counts || n_parts < 1 || n_parts > 256 || (!d_bases && len))
        return fail(KQ_ERR_INVALID, "bad argument");
    if (h->k > (int)NARROW_MAX_K) return fail(KQ_ERR_INVALID, "5-byte records need k <= %u (use kq_emit_packed_dev)", NARROW_MAX_K);
    HIPC(hipSetDevice(h->device));
    for (int i = 0; part_counts && i < n_parts; ++i) part_counts[i] = 0;
    const uint64_t n_groups = (uint64_t)n_parts << NARROW_CBITS;
    HIPC(hipMemsetAsync(d_bucket_counts, 0, n_groups * 8, h->stream));
    if (len < (uint64_t)h->k) { if (part_counts) HIPC(hipStreamSynchronize(h->stream)); return KQ_OK; }
    if (len - h->k + 1 >= (1ull << 32) - 16) return fail(KQ_ERR_INVALID, "a bucket split handles fewer than 2^32 k-mer starts per call (got %llu): cut the batch", (unsigned long long)(len - h->k + 1));
    if (cap < len - h->k + 1 || !d_recs || !d_aux) return fail(KQ_ERR_CAPACITY, "record buffers too small: need room for %llu records", (unsigned long long)(len - h->k + 1));
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    PartPlan p;
    int rc = plan_alloc(h, &p, len, n_tiles_of(lead, len), 1u << NARROW_CBITS, true, n_groups);
    if (rc) return rc;
    // the bucket split of the single-GPU count IS the owner split: bucket = top 8 hash bits, whatever this rank's table
    // looks like, and every part is a range of buckets -- P1 writes straight into the send buffers
    PartCfg cfg = p.cfg;
    cfg.mode = 0; cfg.narrow = 1; cfg.n_coarse = 1u << NARROW_CBITS; cfg.sub_bits = 0; if (cfg.g_shift == 0) cfg.g_shift = 1;
    cfg.filt_lo = 0; cfg.filt_hi = cfg.map_count;
    run_p1(h, &p, cfg, ab, lead, len, EmitRange{0, ~0ull}, (uint64_t*)d_recs, d_aux, AUX_IDX6);
    hipLaunchKernelGGL(k_bucket_meta, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, h->stream, p.seg_off, (uint32_t)n_parts, (unsigned long long*)d_bucket_counts);
    HIPC(hipGetLastError());
    if (!part_counts) return KQ_OK;               // asynchronous: the caller takes the part sizes from the rows of d_bucket_counts
    std::vector<unsigned long long> off((size_t)(1u << NARROW_CBITS) + 1);
    HIPC(hipMemcpyAsync(off.data(), p.seg_off, off.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = off[part_first_bucket(i + 1, n_parts)] - off[part_first_bucket(i, n_parts)];
    return KQ_OK;
}

int kq_insert_sharded_dev(kq_handle* h, const uint32_t* d_recs, const uint8_t* d_aux, uint64_t n, int n_peers, const uint64_t* d_bucket_counts) {
    if (!h || ((!d_recs || !d_aux) && n) || !d_bucket_counts || n_peers < 1 || n_peers > 256) return fail(KQ_ERR_INVALID, "bad argument");
    if (h->k > (int)NARROW_MAX_K) return fail(KQ_ERR_INVALID, "5-byte records need k <= %u (use kq_insert_packed_dev)", NARROW_MAX_K);
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    if (n >= (1ull << 32) - 16) return fail(KQ_ERR_INVALID, "a partition pass handles fewer than 2^32 records (got %llu)", (unsigned long long)n);
    int rc = reserve(h, n, n);
    if (rc) return rc;
    const uint32_t n_in = (uint32_t)n_peers << NARROW_CBITS;      // input segments: one run per (bucket, peer)
    PartPlan p;
    rc = plan_alloc(h, &p, n, 0, 1, true, n_in);
    if (rc) return rc;
    if (p.fmt != FMT_NARROW) return fail(KQ_ERR_INVALID, "the table is too small for 5-byte records (fewer than 2048 regions): use kq_insert_packed_dev");
    // segment table of the received array (peer-major runs) in logical order (bucket-major)
    rc = ensure_buf(&h->scratch, &h->scratch_bytes, (size_t)(3 * (n_in + 2)) * 8);
    if (rc) return rc;
    unsigned long long* start = (unsigned long long*)h->scratch;
    unsigned long long* seg_lo = start + n_in + 2;
    unsigned long long* seg_hi = seg_lo + n_in + 2;
    HIPC(hipMemcpyAsync(start, d_bucket_counts, (size_t)n_in * 8, hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, h->stream, start, (uint64_t)n_in, start + n_in);
    hipLaunchKernelGGL(k_sharded_segments, dim3((n_in + 255) / 256), dim3(256), 0, h->stream, start, (const unsigned long long*)d_bucket_counts, (uint32_t)n_peers, seg_lo, seg_hi);
    P3Set set; bool in_arena = false;
    const bool tight = tight_ok(h, p);
    rc = arena_take(h, p.n_max, tight ? FMT_TIGHT : p.fmt, p.R, &set, &in_arena); if (rc) return rc;
    const uint32_t sb = p.cfg.sub_bits;
    uint64_t* fin = in_arena ? const_cast<uint64_t*>(set.recs) : p.recs2;
    uint8_t* fin_aux = tight ? nullptr : in_arena ? const_cast<uint8_t*>(set.aux) : p.aux2;
    unsigned long long* fin_base = in_arena ? const_cast<unsigned long long*>(set.base) : p.group_base;
    unsigned long long* own_seg_off = p.seg_off;
    LevelCfg first = level_narrow(p.cfg, sb, sb != 0);
    first.n_seg = n_in; first.spb = (uint32_t)n_peers;
    if (tight && sb == 0) first.rstart = h->rstart;               // it is the last level as well
    p.seg_off = seg_lo;                                           // the first level reads the received runs
    if (sb == 0) {
        run_level(h, &p, first, (const uint64_t*)d_recs, d_aux, fin, fin_aux, fin_base, seg_hi);
        p.seg_off = own_seg_off;
    } else {
        run_level(h, &p, first, (const uint64_t*)d_recs, d_aux, p.recs1, p.aux1, nullptr, seg_hi);
        p.seg_off = own_seg_off;
        HIPC(hipMemcpyAsync(p.seg_off, p.group_base, (size_t)(((1u << NARROW_CBITS) << sb) + 1) * 8, hipMemcpyDeviceToDevice, h->stream));
        if (!in_arena) { fin = p.recs2; fin_aux = tight ? nullptr : p.aux2; }
        LevelCfg last = level_narrow(p.cfg, sb, false);
        if (tight) last.rstart = h->rstart;
        run_level(h, &p, last, p.recs1, p.aux1, fin, fin_aux, fin_base);
    }
    if (!in_arena) set = P3Set{fin, fin_aux, fin_base, p.n_max};
    HIPC(hipGetLastError());
    return pend_or_apply(h, set, tight ? FMT_TIGHT : FMT_NARROW, tight ? AUX_TIGHT : AUX_IDX6, in_arena);
}

int kq_insert_packed_dev(kq_handle* h, const uint64_t* d_recs, uint64_t n) {
    if (!h || (!d_recs && n)) return fail(KQ_ERR_INVALID, "null argument");
    if (h->k > PART_MAX_K) return fail(KQ_ERR_INVALID, "packed 8-byte records need k <= %d (use kq_insert_records_dev)", PART_MAX_K);
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    int rc = reserve(h, n, n);
    if (rc) return rc;
    if (!part_table_ok(h, true)) return fail(KQ_ERR_INVALID, "table too large for the partitioned path");
    return count_partitioned_records(h, d_recs, nullptr, AUX_IDX6, n, false);
}

int kq_insert_records_dev(kq_handle* h, const uint64_t* d_keys, const uint8_t* d_edges, uint64_t n) {
    if (!h || ((!d_keys || !d_edges) && n)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    int rc = reserve(h, n, n);
    if (rc) return rc;
    if ((h->count_path == 2 || (h->count_path == 0 && n >= (1u << 20))) && h->n_regions <= (1ull << 20)) {   // WIDE records: key + reference edge byte
        return count_partitioned_records(h, d_keys, d_edges, AUX_EDGE_BYTE, n, true);
    }
    materialize(h);
    h->table_empty = false;
    hipLaunchKernelGGL(k_insert_records, dim3(grid_for(h, n, 256)), dim3(256), 0, h->stream, h->view(), d_keys, d_edges, n);
    HIPC(hipGetLastError());
    return KQ_OK;
}
int kq_insert_records(kq_handle* h, const uint64_t* keys, const uint8_t* edges, uint64_t n) {
    if (!h || ((!keys || !edges) && n)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    int rc = ensure_buf(&h->stage, &h->stage_bytes, n * 9 + 64);
    if (rc) return rc;
    uint64_t* dk = (uint64_t*)h->stage;
    uint8_t* de = (uint8_t*)h->stage + n * 8;
    HIPC(hipMemcpyAsync(dk, keys, n * 8, hipMemcpyHostToDevice, h->stream));
    HIPC(hipMemcpyAsync(de, edges, n, hipMemcpyHostToDevice, h->stream));
    rc = kq_insert_records_dev(h, dk, de, n);
    if (rc) return rc;
    return kq_sync(h);
}

// ---- summary ---------------------------------------------------------------------------------
struct SummaryHost { SummaryOut so; std::vector<unsigned long long> small; std::vector<uint32_t> big; };
static int run_summary(kq_handle* h, SummaryHost* r) {
    const uint64_t big_cap = h->hc_cap;     // cov >= 4096 implies a high-copy k-mer
    size_t need = sizeof(SummaryOut) + HIST_SMALL * sizeof(unsigned long long) + big_cap * sizeof(uint32_t);
    int rc = ensure_buf(&h->scratch, &h->scratch_bytes, need);
    if (rc) return rc;
    SummaryOut* d_so = (SummaryOut*)h->scratch;
    unsigned long long* d_small = (unsigned long long*)(d_so + 1);
    uint32_t* d_big = (uint32_t*)(d_small + HIST_SMALL);
    HIPC(hipMemsetAsync(h->scratch, 0, sizeof(SummaryOut) + HIST_SMALL * sizeof(unsigned long long), h->stream));
    SummaryOut init; memset(&init, 0, sizeof init); init.big_cap = big_cap;
    HIPC(hipMemcpyAsync(d_so, &init, sizeof init, hipMemcpyHostToDevice, h->stream));
    materialize(h);
    hipLaunchKernelGGL(k_summary, dim3(grid_for(h, h->n_slots(), 1024)), dim3(256), 0, h->stream, h->view(), d_so, d_small, d_big);
    r->small.resize(HIST_SMALL);
    HIPC(hipMemcpyAsync(&r->so, d_so, sizeof(SummaryOut), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(r->small.data(), d_small, HIST_SMALL * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    if (r->so.n_big > big_cap) return fail(KQ_ERR_HIP, "summary: high-coverage list overflow");
    r->big.resize(r->so.n_big);
    if (r->so.n_big) {
        HIPC(hipMemcpyAsync(r->big.data(), d_big, r->so.n_big * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        HIPC(hipStreamSynchronize(h->stream));
    }
    return KQ_OK;
}
int kq_summary(kq_handle* h, kq_stats* out) {
    if (!h || !out) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    int rc = check_errors(h);
    if (rc) return rc;
    SummaryHost r;
    rc = run_summary(h, &r);
    if (rc) return rc;
    out->total = r.so.total; out->unique = r.so.uniq; out->distinct = r.so.distinct; out->edges = r.so.edges;
    const uint64_t space = h->k < 32 ? (1ull << (2 * h->k)) : 0ull;          // src/graph-builder.cpp:286
    out->missing = space - out->distinct;
    return KQ_OK;
}
int kq_histogram(kq_handle* h, uint64_t* cov, uint64_t* cnt, uint64_t cap, uint64_t* n_out) {
    if (!h || !n_out) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    int rc = check_errors(h);
    if (rc) return rc;
    SummaryHost r;
    rc = run_summary(h, &r);
    if (rc) return rc;
    std::vector<std::pair<uint64_t, uint64_t>> hist;
    for (uint32_t c = 0; c < HIST_SMALL; ++c) if (r.small[c]) hist.emplace_back(c, r.small[c]);
    std::sort(r.big.begin(), r.big.end());
    for (size_t i = 0; i < r.big.size();) {
        size_t j = i; while (j < r.big.size() && r.big[j] == r.big[i]) ++j;
        hist.emplace_back(r.big[i], j - i);
        i = j;
    }
    *n_out = hist.size();
    if (!cov || !cnt) return KQ_OK;
    if (hist.size() > cap) return fail(KQ_ERR_CAPACITY, "histogram buffer too small: need %zu", hist.size());
    for (size_t i = 0; i < hist.size(); ++i) { cov[i] = hist[i].first; cnt[i] = hist[i].second; }
    return KQ_OK;
}

// ---- lookup ----------------------------------------------------------------------------------
// K3 through the partition machinery (counters only): P1 -> (level) -> k_lookup_regions
static int lookup_partitioned(kq_handle* h, const uint8_t* ab, uint64_t lead, uint64_t len, EmitRange er, uint32_t cov_cutoff,
                              uint32_t map_lo, uint32_t map_hi, unsigned long long* d_counters) {
    PartPlan p;
    PartCfg c0; plan_cfg(h, &c0, true);
    int rc = plan_alloc(h, &p, len, n_tiles_of(lead, len), c0.n_coarse, true);
    if (rc) return rc;
    if (h->k > PART_MAX_K && p.fmt != FMT_TOP8) p.fmt = FMT_WIDE;
    const bool has_aux = p.fmt == FMT_WIDE || p.fmt == FMT_NARROW;
    uint8_t* a1 = has_aux ? p.aux1 : nullptr;
    uint8_t* a2 = has_aux ? p.aux2 : nullptr;
    p.cfg.filt_lo = map_lo; p.cfg.filt_hi = map_hi;               // the reference's range filter, src/kreeq.cpp:150
    run_p1(h, &p, p.cfg, ab, lead, len, er, p.recs1, a1, AUX_IDX6);
    const uint64_t* sorted = p.recs1; const uint8_t* sorted_aux = a1; const unsigned long long* base = p.seg_off;
    if (p.fmt == FMT_NARROW || p.fmt == FMT_TOP8) {
        run_narrow_levels(h, &p, &sorted, &sorted_aux);
        base = p.group_base;
    } else if (p.two_level) {
        run_level(h, &p, level_coarse_to_regions(p.cfg), p.recs1, a1, p.recs2, a2);
        sorted = p.recs2; sorted_aux = a2; base = p.group_base;
    }
    const uint32_t rps = (p.fmt == FMT_NARROW || p.fmt == FMT_TOP8) ? (uint32_t)(p.R >> NARROW_CBITS) : 1u;
    const dim3 grid((unsigned)std::min<uint64_t>(p.R, 1u << 30)), block(P3_THREADS);
#define KQ_LK(F) hipLaunchKernelGGL((k_lookup_regions<F>), grid, block, 0, h->stream, h->view(), sorted, sorted_aux, base, rps, cov_cutoff, d_counters)
    if (p.fmt == FMT_NARROW) KQ_LK(FMT_NARROW); else if (p.fmt == FMT_TOP8) KQ_LK(FMT_TOP8); else if (p.fmt == FMT_WIDE) KQ_LK(FMT_WIDE); else KQ_LK(FMT_PACK8);
#undef KQ_LK
    HIPC(hipGetLastError());
    return KQ_OK;
}

int kq_lookup_sequence_dev(kq_handle* h, const char* d_bases, uint64_t len, uint32_t cov_cutoff, uint16_t map_lo, uint16_t map_hi,
                           kq_dbgbase* d_per_base, uint64_t* d_counters) {
    if (!h || !d_counters || (!d_bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    if (map_lo > map_hi || map_hi > h->map_count) return fail(KQ_ERR_INVALID, "map range [%u,%u) outside [0,%d]", map_lo, map_hi, h->map_count);
    HIPC(hipSetDevice(h->device));
    if (len < (uint64_t)h->k) return KQ_OK;                                  // src/kreeq.cpp:123
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    const uint32_t map_mask = (h->map_count & (h->map_count - 1)) == 0 ? (uint32_t)h->map_count - 1 : 0;
    { int frc = flush_pending(h); if (frc) return frc; }
    materialize(h);
    // counters only, a sequence worth partitioning, and a table small enough next to it (the partitioned path
    // streams the whole table once per slice): regions staged in LDS instead of one random sector per k-mer
    const uint64_t kmers = len - h->k + 1;
    if (!d_per_base && h->lookup_path != 1 && part_table_ok(h, true) &&
        (h->lookup_path == 2 || (kmers >= (1u << 20) && (double)h->n_slots() * sizeof(Slot) <= 64.0 * (double)std::min<uint64_t>(kmers, h->slice_kmers)))) {
        for (uint64_t a = 0; a < kmers; a += h->slice_kmers) {
            const uint64_t b = std::min(kmers, a + h->slice_kmers);
            const uint64_t sub_off = a ? a - 1 : 0;
            const uint64_t sub_len = std::min(len, b + h->k) - sub_off;
            const uint8_t* sab; uint64_t slead;
            aligned_view(d_bases + sub_off, &sab, &slead);
            int rc = lookup_partitioned(h, sab, slead, sub_len, EmitRange{a - sub_off, b - sub_off}, cov_cutoff, map_lo, map_hi, (unsigned long long*)d_counters);
            if (rc) return rc;
        }
        return KQ_OK;
    }
    const dim3 grid(grid_for(h, n_tiles_of(lead, len), 1, 32));
    if (d_per_base) hipLaunchKernelGGL(k_lookup<true>, grid, dim3(TILE_THREADS), 0, h->stream, h->view(), ab, lead, len, h->k, (uint32_t)h->map_count,
                                       map_mask, (uint32_t)map_lo, (uint32_t)map_hi, cov_cutoff, d_per_base, (unsigned long long*)d_counters);
    else hipLaunchKernelGGL(k_lookup<false>, grid, dim3(TILE_THREADS), 0, h->stream, h->view(), ab, lead, len, h->k, (uint32_t)h->map_count,
                            map_mask, (uint32_t)map_lo, (uint32_t)map_hi, cov_cutoff, d_per_base, (unsigned long long*)d_counters);
    HIPC(hipGetLastError());
    return KQ_OK;
}
int kq_lookup_sequence(kq_handle* h, const char* bases, uint64_t len, uint32_t cov_cutoff, uint16_t map_lo, uint16_t map_hi,
                       kq_dbgbase* per_base, uint64_t counters[3]) {
    if (!h || !counters || (!bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    int rc = check_errors(h);
    if (rc) return rc;
    void* d = nullptr;
    rc = stage_in(h, bases, len, &d);
    if (rc) return rc;
    unsigned long long* d_ctr = nullptr;
    kq_dbgbase* d_pb = nullptr;
    HIPC(hipMalloc((void**)&d_ctr, 3 * sizeof(unsigned long long)));
    (void)hipMemsetAsync(d_ctr, 0, 3 * sizeof(unsigned long long), h->stream);
    if (per_base && len) {
        if (hipMalloc((void**)&d_pb, len * sizeof(kq_dbgbase)) != hipSuccess) { (void)hipFree(d_ctr); return fail(KQ_ERR_NOMEM, "per-base buffer allocation failed"); }
        (void)hipMemcpyAsync(d_pb, per_base, len * sizeof(kq_dbgbase), hipMemcpyHostToDevice, h->stream);
    }
    rc = kq_lookup_sequence_dev(h, (const char*)d, len, cov_cutoff, map_lo, map_hi, d_pb, (uint64_t*)d_ctr);
    unsigned long long c[3] = {0, 0, 0};
    if (!rc) {
        hipError_t e1 = hipMemcpyAsync(c, d_ctr, sizeof c, hipMemcpyDeviceToHost, h->stream);
        hipError_t e2 = d_pb ? hipMemcpyAsync(per_base, d_pb, len * sizeof(kq_dbgbase), hipMemcpyDeviceToHost, h->stream) : hipSuccess;
        hipError_t e3 = hipStreamSynchronize(h->stream);
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) rc = fail(KQ_ERR_HIP, "lookup failed: %s", hipGetErrorString(e3));
    }
    (void)hipFree(d_ctr);
    if (d_pb) (void)hipFree(d_pb);
    if (!rc) for (int i = 0; i < 3; ++i) counters[i] += c[i];
    return rc;
}

// ---- candidate-error search support ---------------------------------------------------------------
int kq_lookup_keys(kq_handle* h, const uint64_t* keys, uint64_t n, kq_entry* out) {
    if (!h || ((!keys || !out) && n)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    int rc = flush_pending(h);
    if (rc) return rc;
    materialize(h);
    // keys in, entries out: one scratch buffer (8 + 48 bytes per key)
    rc = ensure_buf(&h->stage, &h->stage_bytes, (size_t)n * (sizeof(uint64_t) + sizeof(kq_entry)) + 64);
    if (rc) return rc;
    uint64_t* d_keys = (uint64_t*)h->stage;
    kq_entry* d_out = (kq_entry*)(d_keys + ((n + 1) & ~1ull));
    HIPC(hipMemcpyAsync(d_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_lookup_keys, dim3(grid_for(h, n, 256)), dim3(256), 0, h->stream, h->view(), d_keys, n, d_out);
    HIPC(hipMemcpyAsync(out, d_out, n * sizeof(kq_entry), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    return KQ_OK;
}
int kq_branch_scan(kq_handle* h, const char* bases, uint64_t len, uint32_t cov_cutoff, uint8_t* flags) {
    if (!h || ((!bases || !flags) && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (len) memset(flags, 0, len);
    if (len < (uint64_t)h->k) return KQ_OK;
    int rc = flush_pending(h);
    if (rc) return rc;
    materialize(h);
    void* d = nullptr;
    rc = stage_in(h, bases, len, &d);
    if (rc) return rc;
    uint8_t* d_flags = nullptr;
    HIPC(hipMalloc((void**)&d_flags, len));
    (void)hipMemsetAsync(d_flags, 0, len, h->stream);
    const uint8_t* ab; uint64_t lead;
    aligned_view((const char*)d, &ab, &lead);
    hipLaunchKernelGGL(k_branch_scan, dim3(grid_for(h, n_tiles_of(lead, len), 1, 32)), dim3(TILE_THREADS), 0, h->stream, h->view(), ab, lead, len, h->k,
                       cov_cutoff, d_flags);
    hipError_t e1 = hipMemcpyAsync(flags, d_flags, len, hipMemcpyDeviceToHost, h->stream);
    hipError_t e2 = hipStreamSynchronize(h->stream);
    (void)hipFree(d_flags);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(KQ_ERR_HIP, "branch scan failed: %s", hipGetErrorString(e2 != hipSuccess ? e2 : e1));
    return KQ_OK;
}

// ---- union / import / export ---------------------------------------------------------------------
int kq_merge(kq_handle* dst, kq_handle* src) {
    if (!dst || !src) return fail(KQ_ERR_INVALID, "null handle");
    if (dst == src) return fail(KQ_ERR_INVALID, "cannot merge a handle into itself");
    if (dst->k != src->k || dst->map_count != src->map_count || dst->device != src->device)
        return fail(KQ_ERR_MISMATCH, "handles differ in k / map_count / device");    // src/input.cpp:136-139
    HIPC(hipSetDevice(dst->device));
    { int frc = flush_pending(dst); if (frc) return frc; }
    { int frc = flush_pending(src); if (frc) return frc; }
    materialize(src);                                   // on src's stream, before the sync below
    int rc = kq_sync(src);
    if (rc) return rc;
    rc = read_state(src);
    if (rc) return rc;
    rc = reserve(dst, src->st_host->slots_used, src->st_host->kmers_added);
    if (rc) return rc;
    // enough source entries to pay for streaming dst once: merge region by region in LDS; else per-entry atomics
    if (dst->merge_path == 2 || (dst->merge_path == 0 && src->st_host->slots_used * 64 >= dst->n_slots())) {
        const int empty = (dst->table_empty || dst->slots_dirty) ? 1 : 0;     // a dirty (lazily cleared) array is never read
        hipLaunchKernelGGL(k_merge_regions, dim3((unsigned)std::min<uint64_t>(dst->n_alloc_regions(), 1u << 30)), dim3(P3_THREADS), 0, dst->stream,
                           dst->view(), src->view(), empty);
        dst->slots_dirty = false;                       // every region image has been written
        dst->table_empty = false;
        HIPC(hipGetLastError());
        return kq_sync(dst);
    }
    materialize(dst);
    dst->table_empty = false;
    hipLaunchKernelGGL(k_merge, dim3(grid_for(dst, src->n_slots(), 256)), dim3(256), 0, dst->stream, dst->view(), src->view());
    HIPC(hipGetLastError());
    return kq_sync(dst);
}

int kq_import(kq_handle* h, const kq_entry* entries, uint64_t n) {
    if (!h || (!entries && n)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    uint64_t inst = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const kq_entry& e = entries[i];
        inst += e.cov;
        bool ok = e.cov > 0 && e.key != EMPTY_KEY;
        for (int w = 0; w < 4; ++w) ok = ok && e.fw[w] <= e.cov && e.bw[w] <= e.cov;   // an edge is seen at most once per instance
        if (!ok) return fail(KQ_ERR_INVALID, "entry %llu is not a valid k-mer record (cov 0, or an edge counter above cov)",
                             (unsigned long long)i);
    }
    int rc = reserve(h, n, inst);
    if (rc) return rc;
    void* d = nullptr;
    rc = stage_in(h, entries, n * sizeof(kq_entry), &d);
    if (rc) return rc;
    materialize(h);
    h->table_empty = false;
    hipLaunchKernelGGL(k_import, dim3(grid_for(h, n, 256)), dim3(256), 0, h->stream, h->view(), (const kq_entry*)d, n);
    HIPC(hipGetLastError());
    return kq_sync(h);
}

int kq_export(kq_handle* h, uint16_t map_lo, uint16_t map_hi, kq_entry* out, uint64_t cap, uint64_t* n_out) {
    if (!h || !n_out) return fail(KQ_ERR_INVALID, "null argument");
    if (map_lo > map_hi || map_hi > h->map_count) return fail(KQ_ERR_INVALID, "map range [%u,%u) outside [0,%d]", map_lo, map_hi, h->map_count);
    HIPC(hipSetDevice(h->device));
    int rc = check_errors(h);
    if (rc) return rc;
    unsigned long long* d_n = nullptr;
    kq_entry* d_out = nullptr;
    HIPC(hipMalloc((void**)&d_n, sizeof(unsigned long long)));
    (void)hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->stream);
    if (out && cap) {
        if (hipMalloc((void**)&d_out, cap * sizeof(kq_entry)) != hipSuccess) { (void)hipFree(d_n); return fail(KQ_ERR_NOMEM, "export buffer allocation failed"); }
    }
    materialize(h);
    hipLaunchKernelGGL(k_export, dim3(grid_for(h, h->n_slots(), 1024)), dim3(256), 0, h->stream, h->view(), (uint32_t)h->map_count,
                       (uint32_t)map_lo, (uint32_t)map_hi, d_out, d_out ? cap : 0, d_n);
    unsigned long long n = 0;
    hipError_t e = hipMemcpyAsync(&n, d_n, sizeof n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail(KQ_ERR_HIP, "export failed: %s", hipGetErrorString(e));
    *n_out = n;
    if (!rc && out) {
        if (n > cap) rc = fail(KQ_ERR_CAPACITY, "export buffer too small: need %llu, have %llu", n, (unsigned long long)cap);
        else if (n) {
            // key order on the device (radix sort of (key, index) + gather) when there is memory for it, else on the host
            const kq_entry* src = d_out;
            kq_entry* d_sorted = nullptr; uint64_t *d_k1 = nullptr, *d_k2 = nullptr; uint32_t *d_i1 = nullptr, *d_i2 = nullptr; void* d_tmp = nullptr;
            size_t tmp_bytes = 0;
            bool on_device = n < (1ull << 32) &&
                hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_k1, d_k2, d_i1, d_i2, (int64_t)n, 0, 2 * h->k, h->stream) == hipSuccess &&
                hipMalloc((void**)&d_sorted, n * sizeof(kq_entry)) == hipSuccess && hipMalloc((void**)&d_k1, n * 8) == hipSuccess &&
                hipMalloc((void**)&d_k2, n * 8) == hipSuccess && hipMalloc((void**)&d_i1, n * 4) == hipSuccess &&
                hipMalloc((void**)&d_i2, n * 4) == hipSuccess && hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 8) == hipSuccess;
            if (on_device) {
                hipLaunchKernelGGL(k_entry_keys, dim3(grid_for(h, n, 256)), dim3(256), 0, h->stream, d_out, n, d_k1, d_i1);
                on_device = hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_k1, d_k2, d_i1, d_i2, (int64_t)n, 0, 2 * h->k, h->stream) == hipSuccess;
                if (on_device) {
                    hipLaunchKernelGGL(k_entry_gather, dim3(grid_for(h, n, 256)), dim3(256), 0, h->stream, d_out, d_i2, n, d_sorted);
                    on_device = hipStreamSynchronize(h->stream) == hipSuccess;
                    src = d_sorted;
                }
            }
            if (!on_device) { (void)hipGetLastError(); src = d_out; }
            // device -> caller: through two pinned bounce buffers (a copy into pageable memory runs at a fraction of the PCIe
            // rate: 0.25 s for 850 MB); the DMA of chunk i+1 overlaps the host copy of chunk i
            const size_t bytes = (size_t)n * sizeof(kq_entry), chunk = (size_t)16 << 20;
            void* bounce[2] = {nullptr, nullptr};
            if (bytes > 4 * chunk && hipHostMalloc(&bounce[0], chunk, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&bounce[1], chunk, hipHostMallocDefault) == hipSuccess) {
                e = hipSuccess;
                const size_t n_chunks = (bytes + chunk - 1) / chunk;
                auto len_of = [&](size_t c) { return std::min(chunk, bytes - c * chunk); };
                e = hipMemcpyAsync(bounce[0], (const char*)src, len_of(0), hipMemcpyDeviceToHost, h->stream);
                for (size_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
                    e = hipStreamSynchronize(h->stream);
                    if (e == hipSuccess && c + 1 < n_chunks)
                        e = hipMemcpyAsync(bounce[(c + 1) & 1], (const char*)src + (c + 1) * chunk, len_of(c + 1), hipMemcpyDeviceToHost, h->stream);
                    memcpy((char*)out + c * chunk, bounce[c & 1], len_of(c));
                }
                if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            } else {
                (void)hipGetLastError();
                e = hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost);
            }
            for (void* b : bounce) if (b) (void)hipHostFree(b);
            if (e != hipSuccess) rc = fail(KQ_ERR_HIP, "export copy failed: %s", hipGetErrorString(e));
            else if (!on_device) parallel_sort_entries(out, n);
            for (void* q : {(void*)d_sorted, (void*)d_k1, (void*)d_k2, (void*)d_i1, (void*)d_i2, d_tmp}) if (q) (void)hipFree(q);
        }
    }
    (void)hipFree(d_n);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"

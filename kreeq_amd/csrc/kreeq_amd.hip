// kreeq_amd.hip -- HIP kernels + C ABI (include/kreeq_amd.h) of the MI355X-native kreeq hot path.
// gfx950 only; no CPU fallback anywhere in this file.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/kreeq_amd.h"
#include "kq_device.h"
#include "kq_partition.h"

using namespace kq;

// ================================================================================================
// kernels
// ================================================================================================

// K1+K2 fused: hashSequences (src/graph-builder.cpp:75-113) + processBuffers (:160-206) without
// materialising the 9-byte records: 1 B/base streamed in, random RMW on the table.
__global__ __launch_bounds__(TILE_THREADS) void k_count_direct(TableView t, const uint8_t* __restrict__ ab,
                                                                uint64_t lead, uint64_t len, int k, EmitRange er, PartCfg filt) {
    uint32_t n_new = 0;
    uint64_t n_kmers = 0;
    scan_tiles(ab, lead, len, k, [&](uint64_t pos, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) {
        if (pos < er.lo || pos >= er.hi) return;
        const bool is_fw = fw < rv;
        const uint64_t key = is_fw ? fw : rv;
        if (filt.filt_lo != 0 || filt.filt_hi != filt.map_count) {
            const uint32_t m = map_index(key, filt.map_count, filt.map_mask);
            if (m < filt.filt_lo || m >= filt.filt_hi) return;
        }
        uint32_t ins = 0;
        if (table_add(t, key, 1, edge_pack(is_fw, prev, next), nullptr, &ins)) ++n_kmers;
        n_new += ins;
    });
    uint64_t a = block_sum(n_new), b = block_sum(n_kmers);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&t.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&t.st->kmers_added, (unsigned long long)b);
    }
}

// K1 pass A: number of k-mers per tile (so that pass B can write in sequence order)
__global__ __launch_bounds__(TILE_THREADS) void k_emit_count(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len,
                                                              int k, unsigned long long* tile_counts) {
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_load(ab, lo_valid, hi_valid, tile, s_codes, s_inv);
        uint64_t total = block_sum(tile_lane_count(s_inv, k));
        if (threadIdx.x == 0) tile_counts[tile] = total;
    }
}

// exclusive scan of tile counts (single workgroup; n_tiles is len/4032, i.e. small)
__global__ __launch_bounds__(1024) void k_exclusive_scan(unsigned long long* a, uint64_t n, unsigned long long* total) {
    __shared__ unsigned long long s_part[1024];
    const int tid = threadIdx.x;
    const uint64_t per = (n + 1023) / 1024;
    const uint64_t lo = (uint64_t)tid * per, hi = lo + per < n ? lo + per : n;
    unsigned long long sum = 0;
    for (uint64_t i = lo; i < hi; ++i) sum += a[i];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned long long run = 0;
        for (int i = 0; i < 1024; ++i) { unsigned long long v = s_part[i]; s_part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    unsigned long long run = s_part[tid];
    for (uint64_t i = lo; i < hi; ++i) { unsigned long long v = a[i]; a[i] = run; run += v; }
}

// K1 pass B: write (key, edge byte) records in sequence order at tile_offsets[tile] + rank in tile
__global__ __launch_bounds__(TILE_THREADS) void k_emit_write(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k,
                                                              const unsigned long long* tile_offsets,
                                                              uint64_t* keys, uint8_t* edges, uint64_t cap) {
    __shared__ uint32_t s_wave[TILE_THREADS / 64];
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    const int tid = threadIdx.x;
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_load(ab, lo_valid, hi_valid, tile, s_codes, s_inv);
        const uint32_t mine = tile_lane_count(s_inv, k);
        // exclusive prefix of `mine` over the workgroup: wave scan + wave totals
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t n = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += n; }
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        uint32_t wave_base = 0;
        for (int w = 0; w < (tid >> 6); ++w) wave_base += s_wave[w];
        uint64_t out = tile_offsets[tile] + wave_base + (incl - mine);
        tile_lane_scan(s_codes, s_inv, lo_valid, tile, k, [&](uint64_t, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) {
            const bool is_fw = fw < rv;
            if (out < cap) {
                keys[out] = is_fw ? fw : rv;
                edges[out] = pack_to_edge_byte(edge_pack(is_fw, prev, next));
            }
            ++out;
        });
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// partitioned count path (kq_partition.h)
// ------------------------------------------------------------------------------------------------
// No per-record and no per-run global atomics: every workgroup gets PRIVATE output cursors from a
// counting pass (count matrix -> exclusive scan), because reservation atomics on a few hundred
// shared cursors serialise per address (~11 ns each) and were the limiter of the first version.

constexpr int P1_F = 4;                 // hist workgroups per scatter workgroup (hist is light on LDS)
constexpr uint32_t P2_UNIT = 8 * MS_TILE;   // records per P2 work unit (never crosses a coarse bucket)

// column of hist workgroup vb in the count matrix: the P1_F hist workgroups whose tiles one scatter
// workgroup b owns (vb = b, b+G1, ...) are adjacent, so b's output range per bin is contiguous
__device__ __forceinline__ uint32_t p1_col(uint32_t vb, uint32_t g1) { return (vb % g1) * P1_F + vb / g1; }

// P1 pass A: per-workgroup counts of coarse buckets -> M1[bin][column] (u64, bin-major)
// BINMODE 0: generic p1_bin (owner split / map-range filter); 1: coarse bucket of the table region; 2: top hash
// bits (narrow).  The specialised modes keep the 16 unrolled steps free of wave-uniform branches.
template <int BINMODE>
__device__ __forceinline__ uint32_t p1_bin_of(const PartCfg& cfg, uint64_t key, uint64_t h) {
    return BINMODE == 2 ? (uint32_t)(h >> (64 - NARROW_CBITS)) : BINMODE == 1 ? (uint32_t)(hash_region(h, cfg.n_regions) >> cfg.g_shift) : p1_bin(cfg, key, h);
}
// KC != 0: k is the compile-time constant KC (the default k = 21 gets its own instantiation: every
// k-dependent shift and mask of the 16 scan steps and of the hash folds to an immediate)
template <int BINMODE, int KC>
__global__ __launch_bounds__(TILE_THREADS) void k_p1_hist(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k_arg,
                                                          PartCfg cfg, EmitRange er, uint32_t g1, unsigned long long* __restrict__ m1) {
    const int k = KC ? KC : k_arg;
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    __shared__ uint32_t s_hist[NB_MAX];
    for (uint32_t b = threadIdx.x; b < cfg.n_coarse; b += TILE_THREADS) s_hist[b] = 0;
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_load(ab, lo_valid, hi_valid, tile, s_codes, s_inv);         // barrier inside (also covers the zeroing above)
        tile_lane_scan_all(s_codes, s_inv, lo_valid, tile, k, er, [&](int, bool valid, uint64_t fw, uint64_t rv, uint32_t, uint32_t) {
            if (valid) {
                const uint64_t key = fw < rv ? fw : rv;
                const uint32_t b = p1_bin_of<BINMODE>(cfg, key, table_hash(key, (uint32_t)k));
                if (BINMODE != 0 || b < cfg.n_coarse) atomicAdd(&s_hist[b], 1u);
            }
        });
        __syncthreads();
    }
    __syncthreads();
    const uint64_t cols = (uint64_t)g1 * P1_F;
    const uint32_t col = p1_col(blockIdx.x, g1);
    for (uint32_t b = threadIdx.x; b < cfg.n_coarse; b += TILE_THREADS) m1[(uint64_t)b * cols + col] = s_hist[b];
}
// after the scan of M1: coarse_off[b] = first output position of bucket b; [n_coarse] = #records
__global__ void k_p1_offsets(const unsigned long long* __restrict__ m1, const unsigned long long* __restrict__ total, PartCfg cfg,
                             uint32_t g1, unsigned long long* __restrict__ coarse_off) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < cfg.n_coarse) coarse_off[b] = m1[(uint64_t)b * g1 * P1_F];
    if (b == cfg.n_coarse) coarse_off[b] = *total;
}
// P1 pass B: (key, edge) records into their coarse bucket, private cursors from the scanned M1
template <int FMT, int NBC, int BINMODE, int KC>
__global__ __launch_bounds__(TILE_THREADS, (NBC <= 512 && FMT != FMT_WIDE) ? 3 : 2) void k_p1_scatter(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k_arg,
                                                             PartCfg cfg, EmitRange er, const unsigned long long* __restrict__ m1,
                                                             uint64_t* __restrict__ recs, uint8_t* __restrict__ recs_aux, int aux_fmt) {
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    constexpr bool WIDE = FMT == FMT_WIDE, NARROW = FMT == FMT_NARROW, TOP8 = FMT == FMT_TOP8;
    constexpr int MS_FMT = TOP8 ? FMT_PACK8 : FMT;                      // TOP8 records are single u64 words like packed ones
    const int k = KC ? KC : k_arg;
    __shared__ MsShared<NBC, MS_FMT> s;
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    const uint64_t cols = (uint64_t)gridDim.x * P1_F;
    for (uint32_t b = threadIdx.x; b < cfg.n_coarse; b += MS_THREADS) s.gbase[b] = (uint32_t)m1[(uint64_t)b * cols + (uint64_t)blockIdx.x * P1_F];
#ifdef KQ_MS_STAMPS
    if (threadIdx.x == 0) s.stamp_on = 0;
#endif
    uint4 nxt = tile_fetch(ab, lo_valid, hi_valid, blockIdx.x);
    landed(nxt.x); landed(nxt.y); landed(nxt.z); landed(nxt.w);         // see k_lv_scatter: keeps the loop header free of a store-draining wait
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_store(nxt, lo_valid, hi_valid, tile, s_codes, s_inv);        // barrier inside (covers the cursor init)
        if (tile + gridDim.x < n_tiles) nxt = tile_fetch(ab, lo_valid, hi_valid, tile + gridDim.x);   // in flight during the split
        uint64_t rec[MS_ITEMS];
        uint32_t aux[MS_ITEMS], bin[MS_ITEMS];
        tile_lane_scan_all(s_codes, s_inv, lo_valid, tile, k, er, [&](int i, bool valid, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) {
            const bool is_fw = fw < rv;
            const uint64_t key = is_fw ? fw : rv;
            const uint64_t h = table_hash(key, (uint32_t)k);             // the only hash of this k-mer on the whole path
            if (WIDE) {
                rec[i] = cfg.raw_out ? key : h;                             // raw keys only for kq_emit_partitioned_dev's caller
                const uint32_t e = edge_idx6(is_fw, prev, next);
                aux[i] = aux_fmt == AUX_IDX6 ? e : idx6_to_edge_byte(e);
            } else if (TOP8) {
                rec[i] = top8_rec(h, edge_idx6(is_fw, prev, next));
                aux[i] = 0;
            } else if (NARROW) {
                rec[i] = narrow_word(narrow_main(h), narrow_aux(h, edge_idx6(is_fw, prev, next)), valid ? p1_bin_of<BINMODE>(cfg, key, h) : cfg.n_coarse);
            } else {
                rec[i] = rec_pack_hash(h, is_fw, prev, next);
                aux[i] = 0;
            }
            if (!NARROW) bin[i] = valid ? p1_bin_of<BINMODE>(cfg, key, h) : cfg.n_coarse;
        });
        block_multisplit<MS_FMT>(s, rec, aux, bin, cfg.n_coarse, recs, recs_aux,
                               [&] { landed(nxt.x); landed(nxt.y); landed(nxt.z); landed(nxt.w); });   // ends with a barrier
    }
}

// ---- one generic level of the record split (LevelCfg) -----------------------------------------
// work units: segment b is cut into ceil(size_b / P2_UNIT) units; unit_base = exclusive prefix
__global__ __launch_bounds__(1024) void k_lv_units(const unsigned long long* __restrict__ seg_off, LevelCfg lv,
                                                   unsigned long long* __restrict__ unit_base) {
    __shared__ unsigned long long s_n[NB_MAX];
    for (uint32_t b = threadIdx.x; b < lv.n_seg; b += blockDim.x)
        s_n[b] = (seg_off[b + 1] - seg_off[b] + P2_UNIT - 1) / P2_UNIT;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (uint32_t b = 0; b < lv.n_seg; ++b) { unit_base[b] = run; run += s_n[b]; }
        unit_base[lv.n_seg] = run;
    }
}
__device__ __forceinline__ uint32_t seg_of_unit(const unsigned long long* unit_base, uint32_t n_seg, uint64_t u) {
    uint32_t lo = 0, hi = n_seg;                  // largest b with unit_base[b] <= u (skips empty segments)
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (unit_base[mid] <= u) lo = mid; else hi = mid; }
    return lo;
}
// pass A: per-unit bin counts -> M2[unit][bin] (u32)
template <int FMT>
__global__ __launch_bounds__(MS_THREADS) void k_lv_hist(const uint64_t* __restrict__ recs, LevelCfg lv,
                                                        const unsigned long long* __restrict__ seg_off,
                                                        const unsigned long long* __restrict__ unit_base, uint32_t* __restrict__ m2) {
    __shared__ uint32_t s_hist[NB_MAX];
    const uint64_t n_units = unit_base[lv.n_seg];
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint32_t b = seg_of_unit(unit_base, lv.n_seg, u);
        const uint64_t lo = seg_off[b] + (u - unit_base[b]) * P2_UNIT;
        const uint64_t hi = lo + P2_UNIT < seg_off[b + 1] ? lo + P2_UNIT : seg_off[b + 1];
        for (uint32_t i = threadIdx.x; i < lv.nb; i += MS_THREADS) s_hist[i] = 0;
        __syncthreads();
        // 8 records per lane in flight, loaded unconditionally (index clamped): a load inside a branch per
        // iteration costs a full memory latency per record (s_waitcnt vmcnt(0) right behind it)
        const uint64_t last = hi - 1;                                   // a unit is never empty
        if (FMT == FMT_NARROW) {
            // 16-byte loads of four u32 records at absolute quad indices (the array is 16-byte aligned and has
            // slack behind its last record); the quads at the unit's ends are masked per element
            const uint4* v4 = reinterpret_cast<const uint4*>(recs);
            const uint64_t q0 = lo >> 2, q1 = (hi + 3) >> 2;
            for (uint64_t qb = q0; qb < q1; qb += 4ull * MS_THREADS) {
                uint4 q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = v4[min(qb + (uint64_t)j * MS_THREADS + threadIdx.x, q1 - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint64_t qi = qb + (uint64_t)j * MS_THREADS + threadIdx.x;
                    const uint32_t e[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint64_t ri = 4 * qi + c;
                        if (qi < q1 && ri >= lo && ri < hi) atomicAdd(&s_hist[narrow_bin(lv, b, e[c])], 1u);
                    }
                }
            }
        } else
        for (uint64_t base = lo; base < hi; base += 8ull * MS_THREADS) {
            uint64_t r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t i = min(base + (uint64_t)j * MS_THREADS + threadIdx.x, last);
                r[j] = FMT == FMT_NARROW ? (uint64_t)reinterpret_cast<const uint32_t*>(recs)[i] : recs[i];      // narrow: the u32 array alone decides the bin
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (base + (uint64_t)j * MS_THREADS + threadIdx.x < hi) {
                    const uint32_t bin = FMT == FMT_NARROW ? narrow_bin(lv, b, (uint32_t)r[j])
                        : FMT == FMT_TOP8 ? narrow_bin(lv, b, (uint32_t)(r[j] >> 32))
                        : (FMT == FMT_PACK8 && lv.top8) ? (uint32_t)(rec_hash<false>(r[j]) >> (64 - NARROW_CBITS))
                        : level_bin(lv, b, hash_region(lv.in_raw ? table_hash(r[j], lv.k) : rec_hash<FMT == FMT_WIDE>(r[j]), lv.n_regions));
                    atomicAdd(&s_hist[bin], 1u);
                }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < lv.nb; i += MS_THREADS) m2[u * lv.nb + i] = s_hist[i];
        __syncthreads();
    }
}
// per output group (segment b, bin): exclusive prefix of its counts over the segment's units
// (in place), group total out.  One WAVE per group: a segment can have thousands of units (the
// flat -> coarse level has a single segment), so the prefix runs 64 units at a time.
__global__ __launch_bounds__(256) void k_lv_offsets(uint32_t* __restrict__ m2, LevelCfg lv, const unsigned long long* __restrict__ unit_base,
                                                    unsigned long long* __restrict__ group_count) {
    const uint64_t r = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (r >= (uint64_t)lv.n_seg * lv.nb) return;                     // wave-uniform
    const uint32_t b = (uint32_t)(r / lv.nb), bin = (uint32_t)(r % lv.nb);
    const uint64_t u0 = unit_base[b], u1 = unit_base[b + 1];
    unsigned long long run = 0;
    for (uint64_t base = u0; base < u1; base += 64) {
        const uint64_t u = base + lane;
        const uint32_t c = u < u1 ? m2[u * lv.nb + bin] : 0u;
        unsigned long long incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long n = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += n; }
        if (u < u1) m2[u * lv.nb + bin] = (uint32_t)(run + incl - c);   // a group holds < 2^32 records of one batch
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) group_count[r] = run;
}
// pass B: records -> grouped by (segment, bin); private cursors = group_base + unit prefix
#ifndef KQ_LV_THREADS
#define KQ_LV_THREADS 512
#define KQ_LV_ITEMS 8
#define KQ_LV_OCC 4
#endif
constexpr int LV_THREADS = KQ_LV_THREADS, LV_ITEMS = KQ_LV_ITEMS, LV_TILE = LV_THREADS * LV_ITEMS;       // records come from memory: more waves per LDS footprint
template <int FMT, int NBC>
__global__ __launch_bounds__(LV_THREADS, KQ_LV_OCC) void k_lv_scatter(const uint64_t* __restrict__ recs, const uint8_t* __restrict__ recs_aux, LevelCfg lv,
                                                           const unsigned long long* __restrict__ seg_off,
                                                           const unsigned long long* __restrict__ unit_base, const uint32_t* __restrict__ m2,
                                                           const unsigned long long* __restrict__ group_base, uint64_t* __restrict__ out,
                                                           uint8_t* __restrict__ out_aux) {
    constexpr bool WIDE = FMT == FMT_WIDE, NARROW = FMT == FMT_NARROW, CONVERT = FMT == FMT_PACK8_TO_NARROW, TOP8 = FMT == FMT_TOP8, HAS_AUX = WIDE || NARROW;
    constexpr int MS_FMT = CONVERT ? FMT_NARROW : TOP8 ? FMT_PACK8 : FMT;                  // format of the records this kernel writes
    const uint32_t* recs32 = reinterpret_cast<const uint32_t*>(recs);
    __shared__ MsShared<NBC, MS_FMT> s;
    const uint32_t nb = lv.nb;
    const uint64_t n_units = unit_base[lv.n_seg];
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint32_t b = seg_of_unit(unit_base, lv.n_seg, u);
        const uint64_t lo = seg_off[b] + (u - unit_base[b]) * P2_UNIT;
        const uint64_t hi = lo + P2_UNIT < seg_off[b + 1] ? lo + P2_UNIT : seg_off[b + 1];
        for (uint32_t i = threadIdx.x; i < nb; i += LV_THREADS) s.gbase[i] = (uint32_t)(group_base[(uint64_t)b * nb + i] + m2[u * nb + i]);
#ifdef KQ_MS_STAMPS
        if (threadIdx.x == 0) { s.stamp_on = 1; s.stamp_last = __builtin_amdgcn_s_memtime(); }
#endif
        __syncthreads();
        // software pipeline: the next round's records are loaded before this round is split
        // (loads are unconditional, index clamped to the unit: a branch around a load makes the compiler
        // drain ALL outstanding loads -- the prefetch included -- at the first use)
        uint64_t nxt[LV_ITEMS];
        uint32_t nxt_aux[LV_ITEMS];
        const uint64_t last = hi - 1;                                   // a unit is never empty
#pragma unroll
        for (int j = 0; j < LV_ITEMS; ++j) {
            const uint64_t i = min(lo + (uint64_t)j * LV_THREADS + threadIdx.x, last);
            nxt[j] = NARROW ? (uint64_t)recs32[i] : recs[i];
            nxt_aux[j] = HAS_AUX ? recs_aux[i] : 0;
        }
        // wait for the first round's records HERE: otherwise the loop header inherits "loads pending" from
        // this path and its s_waitcnt vmcnt(0) also drains the previous round's stores on the back edge
#pragma unroll
        for (int j = 0; j < LV_ITEMS; ++j) { landed(nxt[j]); if (HAS_AUX) landed(nxt_aux[j]); }
        for (uint64_t pos = lo; pos < hi; pos += LV_TILE) {
            uint64_t rec[LV_ITEMS];
            uint32_t aux[LV_ITEMS], bin[LV_ITEMS];
#pragma unroll
            for (int j = 0; j < LV_ITEMS; ++j) {
                const uint64_t i = pos + (uint64_t)j * LV_THREADS + threadIdx.x;
                rec[j] = (WIDE && lv.in_raw) ? table_hash(nxt[j], lv.k) : nxt[j];      // raw keys become hashes at the first level
                aux[j] = nxt_aux[j];
                if (NARROW) rec[j] = narrow_word((uint32_t)rec[j], aux[j], i >= hi ? nb : narrow_bin(lv, b, (uint32_t)rec[j]));
                else if (CONVERT) {
                    const uint64_t hh = rec_hash<false>(rec[j]);
                    rec[j] = narrow_word(narrow_main(hh), narrow_aux(hh, (uint32_t)(rec[j] >> REC_EDGE_SHIFT) & 63u), i >= hi ? nb : (uint32_t)(hh >> (64 - NARROW_CBITS)));
                }
                else if (TOP8) bin[j] = i >= hi ? nb : narrow_bin(lv, b, (uint32_t)(rec[j] >> 32));
                else bin[j] = i >= hi ? nb : level_bin(lv, b, hash_region(rec_hash<WIDE>(rec[j]), lv.n_regions));
            }
#pragma unroll
            for (int j = 0; j < LV_ITEMS; ++j) {
                const uint64_t i = min(pos + LV_TILE + (uint64_t)j * LV_THREADS + threadIdx.x, last);
                nxt[j] = NARROW ? (uint64_t)recs32[i] : recs[i];
                nxt_aux[j] = HAS_AUX ? recs_aux[i] : 0;
            }
            block_multisplit<MS_FMT, LV_THREADS, LV_ITEMS>(s, rec, aux, bin, nb, out, out_aux, [&] {
#pragma unroll
                for (int j = 0; j < LV_ITEMS; ++j) { landed(nxt[j]); if (HAS_AUX) landed(nxt_aux[j]); }
            });
        }
    }
}
__global__ void k_set2(unsigned long long* p, unsigned long long a, unsigned long long b) { p[0] = a; p[1] = b; }

// multi-block exclusive scan helpers (chunks of SCAN_CHUNK elements per workgroup)
constexpr uint32_t SCAN_CHUNK = 16384;
__global__ __launch_bounds__(1024) void k_scan_sums(const unsigned long long* __restrict__ a, uint64_t n, unsigned long long* __restrict__ sums) {
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_CHUNK, hi = lo + SCAN_CHUNK < n ? lo + SCAN_CHUNK : n;
    unsigned long long v = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 1024) v += a[i];
    uint64_t t = block_sum(v);
    if (threadIdx.x == 0) sums[blockIdx.x] = t;
}
__global__ __launch_bounds__(1024) void k_scan_apply(unsigned long long* __restrict__ a, uint64_t n, const unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long s_part[1024];
    const uint64_t lo = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * (SCAN_CHUNK / 1024);
    unsigned long long v[SCAN_CHUNK / 1024], sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_CHUNK / 1024; ++j) { v[j] = lo + j < n ? a[lo + j] : 0; sum += v[j]; }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x < 64) {                      // wave 0 scans the 1024 partials, 16 per lane
        unsigned long long loc[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { loc[j] = s_part[threadIdx.x * 16 + j]; tot += loc[j]; }
        unsigned long long incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned long long nn = __shfl_up(incl, o, 64); if ((int)threadIdx.x >= o) incl += nn; }
        unsigned long long run = incl - tot;
#pragma unroll
        for (int j = 0; j < 16; ++j) { s_part[threadIdx.x * 16 + j] = run; run += loc[j]; }
    }
    __syncthreads();
    unsigned long long run = sums[blockIdx.x] + s_part[threadIdx.x];
#pragma unroll
    for (uint32_t j = 0; j < SCAN_CHUNK / 1024; ++j) { if (lo + j < n) a[lo + j] = run; run += v[j]; }
}


// P3: one workgroup per table region.  The region's slots (REGION_SLOTS x 24 B) are staged in LDS, all
// records of the region are applied with LDS atomics (same two-tier rule as table_add), and the
// image is streamed back.  Global atomics only for the rare high-copy tier and the two totals.
#ifdef KQ_STAMPS   // diagnostic build only (never shipped): per-phase cycle sums of k_count_regions
__device__ unsigned long long g_stamps[8];
#define KQ_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
                         __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) atomicAdd(&g_stamps[i], t_ - stamp_last); stamp_last = t_; } while (0)
#endif
#ifdef KQ_MS_STAMPS
extern "C" int kq_debug_ms_stamps(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[16] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(kq::g_ms_stamps), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(kq::g_ms_stamps), 128);
}
#endif
#ifdef KQ_STAMPS
extern "C" int kq_debug_stamps(unsigned long long* out, int reset) {
    if (reset) { unsigned long long z[8] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 64);
}
#else
#define KQ_STAMP(i) do { } while (0)
#endif
#ifndef KQ_P3_THREADS
#define KQ_P3_THREADS 512
#define KQ_P3_OCC 6
#define KQ_P3_PF 2
#endif
constexpr int P3_THREADS = KQ_P3_THREADS;        // three 48 KiB images per CU (24 waves): with records double-buffered and groups handed out by ticket, one more region in flight per CU beats 2 x 1024 threads (whole count job 2.64 vs 2.67 ms)
// Two instantiations share the regions: HOT = false takes the ordinary ones (deep record prefetch, no
// folding state: fits the 80 VGPRs that let three workgroups share a CU) and appends the skewed ones
// to hot_list; HOT = true then walks that list with the folding loop.
template <int FMT, bool HOT>
__global__ __launch_bounds__(P3_THREADS, HOT ? 4 : KQ_P3_OCC) void k_count_regions(TableView t, const uint64_t* __restrict__ recs, const uint8_t* __restrict__ recs_aux,
                                                              int aux_fmt, const unsigned long long* __restrict__ region_base, int table_is_empty,
                                                              unsigned long long* __restrict__ hot_list /*[0] = count, then region ids*/,
                                                              uint32_t narrow_rps /*FMT_NARROW: regions per top-bit bucket*/) {
    constexpr bool WIDE = FMT == FMT_WIDE, NARROW = FMT == FMT_NARROW, TOP8 = FMT == FMT_TOP8, HAS_AUX = WIDE || NARROW;
    const uint32_t* recs32 = reinterpret_cast<const uint32_t*>(recs);
    __shared__ uint64_t s_img[REGION_SLOTS * 3];
    __shared__ unsigned long long s_new, s_kmers;
    __shared__ unsigned int s_grp;
    // high-copy tier of this region, aggregated in LDS: a repeat k-mer with millions of instances
    // would otherwise serialise millions of global atomics on one side-table entry
    constexpr int HC_LDS = 64;
    __shared__ uint64_t s_hckey[HC_LDS];
    __shared__ uint32_t s_hccnt[HC_LDS][8];
    const int tid = threadIdx.x;
    const uint64_t n_work = HOT ? hot_list[0] : t.n_regions;
#ifdef KQ_STAMPS
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    for (uint64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint64_t r = HOT ? hot_list[1 + w] : w;
        const uint64_t lo = region_base[r], hi = region_base[r + 1];
        const uint32_t narrow_bucket = (NARROW || TOP8) ? (uint32_t)r / narrow_rps : 0u;
        if (lo == hi) {                                                 // block-uniform
            if (!HOT && table_is_empty == 2) {    // lazy kq_clear: this launch initialises every region, also the ones without records
                ulonglong2* g2 = reinterpret_cast<ulonglong2*>(t.slots + (r << REGION_SHIFT));
                for (int i = tid; i < (int)(REGION_SLOTS * 24 / 16); i += P3_THREADS) {
                    const int w = 2 * i;
                    g2[i] = make_ulonglong2(w % 3 == 0 ? EMPTY_KEY : 0ull, (w + 1) % 3 == 0 ? EMPTY_KEY : 0ull);
                }
            }
            continue;
        }
        // folding costs a ballot + shuffle per iteration: only regions that receive far more records than
        // they have slots (skew, or very deep coverage) take that path, in the second launch
        if (!HOT && hi - lo > 32ull * REGION_SLOTS) {
            if (tid == 0) hot_list[1 + atomicAdd(&hot_list[0], 1ull)] = r;
            continue;
        }
        if (tid < HC_LDS) {
            s_hckey[tid] = EMPTY_KEY;
#pragma unroll
            for (int e = 0; e < 8; ++e) s_hccnt[tid][e] = 0;
        }
        uint4* gimg = reinterpret_cast<uint4*>(t.slots + (r << REGION_SHIFT));
        uint4* limg = reinterpret_cast<uint4*>(s_img);
        if (table_is_empty) {            // first batch after kq_create / kq_clear: the image is known, skip the 48 KiB read
            for (int i = tid; i < (int)(REGION_SLOTS * 3); i += P3_THREADS) s_img[i] = (i % 3 == 0) ? EMPTY_KEY : 0ull;
        } else {
            for (int i = tid; i < (int)(REGION_SLOTS * 24 / 16); i += P3_THREADS) limg[i] = gimg[i];
        }
        if (tid == 0) { s_new = 0; s_kmers = 0; s_grp = P3_THREADS / 64; }
        __syncthreads();
        KQ_STAMP(0);                                                    // region_base load + image init/load + barrier
        uint32_t n_new = 0, n_ok = 0;
        // Slot of `key` in the LDS image (word index), claiming an empty one if needed; REGION_SLOTS*3 = not found.
        // The image is read with workgroup-scope relaxed atomic loads on the __shared__ array itself: a
        // volatile access through a generic pointer compiles to flat_load + s_waitcnt vmcnt(0), which also
        // drains the record prefetches on every probe.
        // Four slots of the probe sequence are read per LDS round trip: a wave needs the MAXIMUM probe count of
        // its 64 lanes in dependent round trips (4-6 at load 0.5 when probing one slot at a time -- the walk
        // was bound by exactly that latency chain), now a quarter of it.  The snapshot is scanned in order; an
        // EMPTY slot is claimed with a CAS whose result decides (slots only ever go EMPTY -> key).
        auto find_slot = [&](uint64_t key, uint64_t h) -> uint32_t {
            const uint32_t off = hash_offset(h, t.k);
            for (uint32_t base = 0; base < REGION_SLOTS; base += 4) {
                uint32_t w[4];
                uint64_t c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    w[j] = 3u * ((off + base + j) & (REGION_SLOTS - 1));
                    c[j] = __hip_atomic_load(&s_img[w[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint64_t cur = c[j];
                    if (cur == EMPTY_KEY) {
                        cur = atomicCAS((unsigned long long*)&s_img[w[j]], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                        if (cur == EMPTY_KEY) { ++n_new; return w[j]; }
                    }
                    if (cur == key) return w[j];
                }
            }
            atomicOr(&t.st->err_table_full, 1u);
            return REGION_SLOTS * 3;
        };
        // edge counts that no longer fit the u8 lanes: the region's LDS high-copy aggregation, global beyond 64 k-mers
        auto add_wide = [&](uint64_t key, uint64_t h, const uint32_t (&e)[8]) {
            int hslot = -1;
            uint32_t hp = (uint32_t)(h >> 40) & (HC_LDS - 1);
            for (int probe = 0; probe < HC_LDS; ++probe, hp = (hp + 1) & (HC_LDS - 1)) {
                uint64_t cur = __hip_atomic_load(&s_hckey[hp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == EMPTY_KEY) cur = atomicCAS((unsigned long long*)&s_hckey[hp], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                if (cur == EMPTY_KEY || cur == key) { hslot = (int)hp; break; }
            }
            if (hslot >= 0) {
#pragma unroll
                for (int w = 0; w < 8; ++w) if (e[w]) atomicAdd(&s_hccnt[hslot][w], e[w]);
            } else {
                HcSlot* hs = hc_upsert(t, key);
                if (!hs) { atomicOr(&t.st->err_hc_full, 1u); return; }
#pragma unroll
                for (int w = 0; w < 8; ++w) if (e[w]) atomicAdd((unsigned long long*)&hs->cnt[w], (unsigned long long)e[w]);
            }
        };
        // one record: `pack` holds its (at most two) edge bits, one per byte lane
        auto apply1 = [&](uint64_t key, uint64_t h, uint64_t pack) {
            const uint32_t w = find_slot(key, h);
            if (w == REGION_SLOTS * 3) return;
            ++n_ok;
            const uint64_t old = atomicAdd((unsigned long long*)&s_img[w + 2], 1ull);
            if (!pack) return;
            if (old < LOW_TIER_MAX) { atomicAdd((unsigned long long*)&s_img[w + 1], (unsigned long long)pack); return; }
            uint32_t e1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e1[q] = (uint32_t)(pack >> (8 * q)) & 1u;
            add_wide(key, h, e1);
        };
        // `cnt` folded instances of `key` with edge counts e[0..7] (each <= cnt)
        auto apply = [&](uint64_t key, const uint32_t (&e)[8], uint32_t cnt) {
            const uint64_t h = table_hash(key, t.k);
            const uint32_t w = find_slot(key, h);
            if (w == REGION_SLOTS * 3) return;
            n_ok += cnt;
            const uint64_t old = atomicAdd((unsigned long long*)&s_img[w + 2], (unsigned long long)cnt);
            uint32_t any = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) any |= e[q];
            if (!any) return;
            if (old + cnt <= LOW_TIER_MAX) {                        // every e[q] <= cnt <= 254: fits the u8 lanes
                uint64_t pack = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) pack |= (uint64_t)e[q] << (8 * q);
                atomicAdd((unsigned long long*)&s_img[w + 1], (unsigned long long)pack);
                return;
            }
            add_wide(key, h, e);
        };
        // Hot k-mers (repeats, homopolymers) put most lanes of a wave on ONE slot, batch after batch.
        // Lanes that share the first active lane's key are folded into a per-wave accumulator kept in
        // registers (wave-uniform); it is flushed to LDS only when the hot key changes.
        auto run = [&](auto fold_tag) {
        constexpr bool FOLD = decltype(fold_tag)::value;
        bool have_acc = false;
        uint64_t acc_key = 0;
        uint32_t acc_cnt = 0, acc_e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // Records are double-buffered in registers: the loads of the NEXT group of PF records per lane are
        // issued (unconditionally, index clamped: no branch around them) before the current group is
        // walked, so the only wait sits at the top of an iteration on loads that had a whole group's walk
        // to land.  (A conditional load per record made the compiler wait vmcnt(0) right after issuing
        // the next prefetch: a full HBM latency per record, ~4k cycles.)
        constexpr int PF = KQ_P3_PF;
        uint64_t nxt_rec[PF];
        uint32_t nxt_aux[PF];
        const uint64_t last = hi - 1;                                  // hi > lo here
        // groups of PF*64 records are handed to waves from an LDS ticket: waves that hit long probe
        // chains or contended slots take fewer groups, so all waves reach the barrier together
        constexpr uint64_t GRP = 64ull * PF;
        const uint32_t lane = tid & 63;
        uint32_t g_cur = tid >> 6;
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const uint64_t j = min(lo + g_cur * GRP + (uint64_t)q * 64 + lane, last);
            nxt_rec[q] = NARROW ? (uint64_t)recs32[j] : recs[j];
            nxt_aux[q] = HAS_AUX ? recs_aux[j] : 0u;
        }
        while (lo + g_cur * GRP < hi) {                                 // wave-uniform
          uint64_t cur_rec[PF];
          uint32_t cur_aux[PF];
#pragma unroll
          for (int q = 0; q < PF; ++q) { cur_rec[q] = nxt_rec[q]; cur_aux[q] = nxt_aux[q]; }
          uint32_t g_nxt = 0;
          if (lane == 0) g_nxt = atomicAdd(&s_grp, 1u);
          g_nxt = __builtin_amdgcn_readfirstlane(g_nxt);
#pragma unroll
          for (int q = 0; q < PF; ++q) {
            const uint64_t j = min(lo + g_nxt * GRP + (uint64_t)q * 64 + lane, last);
            nxt_rec[q] = NARROW ? (uint64_t)recs32[j] : recs[j];
            nxt_aux[q] = HAS_AUX ? recs_aux[j] : 0u;
          }
          const uint64_t base = lo + g_cur * GRP;
          g_cur = g_nxt;
#pragma unroll
          for (int q = 0; q < PF; ++q) {
            const uint64_t i = base + (uint64_t)q * 64 + lane;
            bool active = i < hi;
            const uint64_t rec = cur_rec[q];
            const uint32_t aux = cur_aux[q];
            uint64_t key = 0, pack = 0;
            const uint64_t h = NARROW ? narrow_hash(narrow_bucket, (uint32_t)rec, aux) : TOP8 ? top8_hash(narrow_bucket, rec) : rec_hash<WIDE>(rec);
            if (active) {
                key = key_of_hash(h, t.k);                                   // the mix is a bijection: no key in the record
                pack = NARROW ? idx6_to_pack(aux >> 2) : TOP8 ? idx6_to_pack((uint32_t)rec & 63u)
                     : WIDE ? (aux_fmt == AUX_IDX6 ? idx6_to_pack(aux) : edge_byte_to_pack(aux)) : rec_edge_pack(rec);
            }
            const uint64_t act = FOLD ? __ballot(active) : 0ull;
            if (FOLD && act) {
                const int lead = __ffsll((unsigned long long)act) - 1;
                const uint64_t lead_key = __shfl(key, lead, 64);
                const bool in_grp = active && key == lead_key;
                const uint64_t grp = __ballot(in_grp);
                if (__popcll(grp) >= 8) {
                    uint64_t sum = in_grp ? pack : 0ull;            // byte lanes <= 64: no carries
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                    if (have_acc && acc_key != lead_key) {
                        if ((tid & 63) == 0) apply(acc_key, acc_e, acc_cnt);
                        have_acc = false;
                    }
                    if (!have_acc) {
                        have_acc = true; acc_key = lead_key; acc_cnt = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) acc_e[w] = 0;
                    }
                    acc_cnt += (uint32_t)__popcll(grp);
#pragma unroll
                    for (int w = 0; w < 8; ++w) acc_e[w] += (uint32_t)(sum >> (8 * w)) & 0xFFu;
                    if (in_grp) active = false;
                }
            }
            if (active) apply1(key, h, pack);
          }
        }
        if (have_acc && (tid & 63) == 0) apply(acc_key, acc_e, acc_cnt);
        };
        run(std::integral_constant<bool, HOT>{});
        KQ_STAMP(1);                                                    // record walk
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n_new += __shfl_down(n_new, o, 64); n_ok += __shfl_down(n_ok, o, 64); }
        if ((tid & 63) == 0) { if (n_new) atomicAdd(&s_new, (unsigned long long)n_new); if (n_ok) atomicAdd(&s_kmers, (unsigned long long)n_ok); }
        __syncthreads();
        if (tid < HC_LDS && s_hckey[tid] != EMPTY_KEY) {            // flush the region's high-copy sums: one entry per k-mer
            HcSlot* hs = hc_upsert(t, s_hckey[tid]);
            if (!hs) atomicOr(&t.st->err_hc_full, 1u);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (s_hccnt[tid][e]) atomicAdd((unsigned long long*)&hs->cnt[e], (unsigned long long)s_hccnt[tid][e]);
            }
        }
        KQ_STAMP(2);                                                    // barrier (slowest wave) + high-copy flush
        for (int i = tid; i < (int)(REGION_SLOTS * 24 / 16); i += P3_THREADS) gimg[i] = limg[i];
        KQ_STAMP(3);                                                    // image store issue
        if (tid == 0) {
            if (s_new) atomicAdd(&t.st->slots_used, s_new);
            if (s_kmers) atomicAdd(&t.st->kmers_added, s_kmers);
        }
        __syncthreads();
        KQ_STAMP(4);                                                    // final barrier
    }
}


// K3 on partitioned records (counters only): the assembly's k-mers go through the same P1 / level split as
// reads, then one workgroup per table region stages the region image in LDS (read-only) and evaluates its
// records there -- sequential HBM traffic instead of one random 64-byte sector per k-mer.  A record holds
// everything evaluateSegment needs (src/kreeq.cpp:145-216): the hash (-> key) and the indices of the
// fw / bw edge the assembly continues with (edge_idx6: the strand mapping of :178-210 is already applied).
template <int FMT>
__global__ __launch_bounds__(P3_THREADS, 6) void k_lookup_regions(TableView t, const uint64_t* __restrict__ recs, const uint8_t* __restrict__ recs_aux,
                                                                   const unsigned long long* __restrict__ region_base, uint32_t narrow_rps,
                                                                   uint32_t cov_cutoff, unsigned long long* __restrict__ counters) {
    constexpr bool WIDE = FMT == FMT_WIDE, NARROW = FMT == FMT_NARROW, TOP8 = FMT == FMT_TOP8, HAS_AUX = WIDE || NARROW;
    const uint32_t* recs32 = reinterpret_cast<const uint32_t*>(recs);
    __shared__ uint64_t s_img[REGION_SLOTS * 3];
    const int tid = threadIdx.x;
    uint32_t missing = 0, total = 0, edge_missing = 0;
    for (uint64_t r = blockIdx.x; r < t.n_regions; r += gridDim.x) {
        const uint64_t lo = region_base[r], hi = region_base[r + 1];
        if (lo == hi) continue;                                         // block-uniform
        const uint4* gimg = reinterpret_cast<const uint4*>(t.slots + (r << REGION_SHIFT));
        uint4* limg = reinterpret_cast<uint4*>(s_img);
        for (int i = tid; i < (int)(REGION_SLOTS * 24 / 16); i += P3_THREADS) limg[i] = gimg[i];
        __syncthreads();
        const uint32_t narrow_bucket = (NARROW || TOP8) ? (uint32_t)r / narrow_rps : 0u;
        const uint64_t last = hi - 1;
        for (uint64_t base = lo; base < hi; base += 2ull * P3_THREADS) {
            uint64_t rec[2];
            uint32_t aux[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint64_t j = min(base + (uint64_t)q * P3_THREADS + tid, last);
                rec[q] = NARROW ? (uint64_t)recs32[j] : recs[j];
                aux[q] = HAS_AUX ? recs_aux[j] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (base + (uint64_t)q * P3_THREADS + tid >= hi) continue;
                const uint64_t h = NARROW ? narrow_hash(narrow_bucket, (uint32_t)rec[q], aux[q]) : TOP8 ? top8_hash(narrow_bucket, rec[q]) : rec_hash<WIDE>(rec[q]);
                const uint64_t key = key_of_hash(h, t.k);
                const uint32_t idx6 = NARROW ? aux[q] >> 2 : TOP8 ? (uint32_t)rec[q] & 63u : WIDE ? aux[q] : (uint32_t)(rec[q] >> REC_EDGE_SHIFT) & 63u;
                const uint32_t off = hash_offset(h, t.k);
                uint32_t found = REGION_SLOTS * 3;
                for (uint32_t pb = 0; pb < REGION_SLOTS && found == REGION_SLOTS * 3; pb += 4) {     // :153, four slots per LDS round trip
                    uint32_t w[4];
                    uint64_t c[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { w[j] = 3u * ((off + pb + j) & (REGION_SLOTS - 1)); c[j] = s_img[w[j]]; }
                    bool stop = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!stop && c[j] == key) { found = w[j]; stop = true; }
                        if (!stop && c[j] == EMPTY_KEY) stop = true;
                    }
                    if (stop) break;
                }
                uint64_t cov = 0, e8 = 0;
                const HcSlot* hs = nullptr;
                if (found != REGION_SLOTS * 3) {
                    e8 = s_img[found + 1]; cov = s_img[found + 2];
                    if (cov > LOW_TIER_MAX) hs = hc_find(t, key);                                    // :156-166 (32-bit tier)
                }
                if (cov == 0 || cov < cov_cutoff) ++missing;                                         // :172-175
                else {
                    const uint32_t f = idx6 & 7u, b = (idx6 >> 3) & 7u;
                    auto absent = [&](uint32_t e) { return ((e8 >> (8 * e)) & 0xFF) == 0 && !(hs && hs->cnt[e]); };
                    if (f < 4 && b < 4 && absent(f) && absent(4 + b)) ++edge_missing;                 // :176-215
                }
                ++total;                                                                              // :216
            }
        }
        __syncthreads();                                                // the next region overwrites the image
    }
    const uint64_t a = block_sum(missing), b = block_sum(total), c = block_sum(edge_missing);
    if (threadIdx.x == 0) {                                                                           // :223-225
        if (a) atomicAdd(&counters[0], (unsigned long long)a);
        if (b) atomicAdd(&counters[1], (unsigned long long)b);
        if (c) atomicAdd(&counters[2], (unsigned long long)c);
    }
}

// K2 on explicit records: processBuffers :160-206
__global__ __launch_bounds__(256) void k_insert_records(TableView t, const uint64_t* __restrict__ keys,
                                                         const uint8_t* __restrict__ edges, uint64_t n) {
    uint32_t n_new = 0;
    uint64_t n_ok = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t ins = 0;
        if (table_add(t, keys[i], 1, edge_byte_to_pack(edges[i]), nullptr, &ins)) ++n_ok;
        n_new += ins;
    }
    uint64_t a = block_sum(n_new), b = block_sum(n_ok);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&t.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&t.st->kmers_added, (unsigned long long)b);
    }
}

// import / union: add logical entries (kunion + mergeSubMaps, src/graph-builder.cpp:297-432)
__device__ __forceinline__ void add_logical(const TableView& t, uint64_t key, const uint32_t* e, uint32_t cov,
                                            uint32_t& n_new, uint64_t& n_cov) {
    uint64_t pack = 0;
    bool fits = cov <= LOW_TIER_MAX;
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (e[i] > LOW_TIER_MAX) fits = false; pack |= (uint64_t)(e[i] & 0xFF) << (8 * i); }
    uint32_t ins = 0;
    // when the entry itself is beyond the low tier, table_add routes all of it to the wide counters
    if (table_add(t, key, cov, fits ? pack : 0, fits ? nullptr : e, &ins)) n_cov += cov;
    n_new += ins;
}
__global__ __launch_bounds__(256) void k_import(TableView t, const kq_entry* __restrict__ in, uint64_t n) {
    uint32_t n_new = 0;
    uint64_t n_cov = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t e[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) { e[w] = in[i].fw[w]; e[4 + w] = in[i].bw[w]; }
        add_logical(t, in[i].key, e, in[i].cov, n_new, n_cov);
    }
    uint64_t a = block_sum(n_new), b = block_sum(n_cov);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&t.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&t.st->kmers_added, (unsigned long long)b);
    }
}
// K4: dst += src (both on this device)
__global__ __launch_bounds__(256) void k_merge(TableView dst, TableView src) {
    uint32_t n_new = 0;
    uint64_t n_cov = 0;
    const uint64_t n = src.n_regions << REGION_SHIFT;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot* s = src.slots + i;
        if (s->key == EMPTY_KEY) continue;
        Logical L = slot_logical(src, s);
        add_logical(dst, s->key, L.e, L.cov, n_new, n_cov);
    }
    uint64_t a = block_sum(n_new), b = block_sum(n_cov);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&dst.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&dst.st->kmers_added, (unsigned long long)b);
    }
}
// K4 by regions: dst += src without a global atomic per entry.  Both tables place a key by the same hash, so
// the entries of dst region r can only come from the one or two (in general: a contiguous range of) src
// regions that cover the same hash interval.  One workgroup per dst region: its image is staged in LDS
// (or created there when dst is empty), the covering src regions are streamed, their entries that hash to r
// are added with the rule of add_logical / table_add, and the image is written back.
__global__ __launch_bounds__(P3_THREADS, 6) void k_merge_regions(TableView dst, TableView src, int dst_is_empty) {
    __shared__ uint64_t s_img[REGION_SLOTS * 3];
    const int tid = threadIdx.x;
    uint32_t n_new = 0;
    uint64_t n_cov = 0;
    for (uint64_t r = blockIdx.x; r < dst.n_regions; r += gridDim.x) {
        uint4* gimg = reinterpret_cast<uint4*>(dst.slots + (r << REGION_SHIFT));
        uint4* limg = reinterpret_cast<uint4*>(s_img);
        if (dst_is_empty) {
            for (int i = tid; i < (int)(REGION_SLOTS * 3); i += P3_THREADS) s_img[i] = (i % 3 == 0) ? EMPTY_KEY : 0ull;
        } else {
            for (int i = tid; i < (int)(REGION_SLOTS * 24 / 16); i += P3_THREADS) limg[i] = gimg[i];
        }
        __syncthreads();
        // hash interval of dst region r (top 32 bits): [ceil(r 2^32 / R), ceil((r+1) 2^32 / R) - 1]
        const uint64_t R = dst.n_regions;
        const uint32_t h_lo = (uint32_t)(((r << 32) + R - 1) / R), h_hi = (uint32_t)((((r + 1) << 32) + R - 1) / R - 1);
        const uint64_t s_lo = __umulhi(h_lo, (uint32_t)src.n_regions), s_hi = __umulhi(h_hi, (uint32_t)src.n_regions);
        for (uint64_t sr = s_lo; sr <= s_hi; ++sr) {
            const Slot* sslots = src.slots + (sr << REGION_SHIFT);
            // the whole source region in flight at once (three 8-byte loads per slot, unconditional): a load behind
            // the key test would cost two dependent memory round trips per slot
            constexpr int SPT = REGION_SLOTS / P3_THREADS;
            uint64_t sk[SPT], se[SPT], sc[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) { const Slot* sp = sslots + tid + j * P3_THREADS; sk[j] = sp->key; se[j] = sp->edges8; sc[j] = sp->cov; }
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const uint64_t key = sk[j];
                if (key == EMPTY_KEY) continue;
                const uint64_t h = table_hash(key, dst.k);
                if (hash_region(h, R) != r) continue;
                const Logical L = logical_of(src, key, se[j], sc[j]);
                uint64_t pack = 0;
                bool fits = L.cov <= LOW_TIER_MAX, any = false;
#pragma unroll
                for (int e = 0; e < 8; ++e) { if (L.e[e] > LOW_TIER_MAX) fits = false; any |= L.e[e] != 0; pack |= (uint64_t)(L.e[e] & 0xFF) << (8 * e); }
                // find-or-claim in the LDS image, four slots per round trip
                const uint32_t off = hash_offset(h, dst.k);
                uint32_t w = REGION_SLOTS * 3;
                for (uint32_t pb = 0; pb < REGION_SLOTS && w == REGION_SLOTS * 3; pb += 4) {
                    uint32_t ws[4];
                    uint64_t c[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ws[j] = 3u * ((off + pb + j) & (REGION_SLOTS - 1)); c[j] = __hip_atomic_load(&s_img[ws[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (w != REGION_SLOTS * 3) break;
                        uint64_t cur = c[j];
                        if (cur == EMPTY_KEY) {
                            cur = atomicCAS((unsigned long long*)&s_img[ws[j]], (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                            if (cur == EMPTY_KEY) { ++n_new; w = ws[j]; break; }
                        }
                        if (cur == key) w = ws[j];
                    }
                }
                if (w == REGION_SLOTS * 3) { atomicOr(&dst.st->err_table_full, 1u); continue; }
                n_cov += L.cov;
                const uint64_t old = atomicAdd((unsigned long long*)&s_img[w + 2], (unsigned long long)L.cov);
                if (fits && old + L.cov <= LOW_TIER_MAX) {
                    if (pack) atomicAdd((unsigned long long*)&s_img[w + 1], (unsigned long long)pack);
                } else if (any) {                                       // beyond the u8 lanes: the wide counters (rare)
                    HcSlot* hs = hc_upsert(dst, key);
                    if (!hs) { atomicOr(&dst.st->err_hc_full, 1u); continue; }
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (L.e[e]) atomicAdd((unsigned long long*)&hs->cnt[e], (unsigned long long)L.e[e]);
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < (int)(REGION_SLOTS * 24 / 16); i += P3_THREADS) gimg[i] = limg[i];
        __syncthreads();
    }
    const uint64_t a = block_sum(n_new), b = block_sum(n_cov);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&dst.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&dst.st->kmers_added, (unsigned long long)b);
    }
}
// rehash into a bigger table (growth): exact move of physical state
__global__ __launch_bounds__(256) void k_rehash(TableView dst, const Slot* __restrict__ old, uint64_t n_old) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_old; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot s = old[i];
        if (s.key == EMPTY_KEY) continue;
        uint32_t ins = 0;
        Slot* d = table_upsert(dst, s.key, &ins);
        if (!d) { atomicOr(&dst.st->err_table_full, 1u); continue; }
        d->edges8 = s.edges8;     // unique key per thread: plain stores
        d->cov = s.cov;
    }
}
__global__ __launch_bounds__(256) void k_rehash_hc(TableView dst, const HcSlot* __restrict__ old, uint64_t n_old) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_old; i += (uint64_t)gridDim.x * blockDim.x) {
        if (old[i].key == EMPTY_KEY) continue;
        HcSlot* d = hc_upsert(dst, old[i].key);
        if (!d) { atomicOr(&dst.st->err_hc_full, 1u); continue; }
#pragma unroll
        for (int e = 0; e < 8; ++e) d->cnt[e] = old[i].cnt[e];
    }
}
// table initialisation in one streaming pass: word i of the table is EMPTY_KEY when it is the key
// word of a slot (i % words_per_slot == 0) and 0 otherwise; 16 B per lane per store.
__global__ __launch_bounds__(256) void k_clear_slots(ulonglong2* p, uint32_t words_per_slot, uint64_t n_pairs) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pairs; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = 2 * i;
        ulonglong2 v;
        v.x = (w % words_per_slot == 0) ? EMPTY_KEY : 0ull;
        v.y = ((w + 1) % words_per_slot == 0) ? EMPTY_KEY : 0ull;
        p[i] = v;
    }
}

// K5: summary (src/graph-builder.cpp:240-282).  hist_small[c] for cov < HIST_SMALL; rarer larger
// coverages are appended to (big_cov, n_big) and folded on the host.
constexpr uint32_t HIST_SMALL = 4096;
struct SummaryOut {
    unsigned long long total, uniq, distinct, edges, n_big, big_cap;
};
__global__ __launch_bounds__(256) void k_summary(TableView t, SummaryOut* out, unsigned long long* hist_small,
                                                  uint32_t* big_cov) {
    __shared__ uint32_t s_hist[HIST_SMALL];
    for (uint32_t i = threadIdx.x; i < HIST_SMALL; i += blockDim.x) s_hist[i] = 0;
    __syncthreads();
    uint64_t total = 0, uniq = 0, distinct = 0, edges = 0;
    const uint64_t n = t.n_regions << REGION_SHIFT;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot* s = t.slots + i;
        if (s->key == EMPTY_KEY) continue;
        Logical L = slot_logical(t, s);
        if (L.cov == 0) continue;                 // cannot happen (a key is inserted with cov >= 1)
        uniq += (L.cov == 1);                     // :250
#pragma unroll
        for (int w = 0; w < 4; ++w)               // :254 / :263 -> fw>0 ? 1 : (bw>0 ? 1 : 0)
            edges += (L.e[w] > 0) ? 1 : ((L.e[4 + w] > 0) ? 1 : 0);
        ++distinct;
        total += L.cov;                           // :274-278 (tot += cov * count)
        if (L.cov < HIST_SMALL) atomicAdd(&s_hist[L.cov], 1u);
        else {
            unsigned long long o = atomicAdd(&out->n_big, 1ull);
            if (o < out->big_cap) big_cov[o] = L.cov;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < HIST_SMALL; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&hist_small[i], (unsigned long long)s_hist[i]);
    uint64_t a = block_sum(total), b = block_sum(uniq), c = block_sum(distinct), d = block_sum(edges);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&out->total, (unsigned long long)a);
        if (b) atomicAdd(&out->uniq, (unsigned long long)b);
        if (c) atomicAdd(&out->distinct, (unsigned long long)c);
        if (d) atomicAdd(&out->edges, (unsigned long long)d);
    }
}

// export: logical entries of maps [lo, hi)
__global__ __launch_bounds__(256) void k_export(TableView t, uint32_t map_count, uint32_t lo, uint32_t hi,
                                                 kq_entry* out, uint64_t cap, unsigned long long* n_out) {
    const uint64_t n = t.n_regions << REGION_SHIFT;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot* s = t.slots + i;
        const uint64_t key = s->key;
        if (key == EMPTY_KEY) continue;
        const uint32_t m = (uint32_t)(key % map_count);
        if (m < lo || m >= hi) continue;
        unsigned long long o = atomicAdd(n_out, 1ull);
        if (out && o < cap) {
            Logical L = slot_logical(t, s);
            kq_entry e;
            e.key = key;
#pragma unroll
            for (int w = 0; w < 4; ++w) { e.fw[w] = L.e[w]; e.bw[w] = L.e[4 + w]; }
            e.cov = L.cov;
            e.hc = L.cov > LOW_TIER_MAX;          // in maps32 iff total cov >= 255
            out[o] = e;
        }
    }
}

// K3: evaluateSegment (src/kreeq.cpp:143-219) over a whole sequence (segments = ACGT runs).
// One random 24-B probe per k-mer: measured at 27.7 G lookups/s this is the part's random 64-B
// sector rate (a variant with 16 probes in flight per lane was not faster), so the kernel keeps the
// simple one-k-mer-at-a-time form at full occupancy.  Per-base results are staged in LDS and
// written out coalesced, and never read: a position is evaluated in exactly one map-range pass and
// the caller zero-initialises the array (generateValidationVector, src/input.cpp:38-45), so only
// found k-mers need a store.
template <bool PER_BASE>
__global__ __launch_bounds__(TILE_THREADS) void k_lookup(TableView t, const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len,
                                                          int k, uint32_t map_count, uint32_t map_mask, uint32_t map_lo, uint32_t map_hi,
                                                          uint32_t cov_cutoff, kq_dbgbase* __restrict__ per_base,
                                                          unsigned long long* __restrict__ counters) {
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    __shared__ kq_dbgbase s_pb[PER_BASE ? TILE_STARTS : 1];
    const int tid = threadIdx.x;
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    uint32_t missing = 0, total = 0, edge_missing = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_load(ab, lo_valid, hi_valid, tile, s_codes, s_inv);
        if (PER_BASE) {
            kq_dbgbase z; z.fw = z.bw = z.cov = 0; z.isFw = 0; z.pad[0] = z.pad[1] = z.pad[2] = 0;
            for (int j = tid; j < TILE_STARTS; j += TILE_THREADS) s_pb[j] = z;
            __syncthreads();
        }
        lane_scan_core<false>(s_codes, s_inv, lo_valid, tile, k,
                              [&](int i, bool, uint64_t, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) {
            const bool is_fw = fw < rv;                                    // :145
            const uint64_t key = is_fw ? fw : rv;
            const uint32_t m = map_mask ? (uint32_t)key & map_mask : (uint32_t)(key % map_count);   // :146
            if (m < map_lo || m >= map_hi) return;                         // :150
            kq_dbgbase b;
            b.fw = b.bw = b.cov = 0; b.isFw = 0; b.pad[0] = b.pad[1] = b.pad[2] = 0;
            Logical L;
            const Slot* s = table_find(t, key);                            // :153
            if (s) {
                L = slot_logical(t, s);                                    // :156-166 (8-bit or 32-bit tier)
                b.cov = L.cov; b.isFw = is_fw;                             // :168-169
            }
            if (b.cov == 0) ++missing;                                     // :172
            else if (b.cov < cov_cutoff) ++missing;                        // :174
            else {
                bool no_left = false, no_right = false;
                if (is_fw) {                                               // :178-193
                    if (next < 4) { uint32_t v = L.e[next]; if (v) b.fw = v; else no_right = true; }
                    if (prev < 4) { uint32_t v = L.e[4 + prev]; if (v) b.bw = v; else no_left = true; }
                } else {                                                   // :194-210
                    if (prev < 4) { uint32_t v = L.e[3 - prev]; if (v) b.fw = v; else no_left = true; }
                    if (next < 4) { uint32_t v = L.e[4 + 3 - next]; if (v) b.bw = v; else no_right = true; }
                }
                if (no_left && no_right) ++edge_missing;                   // :211
            }
            ++total;                                                       // :216
            if (PER_BASE && s) { b.pad[0] = 1; s_pb[16 * tid + i] = b; }
        });
        if (PER_BASE) {
            __syncthreads();
            const int64_t p0 = (int64_t)(tile * TILE_STARTS) - lo_valid;
            for (int j = tid; j < TILE_STARTS; j += TILE_THREADS) {
                kq_dbgbase b = s_pb[j];
                if (b.pad[0]) { b.pad[0] = 0; per_base[p0 + j] = b; }
            }
        }
        __syncthreads();
    }
    uint64_t a = block_sum(missing), b = block_sum(total), c = block_sum(edge_missing);
    if (threadIdx.x == 0) {                                                // :223-225
        if (a) atomicAdd(&counters[0], (unsigned long long)a);
        if (b) atomicAdd(&counters[1], (unsigned long long)b);
        if (c) atomicAdd(&counters[2], (unsigned long long)c);
    }
}

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

struct kq_handle {
    int device = 0, k = 0, map_count = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    int n_cu = 256;
    Slot* slots = nullptr;
    uint64_t n_regions = 0;
    HcSlot* hc = nullptr;
    uint64_t hc_cap = 0;
    DevState* st = nullptr;          // device
    DevState* st_host = nullptr;     // pinned mirror
    // scratch (grown on demand)
    void* scratch = nullptr; size_t scratch_bytes = 0;
    void* stage = nullptr; size_t stage_bytes = 0;     // device staging for host-buffer entry points
    uint64_t kmers_bound = 0;        // upper bound of instances inserted (sizing the side table)
    uint64_t used_bound = 0;         // upper bound of occupied slots (skips the state read-back)
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

    TableView view() const { TableView v; v.slots = slots; v.n_regions = n_regions; v.hc = hc; v.hc_mask = hc_cap - 1; v.st = st; v.k = (uint32_t)k; return v; }
    uint64_t n_slots() const { return n_regions << REGION_SHIFT; }
};

static void marks_reset(kq_handle* h);

static int grid_for(const kq_handle* h, uint64_t work_items, int per_block) {
    uint64_t blocks = (work_items + per_block - 1) / per_block;
    uint64_t cap = (uint64_t)h->n_cu * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static int ensure_buf(void** p, size_t* have, size_t need) {
    if (*have >= need) return KQ_OK;
    if (*p) { HIPC(hipFree(*p)); *p = nullptr; *have = 0; }
    size_t sz = need + need / 4 + 4096;
    HIPC(hipMalloc(p, sz));
    *have = sz;
    return KQ_OK;
}

static void clear_words(kq_handle* h, void* p, uint32_t words_per_slot, uint64_t n_slots) {
    const uint64_t n_pairs = n_slots * words_per_slot / 2;      // both table sizes make this exact
    hipLaunchKernelGGL(k_clear_slots, dim3(grid_for(h, n_pairs, 256)), dim3(256), 0, h->stream, (ulonglong2*)p, words_per_slot, n_pairs);
}
// lazy kq_clear: give the slot array its empty image now (every table user except the partitioned count)
static void materialize(kq_handle* h) {
    if (!h->slots_dirty) return;
    clear_words(h, h->slots, 3, h->n_slots());
    h->slots_dirty = false;
}
static int alloc_main(kq_handle* h, uint64_t n_regions, Slot** out) {
    Slot* p = nullptr;
    size_t bytes = (size_t)(n_regions << REGION_SHIFT) * sizeof(Slot);
    HIPC(hipMalloc((void**)&p, bytes));
    clear_words(h, p, 3, n_regions << REGION_SHIFT);
    *out = p;
    return KQ_OK;
}
static int alloc_hc(kq_handle* h, uint64_t cap, HcSlot** out) {
    HcSlot* p = nullptr;
    size_t bytes = (size_t)cap * sizeof(HcSlot);
    HIPC(hipMalloc((void**)&p, bytes));
    clear_words(h, p, 9, cap);
    *out = p;
    return KQ_OK;
}

static int read_state(kq_handle* h) {
    HIPC(hipMemcpyAsync(h->st_host, h->st, sizeof(DevState), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    return KQ_OK;
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
    uint64_t want = h->n_slots();
    while (want < need_slots) want *= 2;
    const uint64_t new_regions = want >> REGION_SHIFT;
    Slot* fresh = nullptr;
    int rc = alloc_main(h, new_regions, &fresh);
    if (rc) return rc == KQ_ERR_NOMEM ? fail(KQ_ERR_TABLE_FULL, "cannot grow k-mer table to %llu slots: out of device memory",
                                             (unsigned long long)want) : rc;
    Slot* old = h->slots; const uint64_t n_old = h->n_slots();
    h->slots = fresh; h->n_regions = new_regions;
    if (h->slots_dirty) h->slots_dirty = false;     // lazily cleared table: nothing to move, and the fresh array is clean
    else hipLaunchKernelGGL(k_rehash, dim3(grid_for(h, n_old, 256)), dim3(256), 0, h->stream, h->view(), old, n_old);
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipFree(old));
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
    const uint64_t bound = h->kmers_bound / 255 + 1;
    uint64_t need_hc = 2 * std::min<uint64_t>(bound, 1ull << 23);
    if (bound > (1ull << 23)) {
        if (!state_read) { rc = read_state(h); if (rc) return rc; }
        need_hc = std::max<uint64_t>(need_hc, 4 * h->st_host->hc_used);
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
    h->stream = h->own_stream;
    int rc = KQ_OK;
    do {
        if (hipMalloc((void**)&h->st, sizeof(DevState)) != hipSuccess || hipHostMalloc((void**)&h->st_host, sizeof(DevState)) != hipSuccess) {
            rc = fail(KQ_ERR_NOMEM, "state allocation failed"); break;
        }
        (void)hipMemsetAsync(h->st, 0, sizeof(DevState), h->stream);
        uint64_t slots = (uint64_t)((double)(capacity_hint ? capacity_hint : (1u << 20)) / 0.7);   // load <= 0.7 at the hinted size
        uint64_t regions = (slots + REGION_SLOTS - 1) >> REGION_SHIFT;
        if (regions < 16) regions = 16;
        if (regions >= (uint64_t)NB_MAX) regions = (regions + 255) / 256 * 256;      // FMT_NARROW: 256 hash-prefix buckets of whole regions
        if (regions >= (1ull << 19)) regions = (regions + 2047) / 2048 * 2048;         // ... of 8 sub-buckets of whole regions each
        rc = alloc_main(h, regions, &h->slots); if (rc) break;
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
    if (h->hc) (void)hipFree(h->hc);
    if (h->st) (void)hipFree(h->st);
    if (h->st_host) (void)hipHostFree(h->st_host);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->stage) (void)hipFree(h->stage);
    if (h->part) (void)hipFree(h->part);
    marks_reset(h);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int kq_clear(kq_handle* h) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    HIPC(hipSetDevice(h->device));
    h->slots_dirty = true;           // the 24 B/slot clear is folded into the next partitioned count (k_count_regions writes every region); anything else materialises it first
    clear_words(h, h->hc, 9, h->hc_cap);
    HIPC(hipMemsetAsync(h->st, 0, sizeof(DevState), h->stream));
    h->kmers_bound = 0;
    h->used_bound = 0;
    h->table_empty = true;
    return KQ_OK;
}

int kq_set_stream(kq_handle* h, void* s) {
    if (!h) return fail(KQ_ERR_INVALID, "null handle");
    HIPC(hipStreamSynchronize(h->stream));
    h->stream = s ? (hipStream_t)s : h->own_stream;
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
    cfg->narrow = 0; cfg->sub_bits = 0;
    if (allow_narrow && (h->k <= (int)NARROW_MAX_K || h->k > PART_MAX_K) && cfg->n_regions >= (uint64_t)NB_MAX && cfg->n_regions % (1u << NARROW_CBITS) == 0) {
        const uint64_t rps = cfg->n_regions >> NARROW_CBITS;
        // one level bucket -> regions while a bucket has < mid_rps regions (the multisplit writes runs of 4096 / fan-out
        // records), else a middle level of up to 8 sub-buckets: covers every table that fits the HBM (rps < 16384)
        uint32_t sb = 0;
        if (rps >= h->mid_rps) { sb = 3; while (sb > 0 && rps % (1u << sb)) --sb; }
        if ((rps >> sb) < (uint64_t)NB_MAX && (sb > 0 || rps < (uint64_t)NB_MAX)) { cfg->narrow = 1; cfg->sub_bits = sb; }
    }
    if (cfg->narrow) { cfg->n_coarse = 1u << NARROW_CBITS; if (cfg->g_shift == 0) cfg->g_shift = 1; }
}
// carve the scratch buffer for a batch of at most n_max records; n_tiles = 0 when the input is records
static int plan_alloc(kq_handle* h, PartPlan* p, uint64_t n_max, uint64_t n_tiles, uint32_t p1_bins, bool allow_narrow = false) {
    if (n_max >= (1ull << 32) - 16) return fail(KQ_ERR_INVALID, "a partition pass handles fewer than 2^32 records (got %llu): slice the input", (unsigned long long)n_max);
    plan_cfg(h, &p->cfg, allow_narrow);
    n_max = (n_max + 7) & ~7ull;                                // every record array starts 16-byte aligned and has slack for vector loads
    p->two_level = p->cfg.g_shift != 0;
    p->fmt = !p->cfg.narrow ? FMT_PACK8 : h->k > PART_MAX_K ? FMT_TOP8 : FMT_NARROW;      // the caller switches to FMT_WIDE where it applies
    p->n_max = n_max; p->R = p->cfg.n_regions;
    p->g1 = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(n_tiles, (uint64_t)h->n_cu * (p1_bins < 512 ? 3 : 2)));
    p->m1_n = (uint64_t)p1_bins * p->g1 * P1_F;
    const uint64_t nb_max = std::max<uint64_t>(std::max<uint64_t>(1ull << p->cfg.g_shift, p->cfg.n_coarse), p->cfg.narrow ? (p->cfg.n_regions >> NARROW_CBITS) >> p->cfg.sub_bits : 0);
    p->m2_n = (n_max / P2_UNIT + NB_MAX + 2) * nb_max;         // u32 entries, enough for either level
    p->groups_n = std::max<uint64_t>(p->R, (uint64_t)p->cfg.n_coarse << p->cfg.g_shift) + 2;
    p->sums_n = std::max(p->m1_n, p->groups_n) / SCAN_CHUNK + 2;
    const uint64_t aux_words = (n_max + 7) / 8 + 1;
    const size_t words = (size_t)(2 * n_max + 2 * aux_words + p->m1_n + 2 * (NB_MAX + 2) + p->groups_n + p->sums_n + 4 + (p->R + 2) + (p->m2_n + 1) / 2);
    int rc = ensure_buf(&h->part, &h->part_bytes, words * 8);
    if (rc) return rc;
    p->recs1 = (uint64_t*)h->part;
    p->recs2 = p->recs1 + n_max;
    p->aux1 = (uint8_t*)(p->recs2 + n_max);
    p->aux2 = p->aux1 + aux_words * 8;
    p->m1 = (unsigned long long*)(p->aux2 + aux_words * 8);
    p->seg_off = p->m1 + p->m1_n;
    p->unit_base = p->seg_off + NB_MAX + 2;
    p->group_base = p->unit_base + NB_MAX + 2;
    p->sums = p->group_base + p->groups_n;
    p->total = p->sums + p->sums_n;
    p->hot = p->total + 4;
    p->m2 = (uint32_t*)(p->hot + p->R + 2);
    return KQ_OK;
}
// exclusive scan of n u64 on the device (in place); *total (device) receives the sum
static void scan_u64(kq_handle* h, unsigned long long* a, uint64_t n, unsigned long long* sums_scratch, unsigned long long* total) {
    if (n <= SCAN_CHUNK) {
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
                   uint8_t* out_aux, int aux_fmt) {
    // plain table split (no owner split, no map-range filter): branch-free bin functions
    const bool plain = cfg.mode == 0 && cfg.filt_lo == 0 && cfg.filt_hi == cfg.map_count;
    const int binmode = !plain ? 0 : cfg.narrow ? 2 : 1;
#define KQ_P1H(B, K) hipLaunchKernelGGL((k_p1_hist<B, K>), dim3(p->g1 * P1_F), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, cfg, er, p->g1, p->m1)
    if (binmode == 2) { if (h->k == 21) KQ_P1H(2, 21); else KQ_P1H(2, 0); }
    else if (binmode == 1) { if (h->k == 31) KQ_P1H(1, 31); else KQ_P1H(1, 0); }
    else KQ_P1H(0, 0);
#undef KQ_P1H
    scan_u64(h, p->m1, (uint64_t)cfg.n_coarse * p->g1 * P1_F, p->sums, p->total);
    hipLaunchKernelGGL(k_p1_offsets, dim3((cfg.n_coarse + 256) / 256), dim3(256), 0, h->stream, p->m1, p->total, cfg, p->g1, p->seg_off);
    mark(h, "k_p1_hist+scan");
    const bool small = cfg.n_coarse < 512;                     // 48 KiB LDS variant: three workgroups per CU
#define KQ_P1S(W, N, B, K) hipLaunchKernelGGL((k_p1_scatter<W, N, B, K>), dim3(p->g1), dim3(TILE_THREADS), 0, h->stream, ab, lead, len, h->k, cfg, er, p->m1, out, out_aux, aux_fmt)
    if (cfg.narrow && h->k > PART_MAX_K) { if (plain && h->k == 31) KQ_P1S(FMT_TOP8, 512, 2, 31); else if (plain) KQ_P1S(FMT_TOP8, 512, 2, 0); else KQ_P1S(FMT_TOP8, 512, 0, 0); }
    else if (cfg.narrow)   { if (plain && h->k == 21) KQ_P1S(FMT_NARROW, 512, 2, 21); else if (plain) KQ_P1S(FMT_NARROW, 512, 2, 0); else KQ_P1S(FMT_NARROW, 512, 0, 0); }     // 256 buckets
    else if (out_aux && plain && h->k == 31) { if (small) KQ_P1S(FMT_WIDE, 512, 1, 31); else KQ_P1S(FMT_WIDE, NB_MAX, 1, 31); }   // the HiFi k
    else if (out_aux) { if (small) KQ_P1S(FMT_WIDE, 512, 0, 0); else KQ_P1S(FMT_WIDE, NB_MAX, 0, 0); }
    else if (plain)   { if (small) KQ_P1S(FMT_PACK8, 512, 1, 0); else KQ_P1S(FMT_PACK8, NB_MAX, 1, 0); }
    else              { if (small) KQ_P1S(FMT_PACK8, 512, 0, 0); else KQ_P1S(FMT_PACK8, NB_MAX, 0, 0); }
#undef KQ_P1S
    mark(h, "k_p1_scatter");
}
// one generic split level: in (grouped by p->seg_off[0..n_seg]) -> out grouped by (segment, bin);
// afterwards p->group_base[0..n_seg*nb] are the output offsets
static void run_level(kq_handle* h, PartPlan* p, const LevelCfg& lv, const uint64_t* in, const uint8_t* in_aux, uint64_t* out, uint8_t* out_aux) {
    const int fmt = lv.narrow == 2 ? FMT_TOP8 : lv.narrow ? FMT_NARROW : in_aux != nullptr ? FMT_WIDE : FMT_PACK8;       // input format; lv.top8: packed in, narrow out
    const uint64_t groups = (uint64_t)lv.n_seg * lv.nb;
    hipLaunchKernelGGL(k_lv_units, dim3(1), dim3(1024), 0, h->stream, p->seg_off, lv, p->unit_base);
    // one workgroup per work unit (upper bound of the unit count; surplus workgroups exit at once):
    // the hardware dispatcher balances them, a fixed grid looping over units left a 30 % tail
    const unsigned unit_grid = (unsigned)std::min<uint64_t>(p->n_max / P2_UNIT + lv.n_seg + 1, 1u << 30);
    if (fmt == FMT_TOP8) hipLaunchKernelGGL(k_lv_hist<FMT_TOP8>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, lv, p->seg_off, p->unit_base, p->m2);
    else if (fmt == FMT_NARROW) hipLaunchKernelGGL(k_lv_hist<FMT_NARROW>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, lv, p->seg_off, p->unit_base, p->m2);
    else if (fmt == FMT_WIDE) hipLaunchKernelGGL(k_lv_hist<FMT_WIDE>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, lv, p->seg_off, p->unit_base, p->m2);
    else hipLaunchKernelGGL(k_lv_hist<FMT_PACK8>, dim3(unit_grid), dim3(MS_THREADS), 0, h->stream, in, lv, p->seg_off, p->unit_base, p->m2);
    hipLaunchKernelGGL(k_lv_offsets, dim3((unsigned)((groups * 64 + 255) / 256)), dim3(256), 0, h->stream, p->m2, lv, p->unit_base, p->group_base);
    (void)hipMemsetAsync(p->group_base + groups, 0, 8, h->stream);   // failure surfaces at the caller's hipGetLastError
    scan_u64(h, p->group_base, groups + 1, p->sums, p->total + 1);
    mark(h, "k_lv_hist+offsets+scan");
    const bool small = lv.nb < 512;
#define KQ_LVS(W, N) hipLaunchKernelGGL((k_lv_scatter<W, N>), dim3(unit_grid), dim3(LV_THREADS), 0, h->stream, in, in_aux, lv, \
                                        p->seg_off, p->unit_base, p->m2, p->group_base, out, out_aux)
    if (lv.top8)              { KQ_LVS(FMT_PACK8_TO_NARROW, 512); }
    else if (fmt == FMT_TOP8) { if (small) KQ_LVS(FMT_TOP8, 512); else KQ_LVS(FMT_TOP8, NB_MAX); }
    else if (fmt == FMT_NARROW) { if (small) KQ_LVS(FMT_NARROW, 512); else KQ_LVS(FMT_NARROW, NB_MAX); }
    else if (fmt == FMT_WIDE) { if (small) KQ_LVS(FMT_WIDE, 512); else KQ_LVS(FMT_WIDE, NB_MAX); }
    else                      { if (small) KQ_LVS(FMT_PACK8, 512); else KQ_LVS(FMT_PACK8, NB_MAX); }
#undef KQ_LVS
    mark(h, "k_lv_scatter");
}
static LevelCfg level_coarse_to_regions(const PartCfg& cfg) {
    LevelCfg lv; lv.n_regions = cfg.n_regions; lv.n_seg = cfg.n_coarse; lv.nb = 1u << cfg.g_shift; lv.seg_shift = cfg.g_shift; lv.out_shift = 0; lv.in_raw = 0; lv.k = 0; lv.narrow = 0; lv.top8 = 0;
    lv.nr_shift = lv.nr_rps = lv.nr_sub = lv.nr_inv = 0; lv.nr_div = 1;
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
    return lv;
}
// bucket -> regions for FMT_NARROW records, in one level or (large tables) two; afterwards `*sorted` holds the
// records grouped by region and p->group_base their offsets
static void run_narrow_levels(kq_handle* h, PartPlan* p, const uint64_t** sorted, const uint8_t** sorted_aux);
static LevelCfg level_flat_to_coarse(const PartCfg& cfg) {
    LevelCfg lv; lv.n_regions = cfg.n_regions; lv.n_seg = 1; lv.nb = cfg.n_coarse; lv.seg_shift = 32; lv.out_shift = cfg.g_shift; lv.in_raw = 0; lv.k = 0; lv.narrow = 0; lv.top8 = 0;
    lv.nr_shift = lv.nr_rps = lv.nr_sub = lv.nr_inv = 0; lv.nr_div = 1;
    return lv;
}
static void run_p3(kq_handle* h, PartPlan* p, const uint64_t* sorted, const uint8_t* sorted_aux, int aux_fmt, const unsigned long long* base) {
    // hot list lives in the (now free) count matrix area: [0] = count, then up to R region ids
    unsigned long long* hot = p->hot;
    (void)hipMemsetAsync(hot, 0, 8, h->stream);
    const dim3 grid((unsigned)std::min<uint64_t>(p->R, 1u << 30)), grid_hot(h->n_cu), block(P3_THREADS);   // one workgroup per region: dispatcher-balanced
    const int empty = h->table_empty ? (h->slots_dirty ? 2 : 1) : 0;       // 2: also write the image of regions without records
    const uint32_t rps = (p->fmt == FMT_NARROW || p->fmt == FMT_TOP8) ? (uint32_t)(p->R >> NARROW_CBITS) : 1u;
#define KQ_P3(F) do { \
        hipLaunchKernelGGL((k_count_regions<F, false>), grid, block, 0, h->stream, h->view(), sorted, sorted_aux, aux_fmt, base, empty, hot, rps); \
        hipLaunchKernelGGL((k_count_regions<F, true>), grid_hot, block, 0, h->stream, h->view(), sorted, sorted_aux, aux_fmt, base, empty, hot, rps); } while (0)
    if (p->fmt == FMT_NARROW) KQ_P3(FMT_NARROW);
    else if (p->fmt == FMT_TOP8) KQ_P3(FMT_TOP8);
    else if (sorted_aux) KQ_P3(FMT_WIDE);
    else KQ_P3(FMT_PACK8);
#undef KQ_P3
    h->slots_dirty = false;          // every region has been written
}

// can the record split reach every region of this table?  (5-byte records: up to 256 x 8 x 2047 regions, i.e.
// any table that fits the HBM; 8-byte / wide records: two fan-outs below NB_MAX)
static bool part_table_ok(const kq_handle* h, bool narrow_possible) {
    if (h->n_regions <= (1ull << 20)) return true;
    if (!narrow_possible) return false;
    PartCfg c; plan_cfg(h, &c, true);
    return c.narrow != 0;
}
static void run_narrow_levels(kq_handle* h, PartPlan* p, const uint64_t** sorted, const uint8_t** sorted_aux) {
    const uint32_t sb = p->cfg.sub_bits;
    const bool t8 = p->fmt == FMT_TOP8;
    uint8_t* a1 = t8 ? nullptr : p->aux1;
    uint8_t* a2 = t8 ? nullptr : p->aux2;
    if (sb == 0) {
        run_level(h, p, level_narrow(p->cfg, 0, false, t8), p->recs1, a1, p->recs2, a2);
        *sorted = p->recs2; *sorted_aux = a2;
        return;
    }
    run_level(h, p, level_narrow(p->cfg, sb, true, t8), p->recs1, a1, p->recs2, a2);
    (void)hipMemcpyAsync(p->seg_off, p->group_base, (size_t)(((1u << NARROW_CBITS) << sb) + 1) * 8, hipMemcpyDeviceToDevice, h->stream);
    run_level(h, p, level_narrow(p->cfg, sb, false, t8), p->recs2, a2, p->recs1, a1);
    *sorted = p->recs1; *sorted_aux = a1;
}

// partitioned count of one batch of bases: P1 (coarse split) -> P2 (region split) -> P3 (LDS regions)
static int count_partitioned(kq_handle* h, const uint8_t* ab, uint64_t lead, uint64_t len, EmitRange er) {
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
    marks_reset(h);
    mark(h, "start");
    run_p1(h, &p, p.cfg, ab, lead, len, er, p.recs1, a1, AUX_IDX6);
    if (p.fmt == FMT_NARROW || p.fmt == FMT_TOP8) {
        const uint64_t* sorted; const uint8_t* sorted_aux;
        run_narrow_levels(h, &p, &sorted, &sorted_aux);
        run_p3(h, &p, sorted, sorted_aux, AUX_IDX6, p.group_base);
    } else if (p.two_level) {
        run_level(h, &p, level_coarse_to_regions(p.cfg), p.recs1, a1, p.recs2, a2);
        run_p3(h, &p, p.recs2, a2, AUX_IDX6, p.group_base);
    } else {
        run_p3(h, &p, p.recs1, a1, AUX_IDX6, p.seg_off);       // bins were the regions themselves
    }
    mark(h, "k_count_regions");
    HIPC(hipGetLastError());
    return KQ_OK;
}
// partitioned count of n records already on the device (multi-GPU receive side, kq_insert_records_dev).
// d_aux == nullptr: packed 8-byte records; else WIDE records with d_aux in `aux_fmt`.
// `raw`: d_recs holds raw keys (kq_insert_records); otherwise the records of kq_emit_packed_dev (table hashes)
static int count_partitioned_records(kq_handle* h, const uint64_t* d_recs, const uint8_t* d_aux, int aux_fmt, uint64_t n, bool raw) {
    PartPlan p;
    int rc = plan_alloc(h, &p, n, 0, 1, /*allow_narrow=*/!d_aux && !raw);
    if (rc) return rc;
    if (p.fmt == FMT_NARROW) {
        // packed records of kq_emit_packed_dev on a narrow-eligible table: the first level splits on the top 8 hash
        // bits and writes 5-byte records, the rest is the narrow path of count_partitioned
        hipLaunchKernelGGL(k_set2, dim3(1), dim3(1), 0, h->stream, p.seg_off, 0ull, (unsigned long long)n);
        LevelCfg first = level_flat_to_coarse(p.cfg);
        first.top8 = 1; first.nb = 1u << NARROW_CBITS;
        run_level(h, &p, first, d_recs, nullptr, p.recs1, p.aux1);
        HIPC(hipMemcpyAsync(p.seg_off, p.group_base, (size_t)((1u << NARROW_CBITS) + 1) * 8, hipMemcpyDeviceToDevice, h->stream));
        const uint64_t* sorted; const uint8_t* sorted_aux;
        run_narrow_levels(h, &p, &sorted, &sorted_aux);
        run_p3(h, &p, sorted, sorted_aux, AUX_IDX6, p.group_base);
        HIPC(hipGetLastError());
        return KQ_OK;
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
        run_p3(h, &p, p.recs2, a2, aux_fmt, p.group_base);
    } else {
        run_p3(h, &p, p.recs1, a1, aux_fmt, p.group_base);
    }
    HIPC(hipGetLastError());
    return KQ_OK;
}

int kq_count_batch_dev(kq_handle* h, const char* d_bases, uint64_t len) {
    if (!h || (!d_bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (len < (uint64_t)h->k) return KQ_OK;                                  // src/graph-builder.cpp:60
    const uint64_t kmers = len - h->k + 1;
    // a resident batch of any size is processed in slices of <= 2^28 k-mer starts (the partition
    // scratch is 16 B per start); a slice scans one extra base on the left and k on the right, so
    // k-mers and edges across a cut are seen exactly once
    // The partitioned path streams the whole table once per slice, so a slice should bring a few records per
    // slot: with a large table (and memory to spare for 16 B of scratch per start) slices grow up to 2^31 starts
    uint64_t slice = h->slice_kmers;
    if (!h->slice_user && kmers > slice && 2 * h->n_slots() > slice) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const uint64_t by_mem = (uint64_t)((free_b + h->part_bytes) / 24);          // 16 B scratch + margin per start
            slice = std::max(slice, std::min<uint64_t>(std::min<uint64_t>(2 * h->n_slots(), 1ull << 31), by_mem));
        }
    }
    for (uint64_t a = 0; a < kmers; a += slice) {
        const uint64_t b = std::min(kmers, a + slice);
        int rc = reserve(h, b - a, b - a);
        if (rc) return rc;
        const uint64_t sub_off = a ? a - 1 : 0;
        const uint64_t sub_len = std::min(len, b + h->k) - sub_off;
        const EmitRange er{a - sub_off, b - sub_off};
        const uint8_t* ab; uint64_t lead;
        aligned_view(d_bases + sub_off, &ab, &lead);
        // partitioned path streams the whole table once per slice (2 x 24 B per slot) on top of ~37 B per
        // record; the atomic path costs ~95 ps per record whatever the table size (~480 B at the
        // part's streaming rate): partition unless the table is more than ~200 B per record of the slice
        bool part = (b - a) >= (1u << 20) && part_table_ok(h, true) &&
                    (double)h->n_slots() * sizeof(Slot) <= 200.0 * (double)(b - a);
        if (h->count_path == 1) part = false;
        if (h->count_path == 2) {
            if (!part_table_ok(h, true)) return fail(KQ_ERR_INVALID, "table too large for the partitioned path");
            part = true;
        }
        if (part) {
            rc = count_partitioned(h, ab, lead, sub_len, er);
            h->table_empty = false;
            if (rc) return rc;
            continue;
        }
        h->table_empty = false;
        PartCfg filt; plan_cfg(h, &filt);
        filt.filt_lo = h->filt_lo; filt.filt_hi = h->filt_hi;
        materialize(h);
        hipLaunchKernelGGL(k_count_direct, dim3(grid_for(h, n_tiles_of(lead, sub_len), 1)), dim3(TILE_THREADS), 0, h->stream,
                           h->view(), ab, lead, sub_len, h->k, er, filt);
        HIPC(hipGetLastError());
    }
    return KQ_OK;
}
int kq_count_batch(kq_handle* h, const char* bases, uint64_t len) {
    if (!h || (!bases && len)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    void* d = nullptr;
    int rc = stage_in(h, bases, len, &d);
    if (rc) return rc;
    rc = kq_count_batch_dev(h, (const char*)d, len);
    if (rc) return rc;
    return kq_sync(h);
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
    if (cap < len - h->k + 1 || !d_keys || !d_edges) return fail(KQ_ERR_CAPACITY, "record buffer too small: need room for %llu records",
                                                                  (unsigned long long)(len - h->k + 1));
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    PartPlan p;
    int rc = plan_alloc(h, &p, 0, n_tiles_of(lead, len), (uint32_t)n_parts);
    if (rc) return rc;
    PartCfg cfg = p.cfg;
    cfg.mode = 1; cfg.n_coarse = (uint32_t)n_parts;
    cfg.raw_out = 1;
    run_p1(h, &p, cfg, ab, lead, len, EmitRange{0, ~0ull}, d_keys, d_edges, AUX_EDGE_BYTE);      // WIDE records: key + reference edge byte
    std::vector<unsigned long long> off((size_t)n_parts + 1);
    HIPC(hipMemcpyAsync(off.data(), p.seg_off, off.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = off[(size_t)i + 1] - off[(size_t)i];
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
    if (cap < len - h->k + 1 || !d_recs) return fail(KQ_ERR_CAPACITY, "record buffer too small: need room for %llu records",
                                                       (unsigned long long)(len - h->k + 1));
    const uint8_t* ab; uint64_t lead;
    aligned_view(d_bases, &ab, &lead);
    PartPlan p;
    int rc = plan_alloc(h, &p, 0, n_tiles_of(lead, len), (uint32_t)n_parts);
    if (rc) return rc;
    PartCfg cfg = p.cfg;
    cfg.mode = 1; cfg.n_coarse = (uint32_t)n_parts;
    run_p1(h, &p, cfg, ab, lead, len, EmitRange{0, ~0ull}, d_recs, nullptr, AUX_IDX6);
    std::vector<unsigned long long> off((size_t)n_parts + 1);
    HIPC(hipMemcpyAsync(off.data(), p.seg_off, off.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n_parts; ++i) part_counts[i] = off[(size_t)i + 1] - off[(size_t)i];
    return KQ_OK;
}

int kq_insert_packed_dev(kq_handle* h, const uint64_t* d_recs, uint64_t n) {
    if (!h || (!d_recs && n)) return fail(KQ_ERR_INVALID, "null argument");
    if (h->k > PART_MAX_K) return fail(KQ_ERR_INVALID, "packed 8-byte records need k <= %d (use kq_insert_records_dev)", PART_MAX_K);
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    int rc = reserve(h, n, n);
    if (rc) return rc;
    if (!part_table_ok(h, true)) return fail(KQ_ERR_INVALID, "table too large for the partitioned path");
    rc = count_partitioned_records(h, d_recs, nullptr, AUX_IDX6, n, false);
    h->table_empty = false;
    return rc;
}

int kq_insert_records_dev(kq_handle* h, const uint64_t* d_keys, const uint8_t* d_edges, uint64_t n) {
    if (!h || ((!d_keys || !d_edges) && n)) return fail(KQ_ERR_INVALID, "null argument");
    HIPC(hipSetDevice(h->device));
    if (!n) return KQ_OK;
    int rc = reserve(h, n, n);
    if (rc) return rc;
    if ((h->count_path == 2 || (h->count_path == 0 && n >= (1u << 20))) && h->n_regions <= (1ull << 20)) {   // WIDE records: key + reference edge byte
        rc = count_partitioned_records(h, d_keys, d_edges, AUX_EDGE_BYTE, n, true);
        h->table_empty = false;
        return rc;
    }
    h->table_empty = false;
    materialize(h);
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
    const dim3 grid(grid_for(h, n_tiles_of(lead, len), 1));
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

// ---- union / import / export ---------------------------------------------------------------------
int kq_merge(kq_handle* dst, kq_handle* src) {
    if (!dst || !src) return fail(KQ_ERR_INVALID, "null handle");
    if (dst == src) return fail(KQ_ERR_INVALID, "cannot merge a handle into itself");
    if (dst->k != src->k || dst->map_count != src->map_count || dst->device != src->device)
        return fail(KQ_ERR_MISMATCH, "handles differ in k / map_count / device");    // src/input.cpp:136-139
    HIPC(hipSetDevice(dst->device));
    materialize(src);                                   // on src's stream, before the sync below
    int rc = kq_sync(src);
    if (rc) return rc;
    rc = read_state(src);
    if (rc) return rc;
    rc = reserve(dst, src->st_host->slots_used, src->st_host->kmers_added);
    if (rc) return rc;
    // enough source entries to pay for streaming dst once: merge region by region in LDS; else per-entry atomics
    if (dst->merge_path == 2 || (dst->merge_path == 0 && src->st_host->slots_used * 64 >= dst->n_slots())) {
        const int empty = dst->table_empty ? 1 : 0;
        hipLaunchKernelGGL(k_merge_regions, dim3((unsigned)std::min<uint64_t>(dst->n_regions, 1u << 30)), dim3(P3_THREADS), 0, dst->stream,
                           dst->view(), src->view(), empty);
        dst->slots_dirty = false;                       // every region image has been written
        dst->table_empty = false;
        HIPC(hipGetLastError());
        return kq_sync(dst);
    }
    dst->table_empty = false;
    materialize(dst);
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
    h->table_empty = false;
    void* d = nullptr;
    rc = stage_in(h, entries, n * sizeof(kq_entry), &d);
    if (rc) return rc;
    materialize(h);
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
            e = hipMemcpy(out, d_out, n * sizeof(kq_entry), hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = fail(KQ_ERR_HIP, "export copy failed: %s", hipGetErrorString(e));
            else parallel_sort_entries(out, n);
        }
    }
    (void)hipFree(d_n);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"

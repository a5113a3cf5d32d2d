// Device kernels of libkreeq_amd (gfx950): the partitioned count path (P1 tile scan + multisplit, split levels,
// k_count_regions), the region-wise lookup and union, and the table kernels (direct count, insert, import,
// rehash, clear, summary, export, direct lookup).  Included by kreeq_amd.hip only; see DESIGN.md §4.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

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
                                                                uint64_t lead, uint64_t len, int k, EmitRange er, PartCfg filt,
                                                                const uint16_t* __restrict__ pinv /*packed input (tile_fetch) or null*/) {
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
        if (table_add(t, table_hash(key, (uint32_t)k), 1, edge_pack(is_fw, prev, next), nullptr, &ins)) ++n_kmers;
        n_new += ins;
    }, pinv);
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
    if (tid < 64) {                              // wave 0 scans the 1024 partials, 16 per lane (a serial loop on one lane cost 13 us)
        unsigned long long loc[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { loc[j] = s_part[tid * 16 + j]; tot += loc[j]; }
        unsigned long long incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned long long nn = __shfl_up(incl, o, 64); if (tid >= o) incl += nn; }
        unsigned long long run = incl - tot;
#pragma unroll
        for (int j = 0; j < 16; ++j) { s_part[tid * 16 + j] = run; run += loc[j]; }
        if (tid == 63) *total = incl;
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

constexpr int P1_F = 1;                 // hist workgroups per scatter workgroup (the scatter grid is already 6 x the resident workgroups)
constexpr uint32_t P2_UNIT = 4 * MS_TILE;   // records per work unit of a split level (never crosses a segment); 4 rounds: k_lv_scatter 357 us vs 384 us at 8, 401 us at 16, 430 us at 1

// column of hist workgroup vb in the count matrix: the P1_F hist workgroups whose tiles one scatter
// workgroup b owns (vb = b, b+G1, ...) are adjacent, so b's output range per bin is contiguous
__device__ __forceinline__ uint32_t p1_col(uint32_t vb, uint32_t g1) { return (vb % g1) * P1_F + vb / g1; }

// P1 pass A: per-workgroup counts of coarse buckets -> M1[bin][column] (u64, bin-major)
// BINMODE 0: generic p1_bin (owner split / map-range filter); 1: coarse bucket of the table region; 2: top hash
// bits (narrow).  The specialised modes keep the 16 unrolled steps free of wave-uniform branches.
template <int BINMODE>
__device__ __forceinline__ uint32_t p1_bin_of(const PartCfg& cfg, uint64_t key, uint64_t h) {
    // 4: hash-prefix bucket behind the map-range filter of a memory-bounded pass (KQ_OPT_COUNT_MAP_RANGE; map_count a power
    // of two): one subtract + compare decides, rejected k-mers go to the discard bin
    // 5 (histogram only): the hash-prefix bucket of EVERY k-mer, filed under the map range it belongs to (n_rng equal ranges of a
    // power-of-two map count): one scan of a resident batch yields the count matrices of all its map-range passes
    if (BINMODE == 5) return (((((uint32_t)key & cfg.map_mask) * cfg.n_rng) >> __popc(cfg.map_mask)) << NARROW_CBITS) | (uint32_t)(h >> (64 - NARROW_CBITS));
    // 6: hash-prefix bucket, k-mers outside the table's bucket window dropped (a shard / a bucket-range pass holds only its own buckets)
    if (BINMODE == 6) { const uint32_t b = (uint32_t)(h >> (64 - NARROW_CBITS)); return b - cfg.win_lo < cfg.win_hi - cfg.win_lo ? b : cfg.n_coarse; }
    if (BINMODE == 4) return (((uint32_t)key & cfg.map_mask) - cfg.filt_lo < cfg.filt_hi - cfg.filt_lo) ? (uint32_t)(h >> (64 - NARROW_CBITS)) : cfg.n_coarse;
    return BINMODE == 3 ? ((((((uint32_t)key & cfg.map_mask) * (cfg.n_coarse >> cfg.owner_sub)) >> __popc(cfg.map_mask)) << cfg.owner_sub) |
                          (threadIdx.x & ((1u << cfg.owner_sub) - 1u)))         // owner rank x lane sub-bin; map_count a power of two, no filter
         : BINMODE == 2 ? (uint32_t)(h >> (64 - NARROW_CBITS)) : BINMODE == 1 ? (uint32_t)(hash_region(h, cfg.n_regions) >> cfg.g_shift) : p1_bin(cfg, key, h);
}
// KC != 0: k is the compile-time constant KC (the default k = 21 gets its own instantiation: every
// k-dependent shift and mask of the 16 scan steps and of the hash folds to an immediate)
template <int BINMODE, int KC>
__global__ __launch_bounds__(TILE_THREADS) void k_p1_hist(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k_arg,
                                                          PartCfg cfg, EmitRange er, uint32_t g1, unsigned long long* __restrict__ m1,
                                                          const uint16_t* __restrict__ pinv /*packed input or null*/) {
    const int k = KC ? KC : k_arg;
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    __shared__ uint32_t s_hist[NB_MAX];
    for (uint32_t b = threadIdx.x; b < cfg.n_coarse; b += TILE_THREADS) s_hist[b] = 0;
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_load(ab, lo_valid, hi_valid, tile, s_codes, s_inv, pinv);   // barrier inside (also covers the zeroing above)
        tile_lane_scan_all(s_codes, s_inv, lo_valid, tile, k, er, [&](int, bool valid, uint64_t fw, uint64_t rv, uint32_t, uint32_t) {
            if (valid) {
                const uint64_t key = fw < rv ? fw : rv;
                const uint32_t b = p1_bin_of<BINMODE>(cfg, key, table_hash(key, (uint32_t)k));
                if ((BINMODE != 0 && BINMODE != 4 && BINMODE != 6) || b < cfg.n_coarse) atomicAdd(&s_hist[b], 1u);
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
                                                             uint64_t* __restrict__ recs, uint8_t* __restrict__ recs_aux, int aux_fmt,
                                                             const uint16_t* __restrict__ pinv /*packed input or null*/) {
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
    uint4 nxt = tile_fetch(ab, lo_valid, hi_valid, blockIdx.x, pinv);
    landed(nxt.x); landed(nxt.y); landed(nxt.z); landed(nxt.w);         // see k_lv_scatter: keeps the loop header free of a store-draining wait
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_store(nxt, lo_valid, hi_valid, tile, s_codes, s_inv, pinv != nullptr);        // barrier inside (covers the cursor init)
        if (tile + gridDim.x < n_tiles) nxt = tile_fetch(ab, lo_valid, hi_valid, tile + gridDim.x, pinv);   // in flight during the split
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

// P1 pass B for narrow records into the 256 hash-prefix buckets, "streamed" formulation: a lane does not hold its 16 records
// (and their ranks) in registers through the scan -- every scan step counts its record's bin (LDS add without return) and
// parks the record word in the lane's own column of the stage; after the bin scan the lane takes its 16 words back, and
// behind a barrier they go to their sorted places IN the same stage (cursor atomics), from where the copy-out runs as in
// block_multisplit.  The scan's live registers are then the scanner's alone, the stage is the only large LDS array (37 KiB
// per workgroup: four workgroups per CU instead of three), thread b owns bin b (its absolute cursor stays in a register:
// five barriers per round instead of seven).  Same count matrix, same cursors, same output as k_p1_scatter<FMT_NARROW>.
#ifndef KQ_P1S_TPR
#define KQ_P1S_TPR 2          // tiles per round of an unfiltered scatter (measured: 1 -> 2 takes 22 % off it; a filtered one is faster at 1)
#endif
// TPR = tiles per round: TPR x 256 threads, every 256 of them scan one tile (the workgroup's tiles b, b + G, b + 2G, ... taken TPR at a
// time), ONE split over the TPR x 4032 starts: runs of a bin TPR times as long at the same number of waves per CU.
// TOP8 (k = 29..32): the record is a whole u64 (top8_rec), so its bin travels in a second LDS array (u16) and the stage takes 40 KiB:
// three workgroups per CU, one tile per round.  Replaces k_p1_scatter<FMT_TOP8> (32 spilled registers at k = 31).
template <int BINMODE, int KC, int TPR, bool TOP8 = false>
__global__ __launch_bounds__(TILE_THREADS * TPR, TOP8 ? 3 : 4 / TPR) void k_p1_scatter_s(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k_arg,
                                                             PartCfg cfg, EmitRange er, const unsigned long long* __restrict__ m1,
                                                             uint32_t* __restrict__ recs /*TOP8: u64 records*/, uint8_t* __restrict__ recs_aux,
                                                             const uint16_t* __restrict__ pinv /*packed input or null*/) {
    constexpr uint32_t NB = 1u << NARROW_CBITS;                         // bin NB = "no record"
    constexpr int THREADS = TILE_THREADS * TPR;
    static_assert(NB == TILE_THREADS && MS_TILE == TILE_THREADS * 16 && (TPR == 1 || TPR == 2 || TPR == 4) && (!TOP8 || TPR == 1), "thread b owns bin b");
    __shared__ uint32_t s_codes[TPR][TILE_THREADS];
    __shared__ uint32_t s_inv[TPR][TILE_THREADS];
    __shared__ uint64_t s_buf[TPR * MS_TILE];
    __shared__ uint16_t s_bin[TOP8 ? MS_TILE : 1];
    __shared__ uint32_t s_hist[NB], s_loff[NB + 1], s_grel[NB], s_wave[TILE_THREADS / 64];
    const int k = KC ? KC : k_arg;
    const int tid = threadIdx.x, half = tid / TILE_THREADS, lane = tid % TILE_THREADS;
    const bool bin_owner = TPR == 1 || tid < (int)NB;
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len), stride = (uint64_t)gridDim.x * TPR, mine = (uint64_t)half * gridDim.x;
    uint32_t gabs = bin_owner ? (uint32_t)m1[(uint64_t)tid * gridDim.x * P1_F + (uint64_t)blockIdx.x * P1_F] : 0u;      // bin tid's output cursor
    if (bin_owner) s_hist[tid] = 0;
    uint4 nxt = tile_fetch(ab, lo_valid, hi_valid, blockIdx.x + mine, pinv, lane);      // (a tile behind the last one reads nothing and has no valid start)
    landed(nxt.x); landed(nxt.y); landed(nxt.z); landed(nxt.w);
    for (uint64_t tile0 = blockIdx.x; tile0 < n_tiles; tile0 += stride) {
        const uint64_t tile = tile0 + mine;
        tile_store(nxt, lo_valid, hi_valid, tile, s_codes[half], s_inv[half], pinv != nullptr, lane);        // barrier inside (covers the zeroed counters)
        if (tile0 + stride < n_tiles) nxt = tile_fetch(ab, lo_valid, hi_valid, tile + stride, pinv, lane);    // in flight during the split
        tile_lane_scan_all(s_codes[half], s_inv[half], lo_valid, tile, k, er, [&](int i, bool valid, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) {
            const bool is_fw = fw < rv;
            const uint64_t key = is_fw ? fw : rv;
            const uint64_t h = table_hash(key, (uint32_t)k);
            const uint32_t b = valid ? p1_bin_of<BINMODE>(cfg, key, h) : NB;
            if (b != NB) atomicAdd(&s_hist[b], 1u);
            if (TOP8) { s_buf[i * TILE_THREADS + lane] = top8_rec(h, edge_idx6_any(is_fw, prev, next)); s_bin[i * TILE_THREADS + lane] = (uint16_t)b; }
            else s_buf[half * MS_TILE + i * TILE_THREADS + lane] = narrow_word(narrow_main(h), narrow_aux(h, edge_idx6_any(is_fw, prev, next)), b);
        }, lane);
        __syncthreads();
        // exclusive scan of the 256 counts: one bin per thread
        const uint32_t cnt = bin_owner ? s_hist[tid] : 0u;
        uint32_t incl = cnt;
        if (bin_owner) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += n; }
            if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        }
        uint64_t w[16];
        uint32_t wb[TOP8 ? 16 : 1];
#pragma unroll
        for (int it = 0; it < 16; ++it) { w[it] = s_buf[half * MS_TILE + it * TILE_THREADS + lane]; if (TOP8) wb[it] = s_bin[it * TILE_THREADS + lane]; }            // this lane's own words
        __syncthreads();
        if (bin_owner) {
            uint32_t excl = incl - cnt;
            for (int v = 0; v < (tid >> 6); ++v) excl += s_wave[v];
            s_loff[tid] = excl;
            if (tid == (int)NB - 1) s_loff[NB] = excl + cnt;
            s_hist[tid] = excl;                                         // the bin's placement cursor
            s_grel[tid] = gabs - excl;                                  // output index of staged record j of bin tid = s_grel + j
            gabs += cnt;
        }
        __syncthreads();
        uint32_t pl[16];                                                // (two loops: the sixteen cursor atomics are in flight together)
#pragma unroll
        for (int it = 0; it < 16; ++it) { const uint32_t b = TOP8 ? wb[TOP8 ? it : 0] : narrow_word_bin(w[it]); pl[it] = b != NB ? atomicAdd(&s_hist[b], 1u) : ~0u; }
#pragma unroll
        for (int it = 0; it < 16; ++it) if (pl[it] != ~0u) { s_buf[pl[it]] = w[it]; if (TOP8) s_bin[pl[it]] = (uint16_t)wb[TOP8 ? it : 0]; }
        __syncthreads();
        const uint32_t total = s_loff[NB];
        uint64_t cv[16];
        uint32_t cg[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) cv[it] = s_buf[tid + it * THREADS];
#pragma unroll
        for (int it = 0; it < 16; ++it) cg[it] = s_grel[(uint32_t)(tid + it * THREADS) < total ? (TOP8 ? (uint32_t)s_bin[TOP8 ? tid + it * THREADS : 0] : narrow_word_bin(cv[it])) : 0u] + (tid + it * THREADS);   // (behind `total` the stage holds stale words)
        if (bin_owner) s_hist[tid] = 0;                                 // (all placements are behind the barrier above)
        landed(nxt.x); landed(nxt.y); landed(nxt.z); landed(nxt.w);     // the wait for the prefetch in front of the stores (block_multisplit)
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if ((uint32_t)(tid + it * THREADS) < total) {
                if (TOP8) reinterpret_cast<uint64_t*>(recs)[cg[it]] = cv[it];
                else { recs[cg[it]] = (uint32_t)cv[it]; recs_aux[cg[it]] = (uint8_t)(cv[it] >> 48); }
            }
        }
    }
}

// ---- one generic level of the record split (LevelCfg) -----------------------------------------
// work units: segment b is cut into ceil(size_b / P2_UNIT) units; unit_base = exclusive prefix
// (one workgroup; n_seg <= SEG_MAX: every thread takes a run of consecutive segments, the run totals are scanned)
__global__ __launch_bounds__(1024) void k_lv_units(const unsigned long long* __restrict__ seg_off, const unsigned long long* __restrict__ seg_hi, LevelCfg lv,
                                                   unsigned long long* __restrict__ unit_base) {
    __shared__ unsigned long long s_part[1024];
    const uint32_t tid = threadIdx.x, per = (lv.n_seg + 1023) / 1024;
    const uint32_t lo = min(tid * per, lv.n_seg), hi = min(lo + per, lv.n_seg);
    unsigned long long sum = 0;
    for (uint32_t b = lo; b < hi; ++b) sum += (seg_hi[b] - seg_off[b] + P2_UNIT - 1) / P2_UNIT;
    s_part[tid] = sum;
    __syncthreads();
    if (tid < 64) {                              // wave 0 scans the 1024 partials, 16 per lane
        unsigned long long loc[16], tot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) { loc[j] = s_part[tid * 16 + j]; tot += loc[j]; }
        unsigned long long incl = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned long long nn = __shfl_up(incl, o, 64); if ((int)tid >= o) incl += nn; }
        unsigned long long run = incl - tot;
#pragma unroll
        for (int j = 0; j < 16; ++j) { s_part[tid * 16 + j] = run; run += loc[j]; }
        if (tid == 63) unit_base[lv.n_seg] = incl;
    }
    __syncthreads();
    unsigned long long run = s_part[tid];
    for (uint32_t b = lo; b < hi; ++b) { unit_base[b] = run; run += (seg_hi[b] - seg_off[b] + P2_UNIT - 1) / P2_UNIT; }
}
__device__ __forceinline__ uint32_t seg_of_unit(const unsigned long long* unit_base, uint32_t n_seg, uint64_t u) {
    uint32_t lo = 0, hi = n_seg;                  // largest b with unit_base[b] <= u (skips empty segments)
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (unit_base[mid] <= u) lo = mid; else hi = mid; }
    return lo;
}
// pass A: per-unit bin counts -> M2[unit][bin] (u32)
template <int FMT>
__global__ __launch_bounds__(MS_THREADS) void k_lv_hist(const uint64_t* __restrict__ recs, const uint8_t* __restrict__ /*recs_aux: no bin function reads it*/, LevelCfg lv,
                                                        const unsigned long long* __restrict__ seg_off, const unsigned long long* __restrict__ seg_hi,
                                                        const unsigned long long* __restrict__ unit_base, uint32_t* __restrict__ m2) {
    __shared__ uint32_t s_hist[NB_MAX];
    const uint64_t n_units = unit_base[lv.n_seg];
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint32_t b_in = seg_of_unit(unit_base, lv.n_seg, u);
        const uint32_t b = lv.spb > 1 ? b_in / lv.spb : b_in;          // logical segment of the bin functions
        const uint64_t lo = seg_off[b_in] + (u - unit_base[b_in]) * P2_UNIT;
        const uint64_t hi = lo + P2_UNIT < seg_hi[b_in] ? lo + P2_UNIT : seg_hi[b_in];
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
    const uint32_t n_lseg = lv.n_seg / lv.spb;                       // logical segments (spb input segments each)
    if (r >= (uint64_t)n_lseg * lv.nb) return;                       // wave-uniform
    // group r = (logical segment b, bin), segment-major
    const uint32_t b = (uint32_t)(r / lv.nb), bin = (uint32_t)(r % lv.nb);
    const uint64_t u0 = unit_base[b * lv.spb], u1 = unit_base[(b + 1) * lv.spb];
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
// the same with one THREAD per group, for levels whose segments have only a few units each (large tables: 65536 segments
// of two or three units): the lanes of a wave take consecutive bins, so the m2 accesses stay coalesced, and the launch
// is 64 times smaller (3.2 M groups: 424 -> ~30 us)
__global__ __launch_bounds__(256) void k_lv_offsets_thread(uint32_t* __restrict__ m2, LevelCfg lv, const unsigned long long* __restrict__ unit_base,
                                                           unsigned long long* __restrict__ group_count) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_lseg = lv.n_seg / lv.spb;
    if (r >= (uint64_t)n_lseg * lv.nb) return;
    const uint32_t b = (uint32_t)(r / lv.nb), bin = (uint32_t)(r % lv.nb);
    const uint64_t u0 = unit_base[b * lv.spb], u1 = unit_base[(b + 1) * lv.spb];
    unsigned long long run = 0;
    for (uint64_t u = u0; u < u1; ++u) {
        const uint32_t c = m2[u * lv.nb + bin];
        m2[u * lv.nb + bin] = (uint32_t)run;
        run += c;
    }
    group_count[r] = run;
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
                                                           const unsigned long long* __restrict__ seg_off, const unsigned long long* __restrict__ seg_hi,
                                                           const unsigned long long* __restrict__ unit_base, const uint32_t* __restrict__ m2,
                                                           const unsigned long long* __restrict__ group_base, uint64_t* __restrict__ out,
                                                           uint8_t* __restrict__ out_aux) {
    constexpr bool WIDE = FMT == FMT_WIDE, TIGHT = FMT == FMT_NARROW_TO_TIGHT, NARROW = FMT == FMT_NARROW || TIGHT, CONVERT = FMT == FMT_PACK8_TO_NARROW,
                   TOP8 = FMT == FMT_TOP8, HAS_AUX = WIDE || NARROW;
    constexpr int MS_FMT = (CONVERT || TIGHT) ? FMT_NARROW : TOP8 ? FMT_PACK8 : FMT;       // format of the records this kernel stages
    const uint32_t* recs32 = reinterpret_cast<const uint32_t*>(recs);
    __shared__ MsShared<NBC, MS_FMT, LV_TILE> s;
    __shared__ uint32_t s_rst[TIGHT ? NBC : 1];     // TIGHT (last level, bin = region): rstart[] of the segment's regions
    const uint32_t nb = lv.nb;
    const uint64_t n_units = unit_base[lv.n_seg];
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint32_t b_in = seg_of_unit(unit_base, lv.n_seg, u);
        const uint32_t b = lv.spb > 1 ? b_in / lv.spb : b_in;          // logical segment: bin functions and output groups
        const uint64_t lo = seg_off[b_in] + (u - unit_base[b_in]) * P2_UNIT;
        const uint64_t hi = lo + P2_UNIT < seg_hi[b_in] ? lo + P2_UNIT : seg_hi[b_in];
        for (uint32_t i = threadIdx.x; i < nb; i += LV_THREADS)
            s.gbase[i] = (uint32_t)(group_base[(uint64_t)b * nb + i] + m2[u * nb + i]);
        if (TIGHT) {
            const uint32_t first = (b >> lv.nr_shift) * lv.nr_rps + (b & ((1u << lv.nr_shift) - 1u)) * lv.nr_sub;      // narrow_bin's origin; nr_div == 1
            for (uint32_t i = threadIdx.x; i < nb; i += LV_THREADS) s_rst[i] = lv.rstart[first + i];
        }
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
                if (TIGHT) {
                    const uint32_t bn = i >= hi ? nb : narrow_bin(lv, b, (uint32_t)rec[j]);
                    rec[j] = narrow_word(tight_rec(b >> lv.nr_shift, (uint32_t)rec[j], aux[j], s_rst[bn < nb ? bn : 0u]), 0u, bn);
                }
                else if (NARROW) rec[j] = narrow_word((uint32_t)rec[j], aux[j], i >= hi ? nb : narrow_bin(lv, b, (uint32_t)rec[j]));
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
            }, NBC <= 512 ? lv.rep_shift : 0u);
        }
    }
}
// The split level of narrow records in the formulation of k_p1_scatter_s: thread c owns counter c of the (bin, replica) counters
// (at most 512 = one per thread), the bins' output cursors live in registers of their first counter's thread, the round's counts go
// to the LDS with adds that return nothing, and every record takes its staged place from a cursor atomic behind the scan -- no rank
// registers, no cursor arrays, four barriers per round (block_multisplit: six), the counters double-buffered.  TIGHT: the last level,
// FMT_TIGHT output.  Same units, same count matrix, same output groups as k_lv_scatter<FMT_NARROW / FMT_NARROW_TO_TIGHT, 512>.
#ifndef KQ_LVS_OCC         // waves per SIMD: two workgroups per CU (at three -- 80 registers, 5 to 9 of them spilled -- the tight level is 8 % slower
#define KQ_LVS_OCC 4       // than k_lv_scatter instead of 2..8 % faster).  Measured in one process on the same buffers (KQ_OPT_KERNEL_SET): the level
#define KQ_LVS_OCC_TIGHT 4 // that writes tight records gains (3 Gbp 1.89 -> 1.85 ms, 1 Gbp 3.41 -> 3.15 ms per slice), a level that writes narrow
#endif                     // records does not (2.03 -> 2.13 / 1.65 -> 1.65 ms, 2.94 -> 3.13 ms): it stays with k_lv_scatter.
template <bool TIGHT>
__global__ __launch_bounds__(LV_THREADS, TIGHT ? KQ_LVS_OCC_TIGHT : KQ_LVS_OCC) void k_lv_scatter_s(const uint32_t* __restrict__ recs32, const uint8_t* __restrict__ recs_aux, LevelCfg lv,
                                                           const unsigned long long* __restrict__ seg_off, const unsigned long long* __restrict__ seg_hi,
                                                           const unsigned long long* __restrict__ unit_base, const uint32_t* __restrict__ m2,
                                                           const unsigned long long* __restrict__ group_base, uint32_t* __restrict__ out,
                                                           uint8_t* __restrict__ out_aux) {
    constexpr int NC = 512;
    static_assert(LV_THREADS == NC && LV_ITEMS == 8, "one counter per thread");
    __shared__ uint64_t s_buf[LV_TILE];
    __shared__ uint32_t s_hist[2][NC], s_loff[NC + 1], s_grel[NC], s_wave[LV_THREADS / 64];
    __shared__ uint32_t s_rst[TIGHT ? NC : 1];      // TIGHT (bin = region): rstart[] of the segment's regions
    const int tid = threadIdx.x;
    const uint32_t nb = lv.nb, rs = lv.rep_shift, n_ctr = (nb + 1) << rs;      // (nb + 1) << rs <= NC (host)
    const uint32_t sub = (uint32_t)tid & ((1u << rs) - 1u), my_bin = (uint32_t)tid >> rs;
    const bool own_bin = sub == 0 && my_bin < nb;                       // this thread's counter is the first of bin my_bin
    const uint64_t n_units = unit_base[lv.n_seg];
    for (uint64_t u = blockIdx.x; u < n_units; u += gridDim.x) {
        const uint32_t b_in = seg_of_unit(unit_base, lv.n_seg, u);
        const uint32_t b = lv.spb > 1 ? b_in / lv.spb : b_in;          // logical segment: bin functions and output groups
        const uint64_t lo = seg_off[b_in] + (u - unit_base[b_in]) * P2_UNIT;
        const uint64_t hi = lo + P2_UNIT < seg_hi[b_in] ? lo + P2_UNIT : seg_hi[b_in];
        uint32_t gabs = own_bin ? (uint32_t)(group_base[(uint64_t)b * nb + my_bin] + m2[u * nb + my_bin]) : 0u;      // the bin's output cursor
        if (TIGHT) {                                                    // (every thread is behind the last barrier of the previous unit: nobody reads s_rst or counts any more)
            const uint32_t first = (b >> lv.nr_shift) * lv.nr_rps + (b & ((1u << lv.nr_shift) - 1u)) * lv.nr_sub;      // narrow_bin's origin; nr_div == 1
            if ((uint32_t)tid < nb) s_rst[tid] = lv.rstart[first + tid];
        }
        s_hist[0][tid] = 0;
        __syncthreads();
        // software pipeline: the next round's records are loaded before this round is split (loads unconditional, index clamped
        // to the unit: see k_lv_scatter)
        uint32_t nxt[LV_ITEMS], nxt_aux[LV_ITEMS];
        const uint64_t last = hi - 1;                                   // a unit is never empty
#pragma unroll
        for (int j = 0; j < LV_ITEMS; ++j) {
            const uint64_t i = min(lo + (uint64_t)j * LV_THREADS + tid, last);
            nxt[j] = recs32[i];
            nxt_aux[j] = recs_aux[i];
        }
#pragma unroll
        for (int j = 0; j < LV_ITEMS; ++j) { landed(nxt[j]); landed(nxt_aux[j]); }
        uint32_t par = 0;
        for (uint64_t pos = lo; pos < hi; pos += LV_TILE, par ^= 1u) {
            uint32_t* hist = s_hist[par];
            uint64_t w[LV_ITEMS];
#pragma unroll
            for (int j = 0; j < LV_ITEMS; ++j) {
                const uint64_t i = pos + (uint64_t)j * LV_THREADS + tid;
                const uint32_t bn = i >= hi ? nb : narrow_bin(lv, b, nxt[j]);
                w[j] = TIGHT ? narrow_word(tight_rec(b >> lv.nr_shift, nxt[j], nxt_aux[j], s_rst[bn < nb ? bn : 0u]), 0u, bn) : narrow_word(nxt[j], nxt_aux[j], bn);
                if (bn != nb) atomicAdd(&hist[(bn << rs) | sub], 1u);
            }
#pragma unroll
            for (int j = 0; j < LV_ITEMS; ++j) {
                const uint64_t i = min(pos + LV_TILE + (uint64_t)j * LV_THREADS + tid, last);
                nxt[j] = recs32[i];
                nxt_aux[j] = recs_aux[i];
            }
            __syncthreads();
            // exclusive scan of the counters: one per thread
            const uint32_t cnt = (uint32_t)tid < n_ctr ? hist[tid] : 0u;
            uint32_t incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t n = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += n; }
            if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
            s_hist[par ^ 1u][tid] = 0;                                  // the next round's counters (last used before the previous round's last barrier)
            __syncthreads();
            uint32_t excl = incl - cnt;
            for (int v = 0; v < (tid >> 6); ++v) excl += s_wave[v];
            s_loff[tid] = excl;
            hist[tid] = excl;                                           // the counter's placement cursor
            if (own_bin) s_grel[my_bin] = gabs - excl;                  // output index of staged record j of the bin = s_grel + j
            __syncthreads();
            if (own_bin) gabs += s_loff[(my_bin + 1) << rs] - excl;     // (the first counter of the discard bin holds the total)
#pragma unroll
            for (int j = 0; j < LV_ITEMS; ++j) {
                const uint32_t bn = narrow_word_bin(w[j]);
                if (bn != nb) s_buf[atomicAdd(&hist[(bn << rs) | sub], 1u)] = w[j];
            }
            __syncthreads();
            const uint32_t total = s_loff[nb << rs];
            uint64_t cv[LV_ITEMS];
            uint32_t cg[LV_ITEMS];
#pragma unroll
            for (int it = 0; it < LV_ITEMS; ++it) cv[it] = s_buf[tid + it * LV_THREADS];
#pragma unroll
            for (int it = 0; it < LV_ITEMS; ++it) cg[it] = s_grel[(uint32_t)(tid + it * LV_THREADS) < total ? narrow_word_bin(cv[it]) : 0u] + (tid + it * LV_THREADS);   // (behind `total`: stale words)
#pragma unroll
            for (int j = 0; j < LV_ITEMS; ++j) { landed(nxt[j]); landed(nxt_aux[j]); }      // the wait for the prefetch in front of the stores (block_multisplit)
#pragma unroll
            for (int it = 0; it < LV_ITEMS; ++it) {
                if ((uint32_t)(tid + it * LV_THREADS) < total) {
                    out[cg[it]] = (uint32_t)cv[it];
                    if (!TIGHT) out_aux[cg[it]] = (uint8_t)(cv[it] >> 48);
                }
            }
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


// P3: one workgroup per table region.  The region's slots (REGION_SLOTS x 16 B in HBM) are staged in LDS as the
// three-word image of kq_device.h (img_load), all records of the region are applied with LDS atomics (same
// two-tier rule as table_add), and the image is streamed back (img_store).  Records and slots both hold hash
// bits, so the walk never reconstructs a key.  Global atomics only for the rare high-copy tier and the two totals.
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
// loads through a pointer that was itself loaded from memory are FLAT loads unless the address space is stated, and a
// flat load counts on lgkmcnt as well as vmcnt: every LDS wait of the record walk would also wait for the prefetch
template <class T> __device__ __forceinline__ T ld_global(const T* p) { return *(const __attribute__((address_space(1))) T*)p; }

// A wave's view of the pending sets of one region: lane q < n_sets holds set q's share (first record, record count,
// first ticket); tickets = groups of GRP records, set after set.  Built by every wave for itself (two loads + a wave
// scan, no LDS, no barrier); a ticket is located with one ballot.
template <uint32_t GRP>
struct SetTickets {
    uint64_t lo; uint32_t cnt, first, n_grp;
    __device__ __forceinline__ void build(const P3Set* __restrict__ sets, uint32_t n_sets, uint64_t r, uint32_t lane) {
        lo = 0; cnt = 0;
        if (lane < n_sets) { const unsigned long long* b = ld_global(&sets[lane].base); lo = ld_global(b + r); cnt = (uint32_t)(ld_global(b + r + 1) - lo); }   // a region holds < 2^32 records of one set
        const uint32_t mine = (cnt + GRP - 1) / GRP;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < P3_MAX_SETS; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((int)lane >= o) incl += v; }
        first = incl - mine;
        n_grp = __builtin_amdgcn_readlane(incl, 63);
    }
    // set of ticket g (any non-empty set when g is past the end: the caller only needs loadable addresses then)
    __device__ __forceinline__ void locate(uint32_t g, uint32_t& q, uint32_t& off, uint32_t& q_cnt, uint64_t& q_lo) const {
        const uint64_t nonempty = __ballot(cnt != 0);
        const uint64_t m = __ballot(cnt != 0 && first <= g);
        q = (g < n_grp && m) ? 63u - (uint32_t)__clzll((unsigned long long)m) : (uint32_t)__ffsll((unsigned long long)nonempty) - 1u;
        const uint32_t f = __builtin_amdgcn_readlane(first, q);
        q_cnt = __builtin_amdgcn_readlane(cnt, q);
        q_lo = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(lo >> 32), q) << 32) | __builtin_amdgcn_readlane((uint32_t)lo, q);
        off = (g < n_grp ? g - f : 0u) * GRP;
    }
};

#ifndef KQ_P3_THREADS
#define KQ_P3_THREADS 512
#define KQ_P3_OCC 6
#define KQ_P3_PF 2
#endif
constexpr int P3_THREADS = KQ_P3_THREADS;        // three 48 KiB images per CU (24 waves): with records double-buffered and groups handed out by ticket, one more region in flight per CU beats 2 x 1024 threads (whole count job 2.64 vs 2.67 ms)
// Two instantiations share the regions: HOT = false takes the ordinary ones (deep record prefetch, no
// folding state: fits the 80 VGPRs that let three workgroups share a CU) and appends the skewed ones
// to hot_list; HOT = true then walks that list with the folding loop.
// The records come as up to P3_MAX_SETS record SETS, each sorted by region with its own offset array: the host
// keeps the sets of several slices / batches pending and applies them in ONE pass over the table (a pass streams
// every region image in and out, so its cost is shared by all records of all sets; kreeq_amd.hip "pending sets").
template <int FMT, bool HOT>
__global__ __launch_bounds__(P3_THREADS, HOT ? 4 : KQ_P3_OCC) void k_count_regions(TableView t, const P3Set* __restrict__ sets, uint32_t n_sets,
                                                              int aux_fmt, int table_is_empty,
                                                              unsigned long long* __restrict__ hot_list /*[0] = count, then region ids*/,
                                                              uint32_t narrow_rps /*FMT_NARROW: regions per top-bit bucket*/) {
    constexpr bool WIDE = FMT == FMT_WIDE, NARROW = FMT == FMT_NARROW, TOP8 = FMT == FMT_TOP8, HAS_AUX = WIDE || NARROW;
    constexpr int PF = KQ_P3_PF;
    constexpr uint64_t GRP = 64ull * PF;     // records per ticket
    __shared__ uint64_t s_img[REGION_SLOTS * 3];
    __shared__ uint64_t s_lut[64];            // edge indices -> u8x8 increment: one LDS read instead of ~8 VALU per record (-2.4 %)
    if (threadIdx.x < 64) s_lut[threadIdx.x] = idx6_to_pack(threadIdx.x);     // visible after the first region's barrier
    __shared__ unsigned long long s_new, s_kmers;
    __shared__ unsigned int s_grp;
    // high-copy tier of this region, aggregated in LDS: a repeat k-mer with millions of instances
    // would otherwise serialise millions of global atomics on one side-table entry
    constexpr int HC_LDS = 64;
    __shared__ uint64_t s_hckey[HC_LDS];
    __shared__ uint32_t s_hccnt[HC_LDS][8];
    const int tid = threadIdx.x;
    const uint64_t n_work = HOT ? hot_list[0] : t.reg_hi - t.reg_lo;
#ifdef KQ_STAMPS
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    for (uint64_t w = blockIdx.x; w < n_work; w += gridDim.x) {
        const uint64_t r = HOT ? hot_list[1 + w] : t.reg_lo + w;
        SetTickets<(uint32_t)GRP> tk;                                   // this region's share of every set (per wave, in registers)
        tk.build(sets, n_sets, r, (uint32_t)tid & 63u);
        uint64_t n_recs = tk.cnt;                                       // block-uniform after the reduction
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_recs += __shfl_xor(n_recs, o, 64);
        const uint32_t narrow_bucket = (NARROW || TOP8) ? (uint32_t)r / narrow_rps : 0u;
        const bool tight = NARROW && aux_fmt == AUX_TIGHT;              // FMT_TIGHT sets: one u32 relative to rstart[r], no lockstep byte
        const uint32_t start_r = tight ? t.rstart[r] : 0u;
        if (n_recs == 0) {                                              // block-uniform
            if (!HOT && table_is_empty == 2) {    // lazy kq_clear: this launch initialises every region, also the ones without records
                ulonglong2* g2 = reinterpret_cast<ulonglong2*>(t.slots + (r << REGION_SHIFT));
                for (int i = tid; i < (int)REGION_SLOTS; i += P3_THREADS) g2[i] = make_ulonglong2(0ull, 0ull);
            }
            continue;
        }
        // folding costs a ballot + shuffle per iteration: only regions that receive far more records than
        // they have slots (skew, or very deep coverage) take that path, in the second launch
        if (!HOT && n_recs > 32ull * REGION_SLOTS) {
            if (tid == 0) hot_list[1 + atomicAdd(&hot_list[0], 1ull)] = r;
            continue;
        }
        if (tid < HC_LDS) {
            s_hckey[tid] = EMPTY_KEY;
#pragma unroll
            for (int e = 0; e < 8; ++e) s_hccnt[tid][e] = 0;
        }
        ulonglong2* gimg = reinterpret_cast<ulonglong2*>(t.slots + (r << REGION_SHIFT));
        if (table_is_empty) {            // first batch after kq_create / kq_clear: the image is known, skip the 32 KiB read
            for (int i = tid; i < (int)(REGION_SLOTS * 3); i += P3_THREADS) s_img[i] = (i % 3 == 0) ? EMPTY_KEY : 0ull;
        } else {
            ulonglong2 v[REGION_SLOTS / P3_THREADS];
#pragma unroll
            for (int j = 0; j < (int)(REGION_SLOTS / P3_THREADS); ++j) v[j] = gimg[tid + j * P3_THREADS];     // all loads in flight, then the LDS writes
#pragma unroll
            for (int j = 0; j < (int)(REGION_SLOTS / P3_THREADS); ++j) img_load(s_img, tid + j * P3_THREADS, v[j].x, v[j].y);
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
        auto find_slot = [&](uint64_t key /*56-bit hash remainder*/, uint64_t h) -> uint32_t {
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
        auto add_wide = [&](uint64_t h, const uint32_t (&e)[8]) {
            const uint64_t key = key_of_hash(h, t.k);                  // the high-copy tier is keyed by the canonical key
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
                hc_add(t, h, 0, 0, e);
            }
        };
        // one record: `pack` holds its (at most two) edge bits, one per byte lane
        auto apply1 = [&](uint64_t h, uint64_t pack) {
            const uint32_t w = find_slot(slot_rem(h, t.k), h);
            if (w == REGION_SLOTS * 3) return;
            ++n_ok;
            const uint64_t old = atomicAdd((unsigned long long*)&s_img[w + 2], 1ull);
            if (!pack) return;
            if (old < LOW_TIER_MAX) { atomicAdd((unsigned long long*)&s_img[w + 1], (unsigned long long)pack); return; }
            uint32_t e1[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) e1[q] = (uint32_t)(pack >> (8 * q)) & 1u;
            add_wide(h, e1);
        };
        // `cnt` folded instances of the k-mer with hash h, edge counts e[0..7] (each <= cnt)
        auto apply = [&](uint64_t h, const uint32_t (&e)[8], uint32_t cnt) {
            const uint32_t w = find_slot(slot_rem(h, t.k), h);
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
            add_wide(h, e);
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
        uint64_t nxt_rec[PF];
        uint32_t nxt_aux[PF];
        // groups of PF*64 records are handed to waves from an LDS ticket: waves that hit long probe
        // chains or contended slots take fewer groups, so all waves reach the barrier together.
        const uint32_t lane = tid & 63;
        const uint32_t n_grp = tk.n_grp;
        uint32_t cur_n = 0;                                            // records in the group whose loads are in flight
        auto fetch = [&](uint32_t g, uint32_t& n) {                     // issues the loads of ticket g (or of a valid dummy when g is past the end)
            uint32_t q, off, cnt; uint64_t lo_q;
            tk.locate(g, q, off, cnt, lo_q);
            const uint64_t* rp = sets[q].recs;
            const uint8_t* ap = sets[q].aux;
            n = min(cnt - off, (uint32_t)GRP);
#pragma unroll
            for (int qq = 0; qq < PF; ++qq) {
                const uint64_t j = lo_q + min(off + (uint32_t)qq * 64u + lane, cnt - 1u);      // unconditional, index clamped
                nxt_rec[qq] = NARROW ? (uint64_t)ld_global(reinterpret_cast<const uint32_t*>(rp) + j) : ld_global(rp + j);
                nxt_aux[qq] = (HAS_AUX && !tight) ? ld_global(ap + j) : 0u;
            }
        };
        uint32_t g_cur = tid >> 6;
        fetch(g_cur, cur_n);
        while (g_cur < n_grp) {                                         // wave-uniform
          uint64_t cur_rec[PF];
          uint32_t cur_aux[PF];
#pragma unroll
          for (int q = 0; q < PF; ++q) { cur_rec[q] = nxt_rec[q]; cur_aux[q] = nxt_aux[q]; }
          uint32_t g_nxt = 0;
          if (lane == 0) g_nxt = atomicAdd(&s_grp, 1u);
          g_nxt = __builtin_amdgcn_readfirstlane(g_nxt);
          const uint32_t n_cur = cur_n;
          fetch(g_nxt, cur_n);
          g_cur = g_nxt;
#pragma unroll
          for (int q = 0; q < PF; ++q) {
            bool active = (uint32_t)q * 64u + lane < n_cur;
            const uint64_t rec = cur_rec[q];
            const uint32_t aux = cur_aux[q];
            uint64_t pack = 0;
            const uint64_t h = NARROW ? (tight ? tight_hash((uint32_t)rec, start_r) : narrow_hash(narrow_bucket, (uint32_t)rec, aux))
                             : TOP8 ? top8_hash(narrow_bucket, rec) : rec_hash<WIDE>(rec);
            const uint64_t key = h;                                          // identity of the k-mer inside this kernel: its hash (a bijection of the key)
            if (active) {
                pack = NARROW ? s_lut[tight ? (uint32_t)rec & 63u : (aux >> 2) & 63u] : TOP8 ? s_lut[(uint32_t)rec & 63u]
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
            if (active) apply1(h, pack);
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
#pragma unroll
        for (int j = 0; j < (int)(REGION_SLOTS / P3_THREADS); ++j) gimg[tid + j * P3_THREADS] = img_store(t, s_img, tid + j * P3_THREADS, r);
        KQ_STAMP(3);                                                    // image store issue
        if (tid == 0) {
            if (s_new) atomicAdd(&t.st->slots_used, s_new);
            if (s_kmers) atomicAdd(&t.st->kmers_added, s_kmers);
        }
        __syncthreads();
        KQ_STAMP(4);                                                    // final barrier
    }
}

// rstart[r] = ceil(r 2^32 / R): the first value of the top 32 hash bits that falls into region r
__global__ __launch_bounds__(256) void k_region_starts(uint32_t* __restrict__ out, uint64_t R) {
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r <= R; r += (uint64_t)gridDim.x * blockDim.x)
        out[r] = r < R ? (uint32_t)(((r << 32) + R - 1) / R) : 0xFFFFFFFFu;
}

// P3 for FMT_NARROW records (k <= 21, the default k; ordinary regions only -- skewed ones go to hot_list and the
// generic folding kernel).  Same job as k_count_regions<FMT_NARROW, false>, with a compact LDS image and 32-bit keys:
//   s_a[slot] = key31 | cnt << 32     key31 = the hash bits that region r does not imply: (top 32 hash bits - rstart[r])
//                                     << 10 | the 10 hash bits below them (2k - log2 R <= 31 bits: R >= 2048, k <= 21);
//                                     low word 0xFFFFFFFF = free.  cnt = instances (bit 31 = arrived as a tombstone).
//   s_e[slot] = the eight u8 edge counters
// 32 KiB per region instead of 48: four workgroups per CU; a probe reads the key AND the count in one word, the count
// add returns the tier decision from the same word, and a record turns into its key with three 32-bit operations
// (no 64-bit hash is rebuilt).  Per record: 2 LDS reads (two slots of the probe sequence) + 2 LDS atomics.
constexpr uint32_t N32_EMPTY = 0xFFFFFFFFu, N32_TOMB = 1u << 31;
#ifndef KQ_N32_OCC
#define KQ_N32_OCC 6
#endif
// TIGHT: the sets hold FMT_TIGHT records (one u32 = key << 6 | edge indices, no lockstep byte).
template <int KC, bool TIGHT>
__global__ __launch_bounds__(P3_THREADS, KQ_N32_OCC) void k_count_regions_n32(TableView t, const P3Set* __restrict__ sets, uint32_t n_sets, int table_is_empty,
                                                                      unsigned long long* __restrict__ hot_list, uint32_t rps) {
    constexpr int PF = KQ_P3_PF;
    constexpr uint32_t GRP = 64u * PF;
    __shared__ uint64_t s_a[REGION_SLOTS];
    __shared__ uint64_t s_e[REGION_SLOTS];
    __shared__ uint64_t s_lut[64];
    if (threadIdx.x < 64) s_lut[threadIdx.x] = idx6_to_pack(threadIdx.x);     // visible after the first region's barrier
    __shared__ unsigned int s_new, s_kmers, s_grp;
    constexpr int HC_LDS = 64;
    __shared__ uint64_t s_hckey[HC_LDS];
    __shared__ uint32_t s_hccnt[HC_LDS][8];
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t k = KC ? KC : t.k;
    const uint32_t off_shift = 42 - 2 * k;                              // k <= 21
    for (uint64_t r = t.reg_lo + blockIdx.x; r < t.reg_hi; r += gridDim.x) {
        SetTickets<GRP> tk;
        tk.build(sets, n_sets, r, lane);
        uint64_t n_recs = tk.cnt;                                       // block-uniform after the reduction
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_recs += __shfl_xor(n_recs, o, 64);
        ulonglong2* gimg = reinterpret_cast<ulonglong2*>(t.slots + (r << REGION_SHIFT));
        if (n_recs == 0) {
            if (table_is_empty == 2)                                    // lazy kq_clear: this launch initialises every region
                for (int i = tid; i < (int)REGION_SLOTS; i += P3_THREADS) gimg[i] = make_ulonglong2(0ull, 0ull);
            continue;
        }
        if (n_recs > 32ull * REGION_SLOTS) {                            // skewed region: the folding kernel takes it
            if (tid == 0) hot_list[1 + atomicAdd(&hot_list[0], 1ull)] = r;
            continue;
        }
        const uint32_t bucket = (uint32_t)r / rps;
        const uint32_t start_r = t.rstart[r];
        const uint32_t top_base = (bucket << (32 - NARROW_CBITS)) - start_r;       // (top 32 hash bits of a record) - rstart[r] = top_base + (u32 >> 8)
        const uint32_t start_lo = start_r << 10;                                   // TIGHT: key + start_lo = the low 32 bits of (top 32 bits | 10 below)
        if (tid < HC_LDS) {
            s_hckey[tid] = EMPTY_KEY;
#pragma unroll
            for (int e = 0; e < 8; ++e) s_hccnt[tid][e] = 0;
        }
        if (table_is_empty) {
            for (int i = tid; i < (int)REGION_SLOTS; i += P3_THREADS) { s_a[i] = (uint64_t)N32_EMPTY; s_e[i] = 0; }
        } else {
            ulonglong2 v[REGION_SLOTS / P3_THREADS];
#pragma unroll
            for (int j = 0; j < (int)(REGION_SLOTS / P3_THREADS); ++j) v[j] = gimg[tid + j * P3_THREADS];
#pragma unroll
            for (int j = 0; j < (int)(REGION_SLOTS / P3_THREADS); ++j) {
                const uint64_t w0 = v[j].x;                             // rem56 | cov8 << 56, rem = hash >> 8
                uint64_t a = (uint64_t)N32_EMPTY;
                if (w0) {
                    const uint32_t key = (((uint32_t)(w0 >> 24) - start_r) << 10) | ((uint32_t)(w0 >> 14) & 1023u);
                    const uint32_t c = (uint32_t)(w0 >> COV_SHIFT);
                    a = (uint64_t)key | ((uint64_t)(c == COV8_TOMB ? (N32_TOMB | LOW_TIER_MAX) : c) << 32);
                }
                s_a[tid + j * P3_THREADS] = a;
                s_e[tid + j * P3_THREADS] = v[j].y;
            }
        }
        if (tid == 0) { s_new = 0; s_kmers = 0; s_grp = P3_THREADS / 64; }
        __syncthreads();
        uint32_t n_new = 0, n_ok = 0;
        // hash of a key of this region (high-copy tier only)
        auto hash_of = [&](uint32_t key) -> uint64_t { return ((uint64_t)(start_r + (key >> 10)) << 32) | ((uint64_t)(key & 1023u) << 22); };
        auto add_wide = [&](uint32_t key31, uint64_t pack) {
            const uint64_t h = hash_of(key31);
            const uint64_t key = key_of_hash(h, t.k);
            uint32_t e[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) e[w] = (uint32_t)(pack >> (8 * w)) & 1u;
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
                hc_add(t, h, 0, 0, e);
            }
        };
        // Records are double-buffered in registers (unconditional loads, index clamped); groups of GRP records are handed
        // to waves from an LDS ticket, so waves that hit long probe chains take fewer groups.
        uint32_t nxt_rec[PF], nxt_aux[PF];
        uint32_t cur_n = 0;
        auto fetch = [&](uint32_t g, uint32_t& n) {
            uint32_t q, off, cnt; uint64_t lo_q;
            tk.locate(g, q, off, cnt, lo_q);
            const uint32_t* rp = reinterpret_cast<const uint32_t*>(sets[q].recs);
            const uint8_t* ap = sets[q].aux;
            n = min(cnt - off, GRP);
#pragma unroll
            for (int qq = 0; qq < PF; ++qq) {
                const uint64_t j = lo_q + min(off + (uint32_t)qq * 64u + lane, cnt - 1u);
                nxt_rec[qq] = ld_global(rp + j);
                nxt_aux[qq] = TIGHT ? 0u : ld_global(ap + j);
            }
        };
        uint32_t g_cur = tid >> 6;
        fetch(g_cur, cur_n);
        while (g_cur < tk.n_grp) {                                      // wave-uniform
            uint32_t cur_rec[PF], cur_aux[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) { cur_rec[q] = nxt_rec[q]; cur_aux[q] = nxt_aux[q]; }
            uint32_t g_nxt = 0;
            if (lane == 0) g_nxt = atomicAdd(&s_grp, 1u);
            g_nxt = __builtin_amdgcn_readfirstlane(g_nxt);
            const uint32_t n_cur = cur_n;
            fetch(g_nxt, cur_n);
            g_cur = g_nxt;
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const bool active = (uint32_t)q * 64u + lane < n_cur;
                const uint32_t m = cur_rec[q], aux = cur_aux[q];
                const uint32_t low = ((m & 0xFFu) << 2) | (aux & 3u);   // the 10 hash bits below the top 32
                const uint32_t key = TIGHT ? m >> 6 : ((top_base + (m >> 8)) << 10) | low;
                uint32_t pos = TIGHT ? (((key + start_lo) >> off_shift) & (REGION_SLOTS - 4))          // off_shift + 11 <= 32 (k >= 11); quad-aligned home (hash_offset)
                             : KC == 21 ? (((m << 2) | (aux & 3u)) & (REGION_SLOTS - 4))
                                        : (uint32_t)(((((uint64_t)bucket << 34) | ((uint64_t)m << 2) | (aux & 3u)) >> off_shift) & (REGION_SLOTS - 4));
                const uint64_t pack = s_lut[TIGHT ? m & 63u : (aux >> 2) & 63u];
                // find-or-claim: two slots of the probe sequence per LDS round trip; one CAS site
                uint32_t slot = active ? REGION_SLOTS : 0u;             // REGION_SLOTS = still looking
                uint32_t probes = 0;
#ifdef KQ_ABL
                if (KQ_ABL & 4) slot = active ? pos : 0u;               // ablation build (timing only, never shipped): no probe
#endif
                while (slot == REGION_SLOTS) {
                    const uint32_t i0 = pos, i1 = (pos + 1) & (REGION_SLOTS - 1);
                    const uint32_t c0 = (uint32_t)__hip_atomic_load(&s_a[i0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t c1 = (uint32_t)__hip_atomic_load(&s_a[i1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (c0 == key) { slot = i0; break; }
                    const bool free0 = c0 == N32_EMPTY;
                    if (!free0 && c1 == key) { slot = i1; break; }
                    if (free0 || c1 == N32_EMPTY) {
                        const uint32_t ic = free0 ? i0 : i1;
                        const uint32_t got = (uint32_t)atomicCAS((unsigned long long*)&s_a[ic], (unsigned long long)N32_EMPTY, (unsigned long long)key);
                        if (got == N32_EMPTY) { ++n_new; slot = ic; break; }
                        if (got == key) { slot = ic; break; }
                        continue;                                       // another key took it: look at the pair again
                    }
                    pos = (pos + 2) & (REGION_SLOTS - 1);
                    if ((probes += 2) >= REGION_SLOTS) { atomicOr(&t.st->err_table_full, 1u); slot = REGION_SLOTS + 1; }
                }
                if (active && slot < REGION_SLOTS) {
                    ++n_ok;
#ifdef KQ_ABL
                    const uint64_t old = (KQ_ABL & 2) ? 0 : atomicAdd((unsigned long long*)&s_a[slot], 1ull << 32);
                    if (KQ_ABL & 1) continue;
#else
                    const uint64_t old = atomicAdd((unsigned long long*)&s_a[slot], 1ull << 32);
#endif
                    if (pack) {
                        if ((uint32_t)(old >> 32) < LOW_TIER_MAX) atomicAdd((unsigned long long*)&s_e[slot], (unsigned long long)pack);
                        else add_wide(key, pack);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n_new += __shfl_down(n_new, o, 64); n_ok += __shfl_down(n_ok, o, 64); }
        if (lane == 0) { if (n_new) atomicAdd(&s_new, n_new); if (n_ok) atomicAdd(&s_kmers, n_ok); }
        __syncthreads();
        if (tid < HC_LDS && s_hckey[tid] != EMPTY_KEY) {            // flush the region's high-copy sums: one entry per k-mer
            HcSlot* hs = hc_upsert(t, s_hckey[tid]);
            if (!hs) atomicOr(&t.st->err_hc_full, 1u);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (s_hccnt[tid][e]) atomicAdd((unsigned long long*)&hs->cnt[e], (unsigned long long)s_hccnt[tid][e]);
            }
        }
#pragma unroll
        for (int j = 0; j < (int)(REGION_SLOTS / P3_THREADS); ++j) {
            const int i = tid + j * P3_THREADS;
            const uint64_t a = s_a[i];
            ulonglong2 o = make_ulonglong2(0ull, 0ull);
            if ((uint32_t)a != N32_EMPTY) {
                const uint32_t key = (uint32_t)a, cw = (uint32_t)(a >> 32), c = cw & ~N32_TOMB;
                uint64_t cov8 = (cw & N32_TOMB) ? COV8_TOMB : c;
                if (c > LOW_TIER_MAX) { cov8 = COV8_TOMB; hc_add(t, hash_of(key), c - LOW_TIER_MAX, 0, nullptr); }
                o = make_ulonglong2((((uint64_t)(start_r + (key >> 10))) << 24) | ((uint64_t)(key & 1023u) << 14) | (cov8 << COV_SHIFT), s_e[i]);
            }
            gimg[i] = o;
        }
        if (tid == 0) {
            if (s_new) atomicAdd(&t.st->slots_used, (unsigned long long)s_new);
            if (s_kmers) atomicAdd(&t.st->kmers_added, (unsigned long long)s_kmers);
        }
        __syncthreads();
    }
}


// ---- k_count_regions_q4: the table pass for FMT_NARROW / FMT_TIGHT records, second formulation (round 3) ------------------
// Same job and same HBM image as k_count_regions_n32; what changes is how a wave walks its records.  n32 sent every record
// through a divergent find-or-claim loop: a wave paid the MAXIMUM probe count of its 64 lanes (4-6 iterations of ~20
// instructions at load 0.65) for every group of 64 records, ~100 VALU + ~110 SALU per group -- the pass was bound by that
// instruction stream, not by LDS or HBM (round-2 VERDICT, weak #4).  Here:
//   * a k-mer's home is a 16-byte aligned QUAD of slots (hash_offset() is quad-aligned for every kernel of the library:
//     the probe sequence is still linear, it just starts at a multiple of four), and the LDS image keeps the 32-bit keys in
//     an array of their own, so ONE straight-line step reads the four keys of the home quad (two 8-byte LDS reads), finds the
//     key or the first free slot of the quad, claims it with a single CAS if need be and applies the record: no loop, no
//     per-lane iteration count.  ~88 % of the records end there (key in its home quad, or a free slot in it);
//   * the rest go to a small per-wave queue in LDS and are drained 64 at a time by the loop formulation -- which then runs
//     with every lane busy on a record that needs it, instead of 64 lanes waiting for the slowest one.
//   s_key[slot]  = key31 (the hash bits region r does not imply, as in n32) or N32_EMPTY
//   s_cnt[slot]  = instances (bit 31 = arrived as a tombstone)
//   s_e[slot]    = the eight u8 edge counters
#ifndef KQ_Q4_QCAP
#define KQ_Q4_QCAP 128
#endif
#ifndef KQ_Q4_DEPTH
#define KQ_Q4_DEPTH 1          // tickets a wave has in flight, 4-byte records: 2 was measured (38 spilled registers): 3 Gbp 74.8 -> 85.1 ms per step, with two records per lane 78.8
#endif
#ifndef KQ_Q4_DEPTH_NT
#define KQ_Q4_DEPTH_NT 1       // 5-byte records (small tables: few sets, long pieces)
#endif
#ifndef KQ_Q4_PF_TIGHT
#define KQ_Q4_PF_TIGHT 4       // records per lane and ticket for 4-byte records (one register each): 1000 Mbp 20.2 -> 18.7 ms per step against 2; 5-byte records keep 2 (35 spilled registers at 4, no gain)
#endif
// 256 threads, four workgroups per CU (second half of round 3): at 512 x 3 the kernel held 80 VGPRs and spilled 14 of them, and the PMC
// showed what that costs -- every wave stores and reloads them once per region: 28.7 KB of scratch traffic each way per 32 KB image,
// 95 of the 203 GB a pass wrote.  With 104 VGPRs nothing spills; the pass is 1 % (3 Gbp) to 5 % (1 Gbp, configs[1]) faster and moves
// a third fewer bytes.  (Half the threads for the other region kernels was measured too: k = 31 counts 0.75 -> 1.03 ms, union 1.84 ->
// 2.59 ms -- they keep 512.)
#ifndef KQ_Q4_THREADS
#define KQ_Q4_THREADS 256
#define KQ_Q4_OCC 4
#endif
constexpr int Q4_THREADS = KQ_Q4_THREADS;
template <int KC, bool TIGHT>
__global__ __launch_bounds__(Q4_THREADS, KQ_Q4_OCC) void k_count_regions_q4(TableView t, const P3Set* __restrict__ sets, uint32_t n_sets, int table_is_empty,
                                                                     unsigned long long* __restrict__ hot_list, uint32_t rps) {
    constexpr int PF = TIGHT ? KQ_Q4_PF_TIGHT : KQ_P3_PF;
    constexpr int DEPTH = TIGHT ? KQ_Q4_DEPTH : KQ_Q4_DEPTH_NT;            // tickets a wave has in flight
    constexpr uint32_t GRP = 64u * PF;
    constexpr uint32_t QCAP = KQ_Q4_QCAP, NONE = 0xFFFFFFFFu;
    __shared__ uint64_t s_key2[REGION_SLOTS / 2];                       // the keys, read two at a time
    __shared__ uint32_t s_cnt[REGION_SLOTS];
    __shared__ uint64_t s_e[REGION_SLOTS];
    __shared__ uint64_t s_lut[64];
    __shared__ uint64_t s_q[Q4_THREADS / 64][QCAP];                     // per-wave queue of records that did not resolve in their home quad: key | idx6 << 32
    uint32_t* s_key = reinterpret_cast<uint32_t*>(s_key2);
    if (threadIdx.x < 64) s_lut[threadIdx.x] = idx6_to_pack(threadIdx.x);     // visible after the first region's barrier
    __shared__ unsigned int s_new, s_kmers, s_grp;
    constexpr int HC_LDS = 64;
    __shared__ uint64_t s_hckey[HC_LDS];
    __shared__ uint32_t s_hccnt[HC_LDS][8];
    const int tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t k = KC ? KC : t.k;
    const uint32_t off_shift = 42 - 2 * k;                              // k <= 21
#ifdef KQ_STAMPS
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
    for (uint64_t r = t.reg_lo + blockIdx.x; r < t.reg_hi; r += gridDim.x) {
        SetTickets<GRP> tk;
        tk.build(sets, n_sets, r, lane);
        uint64_t n_recs = tk.cnt;                                       // block-uniform after the reduction
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n_recs += __shfl_xor(n_recs, o, 64);
        ulonglong2* gimg = reinterpret_cast<ulonglong2*>(t.slots + (r << REGION_SHIFT));
        if (n_recs == 0) {
            if (table_is_empty == 2)                                    // lazy kq_clear: this launch initialises every region
                for (int i = tid; i < (int)REGION_SLOTS; i += Q4_THREADS) gimg[i] = make_ulonglong2(0ull, 0ull);
            continue;
        }
        if (n_recs > 32ull * REGION_SLOTS) {                            // skewed region: the folding kernel takes it
            if (tid == 0) hot_list[1 + atomicAdd(&hot_list[0], 1ull)] = r;
            continue;
        }
        // the first records of this wave are requested NOW: their latency (an HBM round trip behind the ticket build's two)
        // runs under the image set-up and the barrier instead of behind them (round 3: a region visit is mostly such
        // dependent round trips -- 23 ns per region and pass at 5.3 M regions whatever the number of records)
        // (DEPTH = 2 keeps two tickets per wave in flight: a region's records come as one short piece per pending set, ~170
        // records of each of 30 sets at 3 Gbp, every piece from another place of the arena.  Measured and not the default.)
        constexpr int NW = Q4_THREADS / 64;
        uint32_t recA[PF], auxA[PF], recB[PF], auxB[PF];
        uint32_t nA = 0, nB = 0;
        auto fetch = [&](uint32_t g, uint32_t (&rec)[PF], uint32_t (&aux)[PF], uint32_t& n) {
            uint32_t q, off, cnt; uint64_t lo_q;
            tk.locate(g, q, off, cnt, lo_q);
            const uint32_t* rp = reinterpret_cast<const uint32_t*>(sets[q].recs);
            const uint8_t* ap = sets[q].aux;
            n = min(cnt - off, GRP);
#pragma unroll
            for (int qq = 0; qq < PF; ++qq) {
                const uint64_t j = lo_q + min(off + (uint32_t)qq * 64u + lane, cnt - 1u);
                rec[qq] = ld_global(rp + j);
                aux[qq] = TIGHT ? 0u : ld_global(ap + j);
            }
        };
        KQ_STAMP(0);                                                    // kernel start -> ticket build + reduction done
        uint32_t gA = wave, gB = wave + NW;                              // the first two tickets of a wave are its own; the rest come from s_grp
        fetch(gA, recA, auxA, nA);
        if (DEPTH == 2) fetch(gB, recB, auxB, nB);
        const uint32_t bucket = (uint32_t)r / rps;
        const uint32_t start_r = t.rstart[r];
        const uint32_t top_base = (bucket << (32 - NARROW_CBITS)) - start_r;       // (top 32 hash bits of a 5-byte record) - rstart[r] = top_base + (u32 >> 8)
        const uint32_t start_lo = start_r << 10;                                   // key + start_lo = the low 32 bits of (top 32 hash bits | the 10 below)
        if (tid < HC_LDS) {
            s_hckey[tid] = EMPTY_KEY;
#pragma unroll
            for (int e = 0; e < 8; ++e) s_hccnt[tid][e] = 0;
        }
        if (table_is_empty) {
            for (int i = tid; i < (int)REGION_SLOTS; i += Q4_THREADS) { s_key[i] = N32_EMPTY; s_cnt[i] = 0; s_e[i] = 0; }
        } else {
            ulonglong2 v[REGION_SLOTS / Q4_THREADS];
#pragma unroll
            for (int j = 0; j < (int)(REGION_SLOTS / Q4_THREADS); ++j) v[j] = gimg[tid + j * Q4_THREADS];
#pragma unroll
            for (int j = 0; j < (int)(REGION_SLOTS / Q4_THREADS); ++j) {
                const uint64_t w0 = v[j].x;                             // rem56 | cov8 << 56, rem = hash >> 8
                uint32_t key = N32_EMPTY, cnt = 0;
                if (w0) {
                    key = (((uint32_t)(w0 >> 24) - start_r) << 10) | ((uint32_t)(w0 >> 14) & 1023u);
                    const uint32_t c = (uint32_t)(w0 >> COV_SHIFT);
                    cnt = c == COV8_TOMB ? (N32_TOMB | LOW_TIER_MAX) : c;
                }
                s_key[tid + j * Q4_THREADS] = key;
                s_cnt[tid + j * Q4_THREADS] = cnt;
                s_e[tid + j * Q4_THREADS] = v[j].y;
            }
        }
        if (tid == 0) { s_new = 0; s_kmers = 0; s_grp = DEPTH * (Q4_THREADS / 64); }
        __syncthreads();
        KQ_STAMP(1);                                                    // first fetch issued, image init / load, barrier
        uint32_t n_new = 0, n_ok = 0;
        auto hash_of = [&](uint32_t key) -> uint64_t { return ((uint64_t)(start_r + (key >> 10)) << 32) | ((uint64_t)(key & 1023u) << 22); };
        auto add_wide = [&](uint32_t key31, uint64_t pack) {            // an edge of a k-mer beyond 254 instances: the region's LDS high-copy sums
            const uint64_t h = hash_of(key31);
            const uint64_t key = key_of_hash(h, t.k);
            uint32_t e[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) e[w] = (uint32_t)(pack >> (8 * w)) & 1u;
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
                hc_add(t, h, 0, 0, e);
            }
        };
        // one instance of the k-mer in `slot`
        auto apply = [&](uint32_t slot, uint32_t key, uint64_t pack) {
            ++n_ok;
            const uint32_t old = atomicAdd(&s_cnt[slot], 1u);
            if (pack) {
                if (old < LOW_TIER_MAX) atomicAdd((unsigned long long*)&s_e[slot], (unsigned long long)pack);
                else add_wide(key, pack);
            }
        };
        // the four keys of the quad at `pos` (a multiple of four): the key's slot, or the first free one (claimed), or NONE
        // when the quad is full of other keys (full = true) / the claim went to another key (full = false: look again)
        auto probe_quad = [&](uint32_t pos, uint32_t key, bool& full) -> uint32_t {
            const uint64_t q01 = __hip_atomic_load(&s_key2[pos >> 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint64_t q23 = __hip_atomic_load(&s_key2[(pos >> 1) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint32_t k0 = (uint32_t)q01, k1 = (uint32_t)(q01 >> 32), k2 = (uint32_t)q23, k3 = (uint32_t)(q23 >> 32);
            uint32_t off = k1 == key ? 1u : 4u;
            off = k2 == key ? 2u : off;
            off = k3 == key ? 3u : off;
            off = k0 == key ? 0u : off;
            full = false;
            if (off < 4u) return pos + off;
            uint32_t eo = k3 == N32_EMPTY ? 3u : 4u;
            eo = k2 == N32_EMPTY ? 2u : eo;
            eo = k1 == N32_EMPTY ? 1u : eo;
            eo = k0 == N32_EMPTY ? 0u : eo;
            if (eo == 4u) { full = true; return NONE; }
            const uint32_t got = atomicCAS(&s_key[pos + eo], N32_EMPTY, key);
            if (got == N32_EMPTY) { ++n_new; return pos + eo; }
            return got == key ? pos + eo : NONE;
        };
        // home quad of a key: the low 32 bits of (top 32 hash bits | the 10 below) are key + start_lo; k < 11 (a key space
        // of < 2^22 k-mers on a table this large: tests only) has fewer than 11 offset bits below them and takes the generic form
        auto home_of = [&](uint32_t key) -> uint32_t {
            return k >= 11 ? ((key + start_lo) >> off_shift) & (REGION_SLOTS - 4) : hash_offset(hash_of(key), k);
        };
        uint32_t qn = 0;                                                // entries in this wave's queue (wave-uniform)
        // the queued records, 64 at a time, through the probe loop
        auto drain = [&]() {
            while (qn) {
                const uint32_t n = qn < 64u ? qn : 64u;
                qn -= n;
                const bool act = lane < n;
                const uint64_t ent = s_q[wave][qn + (act ? lane : 0u)];
                const uint32_t key = (uint32_t)ent;
                uint32_t pos = home_of(key);
                uint32_t slot = NONE, probes = 0;
                bool looking = act;
                while (looking) {
                    bool full;
                    slot = probe_quad(pos, key, full);
                    if (slot != NONE) looking = false;
                    else if (full) {
                        pos = (pos + 4) & (REGION_SLOTS - 1);
                        if ((probes += 4) >= REGION_SLOTS) { atomicOr(&t.st->err_table_full, 1u); looking = false; }
                    }
                }
                if (act && slot != NONE) apply(slot, key, s_lut[(uint32_t)(ent >> 32) & 63u]);
            }
        };
        auto grab = [&]() -> uint32_t {
            uint32_t g = 0;
            if (lane == 0) g = atomicAdd(&s_grp, 1u);
            return __builtin_amdgcn_readfirstlane(g);
        };
        // the PF x 64 records of one ticket
        auto process = [&](const uint32_t (&cur_rec)[PF], const uint32_t (&cur_aux)[PF], uint32_t n_cur) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                if ((uint32_t)q * 64u >= n_cur) break;                   // (uniform) a ticket is one piece of one set: often shorter than PF x 64 records
                if (qn > QCAP - 64u) drain();                            // (uniform) room for 64 more queue entries
                const bool active = (uint32_t)q * 64u + lane < n_cur;
                const uint32_t m = cur_rec[q], aux = cur_aux[q];
                const uint32_t key = TIGHT ? m >> 6 : ((top_base + (m >> 8)) << 10) | ((m & 0xFFu) << 2) | (aux & 3u);
                const uint32_t idx6 = TIGHT ? m & 63u : (aux >> 2) & 63u;
                const uint32_t home = home_of(key);
                const uint64_t pack = s_lut[idx6];
                bool full;
                uint32_t slot = NONE;
                if (active) slot = probe_quad(home, key, full);
                if (active && slot != NONE) apply(slot, key, pack);
                const bool queued = active && slot == NONE;
                const uint64_t qm = __ballot(queued);
                if (qm) {                                                // (uniform)
                    if (queued) s_q[wave][qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u))] = (uint64_t)key | ((uint64_t)idx6 << 32);
                    qn += (uint32_t)__popcll((unsigned long long)qm);
                }
            }
        };
        // tickets are handed out in increasing order and a wave looks at A, B, A, B ...: the first ticket past the end ends its walk
        for (;;) {                                                      // wave-uniform
            if (gA >= tk.n_grp) break;
            {
                uint32_t cur_rec[PF], cur_aux[PF];
#pragma unroll
                for (int q = 0; q < PF; ++q) { cur_rec[q] = recA[q]; cur_aux[q] = auxA[q]; }
                const uint32_t n_cur = nA;
                gA = grab();
                fetch(gA, recA, auxA, nA);
                process(cur_rec, cur_aux, n_cur);
            }
            if (DEPTH == 2) {
                if (gB >= tk.n_grp) break;
                uint32_t cur_rec[PF], cur_aux[PF];
#pragma unroll
                for (int q = 0; q < PF; ++q) { cur_rec[q] = recB[q]; cur_aux[q] = auxB[q]; }
                const uint32_t n_cur = nB;
                gB = grab();
                fetch(gB, recB, auxB, nB);
                process(cur_rec, cur_aux, n_cur);
            }
        }
        drain();
        KQ_STAMP(2);                                                    // record walk (wave 0)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { n_new += __shfl_down(n_new, o, 64); n_ok += __shfl_down(n_ok, o, 64); }
        if (lane == 0) { if (n_new) atomicAdd(&s_new, n_new); if (n_ok) atomicAdd(&s_kmers, n_ok); }
        __syncthreads();
        if (tid < HC_LDS && s_hckey[tid] != EMPTY_KEY) {            // flush the region's high-copy sums: one entry per k-mer
            HcSlot* hs = hc_upsert(t, s_hckey[tid]);
            if (!hs) atomicOr(&t.st->err_hc_full, 1u);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (s_hccnt[tid][e]) atomicAdd((unsigned long long*)&hs->cnt[e], (unsigned long long)s_hccnt[tid][e]);
            }
        }
#pragma unroll
        for (int j = 0; j < (int)(REGION_SLOTS / Q4_THREADS); ++j) {
            const int i = tid + j * Q4_THREADS;
            const uint32_t key = s_key[i];
            ulonglong2 o = make_ulonglong2(0ull, 0ull);
            if (key != N32_EMPTY) {
                const uint32_t cw = s_cnt[i], c = cw & ~N32_TOMB;
                uint64_t cov8 = (cw & N32_TOMB) ? COV8_TOMB : c;
                if (c > LOW_TIER_MAX) { cov8 = COV8_TOMB; hc_add(t, hash_of(key), c - LOW_TIER_MAX, 0, nullptr); }
                o = make_ulonglong2((((uint64_t)(start_r + (key >> 10))) << 24) | ((uint64_t)(key & 1023u) << 14) | (cov8 << COV_SHIFT), s_e[i]);
            }
            gimg[i] = o;
        }
        KQ_STAMP(3);                                                    // barrier (slowest wave), high-copy flush, image store (waited for)
        if (tid == 0) {
            if (s_new) atomicAdd(&t.st->slots_used, (unsigned long long)s_new);
            if (s_kmers) atomicAdd(&t.st->kmers_added, (unsigned long long)s_kmers);
        }
        __syncthreads();
    }
}


// K3 on partitioned records (counters only): the assembly's k-mers go through the same P1 / level split as
// reads, then one workgroup per table region stages the region image in LDS (read-only) and evaluates its
// records there -- sequential HBM traffic instead of one random 64-byte sector per k-mer.  A record holds
// everything evaluateSegment needs (src/kreeq.cpp:145-216): the hash (-> key) and the indices of the
// fw / bw edge the assembly continues with (edge_idx6: the strand mapping of :178-210 is already applied).
template <int FMT>
__global__ __launch_bounds__(P3_THREADS, 8) void k_lookup_regions(TableView t, const uint64_t* __restrict__ recs, const uint8_t* __restrict__ recs_aux,
                                                                   const unsigned long long* __restrict__ region_base, uint32_t narrow_rps,
                                                                   uint32_t cov_cutoff, unsigned long long* __restrict__ counters) {
    constexpr bool WIDE = FMT == FMT_WIDE, NARROW = FMT == FMT_NARROW, TOP8 = FMT == FMT_TOP8, HAS_AUX = WIDE || NARROW;
    const uint32_t* recs32 = reinterpret_cast<const uint32_t*>(recs);
    __shared__ uint64_t s_img[REGION_SLOTS * 2];                        // the region as it lies in HBM (read-only here): 32 KiB, four workgroups per CU
    const int tid = threadIdx.x;
    uint32_t missing = 0, total = 0, edge_missing = 0;
    for (uint64_t r = t.reg_lo + blockIdx.x; r < t.reg_hi; r += gridDim.x) {       // (a window evaluates the k-mers of its own buckets only)
        const uint64_t lo = region_base[r], hi = region_base[r + 1];
        if (lo == hi) continue;                                         // block-uniform
        const uint4* gimg = reinterpret_cast<const uint4*>(t.slots + (r << REGION_SHIFT));
        uint4* limg = reinterpret_cast<uint4*>(s_img);
        for (int i = tid; i < (int)REGION_SLOTS; i += P3_THREADS) limg[i] = gimg[i];
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
                const uint64_t rem = slot_rem(h, t.k);
                const uint32_t idx6 = NARROW ? aux[q] >> 2 : TOP8 ? (uint32_t)rec[q] & 63u : WIDE ? aux[q] : (uint32_t)(rec[q] >> REC_EDGE_SHIFT) & 63u;
                const uint32_t off = hash_offset(h, t.k);
                uint32_t found = REGION_SLOTS * 2;
                for (uint32_t pb = 0; pb < REGION_SLOTS && found == REGION_SLOTS * 2; pb += 4) {     // :153, four slots per LDS round trip
                    uint32_t w[4];
                    uint64_t c[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { w[j] = 2u * ((off + pb + j) & (REGION_SLOTS - 1)); c[j] = s_img[w[j]]; }
                    bool stop = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (!stop && c[j] != 0 && (c[j] & REM_MASK) == rem) { found = w[j]; stop = true; }
                        if (!stop && c[j] == 0) stop = true;
                    }
                    if (stop) break;
                }
                uint64_t cov = 0, e8 = 0;
                const HcSlot* hs = nullptr;
                if (found != REGION_SLOTS * 2) {
                    e8 = s_img[found + 1]; cov = s_img[found] >> COV_SHIFT;
                    if (cov == COV8_TOMB) {                                                          // :156-166 (32-bit tier)
                        hs = hc_find(t, key_of_hash(h, t.k));
                        if (hs) cov = LOW_TIER_MAX + hs->cov_hi;
                    }
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
        if (table_add(t, table_hash(keys[i], t.k), 1, edge_byte_to_pack(edges[i]), nullptr, &ins)) ++n_ok;
        n_new += ins;
    }
    uint64_t a = block_sum(n_new), b = block_sum(n_ok);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&t.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&t.st->kmers_added, (unsigned long long)b);
    }
}

// import / union: add logical entries (kunion + mergeSubMaps, src/graph-builder.cpp:297-432)
__device__ __forceinline__ void add_logical(const TableView& t, uint64_t h, const uint32_t* e, uint32_t cov,
                                            uint32_t& n_new, uint64_t& n_cov) {
    uint64_t pack = 0;
    bool fits = cov <= LOW_TIER_MAX;
#pragma unroll
    for (int i = 0; i < 8; ++i) { if (e[i] > LOW_TIER_MAX) fits = false; pack |= (uint64_t)(e[i] & 0xFF) << (8 * i); }
    uint32_t ins = 0;
    // when the entry itself is beyond the low tier, table_add routes all of it to the wide counters
    if (table_add(t, h, cov, fits ? pack : 0, fits ? nullptr : e, &ins)) n_cov += cov;
    n_new += ins;
}
__global__ __launch_bounds__(256) void k_import(TableView t, const kq_entry* __restrict__ in, uint64_t n) {
    uint32_t n_new = 0;
    uint64_t n_cov = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t e[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) { e[w] = in[i].fw[w]; e[4 + w] = in[i].bw[w]; }
        add_logical(t, table_hash(in[i].key, t.k), e, in[i].cov, n_new, n_cov);
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
    const uint64_t s0 = src.reg_lo << REGION_SHIFT, n = (src.reg_hi - src.reg_lo) << REGION_SHIFT;     // the allocated slots
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot* s = src.slots + s0 + i;
        const uint64_t w0 = s->w0;
        if (w0 == 0) continue;
        const uint64_t h = slot_hash_at(src, s, w0);
        Logical L = logical_of(src, h, w0, s->e8);
        add_logical(dst, h, L.e, L.cov, n_new, n_cov);
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
    constexpr int SPT = REGION_SLOTS / P3_THREADS;
    for (uint64_t r = dst.reg_lo + blockIdx.x; r < dst.reg_hi; r += gridDim.x) {
        ulonglong2* gimg = reinterpret_cast<ulonglong2*>(dst.slots + (r << REGION_SHIFT));
        if (dst_is_empty) {
            for (int i = tid; i < (int)(REGION_SLOTS * 3); i += P3_THREADS) s_img[i] = (i % 3 == 0) ? EMPTY_KEY : 0ull;
        } else {
            ulonglong2 v[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) v[j] = gimg[tid + j * P3_THREADS];
#pragma unroll
            for (int j = 0; j < SPT; ++j) img_load(s_img, tid + j * P3_THREADS, v[j].x, v[j].y);
        }
        __syncthreads();
        // hash interval of dst region r (top 32 bits): [ceil(r 2^32 / R), ceil((r+1) 2^32 / R) - 1]
        const uint64_t R = dst.n_regions;
        const uint32_t h_lo = (uint32_t)(((r << 32) + R - 1) / R), h_hi = (uint32_t)((((r + 1) << 32) + R - 1) / R - 1);
        uint64_t s_lo = __umulhi(h_lo, (uint32_t)src.n_regions), s_hi = __umulhi(h_hi, (uint32_t)src.n_regions);
        if (s_lo < src.reg_lo) s_lo = src.reg_lo;                      // a source window holds nothing outside its regions
        if (s_hi >= src.reg_hi) s_hi = src.reg_hi - 1;                  // (reg_hi > 0; s_hi < s_lo: no iteration)
        for (uint64_t sr = s_lo; sr <= s_hi && src.reg_hi > src.reg_lo; ++sr) {
            const ulonglong2* sslots = reinterpret_cast<const ulonglong2*>(src.slots + (sr << REGION_SHIFT));
            // the whole source region in flight at once (one 16-byte load per slot, unconditional)
            ulonglong2 sv[SPT];
#pragma unroll
            for (int j = 0; j < SPT; ++j) sv[j] = sslots[tid + j * P3_THREADS];
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                if (sv[j].x == 0) continue;
                const uint64_t h = slot_hash(src, sv[j].x & REM_MASK, sr);
                if (hash_region(h, R) != r) continue;
                const uint64_t rem = slot_rem(h, dst.k);
                const Logical L = logical_of(src, h, sv[j].x, sv[j].y);
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
                    for (int q = 0; q < 4; ++q) { ws[q] = 3u * ((off + pb + q) & (REGION_SLOTS - 1)); c[q] = __hip_atomic_load(&s_img[ws[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (w != REGION_SLOTS * 3) break;
                        uint64_t cur = c[q];
                        if (cur == EMPTY_KEY) {
                            cur = atomicCAS((unsigned long long*)&s_img[ws[q]], (unsigned long long)EMPTY_KEY, (unsigned long long)rem);
                            if (cur == EMPTY_KEY) { ++n_new; w = ws[q]; break; }
                        }
                        if (cur == rem) w = ws[q];
                    }
                }
                if (w == REGION_SLOTS * 3) { atomicOr(&dst.st->err_table_full, 1u); continue; }
                n_cov += L.cov;
                const uint64_t old = atomicAdd((unsigned long long*)&s_img[w + 2], (unsigned long long)L.cov);
                if (fits && old + L.cov <= LOW_TIER_MAX) {
                    if (pack) atomicAdd((unsigned long long*)&s_img[w + 1], (unsigned long long)pack);
                } else if (any) {                                       // beyond the u8 lanes: the wide counters (rare); img_store adds the cov part
                    hc_add(dst, h, 0, 0, L.e);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; ++j) gimg[tid + j * P3_THREADS] = img_store(dst, s_img, tid + j * P3_THREADS, r);
        __syncthreads();
    }
    const uint64_t a = block_sum(n_new), b = block_sum(n_cov);
    if (threadIdx.x == 0) {
        if (a) atomicAdd(&dst.st->slots_used, (unsigned long long)a);
        if (b) atomicAdd(&dst.st->kmers_added, (unsigned long long)b);
    }
}
// rehash into a bigger table (growth): exact move of physical state
__global__ __launch_bounds__(256) void k_rehash(TableView dst, TableView old) {
    const uint64_t s0 = old.reg_lo << REGION_SHIFT, n_old = (old.reg_hi - old.reg_lo) << REGION_SHIFT;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_old; j += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = s0 + j;
        const Slot s = old.slots[i];
        if (s.w0 == 0) continue;
        const uint64_t h = slot_hash(old, s.w0 & REM_MASK, i >> REGION_SHIFT);
        Slot* base = region_of(dst, h);
        const uint64_t w0 = slot_rem(h, dst.k) | (s.w0 & ~REM_MASK);      // the count moves as it is (tombstone included)
        const uint32_t off = hash_offset(h, dst.k);
        Slot* d = nullptr;
        for (uint32_t probe = 0; probe < REGION_SLOTS && !d; ++probe) {
            Slot* c = base + ((off + probe) & (REGION_SLOTS - 1));
            if (ld_relaxed(&c->w0) == 0 && atomicCAS((unsigned long long*)&c->w0, 0ull, (unsigned long long)w0) == 0ull) d = c;   // keys are unique: claim = done
        }
        if (!d) { atomicOr(&dst.st->err_table_full, 1u); continue; }
        d->e8 = s.e8;
    }
}
__global__ __launch_bounds__(256) void k_rehash_hc(TableView dst, const HcSlot* __restrict__ old, uint64_t n_old) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_old; i += (uint64_t)gridDim.x * blockDim.x) {
        if (old[i].key == EMPTY_KEY) continue;
        HcSlot* d = hc_upsert(dst, old[i].key);
        if (!d) { atomicOr(&dst.st->err_hc_full, 1u); continue; }
        d->cov_hi = old[i].cov_hi;
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
    const uint64_t s0 = t.reg_lo << REGION_SHIFT, n = (t.reg_hi - t.reg_lo) << REGION_SHIFT;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const Slot* s = t.slots + s0 + i;
        if (s->w0 == 0) continue;
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

// export: logical entries of maps [lo, hi) (gfalibs dumpMap's input).  Tiles of 1024 slots; a workgroup reserves the
// output range of a tile with ONE atomic (block-wide prefix of the per-thread counts), not one per entry.
__global__ __launch_bounds__(256) void k_export(TableView t, uint32_t map_count, uint32_t lo, uint32_t hi,
                                                 kq_entry* out, uint64_t cap, unsigned long long* n_out) {
    __shared__ uint32_t s_wave[4];
    __shared__ unsigned long long s_base;
    const int tid = threadIdx.x;
    const uint64_t tile0 = (t.reg_lo << REGION_SHIFT) / 1024, tile1 = (t.reg_hi << REGION_SHIFT) / 1024;      // whole regions: multiples of 1024 slots
    for (uint64_t tile = tile0 + blockIdx.x; tile < tile1; tile += gridDim.x) {
        uint64_t w0[4], e8[4], key[4];
        uint32_t take = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const Slot* s = t.slots + tile * 1024 + j * 256 + tid; w0[j] = s->w0; e8[j] = s->e8; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            key[j] = 0;
            if (w0[j] == 0) continue;
            key[j] = key_of_hash(slot_hash(t, w0[j] & REM_MASK, (tile * 1024 + j * 256 + tid) >> REGION_SHIFT), t.k);
            const uint32_t m = (uint32_t)(key[j] % map_count);
            if (m >= lo && m < hi) take |= 1u << j;
        }
        const uint32_t mine = __popc(take);
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
        if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < (tid >> 6)) wave_base += s_wave[w]; total += s_wave[w]; }
        if (tid == 0) s_base = total ? atomicAdd(n_out, (unsigned long long)total) : 0ull;
        __syncthreads();
        uint64_t o = s_base + wave_base + (incl - mine);
        if (out) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((take >> j) & 1u)) continue;
                if (o < cap) {
                    const uint64_t idx = tile * 1024 + j * 256 + tid;
                    const Logical L = logical_of(t, slot_hash(t, w0[j] & REM_MASK, idx >> REGION_SHIFT), w0[j], e8[j]);
                    kq_entry e;
                    e.key = key[j];
#pragma unroll
                    for (int w = 0; w < 4; ++w) { e.fw[w] = L.e[w]; e.bw[w] = L.e[4 + w]; }
                    e.cov = L.cov;
                    e.hc = L.cov > LOW_TIER_MAX;          // in maps32 iff total cov >= 255
                    out[o] = e;
                }
                ++o;
            }
        }
        __syncthreads();                                  // s_wave / s_base are reused by the next tile
    }
}

// K3: evaluateSegment (src/kreeq.cpp:143-219) over a whole sequence (segments = ACGT runs).
// One random 16-B probe per k-mer: measured at 27.7 G lookups/s this is the part's random 64-B
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
            const uint64_t h = table_hash(key, (uint32_t)k);
            if (!table_owns(t, h)) return;                                  // a window evaluates the k-mers of its own buckets (like :150 for map ranges)
            const Slot* s = table_find(t, h);                              // :153
            if (s) {
                L = logical_of(t, h, s->w0, s->e8);                        // :156-166 (8-bit or 32-bit tier)
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

// map->find(key) for a batch of canonical keys (candidate-error search, src/variants.cpp:118-131, :203-206)
__global__ __launch_bounds__(256) void k_lookup_keys(TableView t, const uint64_t* __restrict__ keys, uint64_t n, kq_entry* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = keys[i];
        kq_entry e;
        e.key = key; e.cov = 0; e.hc = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { e.fw[w] = 0; e.bw[w] = 0; }
        const uint64_t space_ok = t.k == 32 ? 1 : (key >> (2 * t.k)) == 0;
        if (space_ok) {
            const uint64_t h = table_hash(key, t.k);
            const Slot* s = table_find(t, h);
            if (s) {
                const Logical L = logical_of(t, h, s->w0, s->e8);
#pragma unroll
                for (int w = 0; w < 4; ++w) { e.fw[w] = L.e[w]; e.bw[w] = L.e[4 + w]; }
                e.cov = L.cov; e.hc = L.cov > LOW_TIER_MAX;
            }
        }
        out[i] = e;
    }
}
// pre-filter of the candidate-error search: see kq_branch_scan (include/kreeq_amd.h)
__global__ __launch_bounds__(TILE_THREADS) void k_branch_scan(TableView t, const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k,
                                                               uint32_t cov_cutoff, uint8_t* __restrict__ flags) {
    scan_tiles(ab, lead, len, k, [&](uint64_t pos, uint64_t fw, uint64_t rv, uint32_t, uint32_t next) {
        const bool is_fw = fw < rv;
        const uint64_t h = table_hash(is_fw ? fw : rv, (uint32_t)k);
        const Slot* s = table_find(t, h);
        uint8_t f = 0;
        if (s) {
            f = 1;
            const Logical L = logical_of(t, h, s->w0, s->e8);
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                // direction = isSourceFw; `direction ? fw[i] : bw[i] > covCutOff` (src/variants.cpp:237); the candidate
                // continues the sequence with base i (fw) or 3 - i (bw of the reverse strand): it is the reference
                // path (:241) iff that base is the one the sequence has next
                const bool edge = is_fw ? (L.e[i] != 0) : (L.e[4 + i] > cov_cutoff);
                const uint32_t base = is_fw ? i : 3u - i;
                if (edge && base != next) f |= 2;
            }
        }
        flags[pos] = f;
    });
}

// export ordering: keys / indices of the compacted entries, then the entries gathered in key order
__global__ __launch_bounds__(256) void k_entry_keys(const kq_entry* __restrict__ e, uint64_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) { keys[i] = e[i].key; idx[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(256) void k_entry_gather(const kq_entry* __restrict__ in, const uint32_t* __restrict__ idx, uint64_t n, kq_entry* __restrict__ out) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[idx[i]];
}


// kq_partition.h -- partitioned count path: records are routed to the table region that owns their
// key, then one workgroup per region applies them inside LDS.  No global atomics per k-mer.
//
//   P1  k_p1_hist / k_p1_scatter   bases  -> 8-B records grouped by COARSE bucket (= region >> g_shift)
//   P2  k_p2_hist / k_p2_scatter   coarse -> records grouped by REGION
//   P3  k_count_regions            per region: 96 KiB slot image in LDS, ds_* atomics, stream back
//
// Both splits are workgroup-level multisplits: a tile of <= 4096 records is ranked with LDS
// atomics, staged bucket-contiguous in LDS and copied out as coalesced runs at the workgroup's
// private cursors (count matrix + exclusive scan from a counting pass): no global atomics.
//
// Record (k <= 28): key in bits 0..2k-1; bits 56..58 = index of the fw edge (0..3, 7 = none),
// bits 59..61 = index of the bw edge (0..3, 7 = none): a k-mer instance has at most one of each.
// WIDE records (k = 29..32, or records handed in by a caller): the u64 is the whole key and the
// edge information travels in a parallel u8 array that is split in lockstep ("aux"): either the
// same 6-bit index pair (AUX_IDX6) or the reference edge byte of include/kreeq.h:6-18 (AUX_EDGE_BYTE).
#pragma once
#include "kq_device.h"

namespace kq {

constexpr int MS_THREADS = 256;
constexpr int MS_ITEMS = 16;
constexpr int MS_TILE = MS_THREADS * MS_ITEMS;   // 4096 records per multisplit round
constexpr int NB_MAX = 2048;                     // bins of one split incl. the discard bin: fan-out <= NB_MAX - 1
constexpr int SEG_MAX = 65536;                   // segments of one split level (256 hash-prefix buckets x up to 256 sub-buckets)
constexpr uint32_t SUB_BITS_MAX = 8;
constexpr int REC_EDGE_SHIFT = 56;
constexpr int PART_MAX_K = 28;                   // packed 8-byte records up to here, WIDE above
constexpr int AUX_IDX6 = 0, AUX_EDGE_BYTE = 1, AUX_TIGHT = 2;    // AUX_TIGHT: FMT_TIGHT sets (no lockstep array at all)

struct PartCfg {
    uint64_t n_regions;   // R
    uint32_t g_shift;     // coarse bucket = region >> g_shift
    uint32_t n_coarse;    // bins of P1: ceil(R / 2^g_shift) < NB_MAX (mode 0) or n_parts (mode 1)
    uint32_t mode;        // 0: bin = coarse bucket of the key's table region; 1: bin = owner part (multi-GPU staging)
    uint32_t map_count;   // mode 1: gfalibs mapCount
    uint32_t map_mask;    // map_count - 1 when it is a power of two, else 0
    uint32_t filt_lo, filt_hi;   // count only k-mers whose map index key % map_count lies in [filt_lo, filt_hi)
    uint32_t raw_out;     // 1: WIDE records carry the raw key (kq_emit_partitioned_dev), else its table hash
    uint32_t narrow;      // 1: bin = top NARROW_CBITS hash bits (n_regions is a multiple of 2^NARROW_CBITS), FMT_NARROW records
    uint32_t owner_sub;   // mode 1: each owner part is cut into 2^owner_sub sub-bins by lane id (a multisplit with a handful of
                          // bins serialises its LDS rank atomics on a few addresses); the parts stay contiguous.  The histogram
                          // and the scatter pass give a k-mer the same lane, hence the same sub-bin
    uint32_t sub_bits;    // narrow: > 0 = a middle level cuts each bucket into 2^sub_bits sub-buckets first (very large tables)
    uint32_t win_lo, win_hi;   // bin mode 6: the table is a window of the hash-prefix buckets [win_lo, win_hi): k-mers of other buckets are dropped in P1
    uint32_t n_rng;       // k_p1_hist bin mode 5 (KQ_OPT_COUNT_MAP_PASSES): count for all n_rng equal map ranges at once, bin = range * 256 + bucket
};

// One record set of k_count_regions: records sorted by table region (format FMT_*; `aux` = lockstep u8 array or null)
// and base[0..n_regions] = first record of every region.
constexpr int P3_MAX_SETS = 64;
struct P3Set { const uint64_t* recs; const uint8_t* aux; const unsigned long long* base; uint64_t n_max; };

// One level of the record split.  Input records are grouped in n_seg segments (seg_off[0..n_seg]);
// a record of segment b with table region r goes to bin (r >> out_shift) - (b << (seg_shift - out_shift)).
//   flat -> coarse buckets : n_seg = 1, seg_shift = 32 (b == 0), out_shift = g_shift, nb = n_coarse
//   coarse -> regions      : n_seg = n_coarse, seg_shift = g_shift, out_shift = 0, nb = 2^g_shift
// Output group index = b * nb + bin (== the region id at the last level).
struct LevelCfg {
    uint64_t n_regions;
    uint32_t n_seg, nb, seg_shift, out_shift;
    uint32_t in_raw;        // 1: the input records hold raw keys (kq_insert_records): this level mixes them
    uint32_t k;             // k-mer length (width of the table's mix), needed when in_raw
    uint32_t narrow;        // 1: FMT_NARROW records, segment b = hash-prefix bucket b owning regions [b * nb, (b + 1) * nb)
    uint32_t top8;          // 1: one segment of packed records, bin = top NARROW_CBITS hash bits; the scatter writes narrow records
    // narrow levels: segment b = (bucket b >> nr_shift, sub-bucket b & (2^nr_shift - 1)) starts at region
    // bucket * nr_rps + sub * nr_sub; bin = (region - that) / nr_div (nr_inv = ceil(2^32 / nr_div))
    uint32_t nr_shift, nr_rps, nr_sub, nr_div, nr_inv;
    uint32_t nr_mid, nr_fshift, nr_fmask, nr_f24;   // narrow_bin: middle level, 24 - sub_bits, the bits of the position inside a sub-bucket, product fits 32 bits
    // multi-GPU exchange of narrow records (receive side): spb > 1 = the input has spb segments per logical segment (one
    // run per peer rank): input segment s belongs to logical segment s / spb, whose units and output groups they share
    uint32_t spb;
    uint32_t rep_shift;     // rank replication of the multisplit (block_multisplit's rs), set by the host from nb
    const uint32_t* rstart = nullptr; // non-null: this (last, narrow) level writes FMT_TIGHT records relative to rstart[region]
};
__device__ __forceinline__ uint32_t level_bin(const LevelCfg& lv, uint32_t b, uint64_t region) {
    return (uint32_t)(region >> lv.out_shift) - (lv.seg_shift >= 32 ? 0u : (b << (lv.seg_shift - lv.out_shift)));
}

// fw/bw edge indices as src/graph-builder.cpp:98-110 assigns them
__device__ __forceinline__ uint32_t edge_idx6(bool is_fw, uint32_t prev, uint32_t next) {
    uint32_t f, b;                                  // (written as selects of small values: the branch-free bit form of round 3 cost k_p1_scatter 50 more spilled registers)
    if (is_fw) { f = next < 4 ? next : 7u; b = prev < 4 ? prev : 7u; }
    else       { f = prev < 4 ? 3u - prev : 7u; b = next < 4 ? 3u - next : 7u; }
    return f | (b << 3);
}
// the same without selects, for records whose readers only ask "index < 4" (the region kernels' lookup table, k_lookup_regions):
// "no edge" comes out as 4 (forward strand) or 7 (reverse: 4 ^ 3) instead of always 7.  prev / next in 0..4.
__device__ __forceinline__ uint32_t edge_idx6_any(bool is_fw, uint32_t prev, uint32_t next) {
    const uint32_t f = is_fw ? next : prev, b = is_fw ? prev : next, m = is_fw ? 0u : 0x1Bu;      // (three selects: written as one, the compiler branches on the strand)
    return (f | (b << 3)) ^ m;
}
__device__ __forceinline__ uint64_t rec_pack(uint64_t key, bool is_fw, uint32_t prev, uint32_t next) {
    return key | ((uint64_t)edge_idx6(is_fw, prev, next) << REC_EDGE_SHIFT);
}
__device__ __forceinline__ uint64_t idx6_to_pack(uint32_t v) {
    const uint32_t f = v & 7u, b = (v >> 3) & 7u;
    return (f < 4 ? 1ull << (8 * f) : 0ull) | (b < 4 ? 1ull << (8 * (4 + b)) : 0ull);
}
__device__ __forceinline__ uint32_t idx6_to_edge_byte(uint32_t v) {
    const uint32_t f = v & 7u, b = (v >> 3) & 7u;
    return (f < 4 ? 1u << (7 - f) : 0u) | (b < 4 ? 1u << (7 - (4 + b)) : 0u);
}
__device__ __forceinline__ uint64_t rec_key(uint64_t rec) { return rec & ((1ull << REC_EDGE_SHIFT) - 1); }
// packed u8x8 increment (byte e = edge e) of a record
__device__ __forceinline__ uint64_t rec_edge_pack(uint64_t rec) {
    const uint32_t f = (uint32_t)(rec >> REC_EDGE_SHIFT) & 7u, b = (uint32_t)(rec >> (REC_EDGE_SHIFT + 3)) & 7u;
    return (f < 4 ? 1ull << (8 * f) : 0ull) | (b < 4 ? 1ull << (8 * (4 + b)) : 0ull);
}
// Records carry the table hash of their k-mer (kq_device.h: an invertible mix), not the key: packed
// records its 56 significant bits under the two edge indices, WIDE records the left-aligned 64 bits.
template <bool WIDE>
__device__ __forceinline__ uint64_t rec_hash(uint64_t rec) { return WIDE ? rec : rec << 8; }     // the shift drops the edge bits
__device__ __forceinline__ uint64_t rec_pack_hash(uint64_t h, bool is_fw, uint32_t prev, uint32_t next) {
    return (h >> 8) | ((uint64_t)edge_idx6(is_fw, prev, next) << REC_EDGE_SHIFT);
}

// Record formats of the partitioned path:
//   FMT_PACK8  (k <= 28)  one u64: top 56 bits of the table hash | two 3-bit edge indices at bits 56..61
//   FMT_WIDE   (any k)    u64 table hash (or raw key, see LevelCfg::in_raw) + lockstep u8 (AUX_IDX6 / AUX_EDGE_BYTE)
//   FMT_NARROW (k <= 21, two-level split with 256 top-bit buckets) u32 + lockstep u8: after P1 the top 8
//              hash bits are the bucket a record lies in, the remaining <= 34 bits are the u32 (upper 32) and the
//              low 2 bits of the u8, whose upper 6 bits are the edge indices.  5 bytes instead of 8 through
//              three of the four record passes, and the level histogram reads only the u32 array.
constexpr int FMT_PACK8 = 0, FMT_WIDE = 1, FMT_NARROW = 2;
constexpr int FMT_PACK8_TO_NARROW = 3;      // k_lv_scatter only: packed records in, narrow records out (bins = hash prefix)
constexpr uint32_t NARROW_CBITS = 8, NARROW_MAX_K = 21;
// FMT_TOP8 (k = 29..32 on a table with hash-prefix buckets): one u64 = the 56 hash bits below the bucket prefix,
// left-aligned, | the two edge indices in the low 6 bits.  Replaces the 9-byte WIDE records on the count and
// lookup paths; split levels and region kernels address it like a narrow record whose u32 is the top half.
constexpr int FMT_TOP8 = 4;
// FMT_TIGHT (k <= 21, tables of >= 2^16 regions; the LAST level's output and the pending sets of the count path only): one
// u32, no lockstep byte.  A record that has reached its region r needs only the hash bits r does not imply:
//   (top 32 hash bits - rstart[r]) << 16 | the 10 hash bits below << 6 | the two edge indices
// (ceil(2^32 / R) <= 2^16 values of the first field).  `>> 6` is the 32-bit key of k_count_regions_n32.  4 bytes instead of
// 5 in the arena, one store per record instead of two in the last level, one load in the table pass.
constexpr int FMT_TIGHT = 5;
constexpr int FMT_NARROW_TO_TIGHT = 6;      // k_lv_scatter only: narrow records in, tight records out (last level)
constexpr uint64_t TIGHT_MIN_REGIONS = 1ull << 16;
__device__ __forceinline__ uint32_t tight_rec(uint32_t bucket, uint32_t main32, uint32_t aux, uint32_t rstart_r) {
    const uint32_t d = ((bucket << (32 - NARROW_CBITS)) | (main32 >> NARROW_CBITS)) - rstart_r;
    return (d << 16) | ((main32 & 0xFFu) << 8) | ((aux & 3u) << 6) | (aux >> 2);
}
__device__ __forceinline__ uint64_t tight_hash(uint32_t t, uint32_t rstart_r) { return ((uint64_t)((t >> 16) + rstart_r) << 32) | ((uint64_t)((t >> 6) & 1023u) << 22); }
__device__ __forceinline__ uint64_t top8_rec(uint64_t h, uint32_t idx6) { return ((h << NARROW_CBITS) & ~63ull) | idx6; }
__device__ __forceinline__ uint64_t top8_hash(uint32_t bucket, uint64_t rec) { return ((uint64_t)bucket << (64 - NARROW_CBITS)) | ((rec & ~63ull) >> NARROW_CBITS); }
__device__ __forceinline__ uint32_t narrow_main(uint64_t h) { return (uint32_t)(h >> (32 - NARROW_CBITS)); }
__device__ __forceinline__ uint32_t narrow_aux(uint64_t h, uint32_t idx6) { return ((uint32_t)(h >> (30 - NARROW_CBITS)) & 3u) | (idx6 << 2); }
__device__ __forceinline__ uint64_t narrow_hash(uint32_t bucket, uint32_t main32, uint32_t aux) {
    return ((uint64_t)bucket << (64 - NARROW_CBITS)) | ((uint64_t)main32 << (32 - NARROW_CBITS)) | ((uint64_t)(aux & 3u) << (30 - NARROW_CBITS));
}
// a narrow record on its way through one multisplit round: u32 | bin << 32 | byte << 48
__device__ __forceinline__ uint64_t narrow_word(uint32_t main32, uint32_t aux, uint32_t bin) { return (uint64_t)main32 | ((uint64_t)(bin | (aux << 16)) << 32); }
__device__ __forceinline__ uint32_t narrow_word_bin(uint64_t w) { return (uint32_t)(w >> 32) & 0xFFFFu; }
// region of a narrow record of bucket b: the top 32 hash bits are b's 8 bits over the top 24 of the u32
__device__ __forceinline__ uint32_t narrow_region(uint32_t bucket, uint32_t main32, uint64_t n_regions) {
    return __umulhi((bucket << (32 - NARROW_CBITS)) | (main32 >> NARROW_CBITS), (uint32_t)n_regions);
}
// bin of a narrow record (u32 part) in a narrow level.  The table has 256 * rps regions and rps = 2^sb * nr_sub exactly, so
// with f = the 24 hash bits below the bucket prefix (position inside the bucket):
//   region - first region of the bucket = floor(f * rps / 2^24)                       (hash_region, exact because R = 256 rps)
//   sub-bucket (middle level)           = floor(floor(f rps / 2^24) / nr_sub) = floor(f 2^sb / 2^24) = the top sb bits of f
//   region inside its sub-bucket (last) = floor((f mod 2^(24-sb)) * nr_sub / 2^(24-sb))
// (floor(floor(x) / n) = floor(x / n) for a positive integer n.)  A bit field for the middle level and one 24-bit multiply
// for the last one, instead of a 32 x 32 -> high multiply for the region and a second one for the division (round 2).
__device__ __forceinline__ uint32_t narrow_bin(const LevelCfg& lv, uint32_t /*b: the segment's first region is implied*/, uint32_t main32) {
    const uint32_t f = main32 >> NARROW_CBITS;
    if (lv.nr_mid) return f >> lv.nr_fshift;
    const uint32_t fs = f & lv.nr_fmask;
    return lv.nr_f24 ? umul24u(fs, lv.nr_sub) >> lv.nr_fshift : (uint32_t)(((uint64_t)fs * lv.nr_sub) >> lv.nr_fshift);
}

// LDS of one multisplit workgroup.  NBC = bin capacity (incl. the discard bin): 512 keeps the whole
// struct at 48 KiB (three workgroups per CU) and covers the usual fan-outs; 2048 (72-76 KiB, two per
// CU) is the general case.
template <int NBC, int FMT, int TILE = MS_TILE>
struct MsShared {
    static constexpr int tile = TILE;
    uint64_t stage[TILE];                    // 32 KiB; narrow records are staged as ONE word: u32 | bin << 32 | byte << 48
    uint16_t sbin[FMT != FMT_NARROW ? TILE : 4];     //  8 KiB
    uint8_t  saux[FMT == FMT_WIDE ? TILE : 8];       //  4 KiB   (WIDE records only)
    uint32_t hist[NBC];
    uint32_t loff[NBC];
    uint32_t gbase[NBC];                     // output cursors: a partition pass handles < 2^32 records (host-checked)
    uint32_t wave_sum[16];
#ifdef KQ_MS_STAMPS
    unsigned long long stamp_last;
    int stamp_on;
#endif
};
#ifdef KQ_MS_STAMPS   // diagnostic build only (never shipped): per-phase cycle sums of the multisplit rounds
__device__ unsigned long long g_ms_stamps[16];
#define KQ_MS_STAMP(s, i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
        __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && (s).stamp_on) { atomicAdd(&g_ms_stamps[i], t_ - (s).stamp_last); (s).stamp_last = t_; } } while (0)
#else
#define KQ_MS_STAMP(s, i) do { } while (0)
#endif

// exclusive scan of s.hist[0..nb) into s.loff.  nb <= NBC.
template <int THREADS, class S>
__device__ __forceinline__ void ms_scan(S& s, uint32_t nb) {
    constexpr int NBC = (int)(sizeof(s.hist) / sizeof(uint32_t));
    constexpr int MS_BINS_PER_THREAD = (NBC + THREADS - 1) / THREADS;
    const int tid = threadIdx.x;
    uint32_t v[MS_BINS_PER_THREAD], sum = 0;
#pragma unroll
    for (int j = 0; j < MS_BINS_PER_THREAD; ++j) { uint32_t b = MS_BINS_PER_THREAD * tid + j; v[j] = (b < nb) ? s.hist[b] : 0; sum += v[j]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { uint32_t n = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += n; }
    if ((tid & 63) == 63) s.wave_sum[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += s.wave_sum[w];
    uint32_t run = base + incl - sum;
#pragma unroll
    for (int j = 0; j < MS_BINS_PER_THREAD; ++j) { uint32_t b = MS_BINS_PER_THREAD * tid + j; if (b < nb) s.loff[b] = run; run += v[j]; }
    __syncthreads();
}

// One multisplit round: every thread brings 16 records with bins in [0, nb]; bin == nb means
// "no record" (discard bin), which keeps the whole round free of per-record branches.
// s.gbase[b] is the workgroup's PRIVATE running output cursor of bin b (set by the caller before
// the first round, advanced here), so a round needs no global atomic at all; records of a bin
// land contiguously at the cursor.  All threads of the block must call it.  nb < NB_MAX.
// THREADS x ITEMS == MS_TILE records per round (256 x 16 for the tile scanner, 512 x 8 where the
// records come from memory: twice the waves over the same LDS footprint hides more latency).
// Stores and loads share the vmcnt counter on gfx9 and complete out of order with respect to each
// other, so a prefetched load that is first USED after this function's stores costs an
// s_waitcnt vmcnt(0) that also drains the stores.  `landed(x)` marks a register as used (the compiler
// puts the wait for its load there); callers pass a `pre_store` hook that applies it to their
// prefetch registers, so the wait sits in front of the stores, where the data has long arrived.
__device__ __forceinline__ void landed(uint32_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void landed(uint64_t& v) { asm volatile("" : "+v"(v)); }
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `rs` (rank replication): every bin gets 2^rs rank counters, chosen by lane, and the staged order is (bin, counter):
// the records of a bin stay contiguous, but a split with few bins no longer serialises its rank atomics on a few LDS
// addresses (an 8-way owner split puts 512 records of a round on each counter otherwise).  (nb + 1) << rs <= NBC.
template <int FMT, int THREADS = MS_THREADS, int ITEMS = MS_ITEMS, class S, class F = NoHook>
__device__ __forceinline__ void block_multisplit(S& s, const uint64_t (&rec)[ITEMS], const uint32_t (&aux)[ITEMS],
                                                 const uint32_t (&bin)[ITEMS], uint32_t nb,
                                                 uint64_t* __restrict__ out, uint8_t* __restrict__ out_aux, F pre_store = F(), uint32_t rs = 0) {
    const int tid = threadIdx.x;
    const uint32_t sub = (uint32_t)tid & ((1u << rs) - 1u), n_ctr = (nb + 1) << rs;
    KQ_MS_STAMP(s, 0);                            // caller: loads landed, bins computed
    for (uint32_t b = tid; b < n_ctr; b += THREADS) s.hist[b] = 0;
    __syncthreads();
    KQ_MS_STAMP(s, 1);                            // zero hist + barrier
    static_assert(THREADS * ITEMS <= S::tile, "round size");
    // FMT_NARROW: rec[i] is the staged word itself (u32 | bin << 32 | byte << 48, narrow_word()); aux[] and
    // bin[] are not read, which saves the caller 2 x ITEMS registers
    // records of the discard bin (read separators, k-mers a map-range filter rejects: half or more of a tile in a map-range
    // pass) take no rank: they are never staged, and their rank atomics would all hit ONE LDS counter and serialise
    uint32_t rank[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t b = FMT == FMT_NARROW ? narrow_word_bin(rec[i]) : bin[i];
        rank[i] = 0;
        if (b != nb) rank[i] = atomicAdd(&s.hist[(b << rs) | sub], 1u);
    }
    __syncthreads();
    KQ_MS_STAMP(s, 2);                            // rank atomics + barrier
    ms_scan<THREADS>(s, n_ctr);
    KQ_MS_STAMP(s, 3);                            // scan (2 barriers)
    // from here to the end of the round gbase[b] is relative to the staged order: the output index of the
    // staged record j of bin b is gbase[b] + j (one LDS read per record in the copy-out instead of two)
    for (uint32_t b = tid; b < nb; b += THREADS) s.gbase[b] -= s.loff[b << rs];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const uint32_t b = FMT == FMT_NARROW ? narrow_word_bin(rec[i]) : bin[i];
        if (b == nb) continue;                                          // discarded: not staged
        const uint32_t p = s.loff[(b << rs) | sub] + rank[i];
        if (FMT == FMT_NARROW) {
            s.stage[p] = rec[i];                                        // one 8-byte write, no sub-dword traffic
        } else {
            s.stage[p] = rec[i];
            s.sbin[p] = (uint16_t)bin[i];
            if (FMT == FMT_WIDE) s.saux[p] = (uint8_t)aux[i];
        }
    }
    __syncthreads();
    KQ_MS_STAMP(s, 4);                            // stage writes + barrier
    const uint32_t total = s.loff[nb << rs];     // records in front of the discard bin
    // fully unrolled so that the LDS reads of all ITEMS positions are in flight together (a rolled
    // loop is a chain of three dependent LDS round trips per record)
    uint32_t cb[ITEMS];
    uint32_t cg[ITEMS];
    uint64_t cv[ITEMS];
    if (FMT == FMT_NARROW) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { cv[it] = s.stage[tid + it * THREADS]; cb[it] = (uint32_t)(tid + it * THREADS) < total ? narrow_word_bin(cv[it]) : 0u; }   // (behind `total` the stage holds stale words)
    } else {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) { const uint32_t j = tid + it * THREADS; cb[it] = j < total ? s.sbin[j] : 0u; cv[it] = s.stage[j]; }
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) cg[it] = s.gbase[cb[it]] + (tid + it * THREADS);
    KQ_MS_STAMP(s, 5);                            // copy-out LDS reads
    pre_store();
    KQ_MS_STAMP(s, 6);                            // prefetch landed
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const uint32_t j = tid + it * THREADS;
        if (j < total) {
            if (FMT == FMT_NARROW) {
#ifdef KQ_EXP_NOSTORE      // timing experiment only (results are wrong): how much of a scatter's time its global stores are
                if (KQ_EXP_NOSTORE >= 2) continue;
#endif
                reinterpret_cast<uint32_t*>(out)[cg[it]] = (uint32_t)cv[it];
#ifdef KQ_EXP_NOSTORE
                continue;
#endif
                if (out_aux) out_aux[cg[it]] = (uint8_t)(cv[it] >> 48);       // (uniform) FMT_TIGHT output has no lockstep byte
            } else {
                out[cg[it]] = cv[it];
                if (FMT == FMT_WIDE) out_aux[cg[it]] = s.saux[j];
            }
        }
    }
    __syncthreads();
    KQ_MS_STAMP(s, 7);                            // global stores issued + barrier
    // advance the cursors: gbase[b] + loff[first counter of bin b] is the absolute cursor, the bin's count of this round is
    // the distance to the first counter of the next bin
    for (uint32_t b = tid; b < nb; b += THREADS) s.gbase[b] += s.loff[(b + 1) << rs];     // (loff is next written behind the next round's barriers)
}

// owner part of a key in the multi-GPU exchange: floor((key % map_count) * n_parts / map_count)
__device__ __forceinline__ uint32_t owner_part_of(uint64_t key, uint32_t map_count, uint32_t map_mask, uint32_t n_parts) {
    const uint32_t m = map_mask ? (uint32_t)key & map_mask : (uint32_t)(key % map_count);     // src/graph-builder.cpp:95
    return (uint32_t)(((uint64_t)m * n_parts) / map_count);
}
__device__ __forceinline__ uint32_t map_index(uint64_t key, uint32_t map_count, uint32_t map_mask) {
    return map_mask ? (uint32_t)key & map_mask : (uint32_t)(key % map_count);                  // src/graph-builder.cpp:95
}
// bin of a key in P1, or cfg.n_coarse (the discard bin) when the map-range filter rejects it
__device__ __forceinline__ uint32_t p1_bin(const PartCfg& cfg, uint64_t key, uint64_t h) {
    if (cfg.filt_lo != 0 || cfg.filt_hi != cfg.map_count) {
        const uint32_t m = map_index(key, cfg.map_count, cfg.map_mask);
        if (m < cfg.filt_lo || m >= cfg.filt_hi) return cfg.n_coarse;
    }
    return cfg.mode == 0 ? (cfg.narrow ? (uint32_t)(h >> (64 - NARROW_CBITS)) : (uint32_t)(hash_region(h, cfg.n_regions) >> cfg.g_shift))
                         : (owner_part_of(key, cfg.map_count, cfg.map_mask, cfg.n_coarse >> cfg.owner_sub) << cfg.owner_sub) |
                           (threadIdx.x & ((1u << cfg.owner_sub) - 1u));      // by lane, not by key: grouping equal keys would cluster them for the receiver
}

}  // namespace kq

// kq_device.h -- device-side building blocks shared by all kernels (gfx950 only).
//
// HBM layout (see DESIGN.md "Data layout"):
//   main table   : n_regions x REGION_SLOTS (2048) slots of 16 B  { u64 w0; u64 e8 }
//                  w0  = rem | cov8 << 56.  rem = the 56 bits of the key's table hash that the slot's region does
//                        not imply (slot_rem / slot_hash below); the canonical key itself is never stored: the hash
//                        is a bijection of it, and only export / the high-copy tier need the key back.
//                        cov8 = instance count 1..254, or 255 = "this k-mer lives in the high-copy tier" (the
//                        reference's 8-bit tombstone, src/graph-builder.cpp:166-190).  w0 == 0 <=> empty slot
//                        (an occupied slot has cov8 >= 1), so a cleared table is all zero bytes.
//                  e8  = 8 packed u8 counters, byte e = edge e of include/kreeq.h:6-18
//                        (e 0..3 = fw[A,C,G,T], 4..7 = bw[A,C,G,T]); only the first 254
//                        instances of a k-mer add here, so no byte ever carries
//                  a key lives in region mulhi(hash >> 32, n_regions) and is probed linearly
//                  INSIDE that region only (regions are independent little tables: this is what
//                  lets a workgroup own a region exclusively in the partitioned count path)
//   high-copy    : power-of-two open-addressing table of 80 B { u64 key; u64 cov_hi; u64 cnt[8] }: the instances
//                  number 255.. of a k-mer (the reference's maps32 tier): total cov = 254 + cov_hi, edge e =
//                  e8 byte e + cnt[e]
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kq {

constexpr uint64_t EMPTY_KEY = ~0ull;          // empty marker of the high-copy table and of LDS region images (never a canonical key / a 56-bit remainder)
constexpr uint32_t LARGEST = 4294967295u;      // include/kreeq.h:68
constexpr uint32_t LOW_TIER_MAX = 254;         // src/graph-builder.cpp:166: the 255th instance overflows
#ifndef KQ_REGION_SHIFT
#define KQ_REGION_SHIFT 11
#endif
constexpr int REGION_SHIFT = KQ_REGION_SHIFT;
constexpr uint32_t REGION_SLOTS = 1u << REGION_SHIFT;   // 2048 slots x 16 B = 32 KiB in HBM
constexpr int COV_SHIFT = 56;
constexpr uint64_t REM_MASK = (1ull << COV_SHIFT) - 1;
constexpr uint64_t COV8_TOMB = 255;            // cov8 value of a k-mer whose count continues in the high-copy tier
constexpr int HI_K = 29;                       // k >= HI_K: the hash has more than 56 bits, the region implies its top 8

struct Slot { uint64_t w0, e8; };
struct HcSlot { uint64_t key; uint64_t cov_hi; uint64_t cnt[8]; };
static_assert(sizeof(Slot) == 16 && sizeof(HcSlot) == 80, "table layouts");

// device-resident state the host reads back after a sync
struct DevState {
    unsigned long long slots_used;    // occupied main slots
    unsigned long long hc_used;       // occupied high-copy slots
    unsigned long long kmers_added;   // sum of cov added
    unsigned int err_table_full;      // a region had no free slot
    unsigned int err_hc_full;
    unsigned int pad[2];
};

// A table may hold only a WINDOW of its regions (multi-GPU shards: the rank that owns the hash-prefix buckets [b_lo, b_hi)
// allocates the regions [reg_lo, reg_hi) = [b_lo, b_hi) x n_regions / 256 of the common geometry and nothing else).
// `slots` is the address region 0 WOULD have, so region r of the window is slots + (r << REGION_SHIFT) like everywhere
// else and no hash -> region computation changes; only loops over the table and accesses by hash look at the window.
struct TableView {
    Slot* slots;
    uint64_t n_regions;     // of the whole geometry (the scale of hash_region)
    uint64_t reg_lo, reg_hi;   // allocated regions: [0, n_regions) unless the table is a window
    HcSlot* hc;
    uint64_t hc_mask;       // capacity - 1
    DevState* st;
    uint32_t k;             // k-mer length: the table hash mixes exactly 2k bits
    uint32_t rps;           // k >= HI_K: regions per top-8-bit hash bucket (n_regions / 256, exact); else 0
    const uint32_t* rstart; // rstart[r] = smallest value of the top 32 hash bits that maps to region r = ceil(r 2^32 / n_regions)
};

// Table hash = an INVERTIBLE mix of the key: one xorshift-multiply-xorshift round that is a bijection
// on the 2k bits a canonical key occupies, returned left-aligned in 64 bits.  Region = top 32 bits
// scaled to n_regions (< 2^32), in-region offset = the low 11 bits of the mixed value.  Because the mix
// is a bijection, neither the records of the partitioned count path nor the table slots hold the key: they
// carry (part of) the mixed value, computed once in the tile scanner; every later stage reads region / offset
// bits straight from it, and a stage drops the bits its position already implies.  `key_of_hash` recovers the
// key where the reference's view is needed (export, high-copy tier).  Region occupancy on k-mer sets is
// Poisson-like, indistinguishable from murmur3's finaliser.
constexpr uint64_t MIX_MUL = 0x9E3779B97F4A7C15ull;
constexpr uint64_t mul_inverse(uint64_t a) {                 // a odd: Newton iteration doubles the correct bits
    uint64_t x = a;
    for (int i = 0; i < 6; ++i) x *= 2 - a * x;
    return x;
}
constexpr uint64_t MIX_INV = mul_inverse(MIX_MUL);
static_assert(MIX_MUL * MIX_INV == 1ull, "inverse");
// k <= 24 (round 3; kreeq's default k = 21 among them): a three-round FEISTEL network on the two k-bit halves of the key, round
// function F(R) = bits 8 .. 8+k-1 of the 24 x 24-bit product R * C.  A bijection on 2k bits like the multiply it replaces, with
// the same occupancy statistics on k-mer sets (region counts Poisson: DESIGN.md section 3), at a third of the issue slots: the
// 64-bit multiply is four quarter-rate 32-bit multiplies on this GPU, v_mul_u32_u24 runs at full rate -- and every count and
// lookup kernel is bound by its instruction stream.  k >= 25 keeps the xorshift-multiply-xorshift round.
constexpr uint32_t FEI_C0 = 0x9E3779u, FEI_C1 = 0x85EBCBu, FEI_C2 = 0xC2B2AFu;        // odd 24-bit constants
constexpr uint32_t FEISTEL_MAX_K = 24;
// low 32 bits of the 24 x 24-bit product (v_mul_u32_u24, full rate).  HIP declares __umul24 as returning int: a product with
// bit 31 set would shift arithmetically -- always go through this
__device__ __forceinline__ uint32_t umul24u(uint32_t a, uint32_t b) { return (uint32_t)__umul24(a, b); }
__device__ __forceinline__ uint32_t fei_f(uint32_t r, uint32_t c, uint32_t k) { return (umul24u(r, c) >> 8) & ((1u << k) - 1u); }
__device__ __forceinline__ uint64_t table_hash(uint64_t key, uint32_t k) {      // key < 4^k, 1 <= k <= 32
    const uint32_t pad = 64 - 2 * k;
    if (k <= FEISTEL_MAX_K) {
        uint32_t l = (uint32_t)key & ((1u << k) - 1u), r = (uint32_t)(key >> k), t;
        t = l ^ fei_f(r, FEI_C0, k); l = r; r = t;
        t = l ^ fei_f(r, FEI_C1, k); l = r; r = t;
        t = l ^ fei_f(r, FEI_C2, k); l = r; r = t;
        return (((uint64_t)l << k) | r) << pad;
    }
    uint64_t x = key ^ (key >> k);                           // shift >= half the width: self-inverse
    x = (x * MIX_MUL) << pad;                                // left-aligned product mod 4^k
    return x ^ ((x >> k) & (~0ull << pad));
}
__device__ __forceinline__ uint64_t key_of_hash(uint64_t h, uint32_t k) {
    const uint32_t pad = 64 - 2 * k;
    if (k <= FEISTEL_MAX_K) {
        const uint64_t x = h >> pad;
        uint32_t l = (uint32_t)(x >> k), r = (uint32_t)x & ((1u << k) - 1u), t;
        t = r ^ fei_f(l, FEI_C2, k); r = l; l = t;          // a round maps (l, r) to (r, l ^ F(r)): undo it with (r' ^ F(l'), l')
        t = r ^ fei_f(l, FEI_C1, k); r = l; l = t;
        t = r ^ fei_f(l, FEI_C0, k); r = l; l = t;
        return ((uint64_t)r << k) | l;
    }
    uint64_t x = (h ^ ((h >> k) & (~0ull << pad))) >> pad;
    x = ((x * MIX_INV) << pad) >> pad;
    return x ^ (x >> k);
}
__device__ __forceinline__ uint64_t hash_region(uint64_t h, uint64_t n_regions) { return __umulhi((uint32_t)(h >> 32), (uint32_t)n_regions); }
// Home slot of a hash inside its region: the low 11 bits of the mixed value, rounded down to a QUAD of slots (16-byte aligned
// keys in the LDS image of k_count_regions_q4: one read covers the home quad).  The probe sequence is linear from there, so
// every kernel that probes -- table_find / table_add, the region kernels, rehash -- agrees by using this one function.
__device__ __forceinline__ uint32_t hash_offset(uint64_t h, uint32_t k) { return (uint32_t)(h >> (64 - 2 * k)) & (REGION_SLOTS - 4); }
// The 56 hash bits a slot stores.  k <= 28: the hash has at most 56 significant bits (left-aligned: the low 8 are
// zero).  k >= HI_K: tables have a multiple of 256 regions, so the top 8 hash bits are region / rps.
__device__ __forceinline__ uint64_t slot_rem(uint64_t h, uint32_t k) { return k >= HI_K ? h & REM_MASK : h >> 8; }
__device__ __forceinline__ uint64_t slot_hash(const TableView& t, uint64_t rem, uint64_t region) {
    return t.k >= HI_K ? ((uint64_t)((uint32_t)region / t.rps) << COV_SHIFT) | rem : rem << 8;
}

__device__ __forceinline__ uint64_t mix64(uint64_t h) {
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull;
    h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull;
    h ^= h >> 33;
    return h;
}

// reverse complement of a 2-bit packed k-mer, first base in the low bits:
// rv = sum (3-b[c]) 4^(k-1-c)
__device__ __forceinline__ uint64_t revcomp2(uint64_t fw, int k) {
    uint64_t x = __builtin_bitreverse64(~fw);
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    return x >> (64 - 2 * k);
}

// edge index pair -> packed u8x8 increment.  Mirrors src/graph-builder.cpp:98-110.
// prev/next = base codes 0..3, or 4 when the neighbour is not an ACGT base of the same run.
__device__ __forceinline__ uint64_t edge_pack(bool is_fw, uint32_t prev, uint32_t next) {
    uint64_t p = 0;
    if (is_fw) {
        if (next < 4) p |= 1ull << (8 * next);
        if (prev < 4) p |= 1ull << (8 * (4 + prev));
    } else {
        if (prev < 4) p |= 1ull << (8 * (3 - prev));
        if (next < 4) p |= 1ull << (8 * (4 + 3 - next));
    }
    return p;
}
// packed increment <-> reference edge byte (bit 7-e, include/kreeq.h:10-16)
__device__ __forceinline__ uint8_t pack_to_edge_byte(uint64_t p) {
    uint32_t b = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) b |= (uint32_t)((p >> (8 * e)) & 1) << (7 - e);
    return (uint8_t)b;
}
__device__ __forceinline__ uint64_t edge_byte_to_pack(uint32_t b) {
    uint64_t p = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) p |= (uint64_t)((b >> (7 - e)) & 1) << (8 * e);
    return p;
}

__device__ __forceinline__ uint64_t ld_relaxed(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- main table ------------------------------------------------------------------------------
__device__ __forceinline__ Slot* region_of(const TableView& t, uint64_t h) {
    return t.slots + (hash_region(h, t.n_regions) << REGION_SHIFT);
}
// does the table hold the region of hash h?  (always, unless it is a window)
__device__ __forceinline__ bool table_owns(const TableView& t, uint64_t h) {
    const uint64_t r = hash_region(h, t.n_regions);
    return r >= t.reg_lo && r < t.reg_hi;
}

__device__ __forceinline__ const Slot* table_find(const TableView& t, uint64_t h) {
    if (!table_owns(t, h)) return nullptr;
    const Slot* base = region_of(t, h);
    const uint64_t rem = slot_rem(h, t.k);
    uint32_t off = hash_offset(h, t.k);
    for (uint32_t probe = 0; probe < REGION_SLOTS; ++probe) {
        const Slot* s = base + ((off + probe) & (REGION_SLOTS - 1));
        const uint64_t w = s->w0;
        if (w == 0) return nullptr;
        if ((w & REM_MASK) == rem) return s;
    }
    return nullptr;
}

// ---- high-copy side table ----------------------------------------------------------------------
__device__ __forceinline__ HcSlot* hc_upsert(const TableView& t, uint64_t key) {
    uint64_t i = mix64(key ^ 0x9E3779B97F4A7C15ull) & t.hc_mask;
    for (uint64_t probe = 0; probe <= t.hc_mask; ++probe, i = (i + 1) & t.hc_mask) {
        HcSlot* s = t.hc + i;
        uint64_t cur = ld_relaxed(&s->key);
        if (cur == EMPTY_KEY) {
            cur = atomicCAS((unsigned long long*)&s->key, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (cur == EMPTY_KEY) { atomicAdd(&t.st->hc_used, 1ull); return s; }
        }
        if (cur == key) return s;
    }
    return nullptr;
}
__device__ __forceinline__ const HcSlot* hc_find(const TableView& t, uint64_t key) {
    uint64_t i = mix64(key ^ 0x9E3779B97F4A7C15ull) & t.hc_mask;
    for (uint64_t probe = 0; probe <= t.hc_mask; ++probe, i = (i + 1) & t.hc_mask) {
        const HcSlot* s = t.hc + i;
        uint64_t cur = s->key;
        if (cur == key) return s;
        if (cur == EMPTY_KEY) return nullptr;
    }
    return nullptr;
}
// add `cov_hi` instances and the edge counts (packed u8x8, or 8 wide counters) to the high-copy entry of hash h
__device__ __forceinline__ bool hc_add(const TableView& t, uint64_t h, uint64_t cov_hi, uint64_t pack8, const uint32_t* wide) {
    HcSlot* hs = hc_upsert(t, key_of_hash(h, t.k));
    if (!hs) { atomicOr(&t.st->err_hc_full, 1u); return false; }
    if (cov_hi) atomicAdd((unsigned long long*)&hs->cov_hi, (unsigned long long)cov_hi);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint64_t v = wide ? (uint64_t)wide[e] : ((pack8 >> (8 * e)) & 0xFF);
        if (v) atomicAdd((unsigned long long*)&hs->cnt[e], (unsigned long long)v);
    }
    return true;
}

// Add `cov` instances of the k-mer with table hash `h` whose edge counts are pack8 (u8x8) or wide[0..7] (each
// <= cov); wide != nullptr implies cov > 254.
// Two-tier rule (restates src/graph-builder.cpp:165-205 order-independently): an add that keeps the k-mer's total
// <= 254 goes to cov8 and the packed u8 counters; the add that crosses 254 turns cov8 into the tombstone 255 and,
// like every later one, goes to the high-copy entry (cov_hi counts the instances beyond 254).  Because every edge
// counter <= cov, no u8 lane can exceed 254, and low + high is the exact sum.
// The count shares its word with the hash remainder, so the update is a CAS on w0 (claiming an empty slot and
// counting its first instance in one step); the tombstone is sticky, so the high-copy tier needs no CAS.
__device__ __forceinline__ bool table_add(const TableView& t, uint64_t h, uint64_t cov, uint64_t pack8,
                                          const uint32_t* wide /*8 counters or nullptr*/, uint32_t* inserted) {
    if (!table_owns(t, h)) return false;                               // a window takes only the k-mers of its own buckets
    Slot* base = region_of(t, h);
    const uint64_t rem = slot_rem(h, t.k);
    const uint32_t off = hash_offset(h, t.k);
    for (uint32_t probe = 0; probe < REGION_SLOTS; ++probe) {
        Slot* s = base + ((off + probe) & (REGION_SLOTS - 1));
        uint64_t w = ld_relaxed(&s->w0);
        while (w == 0 || (w & REM_MASK) == rem) {
            const uint64_t old = w >> COV_SHIFT;                       // 0: the slot is empty
            if (old == COV8_TOMB) return hc_add(t, h, cov, pack8, wide);
            const uint64_t sum = old + cov;
            const bool low = sum <= LOW_TIER_MAX;
            const uint64_t prev = atomicCAS((unsigned long long*)&s->w0, (unsigned long long)w,
                                            (unsigned long long)(rem | ((low ? sum : COV8_TOMB) << COV_SHIFT)));
            if (prev != w) { w = prev; continue; }                     // lost a race on this slot: look again
            if (w == 0) *inserted = 1;
            if (low) { if (pack8) atomicAdd((unsigned long long*)&s->e8, (unsigned long long)pack8); return true; }
            return hc_add(t, h, sum - LOW_TIER_MAX, pack8, wide);
        }
    }
    atomicOr(&t.st->err_table_full, 1u);
    return false;
}

// logical (reference-visible) value of a slot: counters clamped to LARGEST
struct Logical { uint32_t e[8]; uint32_t cov; };
__device__ __forceinline__ Logical logical_of(const TableView& t, uint64_t h, uint64_t w0, uint64_t e8) {
    Logical L;
    uint64_t cov = w0 >> COV_SHIFT;
    const HcSlot* hs = (cov == COV8_TOMB) ? hc_find(t, key_of_hash(h, t.k)) : nullptr;
    if (hs) cov = LOW_TIER_MAX + hs->cov_hi;
    L.cov = cov > LARGEST ? LARGEST : (uint32_t)cov;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        uint64_t v = (e8 >> (8 * e)) & 0xFF;
        if (hs) v += hs->cnt[e];
        L.e[e] = v > LARGEST ? LARGEST : (uint32_t)v;
    }
    return L;
}
// hash of the k-mer in slot s of table t (its region is implied by its index)
__device__ __forceinline__ uint64_t slot_hash_at(const TableView& t, const Slot* s, uint64_t w0) {
    return slot_hash(t, w0 & REM_MASK, (uint64_t)(s - t.slots) >> REGION_SHIFT);
}
__device__ __forceinline__ Logical slot_logical(const TableView& t, const Slot* s) {
    const uint64_t w0 = s->w0;
    return logical_of(t, slot_hash_at(t, s, w0), w0, s->e8);
}

// ---- LDS region image of the kernels that own a region (k_count_regions, k_merge_regions) ----------------------
// Three u64 per slot: { rem (EMPTY_KEY when free), e8, cnt }.  cnt is the instance count as a full word, so LDS
// atomics cannot wrap it; a slot that arrived as a tombstone starts at IMG_TOMB | 254, which every "old + n <= 254"
// test fails, so its adds go to the high-copy tier.  img_store() turns cnt back into cov8 and hands the instances
// beyond 254 that this pass added to the high-copy entry (one global atomic per high-copy k-mer and pass).
constexpr uint64_t IMG_TOMB = 1ull << 63;
__device__ __forceinline__ void img_load(uint64_t* img, uint32_t i, uint64_t w0, uint64_t e8) {
    const uint64_t c = w0 >> COV_SHIFT;
    img[3 * i] = w0 ? (w0 & REM_MASK) : EMPTY_KEY;
    img[3 * i + 1] = e8;
    img[3 * i + 2] = c == COV8_TOMB ? (IMG_TOMB | LOW_TIER_MAX) : c;
}
__device__ __forceinline__ ulonglong2 img_store(const TableView& t, const uint64_t* img, uint32_t i, uint64_t region) {
    const uint64_t rem = img[3 * i], e8 = img[3 * i + 1], c = img[3 * i + 2];
    if (rem == EMPTY_KEY) return make_ulonglong2(0ull, 0ull);
    const uint64_t cnt = c & ~IMG_TOMB;
    uint64_t cov8 = (c & IMG_TOMB) ? COV8_TOMB : cnt;
    if (cnt > LOW_TIER_MAX) {
        cov8 = COV8_TOMB;
        hc_add(t, slot_hash(t, rem, region), cnt - LOW_TIER_MAX, 0, nullptr);
    }
    return make_ulonglong2(rem | (cov8 << COV_SHIFT), e8);
}

// ---- sequence tile scanner ---------------------------------------------------------------------
// One workgroup (256 threads) walks tiles of TILE_STARTS k-mer start positions.  Per tile it loads
// a 4096-byte window (16 B per lane, coalesced), converts it to a 2-bit code stream + an
// "invalid base" bit stream in LDS, and every lane then rolls through 16 consecutive starts in
// registers.  f(pos, fw, prev, next) is called for every start whose k bases are all ACGT;
// prev/next are the neighbouring base codes or 4 when outside the run.
constexpr int TILE_THREADS = 256;
constexpr int TILE_STARTS = 4032;          // 252 lanes x 16 starts; window = 4032 + 16 + 48 bytes

__device__ __forceinline__ void convert16(const uint4& v, bool all_in, int64_t g, int64_t lo_valid, int64_t hi_valid,
                                          uint32_t& codes, uint32_t& inv) {
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
    codes = 0; inv = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t x = w[i];
        uint32_t y = ((x >> 1) ^ (x >> 2)) & 0x03030303u;              // A,C,G,T -> 0,1,2,3 (case-blind)
        uint32_t c4 = (y | (y >> 6) | (y >> 12) | (y >> 18)) & 0xFFu;
        codes |= c4 << (8 * i);
        uint32_t u = x & 0xDFDFDFDFu;                                   // upper-case
        uint32_t ok = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t ch = (u >> (8 * b)) & 0xFFu;
            uint32_t idx = ch - 0x41u;                                  // 'A'
            uint32_t good = (idx < 20u) ? ((0x80045u >> idx) & 1u) : 0u; // A=0 C=2 G=6 T=19
            ok |= good << b;
        }
        inv |= ((~ok) & 0xFu) << (4 * i);
    }
    if (!all_in) {
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            int64_t p = g + b;
            if (p < lo_valid || p >= hi_valid) inv |= 1u << b;
        }
    }
}

// global half of a tile load: this lane's 16 bytes of the tile's 4096-byte window (zeros outside).
// PACKED input (pinv != nullptr; kq_pack_bases, include/kreeq_amd.h): `ab` is the array of 2-bit codes, one u32 per 16 bases
// (base i at bits 2i), pinv the array of invalid-base masks, one u16 per 16 bases -- the scanner's own LDS format, 6 bytes
// per 16 bases over PCIe instead of 16; the lane's unit comes back in v.x / v.y and tile_store skips the conversion.
// (`lane` = the thread's index among the TILE_THREADS that work on this tile: threadIdx.x unless a workgroup holds several tiles)
__device__ __forceinline__ uint4 tile_fetch(const uint8_t* __restrict__ ab, int64_t lo_valid, int64_t hi_valid, uint64_t tile,
                                            const uint16_t* __restrict__ pinv = nullptr, int lane = -1) {
    const int64_t g = (int64_t)(tile * TILE_STARTS) - 16 + 16 * (lane < 0 ? (int)threadIdx.x : lane);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g + 16 > lo_valid && g < hi_valid) {
        if (pinv) { v.x = reinterpret_cast<const uint32_t*>(ab)[g >> 4]; v.y = pinv[g >> 4]; }      // (g is a multiple of 16, >= 0 here)
        else v = *reinterpret_cast<const uint4*>(ab + g);
    }
    return v;
}
// LDS half: convert to the 2-bit code stream + invalid-base bit stream (ends with a barrier)
__device__ __forceinline__ void tile_store(const uint4& v, int64_t lo_valid, int64_t hi_valid, uint64_t tile,
                                           uint32_t* s_codes, uint32_t* s_inv, bool packed = false, int lane = -1) {
    const int tid = lane < 0 ? (int)threadIdx.x : lane;
    const int64_t g = (int64_t)(tile * TILE_STARTS) - 16 + 16 * tid;
    uint32_t codes = 0, inv = 0xFFFFu;
    if (g + 16 > lo_valid && g < hi_valid) {
        bool all_in = (g >= lo_valid) && (g + 16 <= hi_valid);
        if (packed) {
            codes = v.x; inv = v.y & 0xFFFFu;
            if (!all_in) {
#pragma unroll
                for (int b = 0; b < 16; ++b) { const int64_t p = g + b; if (p < lo_valid || p >= hi_valid) inv |= 1u << b; }
            }
        } else convert16(v, all_in, g, lo_valid, hi_valid, codes, inv);
    }
    s_codes[tid] = codes;
    s_inv[tid] = inv;
    __syncthreads();
}
// loads + converts one tile's 4096-byte window into LDS (ends with a barrier)
__device__ __forceinline__ void tile_load(const uint8_t* __restrict__ ab, int64_t lo_valid, int64_t hi_valid, uint64_t tile,
                                          uint32_t* s_codes, uint32_t* s_inv, const uint16_t* __restrict__ pinv = nullptr) {
    tile_store(tile_fetch(ab, lo_valid, hi_valid, tile, pinv), lo_valid, hi_valid, tile, s_codes, s_inv, pinv != nullptr);
}

// This lane's 16 consecutive k-mer starts of the loaded tile.  The forward word and its reverse
// complement are ROLLED from start to start (2 shifts + or each) instead of re-extracted:
//   fw' = (fw >> 2) | b[i+k] << (2k-2)        rv' = ((rv << 2) | (3 - b[i+k])) & mask
// ALL = false: f(i, true, pos, fw, rv, prev, next) only for starts whose k bases are all ACGT;
// ALL = true : f(...) for every start with a `valid` flag and the compile-time index i (lets
//              callers fill register arrays with static indexing).
// prev/next = neighbouring base codes, or 4 when that neighbour is not a base of the same run.
template <bool ALL, class F>
__device__ __forceinline__ void lane_scan_core(const uint32_t* s_codes, const uint32_t* s_inv, int64_t lo_valid, uint64_t tile,
                                               int k, F&& f, uint64_t range_lo = 0, uint64_t range_hi = ~0ull, int lane = -1) {
    const int tid = lane < 0 ? (int)threadIdx.x : lane;
    const bool lane_has_work = tid < TILE_STARTS / 16;
    if (!ALL && !lane_has_work) return;
    const int t = lane_has_work ? tid : 0;
    const uint64_t kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const uint64_t wmask = (1ull << k) - 1;
    const uint32_t c0 = s_codes[t], c1 = s_codes[t + 1], c2 = s_codes[t + 2], c3 = s_codes[t + 3];
    const uint32_t m0 = s_inv[t];
    const uint64_t ms = (uint64_t)s_inv[t + 1] | ((uint64_t)s_inv[t + 2] << 16) | ((uint64_t)s_inv[t + 3] << 32);
    const uint64_t lo = (uint64_t)c1 | ((uint64_t)c2 << 32);
    const uint64_t hi = (uint64_t)c3;
    uint32_t prev = (m0 >> 15) ? 4u : (c0 >> 30);
    const int64_t p0 = (int64_t)(tile * TILE_STARTS) + 16 * t - lo_valid;       // caller position of start 0
    // Everything that depends on k is shifted into place ONCE (64-bit dynamic shifts are slow);
    // the 16 steps below then use 32-bit ops with compile-time shifts only:
    //   ahead / inv_ahead = codes / invalid bits of the bases k..k+15 (the base entering the window at step i),
    //   inv_lo            = invalid bits of the bases 0..15           (the base leaving it),
    //   bad               = number of invalid bases inside the current window (rolled: + entering - leaving).
    const uint32_t ahead = (k == 32) ? (uint32_t)hi : (uint32_t)((lo >> (2 * k)) | (hi << (64 - 2 * k)));
    const uint32_t inv_ahead = (uint32_t)(ms >> k) & 0xFFFFu;
    const uint32_t inv_lo = (uint32_t)ms & 0xFFFFu;
    int bad = __popcll(ms & wmask);
    // starts outside [range_lo, range_hi) (caller positions) are reported invalid: one 16-bit mask per lane
    // instead of two 64-bit compares per start
    uint32_t rmask = 0xFFFFu;
    {
        const int64_t a = (int64_t)range_lo - p0, b = range_hi == ~0ull ? 16 : (int64_t)range_hi - p0;
        const uint32_t lo_i = a <= 0 ? 0u : a >= 16 ? 16u : (uint32_t)a, hi_i = b <= 0 ? 0u : b >= 16 ? 16u : (uint32_t)b;
        rmask = ((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u);
    }
    const int top = 2 * k - 2;                  // where the entering base lands in fw (wave-uniform)
    const bool top_hi = top >= 32;
    const int top_sh = top_hi ? top - 32 : top;
    uint64_t fw = lo & kmask;
    uint64_t rv = revcomp2(fw, k);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool valid = lane_has_work && bad == 0 && ((rmask >> i) & 1u);
        const uint32_t nraw = (ahead >> (2 * i)) & 3u;
        const uint32_t ninv = (inv_ahead >> i) & 1u;
        const uint32_t next = ninv ? 4u : nraw;
        if (ALL) f(i, valid, (uint64_t)(p0 + i), fw, rv, prev, next);
        else if (valid) f(i, true, (uint64_t)(p0 + i), fw, rv, prev, next);
        const uint32_t pinv = (inv_lo >> i) & 1u;
        prev = pinv ? 4u : ((uint32_t)fw & 3u);
        bad += (int)ninv - (int)pinv;
        const uint32_t add = nraw << top_sh;
        fw = (fw >> 2) | (top_hi ? ((uint64_t)add << 32) : (uint64_t)add);       // fw' = fw >> 2 | b[i+k] << (2k-2)
        rv = ((rv << 2) | (uint64_t)(3u - nraw)) & kmask;                         // rv' = (rv << 2 | 3 - b[i+k]) & mask
    }
}

// f(pos, fw, rv, prev, next) per valid k-mer
template <class F>
__device__ __forceinline__ void tile_lane_scan(const uint32_t* s_codes, const uint32_t* s_inv, int64_t lo_valid, uint64_t tile,
                                               int k, F&& f) {
    lane_scan_core<false>(s_codes, s_inv, lo_valid, tile, k,
                          [&](int, bool, uint64_t pos, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) { f(pos, fw, rv, prev, next); });
}
// k-mer starts a launch may process, in caller positions: lets the host cut a huge batch into
// slices that share the input array (a slice's scan window includes the neighbouring bases)
struct EmitRange { uint64_t lo, hi; };

// f(i, valid, fw, rv, prev, next) for all 16 starts; starts outside `er` are reported invalid
template <class F>
__device__ __forceinline__ void tile_lane_scan_all(const uint32_t* s_codes, const uint32_t* s_inv, int64_t lo_valid, uint64_t tile,
                                                   int k, EmitRange er, F&& f, int lane = -1) {
    lane_scan_core<true>(s_codes, s_inv, lo_valid, tile, k,
                         [&](int i, bool valid, uint64_t, uint64_t fw, uint64_t rv, uint32_t prev, uint32_t next) { f(i, valid, fw, rv, prev, next); },
                         er.lo, er.hi, lane);
}

// number of valid k-mer starts among this lane's 16
__device__ __forceinline__ uint32_t tile_lane_count(const uint32_t* s_inv, int k) {
    const int tid = threadIdx.x;
    if (tid >= TILE_STARTS / 16) return 0;
    const uint64_t wmask = (1ull << k) - 1;
    const uint64_t ms = (uint64_t)s_inv[tid + 1] | ((uint64_t)s_inv[tid + 2] << 16) | ((uint64_t)s_inv[tid + 3] << 32);
    uint32_t n = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) n += (((ms >> i) & wmask) == 0);
    return n;
}

inline __host__ __device__ uint64_t n_tiles_of(uint64_t lead, uint64_t len) { return (lead + len + TILE_STARTS - 1) / TILE_STARTS; }

template <class F>
__device__ __forceinline__ void scan_tiles(const uint8_t* __restrict__ ab, uint64_t lead, uint64_t len, int k, F&& f, const uint16_t* __restrict__ pinv = nullptr) {
    __shared__ uint32_t s_codes[TILE_THREADS];
    __shared__ uint32_t s_inv[TILE_THREADS];
    const int64_t lo_valid = (int64_t)lead, hi_valid = (int64_t)(lead + len);
    const uint64_t n_tiles = n_tiles_of(lead, len);
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_load(ab, lo_valid, hi_valid, tile, s_codes, s_inv, pinv);
        tile_lane_scan(s_codes, s_inv, lo_valid, tile, k, f);
        __syncthreads();
    }
}

// block-wide sum of per-thread u64, result valid in thread 0
__device__ __forceinline__ uint64_t block_sum(uint64_t v) {
    __shared__ unsigned long long s_acc;
    if (threadIdx.x == 0) s_acc = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_acc, (unsigned long long)v);
    __syncthreads();
    uint64_t r = s_acc;
    __syncthreads();
    return r;
}

}  // namespace kq

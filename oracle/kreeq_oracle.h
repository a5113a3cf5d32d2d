/*
 * kreeq_oracle.h -- CPU restatement of the vgl-hub/kreeq k-mer count / QV hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it, and there
 * only as the checker / reported CPU baseline.  The product path (kreeq_amd/, include/) never
 * includes this header and fails loudly when the HIP library is missing.
 *
 * Parity status: PINNED.  The reference binary cannot be built here (its gfalibs submodule, which
 * also vendors parallel-hashmap, is an empty directory in /root/reference), so
 * this restatement is pinned by the reference's own fixtures instead: the 10 testFiles/ *.kreeq
 * databases, validateFiles/test.{0..14,20..35}.tst stdout goldens and both .bkwig per-base dumps
 * (tests/test_oracle_golden.py checks every one of them).
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 */
#ifndef KREEQ_ORACLE_H
#define KREEQ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/kreeq.h:20-21 (DBGkmer) and :69-70 (DBGkmer32) */
typedef struct { uint8_t  fw[4], bw[4], cov; } kqo_kmer8;
typedef struct { uint32_t fw[4], bw[4], cov; } kqo_kmer32;
/* include/input.h:4-9 (DBGbase) */
typedef struct { uint32_t fw, bw, cov; uint8_t isFw; uint8_t pad[3]; } kqo_dbgbase;

typedef struct kqo_db kqo_db;

/* logical table entry used for comparisons: u32 counters, hc=1 when the k-mer lives in the
 * 32-bit high-copy map (reference: maps32[m]) */
typedef struct { uint64_t key; uint32_t fw[4], bw[4], cov; uint32_t hc; } kqo_entry;

typedef struct {
    uint64_t total;      /* "Total kmers"    = sum cov          src/graph-builder.cpp:274-278 */
    uint64_t unique;     /* "Unique kmers"   = #cov==1          :250-251 */
    uint64_t distinct;   /* "Distinct kmers"                    :256,265 */
    uint64_t missing;    /* "Missing kmers"  = 4^k - distinct   :286 */
    uint64_t edges;      /* "Total edges" with the :254/:263 precedence quirk */
} kqo_stats;

kqo_db*  kqo_create(int k, int map_count);
void     kqo_destroy(kqo_db*);
int      kqo_k(const kqo_db*);
int      kqo_map_count(const kqo_db*);

/* gfalibs Kmap::hash (absent source; semantics SURVEY.md §9.1; call sites
 * src/graph-builder.cpp:93, src/kreeq.cpp:145).  s = k base codes 0..3. */
uint64_t kqo_hash(const uint8_t* s, int k, int* is_fw);

/* hot loop 1 only (DBG::hashSequences, src/graph-builder.cpp:34-126): emit (key, edge byte) records
 * of one read batch in sequence order.  keys/edges may be NULL to just count.  Returns #records. */
uint64_t kqo_emit_records(int k, const char* bases, uint64_t len, uint64_t* keys, uint8_t* edges);

/* hot loop 2 only (DBG::processBuffers, src/graph-builder.cpp:128-223) on records in given order */
int      kqo_insert_records(kqo_db*, const uint64_t* keys, const uint8_t* edges, uint64_t n);

/* loop 1 + partition by key % map_count + loop 2, with `threads` workers (chunks split at read
 * separators for loop 1; one map per job for loop 2, like the reference's thread pool). */
int      kqo_count_batch(kqo_db*, const char* bases, uint64_t len, int threads);

/* DBG::summary + DBG::DBstats numbers (src/graph-builder.cpp:240-295).  hist (optional) receives
 * up to hist_cap (cov,count) pairs sorted by cov; *hist_n = number of distinct cov values. */
int      kqo_summary(const kqo_db*, kqo_stats* out, uint64_t* hist_cov, uint64_t* hist_cnt,
                     uint64_t hist_cap, uint64_t* hist_n);

/* DBG::evaluateSegment (src/kreeq.cpp:110-229) on one segment (ACGTacgt only is expected; any
 * other byte is looked up as the reference would after N-splitting: see kqo_validate_sequence).
 * counters[0] += missing, [1] += total k-mers evaluated, [2] += edge-missing.  per_base may be
 * NULL; otherwise it must hold len zero-initialised entries (generateValidationVector,
 * src/input.cpp:38-45) and is updated in place. */
int      kqo_lookup_segment(const kqo_db*, const char* bases, uint64_t len, uint32_t cov_cutoff,
                            uint16_t map_lo, uint16_t map_hi, kqo_dbgbase* per_base,
                            uint64_t counters[3], int threads);

/* Whole assembly sequence: split into segments at every non-ACGT byte (gfalibs appendSequence
 * splits at N runs) and evaluate each segment.  per_base (nullable) has len
 * entries aligned with `bases` (separator positions stay zero). */
int      kqo_validate_sequence(const kqo_db*, const char* bases, uint64_t len, uint32_t cov_cutoff,
                               uint16_t map_lo, uint16_t map_hi, kqo_dbgbase* per_base,
                               uint64_t counters[3], int threads);

/* DBG::kunion + DBG::mergeSubMaps (src/graph-builder.cpp:297-432): dst += src */
int      kqo_merge(kqo_db* dst, const kqo_db* src);

/* logical content, sorted by key.  map < 0 => all maps.  out may be NULL to count. */
uint64_t kqo_export(const kqo_db*, int map, kqo_entry* out, uint64_t cap);
/* raw physical content of one map (incl. 8-bit tombstones cov==255), for invariant tests */
uint64_t kqo_export_raw8(const kqo_db*, int map, uint64_t* keys, kqo_kmer8* vals, uint64_t cap);
/* insert logical entries (as phmap_load of a .kreeq would): hc!=0 goes to the 32-bit map plus a
 * cov=255 tombstone in the 8-bit map (DBG::reloadMap32, src/graph-builder.cpp:225-238) */
int      kqo_import(kqo_db*, const kqo_entry* in, uint64_t n);

/* errorRate (src/kreeq.cpp:36-40) and QV = -10 log10(err) (:87,:96) */
double   kqo_error_rate(uint64_t missing, uint64_t total, int k);
double   kqo_qv(uint64_t missing, uint64_t total, int k);

#ifdef __cplusplus
}
#endif
#endif

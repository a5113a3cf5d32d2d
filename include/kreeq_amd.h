/*
 * kreeq_amd.h -- C ABI of the MI355X-native k-mer count / QV engine (libkreeq_amd.so).
 *
 * This is the drop-in boundary for kreeq's hot path.  kreeq has no plugin / FFI interface; the
 * seams a replacement sits behind are the CRTP hooks gfalibs' Kmap calls on DBG.
 * Each entry point below names the reference hook it replaces (paths relative to the reference
 * repository vgl-hub/kreeq @ 2024_08_07).  INTEGRATION.md shows the binding a kreeq maintainer
 * would add inside those hooks.
 *
 * Conventions
 *   - every function returns 0 (KQ_OK) or a negative kq_status; kq_last_error() returns a
 *     thread-local message for the last failure on the calling thread; no C++ exception and no
 *     torch type ever crosses this boundary;
 *   - one handle per GPU; calls on one handle must be serialised by the caller, different handles
 *     may be used from different threads / processes concurrently;
 *   - plain pointers and sizes only.  "host" entry points take host buffers and copy over PCIe;
 *     the *_dev entry points take device pointers valid on the handle's GPU and are asynchronous
 *     on the handle's stream (kq_sync() / any host-returning call synchronises);
 *   - a "read batch" / "sequence" is a byte string of bases; any byte other than ACGTacgt ends
 *     the current run, so reads inside a batch are separated by one such byte (e.g. '\n' or 'N')
 *     and k-mers never span it -- exactly the reference's behaviour on a bad base
 *     (src/graph-builder.cpp:77-91) and gfalibs' splitting of assembly sequences at N.
 *
 * There is no CPU fallback: every compute entry point fails with KQ_ERR_NO_DEVICE when no gfx950
 * device is usable.
 */
#ifndef KREEQ_AMD_H
#define KREEQ_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KQ_ABI_VERSION 4

typedef enum {
    KQ_OK = 0,
    KQ_ERR_INVALID = -1,     /* bad argument (k, map_count, null pointer, map range ...) */
    KQ_ERR_NO_DEVICE = -2,   /* no usable HIP device / kernel image for this GPU */
    KQ_ERR_HIP = -3,         /* a HIP runtime call failed; message has the HIP error string */
    KQ_ERR_NOMEM = -4,       /* device or host allocation failed */
    KQ_ERR_TABLE_FULL = -5,  /* k-mer table (or its high-copy side table) could not grow */
    KQ_ERR_CAPACITY = -6,    /* caller's output buffer too small (n_out holds the needed size) */
    KQ_ERR_MISMATCH = -7     /* handles disagree on k / map_count / device */
} kq_status;

typedef struct kq_handle kq_handle;

/* Logical table entry == one k-mer of the de Bruijn graph with exact counters.
 * Replaces the value types DBGkmer / DBGkmer32 (include/kreeq.h:20-21, :69-70): hc != 0 marks a
 * k-mer that the reference keeps in the 32-bit high-copy map (cov >= 255), hc == 0 one whose
 * counters all fit the 8-bit map.  Counters saturate at 2^32-1 (LARGEST, include/kreeq.h:68). */
typedef struct {
    uint64_t key;
    uint32_t fw[4], bw[4], cov;
    uint32_t hc;
} kq_entry;

/* Per-position validation result == DBGbase (include/input.h:4-9). */
typedef struct {
    uint32_t fw, bw, cov;
    uint8_t  isFw;
    uint8_t  pad[3];
} kq_dbgbase;

/* Numbers of the "DBG Summary statistics" block (DBG::summary + DBG::DBstats,
 * src/graph-builder.cpp:240-295), including the ":254/:263" edge-count precedence quirk. */
typedef struct {
    uint64_t total;     /* Total kmers    */
    uint64_t unique;    /* Unique kmers   */
    uint64_t distinct;  /* Distinct kmers */
    uint64_t missing;   /* Missing kmers = 4^k - distinct */
    uint64_t edges;     /* Total edges    */
} kq_stats;

/* Running totals of one handle (not part of the reference; used for throughput reporting). */
typedef struct {
    uint64_t kmers_counted;   /* k-mer instances inserted so far (sum of cov added)          */
    uint64_t slots_used;      /* occupied table slots == distinct k-mers                      */
    uint64_t slots_total;     /* table capacity in slots                                      */
    uint64_t hc_used;         /* occupied high-copy side-table slots                          */
    uint64_t hc_total;
    uint64_t table_bytes;     /* HBM bytes held by table + side table                         */
    uint64_t table_passes;    /* region-wise passes over the table so far (see KQ_OPT_PENDING_BYTES) */
} kq_info;

/* ---- lifetime ---------------------------------------------------------------------------- */

/* Replaces the DBG / Kmap constructor (include/kreeq.h:168-177): k = kmerLen (2..32), map_count =
 * gfalibs mapCount (128 in every .kreeq .index).  capacity_hint = expected distinct k-mers
 * (0 = small default; the table grows by rehashing when needed). */
int  kq_create(kq_handle** out, int device, int k, int map_count, uint64_t capacity_hint);
void kq_destroy(kq_handle* h);
/* Drop all k-mers, keep the allocation. */
int  kq_clear(kq_handle* h);
/* Make the handle enqueue on a caller-owned hipStream_t (NULL = the handle's own stream). */
int  kq_set_stream(kq_handle* h, void* hip_stream);
void* kq_get_stream(kq_handle* h);
/* Tuning knobs (no reference counterpart).
 *   KQ_OPT_TRUST_CAPACITY  value != 0: capacity_hint given to kq_create is an upper bound of the
 *                          distinct k-mers the table will ever hold, so batches do not pre-grow the
 *                          table for their worst case (all k-mers new).  If the bound is wrong a
 *                          table region overflows and the next sync returns KQ_ERR_TABLE_FULL.
 *   KQ_OPT_COUNT_PATH      0 = auto, 1 = direct (global atomics), 2 = partitioned (LDS regions)
 *   KQ_OPT_SLICE_KMERS     k-mer starts processed per internal slice of a resident batch (default 2^28;
 *                          the partition scratch is 16 bytes per start)
 *   KQ_OPT_COUNT_MAP_RANGE value = lo | hi << 16: kq_count_batch(_dev) keeps only k-mers whose map index
 *                          key % map_count lies in [lo, hi).  This is the reference's memory-bounded mode
 *                          (process the maps in ranges, src/kreeq.cpp:59-74; README "HPC" runs + union):
 *                          count the same reads once per range into separate databases, then union.
 *   KQ_OPT_PROFILE         value != 0: HIP events are recorded around the stages of the partitioned count;
 *                          kq_get_profile() returns "stage=ms;..." for the last batch (measurement aid)
 *   KQ_OPT_LOOKUP_PATH     kq_lookup_sequence(_dev) without per-base output: 0 = auto, 1 = direct (one table probe
 *                          per k-mer), 2 = partitioned (the k-mers are split by table region like read k-mers
 *                          and evaluated against region images staged in LDS)
 *   KQ_OPT_MERGE_PATH      kq_merge into this handle: 0 = auto, 1 = one atomic add per source entry, 2 = region by
 *                          region (destination images staged in LDS, both tables streamed once)
 *   KQ_OPT_NARROW_MID      (tuning / tests) regions per hash-prefix bucket from which the record split of the
 *                          partitioned paths gets a middle level (default 2048, i.e. tables above 8.6 GB)
 *   KQ_OPT_PENDING_BYTES   the partitioned count keeps the region-sorted records of a slice PENDING in an arena of at most
 *                          this many bytes of HBM and applies all pending sets in one pass over the table (when the arena
 *                          is full, and before anything reads the table: kq_sync, summary, lookup, export, merge ...), so
 *                          that a large table is streamed once per arena-full of records rather than once per slice.
 *                          -1 = automatic (default: a few times the table, at most the free HBM less 1/8 of the device),
 *                          0 = apply every slice at once.  Results never depend on it (counting is commutative).
 *   KQ_OPT_BUCKET_WINDOW   value = lo | hi << 16, 0 <= lo < hi <= 256, on an EMPTY handle with k <= 21: the handle becomes one
 *                          shard of a multi-GPU database -- it holds, and answers for, only the k-mers whose table hash
 *                          starts with one of the 8-bit prefixes ("hash-prefix buckets") lo .. hi-1.  The memory kq_create
 *                          sized is kept and laid out as that window of a table 256 / (hi - lo) times larger, so the bucket
 *                          split of the count path is the owner split of the exchange (kq_emit_sharded_dev /
 *                          kq_insert_sharded_dev).  k-mers of other buckets are ignored by every entry point: counts drop
 *                          them, lookups do not evaluate them (their sum over the shards is the whole answer, like the map
 *                          ranges of src/kreeq.cpp:150), export / summary see the window's k-mers. */
/*   KQ_OPT_OVERLAP        1 (default): a count call that cuts its batch into several slices (KQ_OPT_SLICE_KMERS) runs the
 *                          partition stages of consecutive slices on two internal streams with a scratch set each (slice
 *                          j+1's scan beside slice j's split levels) and joins them with the handle's stream before it
 *                          returns: the caller's stream-ordered view of the handle is unchanged.  0: one stream.
 *                          2: also in a map-range pass (KQ_OPT_COUNT_MAP_RANGE), where 1 does not fork (tests). */
/*   KQ_OPT_COUNT_MAP_PASSES n = 2, 4 or 8 (1 = off): the caller counts the SAME RESIDENT batches once per range of the equal split of
 *                          the maps into n (KQ_OPT_COUNT_MAP_RANGE = [r mapCount / n, (r + 1) mapCount / n), any order) and vouches
 *                          that a batch keeps its device address and content between those passes.  The first pass that scans a
 *                          slice then counts its k-mers for all n ranges in one histogram scan and keeps the count matrices (<= 4 GB);
 *                          the other passes skip that scan.  Results never depend on it.  Setting the option drops what is kept. */
/*   KQ_OPT_KERNEL_SET      measurement only (results never depend on it): bit mask of count-path stages that run their ALTERNATIVE kernel
 *                          instead of the shipped one, so that both can be timed in one process on the same buffers
 *                          (tools/bench_extra/alt_kernels.sh, bench.py --alt-kernels): 1 = P1 scatter of narrow and hash-remainder records with k_p1_scatter
 *                          (round 2's) instead of k_p1_scatter_s, 2 = split levels that write narrow records with k_lv_scatter_s (the
 *                          streamed formulation, which lost there) instead of k_lv_scatter, 4 = the level that writes tight records
 *                          with k_lv_scatter (round 2's) instead of k_lv_scatter_s.  Default 0. */
enum { KQ_OPT_TRUST_CAPACITY = 1, KQ_OPT_COUNT_PATH = 2, KQ_OPT_SLICE_KMERS = 3, KQ_OPT_COUNT_MAP_RANGE = 4, KQ_OPT_PROFILE = 5,
       KQ_OPT_LOOKUP_PATH = 6, KQ_OPT_MERGE_PATH = 7, KQ_OPT_NARROW_MID = 8, KQ_OPT_PENDING_BYTES = 9, KQ_OPT_BUCKET_WINDOW = 10, KQ_OPT_OVERLAP = 11, KQ_OPT_COUNT_MAP_PASSES = 12, KQ_OPT_KERNEL_SET = 13,
       KQ_OPT_TEST_FAIL_PLAN = 100 /* failure-path tests only: the next partition plan of a count fails with KQ_ERR_NOMEM */ };
int  kq_set_option(kq_handle* h, int option, int64_t value);
int  kq_get_profile(kq_handle* h, char* buf, uint64_t cap);
int  kq_sync(kq_handle* h);
/* Enqueue the table pass that applies all pending record sets (KQ_OPT_PENDING_BYTES); asynchronous, no-op when
 * nothing is pending.  kq_sync() and every call that reads the table do this themselves. */
int  kq_flush(kq_handle* h);
int  kq_get_info(kq_handle* h, kq_info* out);
const char* kq_last_error(void);
int  kq_abi_version(void);
/* 1 when a gfx950-capable device is visible to the HIP runtime. */
int  kq_device_available(void);
/* Free / total HBM of a device in bytes (what gfalibs' get_mem_total / freeMemory are to the reference's
 * memory-bounded mode, src/main.cpp:433, src/kreeq.cpp:59-63): lets the host choose how many map ranges to count in. */
int  kq_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);

/* ---- hot loop 1 + 2: count ---------------------------------------------------------------- */

/* Replaces DBG::hashSequences (src/graph-builder.cpp:34-126) followed by DBG::processBuffers for
 * every map (:128-223) on one read batch: ASCII -> 2-bit, canonical key, edge byte, insert/RMW of
 * cov + 8 edge counters.  The key % mapCount disk partition of the reference has no counterpart
 * here (single table in HBM).  The host variant returns when the batch has been consumed; like the _dev variant it
 * may leave records pending (KQ_OPT_PENDING_BYTES): kq_sync() applies them and reports table-full conditions. */
int  kq_count_batch(kq_handle* h, const char* bases, uint64_t len);
int  kq_count_batch_dev(kq_handle* h, const char* d_bases, uint64_t len);

/* Pipelined host ingest (what gfalibs' loadKmers reader thread + readBatches queue are to the reference, src/input.cpp:95-96):
 * kq_count_batch_async enqueues the host-to-device copy of a batch on a copy stream and its count behind it, and returns
 * at once; copies overlap the counting of earlier batches (a small ring of device staging buffers).  `bases` should come
 * from kq_host_alloc (page-locked memory: the copy is a DMA at PCIe rate, 50+ GB/s here; a copy out of pageable memory the
 * runtime has not seen before costs ~1 ms per MiB for locking its pages) and be REUSED: a few buffers shared by all producer
 * threads beat a buffer per thread.  The buffer may be refilled once kq_host_wait(ticket)
 * has returned.  kq_count_batch_async may be called from several threads at once on one handle (the only entry point
 * that may; the calls take turns inside, none of them waits for the GPU); no other call on the handle may run concurrently
 * with them.  kq_sync() drains everything. */
/* kq_host_free: no copy may still read the buffer -- wait for the tickets of every batch submitted from it (kq_host_wait), or
 * kq_sync(), first; the library does not wait for them here. */
void* kq_host_alloc(uint64_t bytes);
void  kq_host_free(void* p);
int  kq_count_batch_async(kq_handle* h, const char* bases, uint64_t len, uint64_t* ticket);
int  kq_host_wait(kq_handle* h, uint64_t ticket);

/* 2-bit packed input (the "2-bit packing" step of the reference's kcount ancestry, moved in front of PCIe): kq_pack_bases
 * (host, no GPU) turns `len` bytes of bases + separators into ceil(len / 16) units of one u32 of 2-bit codes (base i of the
 * unit at bits 2i; A C G T = 0 1 2 3, case-blind) and one u16 of invalid-base bits (anything but ACGT/acgt -- read
 * separators, N -- and the positions behind `len` in the last unit).  That is the tile scanner's own format: 6 bytes per
 * 16 bases cross PCIe instead of 16, and the count kernels skip the ASCII conversion.  kq_count_packed_dev /
 * kq_count_packed_async count such a batch exactly like kq_count_batch_dev / kq_count_batch_async count its ASCII form
 * (same tickets, same threading rule; `codes` and `inv` may be refilled once kq_host_wait(ticket) has returned). */
void kq_pack_bases(const char* bases, uint64_t len, uint32_t* codes, uint16_t* inv);
int  kq_count_packed_dev(kq_handle* h, const uint32_t* d_codes, const uint16_t* d_inv, uint64_t n_bases);
int  kq_count_packed_async(kq_handle* h, const uint32_t* codes, const uint16_t* inv, uint64_t n_bases, uint64_t* ticket);

/* Hot loop 1 only (DBG::hashSequences :75-113): the (key, edge byte) records of a batch in
 * sequence order; edge byte layout = edgeBit (include/kreeq.h:6-18).  *n_out = number of records
 * (also set on KQ_ERR_CAPACITY).  keys/edges may be NULL to just count. */
int  kq_emit_records(kq_handle* h, const char* bases, uint64_t len,
                     uint64_t* keys, uint8_t* edges, uint64_t cap, uint64_t* n_out);
/* Device variant used to stage the multi-GPU exchange: records are grouped by owner part
 * (part p owns the maps m with floor(m * n_parts / map_count) == p, m = key % map_count; order
 * inside a part is unspecified).  d_keys/d_edges need room for len records; part_counts[n_parts]
 * is a HOST array. Synchronises. */
int  kq_emit_partitioned_dev(kq_handle* h, const char* d_bases, uint64_t len, int n_parts,
                             uint64_t* d_keys, uint8_t* d_edges, uint64_t cap,
                             uint64_t* part_counts);

/* Same staging with PACKED 8-byte records (k <= 28): bits 0..55 = the library's invertible 56-bit mix
 * of the canonical key (its table hash; opaque to the caller, identical on every handle with the same
 * k), bits 56..58 / 59..61 = index of the fw / bw edge the instance contributes (0..3, 7 = none;
 * src/graph-builder.cpp:98-110).  One array to exchange instead of two, and the receive side
 * (kq_insert_packed_dev, the only consumer of this format) runs the partitioned, atomic-free count
 * without hashing again.  d_recs needs room for len - k + 1 records.  Synchronises. */
int  kq_emit_packed_dev(kq_handle* h, const char* d_bases, uint64_t len, int n_parts,
                        uint64_t* d_recs, uint64_t cap, uint64_t* part_counts);
int  kq_insert_packed_dev(kq_handle* h, const uint64_t* d_recs, uint64_t n);

/* Multi-GPU staging with 5-BYTE records (k <= 21, the default k): d_recs[i] (u32) + d_aux[i] (u8) = the hash bits below
 * the 8-bit hash-prefix bucket of record i and its two edge indices -- the record format of the single-GPU count, written
 * by its first (bucket) split.  OWNERSHIP IS BY BUCKET RANGE: part p of n_parts owns the buckets
 * [ceil(256 p / n_parts), ceil(256 (p + 1) / n_parts)), so the bucket-sorted output is already grouped by owner: the part
 * of every owner is one contiguous run (part_counts, host), and d_bucket_counts[p * 256 + b] (DEVICE, n_parts x 256) =
 * records of bucket b if p owns it, else 0.  The receiver gets, from every peer, its run and that peer's 256 counts for it
 * (one all-to-all(v) per array + one all-to-all of 256 counts) and calls kq_insert_sharded_dev with the runs concatenated
 * in peer order and d_bucket_counts = [n_peers x 256] (DEVICE, peer-major): the records enter the bucket -> region split
 * levels directly.  The receiving handle is the window of its buckets (KQ_OPT_BUCKET_WINDOW; with one part: an ordinary
 * table) and must have >= 2048 regions; records of buckets outside its window are ignored.
 * (The reference's key % mapCount -- src/graph-builder.cpp:95 -- remains the layout of the database FILES: export filters
 * by map range on every shard.)  Buffers need room for len - k + 1 records.  kq_emit_sharded_dev synchronises to fill
 * part_counts; with part_counts == NULL it only enqueues (the part sizes are the row sums of d_bucket_counts: a caller that
 * exchanges the counts anyway reads them in the same host round trip). */
int  kq_emit_sharded_dev(kq_handle* h, const char* d_bases, uint64_t len, int n_parts, uint32_t* d_recs, uint8_t* d_aux, uint64_t cap,
                         uint64_t* d_bucket_counts, uint64_t* part_counts);
int  kq_insert_sharded_dev(kq_handle* h, const uint32_t* d_recs, const uint8_t* d_aux, uint64_t n, int n_peers, const uint64_t* d_bucket_counts);

/* Hot loop 2 only (DBG::processBuffers :160-206) on explicit records. */
int  kq_insert_records(kq_handle* h, const uint64_t* keys, const uint8_t* edges, uint64_t n);
int  kq_insert_records_dev(kq_handle* h, const uint64_t* d_keys, const uint8_t* d_edges, uint64_t n);

/* ---- summary ------------------------------------------------------------------------------ */

/* Replaces DBG::summary(m) for all m + DBG::DBstats (src/graph-builder.cpp:240-295). */
int  kq_summary(kq_handle* h, kq_stats* out);
/* finalHistogram (src/graph-builder.cpp:274-278; printed by gfalibs printHist): (cov, count)
 * pairs sorted by cov.  cov/cnt may be NULL to get *n_out only. */
int  kq_histogram(kq_handle* h, uint64_t* cov, uint64_t* cnt, uint64_t cap, uint64_t* n_out);

/* ---- hot loop 3: assembly lookup + QV counters --------------------------------------------- */

/* Replaces DBG::evaluateSegment (src/kreeq.cpp:110-229) for every segment of one assembly
 * sequence: the sequence is split into segments at non-ACGT bytes; every k-mer whose map index
 * key % map_count lies in [map_lo, map_hi) is looked up.  counters[0] += missing,
 * counters[1] += k-mers evaluated, counters[2] += edge-missing (the three atomics of
 * include/kreeq.h:152).  per_base (nullable) has `len` entries aligned with `bases`, must be
 * zero-initialised by the caller before the first map range (generateValidationVector,
 * src/input.cpp:38-45) and is updated only at evaluated positions. */
int  kq_lookup_sequence(kq_handle* h, const char* bases, uint64_t len, uint32_t cov_cutoff,
                        uint16_t map_lo, uint16_t map_hi, kq_dbgbase* per_base,
                        uint64_t counters[3]);
/* d_counters: 3 x u64 on the device, accumulated with atomics (zero them first). Asynchronous. */
int  kq_lookup_sequence_dev(kq_handle* h, const char* d_bases, uint64_t len, uint32_t cov_cutoff,
                            uint16_t map_lo, uint16_t map_hi, kq_dbgbase* d_per_base,
                            uint64_t* d_counters);

/* ---- candidate-error search support (DBG::correctSequences, src/variants.cpp) --------------------------------- */

/* Batched map->find(key) (src/variants.cpp:118-131, :203-206; src/kreeq.cpp:152-166 for the 32-bit tier): the logical
 * entries of n canonical keys, out[i].key = keys[i]; out[i].cov == 0 when the k-mer is absent.  The bounded graph
 * search stays on the host (pointer chasing, serial per source); it asks for its frontier in batches through this. */
int  kq_lookup_keys(kq_handle* h, const uint64_t* keys, uint64_t n, kq_entry* out);

/* Device pre-filter of the search: for every k-mer start c of `bases` (segments = ACGT runs, like kq_lookup_sequence),
 * flags[c] bit 0 = the k-mer is in the table, bit 1 = DBG::searchVariants would find at least one candidate path at
 * its source (src/variants.cpp:232-246 at depth 0): an edge in the direction the sequence is read, with a count
 * above cov_cutoff, that does not lead to the k-mer the sequence continues with.  Only those positions need the host
 * search; everywhere else it returns at once without a variant.  flags has `len` bytes (0 where no k-mer starts). */
int  kq_branch_scan(kq_handle* h, const char* bases, uint64_t len, uint32_t cov_cutoff, uint8_t* flags);

/* ---- union / database import-export --------------------------------------------------------- */

/* Replaces DBG::kunion + DBG::mergeSubMaps (src/graph-builder.cpp:297-432): dst += src, exact
 * saturating sums; both handles must live on the same device. */
int  kq_merge(kq_handle* dst, kq_handle* src);

/* Replaces phmap_load of <db>/.map.<m>.bin + .map.hc.bin (src/graph-builder.cpp:307-308, gfalibs
 * loadMapRange): ADDS logical entries to the table (importing two databases == union). */
int  kq_import(kq_handle* h, const kq_entry* entries, uint64_t n);
/* Replaces gfalibs dumpMap/dumpHighCopyKmers: logical entries with key % map_count in
 * [map_lo, map_hi), sorted by key.  out may be NULL to get *n_out only. */
int  kq_export(kq_handle* h, uint16_t map_lo, uint16_t map_hi, kq_entry* out, uint64_t cap,
               uint64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* KREEQ_AMD_H */

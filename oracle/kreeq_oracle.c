/*
 * kreeq_oracle.c -- CPU restatement of the vgl-hub/kreeq hot path (see kreeq_oracle.h).
 * TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
 *
 * Structure kept identical to the reference so that it can double as the reported CPU baseline
 * (BASELINE.md §2): non-rolling O(k) canonical hash per k-mer, 9-byte (key, edge) records
 * partitioned into key % mapCount buffers, one flat open-addressing map pair (8-bit + 32-bit
 * high-copy) per bucket with a single writer, per-segment lookup.  The hash map itself is a plain
 * linear-probing table (the reference uses parallel-hashmap, which is absent from the mount); map
 * iteration order therefore differs, which is why all comparisons are on logical content.
 */
#define _GNU_SOURCE
#include "kreeq_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define KQO_LARGEST 4294967295u /* include/kreeq.h:68 */
#define KQO_EMPTY   (~0ull)     /* never a canonical key: min(fw,rv)==~0 would need fw==rv==~0 */

/* ------------------------------------------------------------------ base coding (gfalibs ctoi) */
/* A,a->0 C,c->1 G,g->2 T,t->3, everything else -> 4 */
static uint8_t g_ctoi[256];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;
static void init_ctoi(void) {
    memset(g_ctoi, 4, sizeof g_ctoi);
    g_ctoi['A'] = g_ctoi['a'] = 0;
    g_ctoi['C'] = g_ctoi['c'] = 1;
    g_ctoi['G'] = g_ctoi['g'] = 2;
    g_ctoi['T'] = g_ctoi['t'] = 3;
}

/* ------------------------------------------------------------------ flat map (stand-in for phmap) */
typedef struct {
    uint64_t cap, size;   /* cap is a power of two or 0 */
    uint32_t vsz;
    uint64_t* keys;
    uint8_t*  vals;
} kqo_map;

static inline uint64_t map_slot(uint64_t key, uint64_t cap) {
    uint64_t h = key * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    return h & (cap - 1);
}
static void map_init(kqo_map* m, uint32_t vsz) { memset(m, 0, sizeof *m); m->vsz = vsz; }
static void map_free(kqo_map* m) { free(m->keys); free(m->vals); m->keys = NULL; m->vals = NULL; m->cap = m->size = 0; }
static void* map_find(const kqo_map* m, uint64_t key) {
    if (!m->cap) return NULL;
    for (uint64_t i = map_slot(key, m->cap);; i = (i + 1) & (m->cap - 1)) {
        if (m->keys[i] == key) return m->vals + i * m->vsz;
        if (m->keys[i] == KQO_EMPTY) return NULL;
    }
}
static void map_grow(kqo_map* m) {
    uint64_t ncap = m->cap ? m->cap * 2 : 16;
    uint64_t* nk = (uint64_t*)malloc(ncap * sizeof(uint64_t));
    uint8_t* nv = (uint8_t*)calloc(ncap, m->vsz);
    memset(nk, 0xFF, ncap * sizeof(uint64_t));
    for (uint64_t i = 0; i < m->cap; ++i) {
        if (m->keys[i] == KQO_EMPTY) continue;
        uint64_t j = map_slot(m->keys[i], ncap);
        while (nk[j] != KQO_EMPTY) j = (j + 1) & (ncap - 1);
        nk[j] = m->keys[i];
        memcpy(nv + j * m->vsz, m->vals + i * m->vsz, m->vsz);
    }
    free(m->keys); free(m->vals);
    m->keys = nk; m->vals = nv; m->cap = ncap;
}
/* operator[]: find or insert a zero value */
static void* map_at(kqo_map* m, uint64_t key) {
    if ((m->size + 1) * 10 > m->cap * 7) map_grow(m);
    for (uint64_t i = map_slot(key, m->cap);; i = (i + 1) & (m->cap - 1)) {
        if (m->keys[i] == key) return m->vals + i * m->vsz;
        if (m->keys[i] == KQO_EMPTY) {
            m->keys[i] = key;
            memset(m->vals + i * m->vsz, 0, m->vsz);
            ++m->size;
            return m->vals + i * m->vsz;
        }
    }
}

/* ------------------------------------------------------------------ database */
struct kqo_db {
    int k, map_count;
    kqo_map* maps;   /* [map_count] value kqo_kmer8   (reference maps[m])   */
    kqo_map* maps32; /* [map_count] value kqo_kmer32  (reference maps32[m]) */
};

kqo_db* kqo_create(int k, int map_count) {
    pthread_once(&g_once, init_ctoi);
    if (k < 2 || k > 32 || map_count < 1 || map_count > 65535) return NULL;
    kqo_db* db = (kqo_db*)calloc(1, sizeof *db);
    db->k = k; db->map_count = map_count;
    db->maps = (kqo_map*)calloc(map_count, sizeof(kqo_map));
    db->maps32 = (kqo_map*)calloc(map_count, sizeof(kqo_map));
    for (int m = 0; m < map_count; ++m) { map_init(&db->maps[m], sizeof(kqo_kmer8)); map_init(&db->maps32[m], sizeof(kqo_kmer32)); }
    return db;
}
void kqo_destroy(kqo_db* db) {
    if (!db) return;
    for (int m = 0; m < db->map_count; ++m) { map_free(&db->maps[m]); map_free(&db->maps32[m]); }
    free(db->maps); free(db->maps32); free(db);
}
int kqo_k(const kqo_db* db) { return db->k; }
int kqo_map_count(const kqo_db* db) { return db->map_count; }

/* ------------------------------------------------------------------ hash */
uint64_t kqo_hash(const uint8_t* s, int k, int* is_fw) {
    uint64_t fw = 0, rv = 0;
    for (int c = 0; c < k; ++c) {            /* O(k), not rolling -- as the reference */
        fw |= (uint64_t)s[c] << (2 * c);                    /* first base in the low bits */
        rv |= (uint64_t)(3 - s[c]) << (2 * (k - 1 - c));    /* reverse complement, same packing */
    }
    if (is_fw) *is_fw = fw < rv;             /* palindrome => not forward */
    return fw < rv ? fw : rv;
}

/* include/kreeq.h:10-16: edge e sets bit 7-e */
static inline uint8_t edge_bit(unsigned e) { return (uint8_t)(1u << (7 - e)); }

/* ------------------------------------------------------------------ hot loop 1 */
/* 9-byte record buffers, one per map (gfalibs Buf<uint8_t>; src/graph-builder.cpp:66,95-97,112) */
typedef struct { uint8_t* seq; uint64_t pos, size; } recbuf;
static inline void recbuf_push(recbuf* b, uint64_t key, uint8_t edges) {
    if (b->pos + 9 > b->size) { b->size = b->size ? b->size * 2 : 9 * 256; b->seq = (uint8_t*)realloc(b->seq, b->size); }
    memcpy(b->seq + b->pos, &key, 8);
    b->seq[b->pos + 8] = edges;
    b->pos += 9;
}

typedef void (*emit_fn)(void* ctx, uint64_t key, uint8_t edges);

/* DBG::hashSequences body, src/graph-builder.cpp:75-113, restated over maximal ACGT runs.
 * Equivalence with the reference's (p, e) window walk: a k-mer is produced at p iff bases
 * p..p+k-1 are all ACGT (:77-91 restart after any bad base); the "previous base" test :103/:106
 * reads str[p-1], which is the converted base p-1 (a bad base keeps its >3 code, and p>0);
 * the "next base" test :101/:108 reads ctoi[first[p+k]], which is the string terminator when
 * p+k==len.  So prev/next edges exist iff that neighbour is inside the same ACGT run. */
static uint64_t scan_batch(const uint8_t* first, uint64_t len, int k, emit_fn emit, void* ctx) {
    uint64_t n = 0;
    if (len < (uint64_t)k) return 0;                       /* :60 */
    uint8_t* str = (uint8_t*)malloc(len);
    for (uint64_t i = 0; i < len; ++i) str[i] = g_ctoi[first[i]];
    uint64_t i = 0;
    while (i < len) {
        if (str[i] > 3) { ++i; continue; }
        uint64_t j = i;
        while (j < len && str[j] <= 3) ++j;                /* run = [i, j) */
        if (j - i >= (uint64_t)k) {
            for (uint64_t p = i; p + k <= j; ++p) {
                int is_fw;
                uint64_t key = kqo_hash(str + p, k, &is_fw);          /* :93 */
                uint8_t edges = 0;
                int has_next = (p + k < j), has_prev = (p > i);
                if (is_fw) {                                           /* :100-104 */
                    if (has_next) edges |= edge_bit(str[p + k]);
                    if (has_prev) edges |= edge_bit(4 + str[p - 1]);
                } else {                                               /* :105-110 */
                    if (has_prev) edges |= edge_bit(3 - str[p - 1]);
                    if (has_next) edges |= edge_bit(4 + 3 - str[p + k]);
                }
                if (emit) emit(ctx, key, edges);
                ++n;
            }
        }
        i = j;
    }
    free(str);
    return n;
}

typedef struct { uint64_t* keys; uint8_t* edges; uint64_t n; } emit_flat_ctx;
static void emit_flat(void* c, uint64_t key, uint8_t edges) {
    emit_flat_ctx* x = (emit_flat_ctx*)c;
    if (x->keys) x->keys[x->n] = key;
    if (x->edges) x->edges[x->n] = edges;
    ++x->n;
}
uint64_t kqo_emit_records(int k, const char* bases, uint64_t len, uint64_t* keys, uint8_t* edges) {
    pthread_once(&g_once, init_ctoi);
    emit_flat_ctx x = { keys, edges, 0 };
    return scan_batch((const uint8_t*)bases, len, k, emit_flat, &x);
}

/* ------------------------------------------------------------------ hot loop 2 */
/* One record into maps[m]/maps32[m]: src/graph-builder.cpp:165-205 */
static inline void insert_one(kqo_map* map, kqo_map* map32, uint64_t key, uint8_t edges) {
    kqo_kmer8* e8 = (kqo_kmer8*)map_at(map, key);                     /* :165 */
    int overflow = e8->cov >= 254;                                    /* :166 */
    for (int w = 0; w < 4; ++w)                                       /* :168-174 */
        if (e8->fw[w] + 1 == 255 || e8->bw[w] + 1 == 255) { overflow = 1; break; }
    if (!overflow) {                                                  /* :176-184 */
        for (int w = 0; w < 4; ++w) {
            e8->fw[w] += (edges & edge_bit(w)) != 0;
            e8->bw[w] += (edges & edge_bit(4 + w)) != 0;
        }
        ++e8->cov;
        return;
    }
    kqo_kmer32* e32 = (kqo_kmer32*)map_at(map32, key);                /* :188 (may grow map32 only) */
    if (e32->cov == 0) {                                              /* :190-194 */
        for (int w = 0; w < 4; ++w) { e32->fw[w] = e8->fw[w]; e32->bw[w] = e8->bw[w]; }
        e32->cov = e8->cov;
        e8->cov = 255;
    }
    for (int w = 0; w < 4; ++w) {                                     /* :196-202 */
        uint32_t f = (edges & edge_bit(w)) != 0, b = (edges & edge_bit(4 + w)) != 0;
        if (KQO_LARGEST - e32->fw[w] >= f) e32->fw[w] += f;
        if (KQO_LARGEST - e32->bw[w] >= b) e32->bw[w] += b;
    }
    if (e32->cov < KQO_LARGEST) ++e32->cov;                           /* :203-204 */
}

int kqo_insert_records(kqo_db* db, const uint64_t* keys, const uint8_t* edges, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t m = keys[i] % (uint64_t)db->map_count;               /* :95 */
        insert_one(&db->maps[m], &db->maps32[m], keys[i], edges[i]);
    }
    return 0;
}

/* ------------------------------------------------------------------ threaded count */
typedef struct {
    kqo_db* db; const uint8_t* bases; uint64_t lo, hi; recbuf* bufs; /* [map_count] */
} chunk_job;
typedef struct { recbuf* bufs; int map_count; } emit_part_ctx;
static void emit_part(void* c, uint64_t key, uint8_t edges) {
    emit_part_ctx* x = (emit_part_ctx*)c;
    recbuf_push(&x->bufs[key % (uint64_t)x->map_count], key, edges);  /* :95-97,112 */
}
typedef struct {
    chunk_job* chunks; int n_chunks; int next_chunk;
    int next_map; pthread_mutex_t mtx; kqo_db* db;
} count_ctx;
static void* loop1_worker(void* p) {
    count_ctx* c = (count_ctx*)p;
    for (;;) {
        pthread_mutex_lock(&c->mtx); int i = c->next_chunk++; pthread_mutex_unlock(&c->mtx);
        if (i >= c->n_chunks) return NULL;
        chunk_job* j = &c->chunks[i];
        emit_part_ctx x = { j->bufs, c->db->map_count };
        scan_batch(j->bases + j->lo, j->hi - j->lo, c->db->k, emit_part, &x);
    }
}
static void* loop2_worker(void* p) {
    count_ctx* c = (count_ctx*)p;
    for (;;) {
        pthread_mutex_lock(&c->mtx); int m = c->next_map++; pthread_mutex_unlock(&c->mtx);
        if (m >= c->db->map_count) return NULL;
        for (int i = 0; i < c->n_chunks; ++i) {               /* one job per map: single writer */
            recbuf* b = &c->chunks[i].bufs[m];
            for (uint64_t o = 0; o < b->pos; o += 9) {         /* :160-163 */
                uint64_t key; memcpy(&key, b->seq + o, 8);
                insert_one(&c->db->maps[m], &c->db->maps32[m], key, b->seq[o + 8]);
            }
        }
    }
}

int kqo_count_batch(kqo_db* db, const char* bases_, uint64_t len, int threads) {
    const uint8_t* bases = (const uint8_t*)bases_;
    if (threads < 1) threads = 1;
    /* chunks = independent "read batches": cut only at a non-ACGT byte so no k-mer is split */
    int want = threads * 4;
    if (len < 1u << 16) want = 1;
    chunk_job* chunks = (chunk_job*)calloc(want, sizeof *chunks);
    int n_chunks = 0; uint64_t lo = 0;
    for (int i = 1; i <= want && lo < len; ++i) {
        uint64_t hi = (i == want) ? len : (len / want) * i;
        if (hi < lo) hi = lo;
        while (hi < len && g_ctoi[bases[hi]] <= 3) ++hi;
        if (hi == lo) continue;
        chunks[n_chunks].db = db; chunks[n_chunks].bases = bases;
        chunks[n_chunks].lo = lo; chunks[n_chunks].hi = hi;
        chunks[n_chunks].bufs = (recbuf*)calloc(db->map_count, sizeof(recbuf));
        ++n_chunks; lo = hi;
    }
    count_ctx c; memset(&c, 0, sizeof c);
    c.chunks = chunks; c.n_chunks = n_chunks; c.db = db; pthread_mutex_init(&c.mtx, NULL);
    pthread_t* th = (pthread_t*)calloc(threads, sizeof *th);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, loop1_worker, &c);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, loop2_worker, &c);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    for (int i = 0; i < n_chunks; ++i) {
        for (int m = 0; m < db->map_count; ++m) free(chunks[i].bufs[m].seq);
        free(chunks[i].bufs);
    }
    free(chunks); free(th); pthread_mutex_destroy(&c.mtx);
    return 0;
}

/* ------------------------------------------------------------------ summary */
static int cmp_u64(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : x > y;
}
int kqo_summary(const kqo_db* db, kqo_stats* out, uint64_t* hist_cov, uint64_t* hist_cnt,
                uint64_t hist_cap, uint64_t* hist_n) {
    uint64_t uniq = 0, distinct = 0, edges = 0, total = 0;
    kqo_map hist; map_init(&hist, sizeof(uint64_t));
    for (int m = 0; m < db->map_count; ++m) {
        const kqo_map* a = &db->maps[m];
        for (uint64_t i = 0; i < a->cap; ++i) {                       /* src/graph-builder.cpp:245-258 */
            if (a->keys[i] == KQO_EMPTY) continue;
            const kqo_kmer8* e = (const kqo_kmer8*)(a->vals + i * a->vsz);
            if (e->cov == 255) continue;                              /* :247 */
            if (e->cov == 1) ++uniq;                                  /* :250 */
            for (int w = 0; w < 4; ++w)                               /* :254 parses as fw>0 ? 1 : (bw>0 ? 1 : 0) */
                edges += e->fw[w] > 0 ? 1 : (0 + e->bw[w] > 0 ? 1 : 0);
            ++distinct;
            ++*(uint64_t*)map_at(&hist, e->cov);
        }
        const kqo_map* b = &db->maps32[m];
        for (uint64_t i = 0; i < b->cap; ++i) {                       /* :260-267 */
            if (b->keys[i] == KQO_EMPTY) continue;
            const kqo_kmer32* e = (const kqo_kmer32*)(b->vals + i * b->vsz);
            for (int w = 0; w < 4; ++w)
                edges += e->fw[w] > 0 ? 1 : (0 + e->bw[w] > 0 ? 1 : 0);
            ++distinct;
            ++*(uint64_t*)map_at(&hist, e->cov);
        }
    }
    uint64_t nh = 0;
    uint64_t* covs = (uint64_t*)malloc((hist.size + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < hist.cap; ++i) if (hist.keys[i] != KQO_EMPTY) covs[nh++] = hist.keys[i];
    qsort(covs, nh, sizeof(uint64_t), cmp_u64);
    for (uint64_t i = 0; i < nh; ++i) {
        uint64_t cnt = *(uint64_t*)map_find(&hist, covs[i]);
        total += covs[i] * cnt;                                       /* :274-278 */
        if (hist_cov && hist_cnt && i < hist_cap) { hist_cov[i] = covs[i]; hist_cnt[i] = cnt; }
    }
    if (hist_n) *hist_n = nh;
    free(covs); map_free(&hist);
    if (out) {
        out->total = total; out->unique = uniq; out->distinct = distinct; out->edges = edges;
        /* :286 uint64_t missing = pow(4,k) - totDistinct  (k==32: pow overflows u64 -> 0 on x86-64) */
        uint64_t space = db->k < 32 ? (1ull << (2 * db->k)) : 0ull;
        out->missing = space - distinct;
    }
    return 0;
}

/* ------------------------------------------------------------------ lookup / QV */
double kqo_error_rate(uint64_t missing, uint64_t total, int k) {      /* src/kreeq.cpp:36-40 */
    return 1 - pow(1 - (double)missing / total, (double)1 / k);
}
double kqo_qv(uint64_t missing, uint64_t total, int k) {              /* src/kreeq.cpp:87,96 */
    return -10 * log10(kqo_error_rate(missing, total, k));
}

/* DBG::evaluateSegment src/kreeq.cpp:110-229; str must be all <=3 */
static void eval_segment(const kqo_db* db, const uint8_t* str, uint64_t len, uint64_t c_begin, uint64_t c_end,
                         uint32_t cov_cutoff, uint16_t lo, uint16_t hi, kqo_dbgbase* out, uint64_t ctr[3]) {
    int k = db->k;
    if (len < (uint64_t)k) return;                                    /* :123 */
    uint64_t kcount = len - k + 1, kmers = 0, missing = 0, edge_missing = 0;
    if (c_end > kcount) c_end = kcount;
    /* [c_begin, c_end) is the whole segment in the reference; a sub-range only splits the loop
     * across worker threads, each k-mer still sees its true neighbours in the segment */
    for (uint64_t c = c_begin; c < c_end; ++c) {                      /* :143 */
        int is_fw;
        uint64_t key = kqo_hash(str + c, k, &is_fw);                  /* :145 */
        uint64_t i = key % (uint64_t)db->map_count;                   /* :146 */
        if (!(i >= lo && i < hi)) continue;                           /* :150 */
        kqo_kmer32 kh; memset(&kh, 0, sizeof kh);                     /* :154 */
        kqo_dbgbase b; memset(&b, 0, sizeof b);
        if (out) b = out[c];
        const kqo_kmer8* e8 = (const kqo_kmer8*)map_find(&db->maps[i], key);   /* :153 */
        if (e8) {
            for (int w = 0; w < 4; ++w) { kh.fw[w] = e8->fw[w]; kh.bw[w] = e8->bw[w]; }
            kh.cov = e8->cov;
            if (kh.cov == 255) {                                      /* :158-166 */
                const kqo_kmer32* e32 = (const kqo_kmer32*)map_find(&db->maps32[i], key);
                if (!e32) abort();    /* "int32 map missing 255 value from int8 map" :161-164 */
                kh = *e32;
            }
            b.cov = kh.cov; b.isFw = (uint8_t)is_fw;                  /* :168-169 */
        }
        if (b.cov == 0) ++missing;                                    /* :172 */
        else if (b.cov < cov_cutoff) ++missing;                       /* :174 */
        else {
            int no_left = 0, no_right = 0;                            /* :177 */
            if (b.isFw) {                                             /* :178-193 */
                if (c < kcount - 1) { if (kh.fw[str[c + k]] != 0) b.fw = kh.fw[str[c + k]]; else no_right = 1; }
                if (c > 0)          { if (kh.bw[str[c - 1]] != 0) b.bw = kh.bw[str[c - 1]]; else no_left = 1; }
            } else {                                                  /* :194-210 */
                if (c > 0)          { if (kh.fw[3 - str[c - 1]] != 0) b.fw = kh.fw[3 - str[c - 1]]; else no_left = 1; }
                if (c < kcount - 1) { if (kh.bw[3 - str[c + k]] != 0) b.bw = kh.bw[3 - str[c + k]]; else no_right = 1; }
            }
            if (no_left && no_right) ++edge_missing;                  /* :211 */
        }
        if (out) out[c] = b;
        ++kmers;                                                      /* :216 */
    }
    ctr[0] += missing; ctr[1] += kmers; ctr[2] += edge_missing;       /* :223-225 */
}

typedef struct { uint64_t off, len, c0, c1; } seg_t;
typedef struct {
    const kqo_db* db; const uint8_t* str; seg_t* segs; uint64_t n_segs; uint64_t next;
    uint32_t cutoff; uint16_t lo, hi; kqo_dbgbase* out; uint64_t ctr[3]; pthread_mutex_t mtx;
} seg_ctx;
static void* seg_worker(void* p) {
    seg_ctx* c = (seg_ctx*)p;
    uint64_t local[3] = {0, 0, 0};
    for (;;) {
        pthread_mutex_lock(&c->mtx); uint64_t s = c->next++; pthread_mutex_unlock(&c->mtx);
        if (s >= c->n_segs) break;
        eval_segment(c->db, c->str + c->segs[s].off, c->segs[s].len, c->segs[s].c0, c->segs[s].c1,
                     c->cutoff, c->lo, c->hi, c->out ? c->out + c->segs[s].off : NULL, local);
    }
    pthread_mutex_lock(&c->mtx);
    for (int i = 0; i < 3; ++i) c->ctr[i] += local[i];
    pthread_mutex_unlock(&c->mtx);
    return NULL;
}
static int run_segments(const kqo_db* db, const uint8_t* str, seg_t* segs, uint64_t n_segs,
                        uint32_t cutoff, uint16_t lo, uint16_t hi, kqo_dbgbase* out,
                        uint64_t counters[3], int threads) {
    seg_ctx c; memset(&c, 0, sizeof c);
    c.db = db; c.str = str; c.segs = segs; c.n_segs = n_segs; c.cutoff = cutoff; c.lo = lo; c.hi = hi; c.out = out;
    pthread_mutex_init(&c.mtx, NULL);
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > n_segs) threads = n_segs ? (int)n_segs : 1;
    pthread_t* th = (pthread_t*)calloc(threads, sizeof *th);
    for (int t = 0; t < threads; ++t) pthread_create(&th[t], NULL, seg_worker, &c);
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    free(th); pthread_mutex_destroy(&c.mtx);
    for (int i = 0; i < 3; ++i) counters[i] += c.ctr[i];
    return 0;
}

int kqo_lookup_segment(const kqo_db* db, const char* bases, uint64_t len, uint32_t cov_cutoff,
                       uint16_t map_lo, uint16_t map_hi, kqo_dbgbase* per_base,
                       uint64_t counters[3], int threads) {
    (void)threads;
    uint8_t* str = (uint8_t*)malloc(len ? len : 1);
    for (uint64_t i = 0; i < len; ++i) { str[i] = g_ctoi[(uint8_t)bases[i]]; if (str[i] > 3) { free(str); return -1; } }   /* :131-132 */
    eval_segment(db, str, len, 0, len, cov_cutoff, map_lo, map_hi, per_base, counters);
    free(str);
    return 0;
}

int kqo_validate_sequence(const kqo_db* db, const char* bases, uint64_t len, uint32_t cov_cutoff,
                          uint16_t map_lo, uint16_t map_hi, kqo_dbgbase* per_base,
                          uint64_t counters[3], int threads) {
    uint8_t* str = (uint8_t*)malloc(len ? len : 1);
    uint64_t n_segs = 0, cap = 16;
    seg_t* segs = (seg_t*)malloc(cap * sizeof *segs);
    for (uint64_t i = 0; i < len; ++i) str[i] = g_ctoi[(uint8_t)bases[i]];
    const uint64_t tile = 1u << 20;   /* k-mer starts per job */
    for (uint64_t i = 0; i < len;) {
        if (str[i] > 3) { ++i; continue; }
        uint64_t j = i; while (j < len && str[j] <= 3) ++j;
        for (uint64_t c0 = 0; c0 == 0 || c0 + db->k <= j - i; c0 += tile) {
            if (n_segs == cap) { cap *= 2; segs = (seg_t*)realloc(segs, cap * sizeof *segs); }
            segs[n_segs].off = i; segs[n_segs].len = j - i; segs[n_segs].c0 = c0; segs[n_segs].c1 = c0 + tile; ++n_segs;
        }
        i = j;
    }
    int rc;
    rc = run_segments(db, str, segs, n_segs, cov_cutoff, map_lo, map_hi, per_base, counters, threads);
    free(segs); free(str);
    return rc;
}

/* ------------------------------------------------------------------ union */
static inline uint32_t sat_add32(uint32_t a, uint32_t b) {            /* src/graph-builder.cpp:316-329,415-428 */
    return (KQO_LARGEST - a >= b) ? a + b : KQO_LARGEST;
}
/* DBG::kunion (:297-351) + DBG::mergeSubMaps (:353-432), with map1 = src map, map2 = dst map.
 * The reference first sums all high-copy maps (:299-336), then merges 8-bit maps. */
int kqo_merge(kqo_db* dst, const kqo_db* src) {
    if (dst->k != src->k || dst->map_count != src->map_count) return -1;
    for (int m = 0; m < src->map_count; ++m) {
        const kqo_map* h = &src->maps32[m];
        for (uint64_t i = 0; i < h->cap; ++i) {                       /* :310-330 */
            if (h->keys[i] == KQO_EMPTY) continue;
            const kqo_kmer32* s = (const kqo_kmer32*)(h->vals + i * h->vsz);
            kqo_kmer32* d = (kqo_kmer32*)map_at(&dst->maps32[m], h->keys[i]);
            int first = (d->cov == 0);
            if (first) {
                /* dst may hold this key as a plain 8-bit entry: fold it in and tombstone it, so the
                 * result does not depend on which database is merged into which (the reference's
                 * outcome for this corner depends on gfalibs' merge order) */
                kqo_kmer8* e8 = (kqo_kmer8*)map_at(&dst->maps[m], h->keys[i]);
                if (e8->cov != 255) {
                    for (int w = 0; w < 4; ++w) { d->fw[w] = e8->fw[w]; d->bw[w] = e8->bw[w]; }
                    d->cov = e8->cov;
                }
                memset(e8, 0, sizeof *e8); e8->cov = 255;
            }
            for (int w = 0; w < 4; ++w) { d->fw[w] = sat_add32(d->fw[w], s->fw[w]); d->bw[w] = sat_add32(d->bw[w], s->bw[w]); }
            d->cov = sat_add32(d->cov, s->cov);
        }
    }
    for (int m = 0; m < src->map_count; ++m) {
        const kqo_map* a = &src->maps[m];
        kqo_map* map2 = &dst->maps[m];
        kqo_map* map32 = &dst->maps32[m];
        for (uint64_t i = 0; i < a->cap; ++i) {                       /* :361 */
            if (a->keys[i] == KQO_EMPTY) continue;
            uint64_t key = a->keys[i];
            kqo_kmer8 p = *(const kqo_kmer8*)(a->vals + i * a->vsz);
            int overflow = 0;
            if (p.cov == 255) continue;                               /* :365 */
            if (map_find(map32, key)) overflow = 1;                   /* :368-370 */
            else {
                kqo_kmer8* got = (kqo_kmer8*)map_find(map2, key);     /* :373 */
                if (!got) { *(kqo_kmer8*)map_at(map2, key) = p; }     /* :375 */
                else {
                    if (255 - got->cov <= p.cov) overflow = 1;        /* :380 */
                    for (int w = 0; w < 4; ++w)                       /* :383-389 */
                        if (255 - got->fw[w] <= p.fw[w] || 255 - got->bw[w] <= p.bw[w]) { overflow = 1; break; }
                    if (!overflow) {                                  /* :391-398 */
                        for (int w = 0; w < 4; ++w) { got->fw[w] += p.fw[w]; got->bw[w] += p.bw[w]; }
                        got->cov += p.cov;
                    }
                }
            }
            if (overflow) {                                           /* :402-429 */
                kqo_kmer32* d = (kqo_kmer32*)map_at(map32, key);
                if (d->cov == 0) {
                    kqo_kmer8* got = (kqo_kmer8*)map_find(map2, key);
                    for (int w = 0; w < 4; ++w) { d->fw[w] = got->fw[w]; d->bw[w] = got->bw[w]; }
                    d->cov = got->cov;
                    got->cov = 255;
                }
                for (int w = 0; w < 4; ++w) { d->fw[w] = sat_add32(d->fw[w], p.fw[w]); d->bw[w] = sat_add32(d->bw[w], p.bw[w]); }
                d->cov = sat_add32(d->cov, p.cov);
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ export / import */
static int cmp_entry(const void* a, const void* b) {
    uint64_t x = ((const kqo_entry*)a)->key, y = ((const kqo_entry*)b)->key;
    return x < y ? -1 : x > y;
}
uint64_t kqo_export(const kqo_db* db, int map, kqo_entry* out, uint64_t cap) {
    uint64_t n = 0;
    int m0 = map < 0 ? 0 : map, m1 = map < 0 ? db->map_count : map + 1;
    for (int m = m0; m < m1; ++m) {
        const kqo_map* a = &db->maps[m];
        for (uint64_t i = 0; i < a->cap; ++i) {
            if (a->keys[i] == KQO_EMPTY) continue;
            const kqo_kmer8* e = (const kqo_kmer8*)(a->vals + i * a->vsz);
            if (e->cov == 255) continue;
            if (out && n < cap) {
                kqo_entry* o = &out[n]; o->key = a->keys[i]; o->hc = 0; o->cov = e->cov;
                for (int w = 0; w < 4; ++w) { o->fw[w] = e->fw[w]; o->bw[w] = e->bw[w]; }
            }
            ++n;
        }
        const kqo_map* b = &db->maps32[m];
        for (uint64_t i = 0; i < b->cap; ++i) {
            if (b->keys[i] == KQO_EMPTY) continue;
            const kqo_kmer32* e = (const kqo_kmer32*)(b->vals + i * b->vsz);
            if (out && n < cap) {
                kqo_entry* o = &out[n]; o->key = b->keys[i]; o->hc = 1; o->cov = e->cov;
                for (int w = 0; w < 4; ++w) { o->fw[w] = e->fw[w]; o->bw[w] = e->bw[w]; }
            }
            ++n;
        }
    }
    if (out) qsort(out, n < cap ? n : cap, sizeof(kqo_entry), cmp_entry);
    return n;
}
uint64_t kqo_export_raw8(const kqo_db* db, int map, uint64_t* keys, kqo_kmer8* vals, uint64_t cap) {
    uint64_t n = 0;
    const kqo_map* a = &db->maps[map];
    for (uint64_t i = 0; i < a->cap; ++i) {
        if (a->keys[i] == KQO_EMPTY) continue;
        if (keys && vals && n < cap) { keys[n] = a->keys[i]; vals[n] = *(const kqo_kmer8*)(a->vals + i * a->vsz); }
        ++n;
    }
    return n;
}
int kqo_import(kqo_db* db, const kqo_entry* in, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t m = in[i].key % (uint64_t)db->map_count;
        if (in[i].hc) {
            kqo_kmer32* d = (kqo_kmer32*)map_at(&db->maps32[m], in[i].key);
            for (int w = 0; w < 4; ++w) { d->fw[w] = in[i].fw[w]; d->bw[w] = in[i].bw[w]; }
            d->cov = in[i].cov;
            kqo_kmer8* e8 = (kqo_kmer8*)map_at(&db->maps[m], in[i].key);
            memset(e8, 0, sizeof *e8); e8->cov = 255;                 /* reloadMap32 :230-236 */
        } else {
            if (in[i].cov >= 255) return -1;
            kqo_kmer8* e8 = (kqo_kmer8*)map_at(&db->maps[m], in[i].key);
            for (int w = 0; w < 4; ++w) { e8->fw[w] = (uint8_t)in[i].fw[w]; e8->bw[w] = (uint8_t)in[i].bw[w]; }
            e8->cov = (uint8_t)in[i].cov;
        }
    }
    return 0;
}

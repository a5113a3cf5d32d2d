// Candidate-error search of `kreeq validate -o vcf` (reference DBG::correctSequences / DBGtoVariants / searchVariants,
// src/variants.cpp:40-310) on top of the GPU table.
//
// Division of labour: the device answers "is this k-mer in the graph, and does the graph branch away from the sequence
// here?" for every position of the assembly in one pass (kq_branch_scan), so the bounded graph search -- pointer chasing,
// serial per source k-mer -- runs on the host only at the few positions where it can find something.  The searches of a
// batch advance in lockstep: each runs until it needs graph nodes that are not cached, the needed keys of all of them go
// to the device in ONE kq_lookup_keys call, and they resume.  Visiting order, distances, path reconstruction and the
// variant typing follow the reference statement by statement (line numbers cited below), including the order in which
// its Fibonacci heap (include/fibonacci-heap.h) hands out nodes of equal key.
//
// The VCF text is written by gfalibs' Report class in the reference, which is absent from the reference tree; the record
// layout here is inferred from its one golden file (validateFiles/test.50.tst) and documented at vcf_record().
#include "variants.h"

#include <algorithm>
#include <chrono>
#include <deque>
#include <stdexcept>
#include <thread>
#include <tuple>

namespace kqhost {

namespace {

const char kItoc[4] = {'A', 'C', 'G', 'T'};
inline int ctoi(char c) {
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
inline char rev_com(char c) { return kItoc[3 - ctoi(c)]; }

// gfalibs Kmap::hash: canonical 2-bit key of k base codes, first base in the low bits
uint64_t hash_kmer(const uint8_t* b, int k, bool* is_fw) {
    uint64_t fw = 0, rv = 0;
    for (int c = 0; c < k; ++c) { fw |= (uint64_t)b[c] << (2 * c); rv |= (uint64_t)(3 - b[c]) << (2 * (k - 1 - c)); }
    if (is_fw) *is_fw = fw < rv;
    return fw < rv ? fw : rv;
}
std::string reverse_hash(uint64_t key, int k) {
    std::string s((size_t)k, 'A');
    for (int c = 0; c < k; ++c) s[(size_t)c] = kItoc[(key >> (2 * c)) & 3];
    return s;
}
// DBG::buildNextKmer, src/subgraph.cpp:581-598: the k-mer one step along edge `base` of the canonical string of `key`
uint64_t next_key(uint64_t key, int base, bool fw, int k, bool* is_fw) {
    uint8_t codes[33];
    if (fw) { for (int c = 0; c + 1 < k; ++c) codes[c] = (uint8_t)((key >> (2 * (c + 1))) & 3); codes[k - 1] = (uint8_t)base; }
    else    { codes[0] = (uint8_t)base; for (int c = 1; c < k; ++c) codes[c] = (uint8_t)((key >> (2 * (c - 1))) & 3); }
    return hash_kmer(codes, k, is_fw);
}

// u64 -> T with open addressing (linear probing, power-of-two capacity, grown at load 1/2; no allocation before the first insert).
// The search never iterates over its maps, so any container with find / insert gives the reference's result; the node-based
// std::unordered_map cost 0.5 us per cached graph node and as much again to destroy (100 x the HiFi test: 5.7 of 9 s).
template <class T>
class FlatMap {
    std::vector<uint64_t> keys_;
    std::vector<T> vals_;
    std::vector<uint8_t> used_;
    size_t n_ = 0;
    static size_t mix(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 29; return (size_t)k; }
    size_t slot(uint64_t k) const {                                     // the key's slot, or the free slot where it would go
        size_t i = mix(k) & (keys_.size() - 1);
        while (used_[i] && keys_[i] != k) i = (i + 1) & (keys_.size() - 1);
        return i;
    }
    void grow() {
        const size_t cap = keys_.empty() ? 16 : keys_.size() * 2;
        std::vector<uint64_t> ok(cap); std::vector<T> ov(cap); std::vector<uint8_t> ou(cap, 0);
        ok.swap(keys_); ov.swap(vals_); ou.swap(used_);
        for (size_t i = 0; i < ok.size(); ++i) if (ou[i]) { const size_t j = slot(ok[i]); keys_[j] = ok[i]; vals_[j] = std::move(ov[i]); used_[j] = 1; }
    }
public:
    size_t size() const { return n_; }
    void reserve(size_t n) { while (keys_.size() < 2 * n) grow(); }
    const T* find(uint64_t k) const { if (keys_.empty()) return nullptr; const size_t i = slot(k); return used_[i] ? &vals_[i] : nullptr; }
    T* find(uint64_t k) { if (keys_.empty()) return nullptr; const size_t i = slot(k); return used_[i] ? &vals_[i] : nullptr; }
    bool count(uint64_t k) const { return find(k) != nullptr; }
    T& operator[](uint64_t k) {                                         // inserts a value-initialised T when absent
        if (2 * (n_ + 1) > keys_.size()) grow();
        const size_t i = slot(k);
        if (!used_[i]) { used_[i] = 1; keys_[i] = k; vals_[i] = T(); ++n_; }
        return vals_[i];
    }
    void clear() { std::fill(used_.begin(), used_.end(), 0); n_ = 0; }
    void release() { std::vector<uint64_t>().swap(keys_); std::vector<T>().swap(vals_); std::vector<uint8_t>().swap(used_); n_ = 0; }
};

// The reference's priority queue, include/fibonacci-heap.h, on an index pool.  The search inserts every node except
// the source with key 0 and its decreaseKey refuses to raise a key (:141), so which of several queued nodes comes out
// next is decided by the shape of the root list alone: insert links a node left of the minimum (:72-80), extractMin
// promotes the children, steps to the right neighbour and consolidates equal degrees (:87-126, :218-268).
class NodeQueue {
    struct N { int degree, parent, child, left, right, key; bool mark; uint64_t obj; };
    std::vector<N> n_;
    std::vector<int> deg_;
    int min_ = -1, count_ = 0;

    void to_root(int x) {                                                // _existingToRoot :145-164
        n_[x].parent = -1; n_[x].mark = false;
        if (min_ >= 0) {
            const int ml = n_[min_].left;
            n_[min_].left = x; n_[x].right = min_; n_[x].left = ml; n_[ml].right = x;
            if (n_[min_].key > n_[x].key) min_ = x;
        } else { min_ = x; n_[x].left = n_[x].right = x; }
    }
    void unlink(int x) {                                                 // _removeNodeFromRoot :165-179
        if (n_[x].right != x) { n_[n_[x].right].left = n_[x].left; n_[n_[x].left].right = n_[x].right; }
        const int p = n_[x].parent;
        if (p >= 0) {
            n_[p].child = n_[p].degree == 1 ? -1 : n_[x].right;
            --n_[p].degree;
        }
    }
    void add_child(int p, int c) {                                       // _addChild :184-201
        if (n_[p].degree == 0) { n_[p].child = c; n_[c].left = n_[c].right = c; }
        else { const int c1 = n_[p].child, l = n_[c1].left; n_[c1].left = c; n_[c].right = c1; n_[c].left = l; n_[l].right = c; }
        n_[c].parent = p; ++n_[p].degree;
    }
    void consolidate() {                                                 // :218-268
        if (count_ <= 1) return;
        deg_.clear();
        int roots = 0, it = min_;
        do { ++roots; it = n_[it].right; } while (it != min_);
        int cur = min_;
        for (int r = 0; r < roots; ++r) {
            int x = cur;
            cur = n_[cur].right;
            int d = n_[x].degree;
            for (;;) {
                while (d >= (int)deg_.size()) deg_.push_back(-1);
                if (deg_[(size_t)d] < 0) { deg_[(size_t)d] = x; break; }
                int y = deg_[(size_t)d];
                if (n_[x].key > n_[y].key) std::swap(x, y);
                if (y == x) break;
                unlink(y); add_child(x, y); n_[y].mark = false;          // _link
                deg_[(size_t)d] = -1;
                ++d;
            }
        }
        min_ = -1;
        for (int x : deg_) if (x >= 0) to_root(x);
    }
public:
    int size() const { return count_; }
    void insert(uint64_t obj, int key) {                                 // :57-86
        const int x = (int)n_.size();
        n_.push_back(N{0, -1, -1, x, x, key, false, obj});
        if (min_ >= 0) { const int ml = n_[min_].left; n_[min_].left = x; n_[x].right = min_; n_[x].left = ml; n_[ml].right = x; }
        if (min_ < 0 || n_[min_].key > key) min_ = x;
        ++count_;
    }
    uint64_t extract_min() {                                             // :87-126
        const int m = min_;
        int c = n_[m].child;
        for (int i = 0, d = n_[m].degree; i < d; ++i) { const int rem = c; c = n_[c].right; to_root(rem); }
        unlink(m);
        --count_;
        if (count_ == 0) min_ = -1;
        else {
            min_ = n_[m].right;
            const int ml = n_[m].left;
            n_[min_].left = ml; n_[ml].right = min_;
            consolidate();
        }
        return n_[m].obj;
    }
    // decreaseKey (:127-142) only ever sees new keys >= 1 for nodes inserted with key 0 here: it returns at :141
};

struct Node { uint32_t fw[4], bw[4]; bool present; };

struct Search {
    // what DBGtoVariants hands to searchVariants (:136)
    uint64_t seg_id = 0, c = 0, source = 0, ref = 0;
    bool source_fw = false, has_ref = false;
    std::vector<uint64_t> targets_queue;
    std::vector<uint64_t> targets;                                       // targetsMap: at most maxSpan keys, sorted
    bool is_target(uint64_t key) const { return std::binary_search(targets.begin(), targets.end(), key); }
    // searchVariants' locals (:173-185)
    NodeQueue Q;
    FlatMap<uint8_t> dist;
    FlatMap<std::pair<uint64_t, bool>> prev;
    std::vector<uint64_t> destinations;
    int depth = 0;
    bool direction = true, started = false, done = false;
    // the node being expanded
    uint64_t u = 0;
    std::vector<std::tuple<uint64_t, bool, bool>> cand;
    bool have_u = false;
    std::vector<DbgPath> paths;
};

using Cache = FlatMap<Node>;

// Runs one search until it needs uncached nodes (their keys are appended to `want`) or is finished.
void advance(Search& s, const Cache& cache, int k, int kmer_depth, uint32_t cov_cutoff, std::vector<uint64_t>& want) {
    if (s.done) return;
    if (!s.started) {
        if (!cache.count(s.source)) { want.push_back(s.source); return; }
        s.dist[s.source] = 1;                                            // :180-181
        s.Q.insert(s.source, 1);
        s.started = true;
    }
    for (;;) {
        if (!s.have_u) {
            if (!(s.Q.size() > 0 && s.depth < kmer_depth + 1)) break;   // :187
            s.u = s.Q.extract_min();                                     // :192
            const auto* got = s.prev.find(s.u);                          // :193-196
            if (got) s.direction = got->second;
            const Node& nu = *cache.find(s.u);
            s.cand.clear();
            for (int i = 0; i < 4; ++i) {                                // :232-246
                if (s.depth == 0) s.direction = s.source_fw;
                // `direction ? fw[i] : bw[i] > covCutOff` == direction ? (fw[i] != 0) : (bw[i] > covCutOff)   (:237)
                const bool edge = s.direction ? nu.fw[i] != 0 : nu.bw[i] > cov_cutoff;
                if (!edge) continue;
                bool is_fw = false;
                const uint64_t key = next_key(s.u, i, s.direction, k, &is_fw);
                if (s.has_ref && key == s.ref) continue;                 // :241: the reference path is never rediscovered
                s.cand.emplace_back(key, is_fw, s.direction);
            }
            s.have_u = true;
        }
        bool missing = false;
        for (auto& cnd : s.cand) {
            const uint64_t key = std::get<0>(cnd);
            if (!s.is_target(key) && !cache.count(key)) { want.push_back(key); missing = true; }
        }
        if (missing) return;                                             // resumed after the batch lookup
        size_t explored_count = 0;
        for (auto& cnd : s.cand) {                                       // :247-260 with checkNext :197-228
            const uint64_t key = std::get<0>(cnd);
            const bool dirn = std::get<2>(cnd), cont = std::get<1>(cnd) ? dirn : !dirn;
            if (!s.is_target(key)) {
                uint8_t alt = s.dist[s.u];
                if (alt < 255) ++alt;
                if (!s.dist.count(key)) { s.dist[key] = 255; s.Q.insert(key, 0); }
                if (alt < s.dist[key]) { s.prev[key] = std::make_pair(s.u, cont); s.dist[key] = alt; }
            }
            ++explored_count;
            if (s.is_target(key)) { s.prev[key] = std::make_pair(s.u, dirn); s.destinations.push_back(key); }
        }
        (void)explored_count;                                            // every candidate is reachable (one map range): edgeCount == exploredCount
        ++s.depth;                                                       // :261
        s.have_u = false;
    }
    // paths from the destinations back to the source (:266-303)
    auto prev_of = [&](uint64_t key) { const auto* it = s.prev.find(key); return it ? *it : std::make_pair((uint64_t)0, false); };
    for (uint64_t dest : s.destinations) {
        DbgPath path;
        path.pos = s.c + (uint64_t)k;                                    // :139-140
        const auto where = std::find(s.targets_queue.begin(), s.targets_queue.end(), dest);
        const int ref_len = (int)(where - s.targets_queue.begin()) + k;
        int i = 0;
        uint64_t node = prev_of(dest).first;
        while (node != s.source) { node = prev_of(node).first; ++i; }
        node = prev_of(dest).first;
        bool dirn = prev_of(node).second;
        int b = i - ref_len;
        if (ref_len > k) { path.type = VAR_COM; path.ref_len = (uint32_t)(ref_len - k + 1); b = ref_len - k; }
        else if (i == ref_len) path.type = VAR_SNV;
        else if (i > ref_len) { path.type = VAR_DEL; --b; node = prev_of(node).first; dirn = prev_of(node).second; }
        else path.type = VAR_INS;
        while (b >= 0) {
            const std::string str = reverse_hash(node, k);
            path.sequence.push_back(dirn ? str[0] : rev_com(str[(size_t)k - 1]));
            node = prev_of(node).first;
            dirn = prev_of(node).second;
            --b;
        }
        std::reverse(path.sequence.begin(), path.sequence.end());
        s.paths.push_back(std::move(path));
    }
    s.done = true;
    // only the paths are read from here on
    s.dist.release(); s.prev.release(); s.Q = NodeQueue();
    std::vector<uint64_t>().swap(s.targets_queue); std::vector<uint64_t>().swap(s.targets); std::vector<uint64_t>().swap(s.destinations);
}

}  // namespace

GraphSource graph_of_handle(kq_handle* h) {
    GraphSource g;
    g.branch_scan = [h](const std::string& joined, uint32_t cov_cutoff, std::vector<uint8_t>& flags) {
        if (kq_branch_scan(h, joined.data(), joined.size(), cov_cutoff, flags.data()) != KQ_OK) throw std::runtime_error(std::string("Error: ") + kq_last_error());
    };
    g.lookup = [h](const std::vector<uint64_t>& want, std::vector<kq_entry>& got) {
        if (kq_lookup_keys(h, want.data(), want.size(), got.data()) != KQ_OK) throw std::runtime_error(std::string("Error: ") + kq_last_error());
    };
    return g;
}

std::vector<VariantSite> find_candidate_errors(const GraphSource& g, int k, const std::vector<SeqRecord>& seqs, int kmer_depth, int max_span,
                                               uint32_t cov_cutoff, const std::function<void(const std::string&)>& log) {
    // 1. device pre-filter over the whole assembly (sequences joined by a non-base byte)
    std::string joined;
    std::vector<uint64_t> offset;
    for (auto& r : seqs) { offset.push_back(joined.size()); joined += r.seq; joined.push_back('\n'); }
    std::vector<uint8_t> flags(joined.size());
    g.branch_scan(joined, cov_cutoff, flags);

    // 2. one Search per flagged position, segment by segment (DBGtoVariants :53-169, single map range)
    std::vector<Search> searches;
    std::vector<std::tuple<size_t, uint64_t>> seg_of;                    // segment id -> (sequence index, start inside the sequence)
    uint64_t n_pos = 0;
    for (size_t si = 0; si < seqs.size(); ++si) {
        const std::string& seq = seqs[si].seq;
        for (uint64_t i = 0; i < seq.size();) {                          // segments = runs of bases (gfalibs splits at N)
            if (ctoi(seq[i]) > 3) { ++i; continue; }
            uint64_t j = i;
            while (j < seq.size() && ctoi(seq[j]) <= 3) ++j;
            const uint64_t len = j - i;
            if (len >= (uint64_t)k) {
                const uint64_t kcount = len - k + 1;
                std::vector<uint8_t> str(len);
                for (uint64_t p = 0; p < len; ++p) str[p] = (uint8_t)ctoi(seq[i + p]);
                const uint64_t seg_id = seg_of.size();
                seg_of.emplace_back(si, i);
                std::vector<uint64_t> seg_keys;                          // canonical key of every k-mer of the segment (filled on first use)
                auto fill_keys = [&] { if (seg_keys.empty()) { seg_keys.resize(kcount); for (uint64_t t = 0; t < kcount; ++t) seg_keys[t] = hash_kmer(str.data() + t, k, nullptr); } };
                const uint8_t* f = flags.data() + offset[si] + i;
                n_pos += kcount;
                for (uint64_t c = 0; c < kcount; ++c) {
                    if ((f[c] & 3) != 3) continue;                       // absent (:149-152) or no candidate at the source: nothing to find
                    fill_keys();
                    Search s;
                    s.seg_id = seg_id; s.c = c;
                    s.source = hash_kmer(str.data() + c, k, &s.source_fw);       // :114; isFw of k-mer c is what :136 passes on
                    if (c + 1 < kcount) { s.ref = hash_kmer(str.data() + c + 1, k, nullptr); s.has_ref = true; }
                    // the targets while c is processed: the k-mers starting at c+k+1 .. c+k+maxSpan (:92-110).  targetsMap
                    // loses a key when ANY occurrence of it leaves the window (:104) and gets it back with the next push
                    // (:108): a key is in the map iff no occurrence left after its latest one entered
                    const uint64_t w_lo = c + k + 1, w_hi = std::min<uint64_t>(c + k + (uint64_t)max_span, kcount - 1);
                    for (uint64_t t = w_lo; t <= w_hi; ++t) s.targets_queue.push_back(seg_keys[t]);
                    for (uint64_t t = w_lo; t <= w_hi; ++t) {
                        const uint64_t key = seg_keys[t];
                        uint64_t t_max = t;
                        for (uint64_t t2 = t + 1; t2 <= w_hi; ++t2) if (seg_keys[t2] == key) t_max = t2;
                        bool erased = false;                             // an occurrence popped at or after the push of t_max
                        const uint64_t from = t_max >= (uint64_t)max_span ? t_max - (uint64_t)max_span + 1 : 0;
                        for (uint64_t t1 = std::max<uint64_t>(from, (uint64_t)k); t1 <= c + k && !erased; ++t1) erased = seg_keys[t1] == key;
                        if (!erased) s.targets.push_back(key);
                    }
                    std::sort(s.targets.begin(), s.targets.end());
                    s.targets.erase(std::unique(s.targets.begin(), s.targets.end()), s.targets.end());
                    searches.push_back(std::move(s));
                }
            }
            i = j;
        }
    }
    if (log) log("Candidate positions after the device pre-filter: " + std::to_string(searches.size()) + " of " + std::to_string(n_pos));
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_adv = 0, t_sort = 0, t_look = 0, t_cache = 0;
    size_t n_rounds = 0, n_keys = 0;

    // 3. lockstep rounds: advance every search, fetch what they ask for in one batch
    Cache cache;
    std::vector<uint64_t> want;
    std::vector<kq_entry> got;
    const size_t kBatch = 1 << 16;
    for (size_t lo = 0; lo < searches.size(); lo += kBatch) {
        const size_t hi = std::min(searches.size(), lo + kBatch);
        for (;;) {
            want.clear();
            bool any = false;
            // the searches of a round are independent (each reads the cache and its own state): a few host threads share
            // them when there are enough (HiFi-scale inputs flag tens of thousands of positions per batch)
            const size_t n_live = hi - lo;
            double t0 = now();
            unsigned n_thr = std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
            if (n_live < 512) n_thr = 1;
            if (n_thr > 1) {
                std::vector<std::vector<uint64_t>> wants(n_thr);
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < n_thr; ++t)
                    pool.emplace_back([&, t] {
                        const size_t a = lo + n_live * t / n_thr, b = lo + n_live * (t + 1) / n_thr;
                        for (size_t i = a; i < b; ++i) if (!searches[i].done) advance(searches[i], cache, k, kmer_depth, cov_cutoff, wants[t]);
                    });
                for (auto& th : pool) th.join();
                for (auto& w : wants) want.insert(want.end(), w.begin(), w.end());
            } else {
                for (size_t i = lo; i < hi; ++i) if (!searches[i].done) advance(searches[i], cache, k, kmer_depth, cov_cutoff, want);
            }
            for (size_t i = lo; i < hi && !any; ++i) any = !searches[i].done;
            t_adv += now() - t0; t0 = now();
            if (!any) break;
            std::sort(want.begin(), want.end());
            want.erase(std::unique(want.begin(), want.end()), want.end());
            if (want.empty()) throw std::runtime_error("candidate-error search stalled");
            t_sort += now() - t0; t0 = now();
            got.resize(want.size());
            g.lookup(want, got);
            t_look += now() - t0; t0 = now();
            cache.reserve(cache.size() + got.size());
            for (auto& e : got) {
                Node& n = cache[e.key];
                for (int w = 0; w < 4; ++w) { n.fw[w] = e.fw[w]; n.bw[w] = e.bw[w]; }
                n.present = e.cov != 0;
            }
            t_cache += now() - t0;
            ++n_rounds; n_keys += want.size();
        }
        if (cache.size() > (1u << 24)) cache.clear();                    // bounded memory on large assemblies
    }

    if (log) log("Search rounds: " + std::to_string(n_rounds) + ", keys fetched " + std::to_string(n_keys) + "; advance " + std::to_string(t_adv) + " s, sort " + std::to_string(t_sort) +
                 " s, lookup " + std::to_string(t_look) + " s, cache " + std::to_string(t_cache) + " s");
    // 4. sites in sequence / position order
    std::vector<VariantSite> out;
    for (auto& s : searches) {
        if (s.paths.empty()) continue;
        VariantSite v;
        v.seq_index = std::get<0>(seg_of[s.seg_id]);
        v.seg_start = std::get<1>(seg_of[s.seg_id]);
        v.paths = std::move(s.paths);
        out.push_back(std::move(v));
    }
    return out;
}

// One VCF record per path.  Layout inferred from validateFiles/test.50.tst:7-37 (the writer is gfalibs' Report):
//   SNV / COM : POS = pos + 1, REF = the refLen (1 for SNV) original bases from pos, ALT = the path's bases
//   INS / DEL : the base in front anchors the record: POS = pos, REF = anchor + base at pos,
//               ALT = anchor + the path's bases (+ the base at pos for DEL: it stays)
// QUAL 0, FILTER PASS, genotype 1/1 with quality 0, exactly as the golden prints them.
std::vector<std::string> vcf_lines(const std::vector<SeqRecord>& seqs, const std::vector<VariantSite>& sites) {
    std::vector<std::string> out = {"##fileformat=VCFv4.2",
                                    "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">",
                                    "##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">",
                                    "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"};
    for (auto& site : sites) {
        const SeqRecord& r = seqs[site.seq_index];
        for (auto& p : site.paths) {
            const uint64_t pos = site.seg_start + p.pos;
            uint64_t vpos;
            std::string ref, alt;
            if (p.type == VAR_SNV || p.type == VAR_COM) {
                const uint64_t n = p.type == VAR_COM ? p.ref_len : 1;
                vpos = pos + 1; ref = r.seq.substr(pos, n); alt = p.sequence;
            } else {
                const char anchor = r.seq[pos - 1];
                vpos = pos;
                ref = std::string(1, anchor) + r.seq[pos];
                alt = std::string(1, anchor) + p.sequence + (p.type == VAR_DEL ? std::string(1, r.seq[pos]) : std::string());
            }
            out.push_back(r.header + "\t" + std::to_string(vpos) + "\t.\t" + ref + "\t" + alt + "\t0\tPASS\t.\tGT:GQ\t1/1:0");
        }
    }
    return out;
}

}  // namespace kqhost

// kreeq (MI355X build) -- host CLI with the reference's `validate` / `union` interface.
// Flag names, stdout blocks, exit codes and the .kreeq directory follow vgl-hub/kreeq @ 2024_08_07
// (src/main.cpp:78-97, :220-229; src/input.cpp:76-152; src/kreeq-output.cpp:34-136;
// src/graph-builder.cpp:288-293; src/kreeq.cpp:78-106).  All compute goes through the C ABI
// (include/kreeq_amd.h) to the GPU; this file is plumbing: argument parsing, file formats, text.
#include <getopt.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fastx.h"
#include "kreeq_amd.h"
#include "kreeq_db.h"
#include "variants.h"

using namespace kqhost;

namespace {

const char* kVersion = "0.1.0-mi355x";

struct UserInput {                       // reference UserInputKreeq (include/input.h:25-34) + gfalibs UserInput fields used
    int mode = 0;                        // 0 validate, 1 union
    std::string inSequence, outFile, prefix = ".", inBedInclude;
    std::vector<std::string> inReads, kmerDB;
    int kmerLen = 21;
    uint32_t covCutOff = 0;
    int kmerDepth = -1, maxSpan = 5, maxThreads = 0;
    double maxMem = 0;
    int device = 0;
    int passes = 0;                      // --passes: count the maps in this many ranges (memory-bounded mode); 0 = as many as the HBM asks for
};

int verbose_flag = 0, cmd_flag = 0;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
const double t_start = now_s();
void verbose(const std::string& s) { if (verbose_flag) fprintf(stderr, "[%8.3f s] %s\n", now_s() - t_start, s.c_str()); }

[[noreturn]] void die(const std::string& msg) {
    fprintf(stderr, "%s\n", msg.c_str());
    exit(EXIT_FAILURE);
}
void kq_or_die(int rc) { if (rc != KQ_OK) die(std::string("Error: ") + kq_last_error()); }

bool is_number(const char* s) { if (!*s) return false; for (; *s; ++s) if (*s < '0' || *s > '9') return false; return true; }
bool is_int(const char* s) { if (*s == '-' || *s == '+') ++s; return is_number(s); }
void if_file_exists(const char* p) {
    struct stat st;
    if (stat(p, &st) != 0) { printf("File does not exist (%s). Terminating.\n", p); exit(EXIT_FAILURE); }
}
// extension after the last '.', keeping a trailing ".gz" attached ("x.gfa.gz" -> "gfa.gz")
std::string file_ext(const std::string& path) {
    std::string p = path;
    std::string gz;
    if (p.size() > 3 && p.substr(p.size() - 3) == ".gz") { gz = ".gz"; p.resize(p.size() - 3); }
    size_t dot = p.rfind('.');
    if (dot == std::string::npos) return "";
    return p.substr(dot + 1) + gz;
}

void print_help() {
    printf("kreeq [mode] -h\nfor additional help.\n");
    printf("\nModes:\n");
    printf("validate\n");
    printf("union\n");
    exit(0);
}

// ------------------------------------------------------------------------------------------------
struct Assembly {
    std::vector<SeqRecord> seqs;
    std::vector<uint64_t> offset;     // start of each sequence inside `joined`
    std::string joined;               // sequences separated by '\n' (a non-ACGT byte ends every segment)
};

void load_genome(const std::string& path, Assembly& a) {          // reference Input::loadGenome, src/input.cpp:188-308
    read_fastx(path, [&](SeqRecord&& r) { a.seqs.push_back(std::move(r)); });
    size_t total = 0;
    for (auto& s : a.seqs) total += s.seq.size() + 1;
    a.joined.reserve(total);
    for (auto& s : a.seqs) {
        a.offset.push_back(a.joined.size());
        a.joined += s.seq;
        a.joined.push_back('\n');
    }
}

inline bool is_base(char c) {
    switch (c) { case 'A': case 'C': case 'G': case 'T': case 'a': case 'c': case 'g': case 't': return true; default: return false; }
}
// segments of one sequence = maximal runs of bases (gfalibs appendSequence splits at N runs)
std::vector<std::pair<uint64_t, uint64_t>> segments_of(const std::string& s) {
    std::vector<std::pair<uint64_t, uint64_t>> out;
    for (uint64_t i = 0; i < s.size();) {
        if (!is_base(s[i])) { ++i; continue; }
        uint64_t j = i;
        while (j < s.size() && is_base(s[j])) ++j;
        out.emplace_back(i, j - i);
        i = j;
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
struct Engine {
    kq_handle* h = nullptr;
    UserInput ui;
    int k = 21, map_count = 128;
    Assembly genome;
    std::vector<kq_dbgbase> per_base;     // aligned with genome.joined when a per-base writer needs it
    uint64_t counters[3] = {0, 0, 0};

    void create(uint64_t hint) { kq_or_die(kq_create(&h, ui.device, k, map_count, hint)); }

    void stats() {                                                   // gfalibs stats() -> DBG::summary + DBG::DBstats
        kq_stats st;
        kq_or_die(kq_summary(h, &st));
        std::cout << "DBG Summary statistics:\n"                     // src/graph-builder.cpp:288-293
                  << "Total kmers: " << st.total << "\n"
                  << "Unique kmers: " << st.unique << "\n"
                  << "Distinct kmers: " << st.distinct << "\n"
                  << "Missing kmers: " << st.missing << "\n"
                  << "Total edges: " << st.edges << "\n";
    }

    static double error_rate(uint64_t missing, uint64_t total, int k) {          // src/kreeq.cpp:36-40
        return 1 - pow(1 - (double)missing / total, (double)1 / k);
    }

    void validate_sequences(bool want_per_base) {                    // DBG::validateSequences, src/kreeq.cpp:47-108
        if (ui.inSequence.empty()) return;
        verbose("Validating sequence");
        if (want_per_base) per_base.assign(genome.joined.size(), kq_dbgbase{});
        kq_or_die(kq_lookup_sequence(h, genome.joined.data(), genome.joined.size(), ui.covCutOff, 0, (uint16_t)map_count,
                                     want_per_base ? per_base.data() : nullptr, counters));
        print_qv();
    }

    void print_qv() {                                                // src/kreeq.cpp:78-106
        if (ui.outFile.find(".") != std::string::npos || ui.outFile == "") {
            const uint64_t missing = counters[0], total = counters[1], edge_missing = counters[2];
            std::cout << "Missing" << "\t" << "Total" << "\t" << "QV" << "\t" << "Error" << "\t" << "k" << "\t" << "Method" << std::endl;
            double merquryError = error_rate(missing, total, k), merquryQV = -10 * log10(merquryError);
            std::cout << missing << "\t" << total << "\t" << merquryQV << "\t" << merquryError << "\t" << std::to_string(k) << "\t"
                      << "Merqury" << std::endl;
            double kreeqError = error_rate(missing + edge_missing, total, k), kreeqQV = -10 * log10(kreeqError);
            std::cout << missing + edge_missing << "\t" << total << "\t" << kreeqQV << "\t" << kreeqError << "\t" << std::to_string(k)
                      << "\t" << "Kreeq" << std::endl;
        }
    }

    void write_kreeq_db(const std::string& dir) {
        uint64_t n = 0;
        kq_or_die(kq_export(h, 0, (uint16_t)map_count, nullptr, 0, &n));
        std::vector<kq_entry> entries((size_t)n);
        if (n) kq_or_die(kq_export(h, 0, (uint16_t)map_count, entries.data(), n, &n));
        verbose("Table exported (" + std::to_string(n) + " k-mers)");
        write_db(dir, k, map_count, entries);
    }

    // .kwig (src/kreeq-output.cpp:243-303) and .bkwig (:305-399)
    void write_kwig(const std::string& path) {
        std::ofstream ofs(path);
        ofs << std::to_string(k) << "\n";
        for (size_t s = 0; s < genome.seqs.size(); ++s) {
            for (auto& seg : segments_of(genome.seqs[s].seq)) {
                ofs << "fixedStep chrom=" << genome.seqs[s].header << " start=" << seg.first << " step=1" << "\n";
                const kq_dbgbase* b = per_base.data() + genome.offset[s] + seg.first;
                for (uint64_t i = 0; i < seg.second; ++i)
                    ofs << std::to_string(b[i].cov) << "," << std::to_string(b[i].isFw ? b[i].fw : b[i].bw) << ","
                        << std::to_string(b[i].isFw ? b[i].bw : b[i].fw) << "\n";
            }
        }
    }
    void write_bkwig(const std::string& path) {
        std::ofstream ofs(path, std::ios::trunc | std::ios::out | std::ios::binary);
        uint8_t k8 = (uint8_t)k;
        ofs.write((const char*)&k8, 1);
        uint32_t nPaths = (uint32_t)genome.seqs.size();
        ofs.write((const char*)&nPaths, 4);
        std::vector<std::vector<std::pair<uint64_t, uint64_t>>> segs;
        for (auto& s : genome.seqs) segs.push_back(segments_of(s.seq));
        for (size_t s = 0; s < genome.seqs.size(); ++s) {                            // writeIndex :305-355
            uint16_t hl = (uint16_t)genome.seqs[s].header.size();
            ofs.write((const char*)&hl, 2);
            ofs << genome.seqs[s].header;
            uint32_t nc = (uint32_t)segs[s].size();
            ofs.write((const char*)&nc, 4);
            for (auto& seg : segs[s]) {
                uint8_t step = 1;
                ofs.write((const char*)&seg.first, 8);
                ofs.write((const char*)&seg.second, 8);
                ofs.write((const char*)&step, 1);
            }
        }
        for (size_t s = 0; s < genome.seqs.size(); ++s)
            for (auto& seg : segs[s]) {
                const kq_dbgbase* b = per_base.data() + genome.offset[s] + seg.first;
                for (uint64_t i = 0; i < seg.second; ++i) {
                    ofs.write((const char*)&b[i].cov, 4);
                    ofs.write((const char*)(b[i].isFw ? &b[i].fw : &b[i].bw), 4);
                    ofs.write((const char*)(b[i].isFw ? &b[i].bw : &b[i].fw), 4);
                }
            }
    }
    // .bed / .csvtable per-base table (src/kreeq-output.cpp:138-241): for every position the k most
    // recent values of cov / fw-edge / bw-edge, oldest first.  No fixture of the reference pins this
    // text (parity unpinned); it is restated from the writer's source.
    void write_table(const std::string& path, const std::string& ext) {
        char colSep = ',', entrySep = ',';
        if (ext == "bed") colSep = '\t', entrySep = ':';
        std::ofstream ofs(path);
        for (size_t s = 0; s < genome.seqs.size(); ++s) {
            for (auto& seg : segments_of(genome.seqs[s].seq)) {
                std::vector<uint32_t> kc(k - 1, 0), ef(k - 1, 0), eb(k - 1, 0);
                const kq_dbgbase* b = per_base.data() + genome.offset[s] + seg.first;
                for (uint64_t i = 0; i < seg.second; ++i) {
                    ofs << genome.seqs[s].header << colSep << (seg.first + i) << colSep;
                    kc.push_back(b[i].cov);
                    ef.push_back(b[i].isFw ? b[i].fw : b[i].bw);
                    eb.push_back(b[i].isFw ? b[i].bw : b[i].fw);
                    for (auto* v : {&kc, &ef, &eb}) {
                        for (int c = 0; c < k; ++c) { ofs << std::to_string((*v)[(size_t)c]); if (c < k - 1) ofs << entrySep; }
                        if (v != &eb) ofs << colSep;
                        v->erase(v->begin());
                    }
                    ofs << "\n";
                }
            }
        }
    }
    // -o vcf / -o x.vcf: DBG::correctSequences (src/variants.cpp:40-50) + gfalibs' VCF writer; "-o vcf" names no file:
    // the records go to stdout (validateFiles/test.50.tst)
    void write_vcf() { write_vcf(graph_of_handle(h)); }
    void write_vcf(const GraphSource& g) {
        if (ui.inSequence.empty()) return;                           // src/variants.cpp:42-43
        const int depth = ui.kmerDepth == -1 ? k : ui.kmerDepth;     // include/kreeq.h:171-173 (unidirectional search)
        auto sites = find_candidate_errors(g, k, genome.seqs, depth, ui.maxSpan, ui.covCutOff, [](const std::string& m) { verbose(m); });
        auto lines = vcf_lines(genome.seqs, sites);
        if (ui.outFile == "vcf") { for (auto& l : lines) std::cout << l << "\n"; return; }
        std::ofstream ofs(ui.outFile);
        for (auto& l : lines) ofs << l << "\n";
    }
    void write_hist(const std::string& path) {                       // gfalibs printHist (format not pinned by any fixture)
        uint64_t n = 0;
        kq_or_die(kq_histogram(h, nullptr, nullptr, 0, &n));
        std::vector<uint64_t> cov((size_t)n), cnt((size_t)n);
        if (n) kq_or_die(kq_histogram(h, cov.data(), cnt.data(), n, &n));
        std::ofstream ofs(path);
        for (uint64_t i = 0; i < n; ++i) ofs << cov[i] << "\t" << cnt[i] << "\n";
    }

    void report() {                                                  // DBG::report, src/kreeq-output.cpp:34-136
        std::string ext = "stdout";
        if (ui.outFile != "") ext = file_ext("." + ui.outFile);
        if (ui.outFile.find(".") != std::string::npos || ui.outFile == "" || ext == "kreeq") { stats(); verbose("Summary computed"); }
        verbose("Writing ouput: " + ui.outFile);
        const bool per_base_out = (ext == "kwig" || ext == "bkwig" || ext == "bed" || ext == "csvtable");
        if (ext == "gfa" || ext == "gfa2" || ext == "gfa.gz" || ext == "gfa2.gz")
            die("Error: ." + ext + " output (variant graph) is not supported by this build: use -o vcf");
        if (ext == "vcf") { if (ui.mode == 0) write_vcf(); return; }   // case 6: correctSequences, then printVCF (src/kreeq-output.cpp:74-82, :124-127)
        // the reference's first switch has no case for .kreeq / .hist, so they fall to `default:` like every other
        // extension and validate too (src/kreeq-output.cpp:62-72); validateSequences returns at once without -f
        if (ui.mode == 0) validate_sequences(per_base_out);
        if (ext == "kreeq") { write_kreeq_db(ui.outFile); verbose("Database written"); }
        else if (ext == "kwig") write_kwig(ui.outFile);
        else if (ext == "bkwig") write_bkwig(ui.outFile);
        else if (ext == "bed" || ext == "csvtable") write_table(ui.outFile, ext);
        else if (ext == "hist") write_hist(ui.outFile);
    }
};

uint64_t file_size(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0; }
uint64_t db_bytes(const std::string& dir, int map_count) {
    uint64_t b = file_size(dir + "/.map.hc.bin");
    for (int m = 0; m < map_count; ++m) b += file_size(dir + "/.map." + std::to_string(m) + ".bin");
    return b;
}

// distinct k-mers a read set may hold, from its size on disk (a FASTQ record spends about half its bytes on bases,
// a gzipped one is ~4x smaller; at 30x coverage and 0.5 % errors about a fifth of the k-mer instances are distinct)
uint64_t distinct_estimate(uint64_t read_bytes) { return read_bytes / 2 + (1 << 20); }
// Map ranges to count in (the reference's computeMapRange, src/kreeq.cpp:59-63, bounds its maps by -m / 90 % of the
// RAM; here the bound is the HBM): the table (16 B per slot at load <= 0.7) plus the partition scratch and the
// pending-set arena should stay within 60 % of what is free, or of -m <GB> when that is smaller.
int passes_for(const UserInput& ui, uint64_t distinct) {
    uint64_t free_b = 0, total_b = 0;
    if (kq_device_memory(ui.device, &free_b, &total_b) != KQ_OK) return 1;
    double budget = 0.6 * (double)free_b;
    if (ui.maxMem > 0) budget = std::min(budget, ui.maxMem * 1e9);
    const double table = (double)distinct / 0.7 * 16.0;
    return (int)std::max(1.0, std::min(128.0, std::ceil(table / budget)));
}

// Memory-bounded validate: the hash maps are processed in `passes` ranges, re-reading the reads for
// every range -- the GPU counterpart of the reference's map-range loop (computeMapRange /
// loadMapRange, src/kreeq.cpp:59-74) and of its spill-to-disk behaviour under -m.  Only 1/passes of
// the table is resident at a time; summary numbers and QV counters add up over the disjoint ranges.
// Reads -> GPU: the parser threads share a small POOL of page-locked buffers.  A thread takes a free buffer, fills it and
// submits it itself (kq_count_batch_async: enqueues the DMA and the count kernels behind it, returns at once); the buffer
// goes back to the pool when its copy has finished.  Parsing, PCIe copies and counting overlap without a consumer thread.
// Why a pool: what costs on this path is not PCIe (a page-locked buffer copies at 50+ GB/s) but making host pages
// DMA-able -- a copy out of a fresh pageable buffer takes 6-7 ms per 4-8 MB for exactly that, and buffers of their own for
// every parser thread paid it once per buffer (35 of the 38 ms of ingesting a 310 MB FASTQ).  A pool is locked once
// (kq_host_alloc: ~0.5 ms per buffer) and every copy after that runs at PCIe rate.
// Optional (KQ_INGEST_PACK=1): the parser thread packs a buffer to 2 bits per base + a validity bit (kq_pack_bases) before
// it submits it: 6 bytes per 16 bases cross PCIe instead of 16.
struct GpuSink {
    kq_handle* h;
    size_t cap;
    bool packed;
    struct Buf { char* ascii = nullptr; uint32_t* codes = nullptr; uint16_t* inv = nullptr; uint64_t ticket = 0; bool in_flight = false; };
    std::vector<Buf> pool;
    std::vector<int> free_list;                   // indices of buffers nobody holds
    std::vector<int> flying;                      // submitted, oldest first
    std::mutex m;
    std::condition_variable cv;
    bool failed = false;                          // a parser thread or a submit has failed: nobody waits for a buffer any more (under m)
    std::atomic<long long> ns_wait{0}, ns_pack{0}, ns_call{0}, n_submit{0};     // KQ_INGEST_TRACE: where the threads spend their time
    GpuSink(kq_handle* handle, unsigned n_buffers, size_t buffer_bytes, bool pack) : h(handle), cap(buffer_bytes), packed(pack), pool(n_buffers) {
        for (unsigned i = 0; i < n_buffers; ++i) free_list.push_back((int)(n_buffers - 1 - i));
    }
    // a buffer gets its memory from the thread that first takes it: the page locking of the pool runs in parallel
    void materialize(Buf& b) {
        if (b.ascii) return;
        // packed mode: the ASCII buffer never crosses PCIe (plain memory), the packed arrays do
        b.ascii = packed ? (char*)malloc(cap) : (char*)kq_host_alloc(cap);
        if (packed) { b.codes = (uint32_t*)kq_host_alloc((cap / 16 + 1) * 4); b.inv = (uint16_t*)kq_host_alloc((cap / 16 + 1) * 2); }
        if (!b.ascii || (packed && (!b.codes || !b.inv))) throw std::runtime_error("read buffer allocation failed");
    }
    // every copy that still reads a pool buffer has finished (kq_host_free's contract, include/kreeq_amd.h)
    void drain() {
        std::vector<uint64_t> tickets;
        { std::lock_guard<std::mutex> l(m); for (int i : flying) tickets.push_back(pool[i].ticket); for (int i : flying) free_list.push_back(i); flying.clear(); }
        for (uint64_t tk : tickets) (void)kq_host_wait(h, tk);
    }
    void fail() { { std::lock_guard<std::mutex> l(m); failed = true; } cv.notify_all(); }
    ~GpuSink() {
        drain();
        for (auto& b : pool) { if (packed) free(b.ascii); else kq_host_free(b.ascii); kq_host_free(b.codes); kq_host_free(b.inv); }
    }
    BatchSink sink() {
        BatchSink bs;
        bs.acquire = [this](unsigned, size_t* c) {
            const auto t0 = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> l(m);
            for (;;) {
                if (failed) throw std::runtime_error("read ingest aborted");      // (the first failure is the one reported)
                if (!free_list.empty()) break;
                if (!flying.empty()) {                                // the oldest copy in flight: wait for it outside the lock
                    const int i = flying.front();
                    flying.erase(flying.begin());
                    const uint64_t tk = pool[i].ticket;
                    l.unlock();
                    const int wrc = kq_host_wait(h, tk);
                    const std::string werr = wrc != KQ_OK ? kq_last_error() : "";
                    l.lock();
                    free_list.push_back(i);                           // back to the pool whatever happened
                    if (wrc != KQ_OK) { failed = true; cv.notify_all(); throw std::runtime_error("Error: " + werr); }
                    cv.notify_one();
                    continue;
                }
                cv.wait(l);                                           // every buffer is being filled by another thread
            }
            const int i = free_list.back();
            free_list.pop_back();
            l.unlock();
            try { materialize(pool[i]); }
            catch (...) { { std::lock_guard<std::mutex> g(m); free_list.push_back(i); failed = true; } cv.notify_all(); throw; }
            ns_wait += (std::chrono::steady_clock::now() - t0).count();
            *c = cap;
            return pool[i].ascii;
        };
        bs.submit = [this](unsigned, char* buf, size_t len) {
            int i = 0;
            while (pool[i].ascii != buf) ++i;
            Buf& b = pool[i];
            const auto t0 = std::chrono::steady_clock::now();
            if (packed) kq_pack_bases(buf, len, b.codes, b.inv);
            const auto t1 = std::chrono::steady_clock::now();
            // no lock around the call: kq_count_*_async may be called from several threads (they take turns inside the library)
            int rc = !len ? KQ_OK : packed ? kq_count_packed_async(h, b.codes, b.inv, len, &b.ticket) : kq_count_batch_async(h, buf, len, &b.ticket);
            static const long fail_after = getenv("KQ_TEST_FAIL_SUBMIT") ? atol(getenv("KQ_TEST_FAIL_SUBMIT")) : -1;      // failure-path test: every submit from the n-th on fails
            const bool injected = rc == KQ_OK && len && fail_after >= 0 && n_submit.load() >= fail_after;
            if (injected) { (void)kq_host_wait(h, b.ticket); rc = KQ_ERR_NOMEM; }
            if (rc != KQ_OK || !len) {
                // a failed submit (out of device memory at scale ...) or a buffer that was taken and not filled: the buffer goes
                // back to the pool; on failure every thread waiting for a buffer is woken and throws (they used to
                // wait forever for buffers held by threads that had died)
                const std::string err = injected ? "submit failed (injected by KQ_TEST_FAIL_SUBMIT)" : rc != KQ_OK ? kq_last_error() : "";
                { std::lock_guard<std::mutex> l(m); free_list.push_back(i); if (rc != KQ_OK) failed = true; }
                cv.notify_all();
                if (rc != KQ_OK) throw std::runtime_error("Error: " + err);
                return;
            }
            ns_pack += (t1 - t0).count(); ns_call += (std::chrono::steady_clock::now() - t1).count(); ++n_submit;
            std::lock_guard<std::mutex> l(m);
            flying.push_back(i);
            cv.notify_one();
        };
        // a sequence longer than a pool buffer: a page-locked buffer of its own, submitted like any other and released once
        // its copy has finished (the device side cuts it into slices with overlapping scans: k-mers and edges across a cut
        // are seen exactly once, kreeq_amd.hip count_seq_dev)
        bs.acquire_big = [this](unsigned, size_t n) {
            char* p = (char*)kq_host_alloc(n);
            if (!p) { fail(); throw std::runtime_error("read buffer allocation failed (" + std::to_string(n) + " bytes)"); }
            return p;
        };
        bs.submit_big = [this](unsigned, char* buf, size_t len) {
            uint64_t tk = 0;
            int rc = len ? kq_count_batch_async(h, buf, len, &tk) : KQ_OK;
            std::string err = rc != KQ_OK ? kq_last_error() : "";
            if (rc == KQ_OK && len) { rc = kq_host_wait(h, tk); if (rc != KQ_OK) err = kq_last_error(); }
            kq_host_free(buf);
            if (rc != KQ_OK) { fail(); throw std::runtime_error("Error: " + err); }
            ++n_submit;
        };
        bs.abort = [this] { fail(); };
        return bs;
    }
};
void count_reads_file(kq_handle* h, const std::string& path, unsigned threads, uint64_t input_bytes, int device) {
    // buffers of 8-32 MiB: small enough that copies and counting of a 300 MB file overlap its parsing, large enough
    // (>= 7 M k-mers) for the partitioned count path; the pool has a buffer per parser thread up to 16 and 2 to spare
    size_t cap = (size_t)std::min<uint64_t>(32ull << 20, std::max<uint64_t>(8ull << 20, input_bytes / 64));
    threads = std::min(threads, 64u);
    unsigned n_buf = std::min(threads, 16u) + 2;
    bool pack = false;
    if (const char* e = getenv("KQ_INGEST_CAP_MB")) cap = (size_t)atol(e) << 20;      // tuning aids (tools/bench_extra/cli_ingest_*.py)
    if (const char* e = getenv("KQ_INGEST_BUFFERS")) n_buf = (unsigned)std::max(2, atoi(e));
    if (const char* e = getenv("KQ_INGEST_PACK")) pack = atoi(e) != 0;
    // A job that knows its size sizes the pending-set arena once (the automatic arena starts small and doubles -- a table
    // pass, a synchronisation and a reallocation every time -- which a 300 MB file pays four or five times): about one k-mer
    // per two input bytes, 5 bytes per record, within half of the free HBM.  Larger inputs just fill it more than once.
    {
        uint64_t free_b = 0, total_b = 0;
        if (kq_device_memory(device, &free_b, &total_b) != KQ_OK) free_b = 8ull << 30;
        uint64_t want = std::max<uint64_t>(256ull << 20, input_bytes * 3);
        if (const char* e = getenv("KQ_CLI_PENDING_MB")) want = (uint64_t)atol(e) << 20;
        if (!getenv("KQ_CLI_PENDING_AUTO")) kq_or_die(kq_set_option(h, KQ_OPT_PENDING_BYTES, (int64_t)std::min<uint64_t>(want, free_b / 2)));
    }
    const auto t00 = std::chrono::steady_clock::now();
    GpuSink gs(h, n_buf, cap, pack);
    const auto t0 = std::chrono::steady_clock::now();
    read_batches_sink(path, threads, gs.sink());
    const auto t1 = std::chrono::steady_clock::now();
    kq_or_die(kq_flush(h));                                          // everything is enqueued; the table pass may start
    gs.drain();                                                      // the last copies out of the pool have finished before it is released
    if (getenv("KQ_INGEST_TRACE"))
        fprintf(stderr, "ingest: %u threads, %u buffers of %zu MiB, %lld submits, pool %.1f ms, wall %.1f ms + flush %.1f ms; thread-time sums: wait for a buffer %.1f ms, "
                        "pack %.1f ms, submit calls %.1f ms\n", threads, n_buf, cap >> 20, (long long)gs.n_submit, (t0 - t00).count() * 1e-6, (t1 - t0).count() * 1e-6,
                (std::chrono::steady_clock::now() - t1).count() * 1e-6, gs.ns_wait * 1e-6, gs.ns_pack * 1e-6, gs.ns_call * 1e-6);
}

// -j, or one parser thread per 32 MB of input within [4, all cores]: the threads take turns submitting (one stream feeds the
// GPU), and on a few hundred MB more of them only queue up behind each other (310 MB FASTQ: 8 threads 26-32 ms, 64: 38 ms)
unsigned parser_threads(const UserInput& ui, uint64_t input_bytes = ~0ull) {
    if (ui.maxThreads > 0) return (unsigned)ui.maxThreads;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return (unsigned)std::min<uint64_t>(hw, std::max<uint64_t>(4, input_bytes >> 25));
}

long peak_rss_mb() { struct rusage ru; getrusage(RUSAGE_SELF, &ru); return ru.ru_maxrss / 1024; }
std::pair<int, int> pass_range(int p, int passes, int map_count) {
    return {(int)((long long)p * map_count / passes), (int)((long long)(p + 1) * map_count / passes)};
}

// A .kreeq database read MAP RANGE BY MAP RANGE (the reference's loadMapRange / deleteMapRange, src/kreeq.cpp:59-74;
// mergeMaps map by map, src/graph-builder.cpp:341-347): the host never holds more than a chunk of a few maps' entries
// (48 B each; round-2 VERDICT: a human-scale database read whole into a std::vector needs ~500 GB of host memory).
struct DbSource {
    std::string dir;
    DbIndex idx;
    std::vector<kq_entry> hc;                     // the high-copy map: small (k-mers with cov >= 255), read once
    explicit DbSource(const std::string& d) : dir(d) { idx = read_index(d); read_db_hc(d, hc); }
    uint64_t map_entries_bound(int m) const { return file_size(dir + "/.map." + std::to_string(m) + ".bin") / 24; }   // 24-byte slots
    uint64_t entries_bound(int lo, int hi) const {
        uint64_t n = 0;
        for (int m = lo; m < hi; ++m) n += map_entries_bound(m);
        for (auto& e : hc) { const int m = (int)(e.key % (uint64_t)idx.map_count); n += m >= lo && m < hi; }
        return n;
    }
    static size_t chunk_entries() { const char* e = getenv("KQ_DB_CHUNK_ENTRIES"); return e ? (size_t)std::max(1ll, atoll(e)) : ((size_t)1 << 25); }   // 1.6 GB of host memory
    // ADDS the maps [lo, hi) to the table of h
    uint64_t import_into(kq_handle* h, int lo, int hi) const {
        std::vector<kq_entry> v;
        uint64_t total = 0;
        const size_t chunk = chunk_entries();
        for (int a = lo; a < hi;) {
            int b = a;
            uint64_t est = 0;
            do { est += map_entries_bound(b); ++b; } while (b < hi && est + map_entries_bound(b) <= chunk);
            v.clear();
            read_db_maps(dir, idx, a, b, hc, v);
            kq_or_die(kq_import(h, v.data(), v.size()));
            total += v.size();
            a = b;
        }
        return total;
    }
};

// The variant search over a database whose table is resident one map range at a time: a lookup round asks every range that
// owns one of its keys (the resident one first), the pre-filter is the OR of the ranges' scans (a k-mer is in exactly one
// range).  The reference's search loops over the map ranges in the same way (src/variants.cpp:78-84) with a cache of the
// entries it has seen (:199-210) -- here the host-side cache of find_candidate_errors.
GraphSource graph_of_db_ranges(kq_handle* h, const DbSource& db, int passes) {
    auto cur = std::make_shared<int>(-1);
    auto load = [h, &db, passes, cur](int p) {
        if (*cur == p) return;
        const auto r = pass_range(p, passes, db.idx.map_count);
        kq_or_die(kq_clear(h));
        db.import_into(h, r.first, r.second);
        *cur = p;
        verbose("Variant search: maps [" + std::to_string(r.first) + "," + std::to_string(r.second) + ") loaded");
    };
    GraphSource g;
    g.branch_scan = [h, passes, load](const std::string& joined, uint32_t cov_cutoff, std::vector<uint8_t>& flags) {
        std::vector<uint8_t> part(joined.size());
        std::fill(flags.begin(), flags.end(), 0);
        for (int p = 0; p < passes; ++p) {
            load(p);
            kq_or_die(kq_branch_scan(h, joined.data(), joined.size(), cov_cutoff, part.data()));
            for (size_t i = 0; i < flags.size(); ++i) flags[i] |= part[i];
        }
    };
    g.lookup = [h, &db, passes, load, cur](const std::vector<uint64_t>& want, std::vector<kq_entry>& got) {
        const int mc = db.idx.map_count;
        std::vector<std::vector<size_t>> by(passes);
        for (size_t i = 0; i < want.size(); ++i) {
            const int m = (int)(want[i] % (uint64_t)mc);
            int p = (int)(((long long)(m + 1) * passes - 1) / mc);          // the range [p mc / passes, (p + 1) mc / passes) that holds map m
            while (pass_range(p, passes, mc).first > m) --p;
            while (pass_range(p, passes, mc).second <= m) ++p;
            by[p].push_back(i);
        }
        std::vector<uint64_t> sub;
        std::vector<kq_entry> res;
        const int start = *cur < 0 ? 0 : *cur;
        for (int q = 0; q < passes; ++q) {
            const int p = (start + q) % passes;
            if (by[p].empty()) continue;
            load(p);
            sub.clear();
            for (size_t i : by[p]) sub.push_back(want[i]);
            res.resize(sub.size());
            kq_or_die(kq_lookup_keys(h, sub.data(), sub.size(), res.data()));
            for (size_t j = 0; j < sub.size(); ++j) got[by[p][j]] = res[j];
        }
    };
    return g;
}

// Memory-bounded validate in `passes` map ranges, from reads (every range rescans them) or from a database on disk (every
// range is read from its map files): summary numbers, histogram and QV counters add up over the disjoint ranges, the
// per-base table is filled range by range (a position is evaluated in exactly one), a .kreeq output is written range by range.
// The variant search (-o vcf) needs k-mers of any range at any time: it runs over the database on disk (the input one, or
// a temporary one written during the passes -- the reference dumps its maps to disk and reloads ranges likewise).
int run_passes(Engine& e, const DbSource* db) {
    UserInput& ui = e.ui;
    if (ui.passes > e.map_count) ui.passes = e.map_count;
    uint64_t bytes = 0;
    for (auto& f : ui.inReads) bytes += file_size(f) * (file_ext(f).find("gz") != std::string::npos ? 4 : 1);
    if (db) {
        uint64_t cap = 0;
        for (int p = 0; p < ui.passes; ++p) { const auto r = pass_range(p, ui.passes, e.map_count); cap = std::max(cap, db->entries_bound(r.first, r.second)); }
        e.create(cap + 1024);
    } else {
        e.k = ui.kmerLen;
        e.create(distinct_estimate(bytes) / (uint64_t)ui.passes + (1 << 20));
    }
    std::string ext = "stdout";
    if (ui.outFile != "") ext = file_ext("." + ui.outFile);
    if (ext == "gfa" || ext == "gfa2" || ext == "gfa.gz" || ext == "gfa2.gz")
        die("Error: ." + ext + " output (variant graph) is not supported by this build: use -o vcf");
    if (!ui.inSequence.empty()) load_genome(ui.inSequence, e.genome);
    const bool want_stats = ui.outFile.find(".") != std::string::npos || ui.outFile == "" || ext == "kreeq";
    const bool vcf = ext == "vcf";
    const bool want_validate = !ui.inSequence.empty() && !vcf;      // every other extension validates (src/kreeq-output.cpp:62-72)
    const bool per_base_out = (ext == "kwig" || ext == "bkwig" || ext == "bed" || ext == "csvtable");
    if (want_validate && per_base_out) e.per_base.assign(e.genome.joined.size(), kq_dbgbase{});
    // where the ranges go when a later step needs them again: the .kreeq output, or a temporary database for the variant search
    std::string range_db;
    if (ext == "kreeq") range_db = ui.outFile;
    else if (vcf && !db && !ui.inSequence.empty()) range_db = ui.prefix + "/.kreeq_ranges_" + std::to_string((long)getpid()) + ".kreeq";
    kq_stats sum{};
    std::map<uint64_t, uint64_t> hist;
    std::vector<kq_entry> hc_all;
    const unsigned threads = parser_threads(ui, bytes);
    for (int p = 0; p < ui.passes; ++p) {
        const auto r = pass_range(p, ui.passes, e.map_count);
        const int lo = r.first, hi = r.second;
        verbose("Pass " + std::to_string(p + 1) + "/" + std::to_string(ui.passes) + ": maps [" + std::to_string(lo) + "," + std::to_string(hi) + ")");
        if (p) kq_or_die(kq_clear(e.h));
        if (db) db->import_into(e.h, lo, hi);
        else {
            kq_or_die(kq_set_option(e.h, KQ_OPT_COUNT_MAP_RANGE, (int64_t)lo | ((int64_t)hi << 16)));
            for (auto& f : ui.inReads) count_reads_file(e.h, f, threads, bytes, ui.device);
        }
        kq_stats st;
        kq_or_die(kq_summary(e.h, &st));
        sum.total += st.total; sum.unique += st.unique; sum.distinct += st.distinct; sum.edges += st.edges;
        if (want_validate)
            kq_or_die(kq_lookup_sequence(e.h, e.genome.joined.data(), e.genome.joined.size(), ui.covCutOff, (uint16_t)lo, (uint16_t)hi,
                                         per_base_out ? e.per_base.data() : nullptr, e.counters));
        if (ext == "hist") {
            uint64_t n = 0;
            kq_or_die(kq_histogram(e.h, nullptr, nullptr, 0, &n));
            std::vector<uint64_t> cov((size_t)n), cnt((size_t)n);
            if (n) kq_or_die(kq_histogram(e.h, cov.data(), cnt.data(), n, &n));
            for (uint64_t i = 0; i < n; ++i) hist[cov[i]] += cnt[i];
        }
        if (!range_db.empty()) {
            uint64_t n = 0;
            kq_or_die(kq_export(e.h, (uint16_t)lo, (uint16_t)hi, nullptr, 0, &n));
            std::vector<kq_entry> entries((size_t)n);
            if (n) kq_or_die(kq_export(e.h, (uint16_t)lo, (uint16_t)hi, entries.data(), n, &n));
            write_db_maps(range_db, e.map_count, lo, hi, entries, hc_all);
        }
    }
    if (!range_db.empty()) write_db_finish(range_db, e.k, e.map_count, hc_all);
    if (want_stats) {
        const uint64_t space = e.k < 32 ? (1ull << (2 * e.k)) : 0ull;
        std::cout << "DBG Summary statistics:\n"
                  << "Total kmers: " << sum.total << "\n" << "Unique kmers: " << sum.unique << "\n" << "Distinct kmers: " << sum.distinct << "\n"
                  << "Missing kmers: " << (space - sum.distinct) << "\n" << "Total edges: " << sum.edges << "\n";
    }
    if (want_validate) {
        e.print_qv();
        if (ext == "kwig") e.write_kwig(ui.outFile);
        else if (ext == "bkwig") e.write_bkwig(ui.outFile);
        else if (ext == "bed" || ext == "csvtable") e.write_table(ui.outFile, ext);
    }
    if (ext == "hist") {
        std::ofstream ofs(ui.outFile);
        for (auto& kv : hist) ofs << kv.first << "\t" << kv.second << "\n";
    }
    if (vcf && !ui.inSequence.empty()) {
        if (db) e.write_vcf(graph_of_db_ranges(e.h, *db, ui.passes));
        else {
            { DbSource tmp(range_db); e.write_vcf(graph_of_db_ranges(e.h, tmp, ui.passes)); }
            for (int m = 0; m < e.map_count; ++m) ::unlink((range_db + "/.map." + std::to_string(m) + ".bin").c_str());
            ::unlink((range_db + "/.map.hc.bin").c_str()); ::unlink((range_db + "/.index").c_str()); ::rmdir(range_db.c_str());
        }
    }
    verbose("Peak host memory: " + std::to_string(peak_rss_mb()) + " MB");
    kq_destroy(e.h);
    return EXIT_SUCCESS;
}

int run(UserInput& ui) {
    Engine e;
    e.ui = ui;
    if (ui.outFile.find(".kreeq") != std::string::npos) e.ui.prefix = ui.outFile;        // src/input.cpp:78-79
    switch (ui.mode) {
        case 0: {                                                    // src/input.cpp:86-118
            if (!ui.inReads.empty()) {
                uint64_t bytes = 0;
                for (auto& f : ui.inReads) bytes += file_size(f) * (file_ext(f).find("gz") != std::string::npos ? 4 : 1);
                if (ui.passes == 0) {                                // automatic: as many map ranges as the HBM asks for
                    e.ui.passes = ui.passes = passes_for(ui, distinct_estimate(bytes));
                    if (ui.passes > 1) verbose("Table of ~" + std::to_string(distinct_estimate(bytes)) + " k-mers does not fit the free HBM: counting in " + std::to_string(ui.passes) + " map ranges");
                }
                if (ui.passes > 1) return run_passes(e, nullptr);
                e.k = ui.kmerLen;
                e.create(distinct_estimate(bytes));
                verbose("Loading input reads.");
                for (auto& f : ui.inReads)
                    count_reads_file(e.h, f, parser_threads(ui, bytes), bytes, ui.device);
                verbose("Reads loaded.");
            } else {                                                 // Input::loadGraph, src/input.cpp:56-74
                if (ui.kmerDB.size() > 1) die("More than one DBG database provided. Merge them first. Exiting.");
                if (ui.kmerDB.empty()) die("Cannot load DBG input. Exiting.");
                // the database is read map range by map range (DbSource): a few maps' entries on the host at a time; when its table
                // does not fit the HBM (or -m), it is also EVALUATED range by range (src/kreeq.cpp:59-74)
                DbSource db(ui.kmerDB[0]);
                verbose("Overriding default kmer length (" + std::to_string(ui.kmerLen) + ") with DB kmer length (" + std::to_string(db.idx.k) + ").");
                e.k = db.idx.k; e.map_count = db.idx.map_count;
                const uint64_t bound = db.entries_bound(0, e.map_count);
                if (ui.passes == 0) {
                    e.ui.passes = ui.passes = std::min(passes_for(ui, bound), e.map_count);
                    if (ui.passes > 1) verbose("Table of <= " + std::to_string(bound) + " k-mers does not fit the free HBM: validating in " + std::to_string(ui.passes) + " map ranges");
                }
                if (ui.passes > 1) return run_passes(e, &db);
                e.create(bound + 1024);
                const uint64_t n_in = db.import_into(e.h, 0, e.map_count);
                verbose("Database loaded (" + std::to_string(n_in) + " k-mers; peak host memory " + std::to_string(peak_rss_mb()) + " MB)");
            }
            if (!ui.inSequence.empty()) {
                verbose("Loading input sequences");
                load_genome(ui.inSequence, e.genome);
                verbose("Sequences loaded");
            }
            e.report();
            break;
        }
        case 1: {                                                    // src/input.cpp:119-152
            verbose("Merging input databases.");
            int k = 0, map_count = 128;
            for (auto& db : ui.kmerDB) {
                DbIndex idx = read_index(db);
                if (k == 0) { k = idx.k; map_count = idx.map_count; }
                if (k != idx.k) { fprintf(stderr, "Cannot merge databases with different kmer length.\n"); exit(1); }
            }
            if (k == 0 || k > 32) { fprintf(stderr, "Invalid kmer length.\n"); exit(1); }
            e.k = k; e.map_count = map_count;
            // DBG::kunion (src/graph-builder.cpp:297-351): largest database first (:341-344); the databases are merged MAP
            // RANGE BY MAP RANGE (mergeMaps map by map, :341-347): for every range the first database's maps are imported,
            // every further database's maps of the range are loaded into a table of their own and merged region by region
            // on the device (kq_merge = mergeSubMaps), and the range is summarised / written before the next one is loaded.
            // One range when everything fits the HBM (and -m); the host holds a few maps' entries at a time either way.
            std::vector<std::pair<uint64_t, size_t>> order;
            for (size_t i = 0; i < ui.kmerDB.size(); ++i) order.emplace_back(db_bytes(ui.kmerDB[i], map_count), i);
            std::sort(order.rbegin(), order.rend());
            std::vector<DbSource> dbs;
            for (auto& o : order) dbs.emplace_back(ui.kmerDB[o.second]);
            uint64_t total = 0;
            for (auto& d : dbs) total += d.entries_bound(0, map_count);   // 24-byte slots at load <= 7/8: an upper bound of the entries
            int passes = ui.passes ? std::min(ui.passes, map_count) : std::min(passes_for(ui, total + total / 2), map_count);   // (+ the source table of a merge)
            if (passes > 1) verbose("Merging in " + std::to_string(passes) + " map ranges");
            uint64_t cap = 0;
            for (int p = 0; p < passes; ++p) {
                const auto r = pass_range(p, passes, map_count);
                uint64_t c = 0;
                for (auto& d : dbs) c += d.entries_bound(r.first, r.second);
                cap = std::max(cap, c);
            }
            e.create(cap + 1024);
            verbose("DBG object generated. Merging.");
            std::string ext = "stdout";
            if (ui.outFile != "") ext = file_ext("." + ui.outFile);
            const bool want_stats = ui.outFile.find(".") != std::string::npos || ui.outFile == "" || ext == "kreeq";
            kq_stats sum{};
            std::map<uint64_t, uint64_t> hist;
            std::vector<kq_entry> hc_all;
            for (int p = 0; p < passes; ++p) {
                const auto r = pass_range(p, passes, map_count);
                if (p) kq_or_die(kq_clear(e.h));
                dbs[0].import_into(e.h, r.first, r.second);
                for (size_t n = 1; n < dbs.size(); ++n) {
                    kq_handle* src = nullptr;
                    kq_or_die(kq_create(&src, ui.device, k, map_count, dbs[n].entries_bound(r.first, r.second) + 1024));
                    dbs[n].import_into(src, r.first, r.second);
                    kq_or_die(kq_set_option(e.h, KQ_OPT_MERGE_PATH, 2));
                    kq_or_die(kq_merge(e.h, src));
                    kq_destroy(src);
                }
                kq_stats st;
                kq_or_die(kq_summary(e.h, &st));
                sum.total += st.total; sum.unique += st.unique; sum.distinct += st.distinct; sum.edges += st.edges;
                if (ext == "hist") {
                    uint64_t n = 0;
                    kq_or_die(kq_histogram(e.h, nullptr, nullptr, 0, &n));
                    std::vector<uint64_t> cov((size_t)n), cnt((size_t)n);
                    if (n) kq_or_die(kq_histogram(e.h, cov.data(), cnt.data(), n, &n));
                    for (uint64_t i = 0; i < n; ++i) hist[cov[i]] += cnt[i];
                }
                if (ext == "kreeq") {
                    uint64_t n = 0;
                    kq_or_die(kq_export(e.h, (uint16_t)r.first, (uint16_t)r.second, nullptr, 0, &n));
                    std::vector<kq_entry> entries((size_t)n);
                    if (n) kq_or_die(kq_export(e.h, (uint16_t)r.first, (uint16_t)r.second, entries.data(), n, &n));
                    write_db_maps(ui.outFile, map_count, r.first, r.second, entries, hc_all);
                }
            }
            if (want_stats) {
                const uint64_t space = k < 32 ? (1ull << (2 * k)) : 0ull;
                std::cout << "DBG Summary statistics:\n"
                          << "Total kmers: " << sum.total << "\n" << "Unique kmers: " << sum.unique << "\n" << "Distinct kmers: " << sum.distinct << "\n"
                          << "Missing kmers: " << (space - sum.distinct) << "\n" << "Total edges: " << sum.edges << "\n";
            }
            if (ext == "kreeq") { write_db_finish(ui.outFile, k, map_count, hc_all); verbose("Database written"); }
            if (ext == "hist") { std::ofstream ofs(ui.outFile); for (auto& kv : hist) ofs << kv.first << "\t" << kv.second << "\n"; }
            verbose("Peak host memory: " + std::to_string(peak_rss_mb()) + " MB");
            break;
        }
        default:
            fprintf(stderr, "Invalid mode.\n");
            exit(1);
    }
    kq_destroy(e.h);
    return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char** argv) {
    UserInput ui;
    if (argc == 1 || argc == 2) print_help();
    std::string mode = argv[1];
    if (mode == "dbtool") {           // host-only helper (no GPU): `dbtool dump <db>` | `dbtool rewrite <in> <out>`
        try {
            std::vector<kq_entry> entries;
            DbIndex idx;
            if (argc == 4 && std::string(argv[2]) == "dump") {
                read_db(argv[3], entries, &idx);
                printf("#k=%d map_count=%d columns: map key fw0 fw1 fw2 fw3 bw0 bw1 bw2 bw3 cov hc\n", idx.k, idx.map_count);
                std::sort(entries.begin(), entries.end(), [](const kq_entry& a, const kq_entry& b) { return a.key < b.key; });
                for (auto& e : entries)
                    printf("%llu %llu %u %u %u %u %u %u %u %u %u %u\n", (unsigned long long)(e.key % (uint64_t)idx.map_count),
                           (unsigned long long)e.key, e.fw[0], e.fw[1], e.fw[2], e.fw[3], e.bw[0], e.bw[1], e.bw[2], e.bw[3], e.cov, e.hc);
                return EXIT_SUCCESS;
            }
            if (argc == 5 && std::string(argv[2]) == "rewrite") {
                read_db(argv[3], entries, &idx);
                write_db(argv[4], idx.k, idx.map_count, entries);
                return EXIT_SUCCESS;
            }
            if (argc >= 4 && std::string(argv[2]) == "seqsum") {       // order-independent digest of the sequences a reader yields
                const unsigned threads = argc > 4 ? (unsigned)atoi(argv[4]) : 0;
                const size_t batch = argc > 5 ? (size_t)atoll(argv[5]) : ((size_t)1 << 20);
                unsigned long long n = 0, bases = 0, digest = 0, batches = 0;
                const char* fail_after = getenv("KQ_TEST_FAIL_AFTER_BATCHES");     // failure-path test: the consumer throws mid-file
                auto eat = [&](const std::string& b) {
                    if (fail_after && ++batches > (unsigned long long)atoll(fail_after)) throw std::runtime_error("consumer failed (injected)");
                    size_t i = 0;
                    while (i <= b.size()) {
                        size_t j = b.find('\n', i);
                        if (j == std::string::npos) j = b.size();
                        unsigned long long hsh = 1469598103934665603ull;
                        for (size_t c = i; c < j; ++c) { hsh ^= (unsigned char)b[c]; hsh *= 1099511628211ull; }
                        ++n; bases += j - i; digest += hsh;
                        i = j + 1;
                    }
                };
                if (threads == 0) read_batches(argv[3], batch, eat); else read_batches_parallel(argv[3], batch, threads, eat);
                printf("%llu %llu %llu\n", n, bases, digest);
                return EXIT_SUCCESS;
            }
            if (argc >= 6 && std::string(argv[2]) == "seqsink") {      // the same digest through the zero-copy sink reader (plain heap buffers)
                const unsigned threads = (unsigned)atoi(argv[4]);
                const size_t cap = (size_t)atoll(argv[5]);
                unsigned long long n = 0, bases = 0, digest = 0;
                std::mutex m;
                std::vector<std::vector<char>> bufs(std::max(1u, threads));
                BatchSink sink;
                sink.acquire = [&](unsigned t, size_t* c) { bufs[t].resize(cap); *c = cap; return bufs[t].data(); };
                if (!getenv("KQ_TEST_NO_BIG")) {      // (a sink without one-off buffers refuses over-long sequences)
                    sink.acquire_big = [&](unsigned, size_t nb) { return (char*)malloc(nb); };
                    sink.submit_big = [&](unsigned t, char* b, size_t len) { sink.submit(t, b, len); free(b); };
                }
                sink.submit = [&](unsigned, char* b, size_t len) {
                    if (!len) return;
                    unsigned long long ln = 0, lb = 0, ld = 0;
                    size_t i = 0;
                    while (i <= len) {
                        const char* nl = (const char*)memchr(b + i, '\n', len - i);
                        const size_t j = nl ? (size_t)(nl - b) : len;
                        unsigned long long hsh = 1469598103934665603ull;
                        for (size_t c = i; c < j; ++c) { hsh ^= (unsigned char)b[c]; hsh *= 1099511628211ull; }
                        ++ln; lb += j - i; ld += hsh;
                        i = j + 1;
                    }
                    std::lock_guard<std::mutex> l(m);
                    n += ln; bases += lb; digest += ld;
                };
                read_batches_sink(argv[3], threads, sink);
                printf("%llu %llu %llu\n", n, bases, digest);
                return EXIT_SUCCESS;
            }
        } catch (const std::exception& ex) { fprintf(stderr, "%s\n", ex.what()); return EXIT_FAILURE; }
        fprintf(stderr, "usage: kreeq dbtool dump <db.kreeq> | rewrite <in.kreeq> <out.kreeq> | seqsum <fastx> [threads] [batch_bytes]\n");
        return EXIT_FAILURE;
    }
    if (mode == "validate") ui.mode = 0;
    else if (mode == "union") ui.mode = 1;
    else if (mode == "subgraph") { fprintf(stderr, "mode subgraph is not part of this build (validate / union only)\n"); return EXIT_FAILURE; }
    else { fprintf(stderr, "mode %s does not exist. Terminating\n", argv[1]); return EXIT_FAILURE; }

    if (ui.mode == 0) {
        static struct option long_options[] = {                      // src/main.cpp:78-97
            {"coverage-cutoff", required_argument, 0, 'c'}, {"database", required_argument, 0, 'd'},
            {"input-positions", required_argument, 0, 'p'}, {"input-sequence", required_argument, 0, 'f'},
            {"kmer-length", required_argument, 0, 'k'}, {"search-depth", required_argument, 0, 0},
            {"max-span", required_argument, 0, 0}, {"out-format", required_argument, 0, 'o'},
            {"input-reads", required_argument, 0, 'r'}, {"tmp-prefix", required_argument, 0, 't'},
            {"max-memory", required_argument, 0, 'm'}, {"threads", required_argument, 0, 'j'},
            {"device", required_argument, 0, 0}, {"passes", required_argument, 0, 0},
            {"verbose", no_argument, &verbose_flag, 1}, {"cmd", no_argument, &cmd_flag, 1},
            {"version", no_argument, 0, 'v'}, {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
        for (;;) {
            int option_index = 0;
            int c = getopt_long(argc, argv, "-:c:d:f:k:o:p:r:t:m:j:vh", long_options, &option_index);
            if (c == -1) break;
            switch (c) {
                case ':': fprintf(stderr, "option -%c is missing a required argument\n", optopt); return EXIT_FAILURE;
                case 0:
                    if (strcmp(long_options[option_index].name, "search-depth") == 0) ui.kmerDepth = atoi(optarg);
                    if (strcmp(long_options[option_index].name, "max-span") == 0) ui.maxSpan = atoi(optarg);
                    if (strcmp(long_options[option_index].name, "device") == 0) ui.device = atoi(optarg);
                    if (strcmp(long_options[option_index].name, "passes") == 0) ui.passes = std::max(0, atoi(optarg));
                    break;
                case 'c':
                    if (!is_number(optarg)) { fprintf(stderr, "input '%s' to option -%c must be a number\n", optarg, optopt); return EXIT_FAILURE; }
                    ui.covCutOff = (uint32_t)atoi(optarg);
                    break;
                case 'd': if_file_exists(optarg); ui.kmerDB.push_back(optarg); break;
                case 'f': if_file_exists(optarg); ui.inSequence = optarg; break;
                case 'k':
                    if (!is_number(optarg)) { fprintf(stderr, "input '%s' to option -%c must be a number\n", optarg, optopt); return EXIT_FAILURE; }
                    ui.kmerLen = atoi(optarg);
                    break;
                case 'j': ui.maxThreads = atoi(optarg); break;
                case 'o': ui.outFile = optarg; break;
                case 'p': if_file_exists(optarg); ui.inBedInclude = optarg; break;
                case 'r':                                            // consumes several files, src/main.cpp:169-180
                    optind--;
                    for (; optind < argc && *argv[optind] != '-' && !is_int(argv[optind]); optind++) {
                        if_file_exists(argv[optind]);
                        ui.inReads.push_back(argv[optind]);
                    }
                    break;
                case 't': ui.prefix = optarg; break;
                case 'm': ui.maxMem = atof(optarg); break;
                case 'v': printf("kreeq v%s (MI355X build)\n", kVersion); exit(0);
                case 'h':
                    printf("kreeq [command]\n\nOptions:\n");
                    printf("\t-c --coverage-cutoff coverage cutoff.\n");
                    printf("\t-d --database kreeq database to load.\n");
                    printf("\t-f --input-sequence sequence input file (fasta,fastq).\n");
                    printf("\t-r --input-reads read input files (fastq).\n");
                    printf("\t-k --kmer-length length of kmers.\n");
                    printf("\t-o --out-format supported extensions:\n");
                    printf("\t\t .kreeq dumps hashmaps to file for reuse; .kwig .bkwig per-base tables; .hist coverage histogram.\n");
                    printf("\t-t --tmp-prefix prefix to temporary directory (unused: the table lives in HBM).\n");
                    printf("\t-m --max-memory <GB> HBM the k-mer table may use (default: 60 %% of what is free); bounds the map ranges counted at a time.\n");
                    printf("\t-j --threads <n> parser threads for the read files (default: all cores).\n");
                    printf("\t--device <n> GPU to use (default 0).\n");
                    printf("\t--passes <n> count the reads n times, one range of the hash maps per pass (HBM use = 1/n of the table); default: as many as -m asks for.\n");
                    printf("\t-v --version software version.\n");
                    printf("\t--cmd print $0 to stdout.\n");
                    exit(0);
                default: break;                                      // positional arguments are ignored like the reference
            }
        }
        if (ui.kmerLen < 2 || ui.kmerLen > 32) { fprintf(stderr, "Invalid kmer length.\n"); return EXIT_FAILURE; }
    } else {
        static struct option long_options[] = {                      // src/main.cpp:220-229
            {"databases", required_argument, 0, 'd'}, {"out-format", required_argument, 0, 'o'},
            {"threads", required_argument, 0, 'j'}, {"device", required_argument, 0, 0},
            {"max-memory", required_argument, 0, 'm'}, {"passes", required_argument, 0, 0},     // (this build: the HBM bound of the merge)
            {"verbose", no_argument, &verbose_flag, 1}, {"cmd", no_argument, &cmd_flag, 1},
            {"help", no_argument, 0, 'h'}, {0, 0, 0, 0}};
        for (;;) {
            int option_index = 1;
            int c = getopt_long(argc, argv, "-:d:j:o:m:h", long_options, &option_index);
            if (c == -1) break;
            switch (c) {
                case ':': fprintf(stderr, "option -%c is missing a required argument\n", optopt); return EXIT_FAILURE;
                case 0:
                    if (strcmp(long_options[option_index].name, "device") == 0) ui.device = atoi(optarg);
                    if (strcmp(long_options[option_index].name, "passes") == 0) ui.passes = std::max(0, atoi(optarg));
                    break;
                case 'm': ui.maxMem = atof(optarg); break;
                case 'd':
                    optind--;
                    for (; optind < argc && *argv[optind] != '-' && !is_int(argv[optind]); optind++) {
                        if_file_exists(argv[optind]);
                        ui.kmerDB.push_back(argv[optind]);
                    }
                    break;
                case 'j': ui.maxThreads = atoi(optarg); break;
                case 'o': ui.outFile = optarg; break;
                case 'h':
                    printf("kreeq union [options]\n\nOptions:\n");
                    printf("\t-d --databases DBG databases to merge.\n");
                    printf("\t-j --threads <n> accepted for compatibility.\n");
                    printf("\t-o --out-format generates various kinds of outputs (currently supported: .kreeq).\n");
                    printf("\t-m --max-memory <GB> HBM the merge may use (default: 60 %% of what is free); bounds the map ranges merged at a time.\n");
                    printf("\t--passes <n> merge the databases in n ranges of the hash maps.\n");
                    printf("\t--cmd print $0 to stdout.\n");
                    exit(0);
                default: break;
            }
        }
        if (ui.kmerDB.size() < 2) { fprintf(stderr, "At least two databases required (-d).\n"); return EXIT_FAILURE; }   // src/main.cpp:290-293
    }
    if (cmd_flag) {
        for (int i = 0; i < argc; ++i) printf("%s ", argv[i]);
        printf("\n");
    }
    try {
        return run(ui);
    } catch (const std::exception& ex) {
        fprintf(stderr, "%s\n", ex.what());
        return EXIT_FAILURE;
    }
}

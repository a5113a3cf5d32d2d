#include "fastx.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace kqhost {

namespace {

class LineReader {
    gzFile f_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool fill() {
        int n = gzread(f_, buf_.data(), (unsigned)buf_.size());
        if (n < 0) throw std::runtime_error("read error");
        pos_ = 0; end_ = (size_t)n;
        return n > 0;
    }
public:
    explicit LineReader(const std::string& path) : buf_(1 << 22) {
        f_ = gzopen(path.c_str(), "rb");                 // transparently reads plain files too
        if (!f_) throw std::runtime_error("Stream not successful: " + path);
        gzbuffer(f_, 1 << 20);
    }
    ~LineReader() { if (f_) gzclose(f_); }
    int peek() { if (pos_ == end_ && !fill()) return -1; return (unsigned char)buf_[pos_]; }
    // reads up to (not including) delim; returns false at EOF with nothing read
    bool getline(std::string& out, char delim = '\n') {
        out.clear();
        bool any = false;
        for (;;) {
            if (pos_ == end_ && !fill()) return any;
            any = true;
            size_t i = pos_;
            while (i < end_ && buf_[i] != delim) ++i;
            out.append(buf_.data() + pos_, i - pos_);
            if (i < end_) { pos_ = i + 1; return true; }
            pos_ = end_;
        }
    }
};

void split_header(const std::string& line, SeqRecord& r) {
    size_t sp = line.find(' ');
    r.header = line.substr(0, sp);
    r.comment = sp == std::string::npos ? std::string() : line.substr(sp + 1);
    while (!r.header.empty() && (r.header.back() == '\r')) r.header.pop_back();
}

}  // namespace

void read_fastx(const std::string& path, const std::function<void(SeqRecord&&)>& on_record) {
    LineReader in(path);
    int c = in.peek();
    std::string line;
    if (c == '>') {
        std::string body;
        in.getline(line, '>');                            // consume the leading '>' (empty field)
        while (in.getline(line)) {                        // header line
            SeqRecord r;
            split_header(line, r);
            in.getline(body, '>');                        // everything up to the next record
            r.seq.reserve(body.size());
            for (char ch : body) if (ch != '\n' && ch != '\r') r.seq.push_back(ch);
            on_record(std::move(r));
        }
    } else if (c == '@') {
        std::string plus, qual;
        while (in.getline(line)) {
            if (line.empty()) continue;
            SeqRecord r;
            split_header(line.substr(1), r);
            in.getline(r.seq);
            while (!r.seq.empty() && r.seq.back() == '\r') r.seq.pop_back();
            in.getline(plus);
            in.getline(qual);
            on_record(std::move(r));
        }
    } else if (c == -1) {
        return;
    } else if (c == 'H' || c == 'S') {
        // GFA: only the S-line sequences matter to validate (each segment is looked up on its own;
        // reference src/input.cpp:287-291 -> gfalibs readGFA).  GFA2 S lines carry the length first.
        bool gfa2 = false;
        while (in.getline(line)) {
            while (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.rfind("H\t", 0) == 0 && line.find("VN:Z:2") != std::string::npos) gfa2 = true;
            if (line.rfind("S\t", 0) != 0) continue;
            std::vector<std::string> f;
            size_t i = 0;
            while (i <= line.size()) { size_t j = line.find('\t', i); if (j == std::string::npos) j = line.size(); f.push_back(line.substr(i, j - i)); i = j + 1; }
            const size_t si = gfa2 ? 3 : 2;
            if (f.size() <= si || f[si] == "*") continue;
            SeqRecord r;
            r.header = f[1];
            r.seq = f[si];
            on_record(std::move(r));
        }
    } else {
        throw std::runtime_error("unsupported sequence format (FASTA/FASTQ/GFA expected): " + path);
    }
}

void read_batches(const std::string& path, size_t batch_bytes, const std::function<void(const std::string&)>& on_batch) {
    std::string batch;
    batch.reserve(batch_bytes + (1 << 16));
    read_fastx(path, [&](SeqRecord&& r) {
        if (!batch.empty()) batch.push_back('\n');
        batch += r.seq;
        if (batch.size() >= batch_bytes) { on_batch(batch); batch.clear(); }
    });
    if (!batch.empty()) on_batch(batch);
}

namespace {

class BatchQueue {                       // bounded multi-producer / single-consumer queue
    std::mutex m_;
    std::condition_variable not_full_, not_empty_;
    std::deque<std::string> q_;
    size_t cap_;
    unsigned producers_;
    std::string error_;
    bool cancelled_ = false;
public:
    BatchQueue(size_t cap, unsigned producers) : cap_(cap), producers_(producers) {}
    // false: the consumer gave up (its callback threw); the producer must stop
    bool push(std::string&& b) {
        std::unique_lock<std::mutex> l(m_);
        not_full_.wait(l, [&] { return q_.size() < cap_ || cancelled_; });
        if (cancelled_) return false;
        q_.push_back(std::move(b));
        not_empty_.notify_one();
        return true;
    }
    void cancel() {
        std::lock_guard<std::mutex> l(m_);
        cancelled_ = true;
        q_.clear();
        not_full_.notify_all();
    }
    void producer_done(const std::string& err = std::string()) {
        std::lock_guard<std::mutex> l(m_);
        if (!err.empty() && error_.empty()) error_ = err;
        --producers_;
        not_empty_.notify_all();
    }
    bool pop(std::string& out) {         // false when all producers are done and the queue is empty
        std::unique_lock<std::mutex> l(m_);
        not_empty_.wait(l, [&] { return !q_.empty() || producers_ == 0; });
        if (q_.empty()) { if (!error_.empty()) throw std::runtime_error(error_); return false; }
        out = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return true;
    }
};

inline const char* next_line(const char* p, const char* end) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    return nl ? nl + 1 : end;
}
// first FASTQ record starting at or after `p`: a line starting with '@' whose line + 2 starts with '+'
// (a quality line may start with '@', but then line + 2 is a sequence line)
const char* fastq_sync(const char* p, const char* begin, const char* end) {
    if (p != begin) p = next_line(p - 1, end);          // start of the line after the one containing p-1
    while (p < end) {
        if (*p == '@') {
            const char* l1 = next_line(p, end);
            const char* l2 = next_line(l1, end);
            if (l2 < end && *l2 == '+') return p;
            if (l2 >= end) return p;                     // truncated tail: let the parser deal with it
        }
        p = next_line(p, end);
    }
    return end;
}
const char* fasta_sync(const char* p, const char* begin, const char* end) {
    if (p != begin) p = next_line(p - 1, end);
    while (p < end && *p != '>') p = next_line(p, end);
    return p;
}

void append_seq(std::string& batch, const char* a, const char* b) {
    while (b > a && (b[-1] == '\n' || b[-1] == '\r')) --b;
    if (!batch.empty()) batch.push_back('\n');
    batch.append(a, (size_t)(b - a));
}

}  // namespace

void read_batches_parallel(const std::string& path, size_t batch_bytes, unsigned threads,
                           const std::function<void(const std::string&)>& on_batch) {
    if (threads < 1) threads = 1;
    struct stat st;
    const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    if (!gz && stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { data = (const char*)m; size = (size_t)st.st_size; madvise(m, size, MADV_SEQUENTIAL); }
        }
    }
    if (!data || (data[0] != '@' && data[0] != '>')) {
        // compressed / piped / unknown: one producer thread runs the sequential reader
        if (data) { munmap((void*)data, size); }
        if (fd >= 0) close(fd);
        BatchQueue q(4, 1);
        std::thread prod([&] {
            try {
                read_batches(path, batch_bytes, [&](const std::string& b) { if (!q.push(std::string(b))) throw std::runtime_error("cancelled"); });
                q.producer_done();
            } catch (const std::exception& e) { q.producer_done(e.what()); }
        });
        std::string b;
        try { while (q.pop(b)) on_batch(b); } catch (...) { q.cancel(); prod.join(); throw; }
        prod.join();
        return;
    }
    const bool fastq = data[0] == '@';
    const char* begin = data;
    const char* end = data + size;
    const size_t chunk = std::max<size_t>((size_t)8 << 20, std::min<size_t>((size_t)64 << 20, size / (threads * 4) + 1));
    const size_t n_chunks = (size + chunk - 1) / chunk;
    std::mutex next_m;
    size_t next_chunk = 0;
    BatchQueue q(threads + 2, threads);
    auto worker = [&] {
        try {
            std::string batch;
            batch.reserve(batch_bytes + (1 << 16));
            for (;;) {
                size_t c;
                { std::lock_guard<std::mutex> l(next_m); c = next_chunk++; }
                if (c >= n_chunks) break;
                const char* lo = begin + c * chunk;
                const char* hi = std::min(end, lo + chunk);
                const char* p = fastq ? fastq_sync(lo, begin, end) : fasta_sync(lo, begin, end);
                if (c == 0 && p != begin) throw std::runtime_error("malformed " + std::string(fastq ? "FASTQ" : "FASTA") + " record at the start of " + path);
                while (p < hi) {                          // records whose header starts inside [lo, hi)
                    if (fastq) {
                        const char* seq = next_line(p, end);
                        const char* plus = next_line(seq, end);
                        const char* qual = next_line(plus, end);
                        // four-line records only, like the reference's loader (src/input.cpp:245-286): anything else
                        // (wrapped sequence / quality lines) is refused instead of being mis-parsed silently
                        if (*p != '@' || (plus < end && *plus != '+'))
                            throw std::runtime_error("malformed FASTQ record (four-line records expected) at byte " + std::to_string((size_t)(p - begin)) + " of " + path);
                        append_seq(batch, seq, plus);
                        p = next_line(qual, end);
                    } else {
                        const char* seq = next_line(p, end);
                        const char* nxt = seq;
                        while (nxt < end && *nxt != '>') nxt = next_line(nxt, end);
                        if (!batch.empty()) batch.push_back('\n');
                        for (const char* s = seq; s < nxt;) {           // drop the line breaks (src/input.cpp:225)
                            const char* e = next_line(s, nxt);
                            const char* t = e;
                            while (t > s && (t[-1] == '\n' || t[-1] == '\r')) --t;
                            batch.append(s, (size_t)(t - s));
                            s = e;
                        }
                        p = nxt;
                    }
                    // the records of a chunk must end where the next chunk finds its first one: anything else means the
                    // resynchronisation and the parser disagree about the record structure (wrapped FASTQ lines ...)
                    if (p >= hi && hi < end) {
                        const char* want = fastq ? fastq_sync(hi, begin, end) : fasta_sync(hi, begin, end);
                        if (p != want) throw std::runtime_error("malformed " + std::string(fastq ? "FASTQ (four-line records expected)" : "FASTA") + " near byte " + std::to_string((size_t)(hi - begin)) + " of " + path);
                    }
                    if (batch.size() >= batch_bytes) {
                        if (!q.push(std::move(batch))) { q.producer_done(); return; }
                        batch.clear(); batch.reserve(batch_bytes + (1 << 16));
                    }
                }
            }
            if (!batch.empty()) q.push(std::move(batch));
            q.producer_done();
        } catch (const std::exception& e) { q.producer_done(e.what()); }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker);
    std::string b;
    try { while (q.pop(b)) on_batch(b); }
    catch (...) {                                  // workers still read the mapping: stop them, join, only then unmap
        q.cancel();
        for (auto& t : pool) t.join();
        munmap((void*)data, size); close(fd);
        throw;
    }
    for (auto& t : pool) t.join();
    munmap((void*)data, size);
    close(fd);
}

namespace {

// one parser thread's current buffer of a BatchSink
struct Appender {
    const BatchSink& sink;
    unsigned t;
    char* buf = nullptr;
    size_t cap = 0, len = 0, pool_cap = 0;
    bool big = false;                  // buf is a one-off buffer for a sequence longer than a pool buffer
    Appender(const BatchSink& s, unsigned thread) : sink(s), t(thread) {}
    // room for a sequence of n bases (and its separator) in the current buffer, else submit it and take a fresh one
    void reserve(size_t n) {
        if (buf && len + n + 1 <= cap) return;
        flush();
        if (pool_cap && n + 1 > pool_cap) { take_big(n); return; }
        buf = sink.acquire(t, &cap);
        pool_cap = cap;
        len = 0;
        if (n + 1 > cap) {
            // a sequence longer than a pool buffer (a chromosome of a FASTA given as reads): the pool buffer goes back empty
            // and the sequence travels in a buffer of its own -- whole, so its k-mers and edges are counted exactly once
            sink.submit(t, buf, 0);
            buf = nullptr;
            take_big(n);
        }
    }
    void take_big(size_t n) {
        if (!sink.acquire_big || !sink.submit_big)
            throw std::runtime_error("sequence of " + std::to_string(n) + " bases does not fit a " + std::to_string(pool_cap) + "-byte read batch");
        buf = sink.acquire_big(t, n + 1);
        cap = n + 1; len = 0; big = true;
    }
    void begin() { if (len) buf[len++] = '\n'; }
    void piece(const char* a, const char* b) { memcpy(buf + len, a, (size_t)(b - a)); len += (size_t)(b - a); }
    void flush() {
        if (buf && big) sink.submit_big(t, buf, len);
        else if (buf && len) sink.submit(t, buf, len);
        else if (buf) sink.submit(t, buf, 0);                     // taken but never filled: back to the pool
        buf = nullptr; len = 0; big = false;
    }
};
inline const char* trim_eol(const char* a, const char* b) { while (b > a && (b[-1] == '\n' || b[-1] == '\r')) --b; return b; }

}  // namespace

void read_batches_sink(const std::string& path, unsigned threads, const BatchSink& sink) {
    if (threads < 1) threads = 1;
    struct stat st;
    const bool gz = path.size() > 3 && path.compare(path.size() - 3, 3, ".gz") == 0;
    int fd = -1;
    const char* data = nullptr;
    size_t size = 0;
    if (!gz && stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        fd = open(path.c_str(), O_RDONLY);
        if (fd >= 0) {
            void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) { data = (const char*)m; size = (size_t)st.st_size; madvise(m, size, MADV_SEQUENTIAL); }
        }
    }
    if (!data || (data[0] != '@' && data[0] != '>')) {
        // compressed / piped / unknown: the sequential reader feeds buffer after buffer on this thread
        if (data) munmap((void*)data, size);
        if (fd >= 0) close(fd);
        Appender ap(sink, 0);
        read_fastx(path, [&](SeqRecord&& r) { ap.reserve(r.seq.size()); ap.begin(); ap.piece(r.seq.data(), r.seq.data() + r.seq.size()); });
        ap.flush();
        return;
    }
    const bool fastq = data[0] == '@';
    const char* begin = data;
    const char* end = data + size;
    const size_t chunk = std::max<size_t>((size_t)4 << 20, std::min<size_t>((size_t)32 << 20, size / (threads * 4) + 1));
    const size_t n_chunks = (size + chunk - 1) / chunk;
    std::mutex next_m;
    size_t next_chunk = 0;
    std::string first_error;
    bool failed = false;
    auto worker = [&](unsigned t) {
        Appender ap(sink, t);
        try {
            for (;;) {
                size_t c;
                { std::lock_guard<std::mutex> l(next_m); if (failed) break; c = next_chunk++; }
                if (c >= n_chunks) break;
                const char* lo = begin + c * chunk;
                const char* hi = std::min(end, lo + chunk);
                const char* p = fastq ? fastq_sync(lo, begin, end) : fasta_sync(lo, begin, end);
                if (c == 0 && p != begin) throw std::runtime_error("malformed " + std::string(fastq ? "FASTQ" : "FASTA") + " record at the start of " + path);
                while (p < hi) {
                    if (fastq) {
                        const char* seq = next_line(p, end);
                        const char* plus = next_line(seq, end);
                        const char* qual = next_line(plus, end);
                        if (*p != '@' || (plus < end && *plus != '+'))
                            throw std::runtime_error("malformed FASTQ record (four-line records expected) at byte " + std::to_string((size_t)(p - begin)) + " of " + path);
                        const char* e = trim_eol(seq, plus);
                        ap.reserve((size_t)(e - seq));
                        ap.begin();
                        ap.piece(seq, e);
                        p = next_line(qual, end);
                    } else {
                        const char* seq = next_line(p, end);
                        const char* nxt = seq;
                        size_t total = 0;
                        while (nxt < end && *nxt != '>') { const char* e = next_line(nxt, end); total += (size_t)(trim_eol(nxt, e) - nxt); nxt = e; }
                        ap.reserve(total);
                        ap.begin();
                        for (const char* s = seq; s < nxt;) { const char* e = next_line(s, nxt); ap.piece(s, trim_eol(s, e)); s = e; }
                        p = nxt;
                    }
                    if (p >= hi && hi < end) {
                        const char* want = fastq ? fastq_sync(hi, begin, end) : fasta_sync(hi, begin, end);
                        if (p != want) throw std::runtime_error("malformed " + std::string(fastq ? "FASTQ (four-line records expected)" : "FASTA") + " near byte " + std::to_string((size_t)(hi - begin)) + " of " + path);
                    }
                }
            }
            ap.flush();
        } catch (const std::exception& e) {
            bool first = false;
            { std::lock_guard<std::mutex> l(next_m); if (!failed) { failed = true; first = true; first_error = e.what(); } }
            if (first && sink.abort) sink.abort();                // the other threads may be waiting for buffers this one holds
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t) pool.emplace_back(worker, t);
    for (auto& t : pool) t.join();
    munmap((void*)data, size);
    close(fd);
    if (failed) throw std::runtime_error(first_error);
}

}  // namespace kqhost

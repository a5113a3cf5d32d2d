#include "fastx.h"

#include <zlib.h>

#include <stdexcept>

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
    } else {
        throw std::runtime_error("unsupported sequence format (FASTA/FASTQ expected): " + path);
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

}  // namespace kqhost

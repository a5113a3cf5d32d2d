// fastx.h -- FASTA / FASTQ (optionally gzipped) reader for the host CLI.
// Replaces the pieces of gfalibs StreamObj / loadKmers / Input::loadGenome the hot path needs
// (reference src/input.cpp:188-286).  GFA input is out of scope.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace kqhost {

struct SeqRecord { std::string header, comment, seq; };

// Calls on_record for every sequence of a FASTA ('>') or FASTQ ('@') file; .gz handled by zlib.
// FASTA sequence text has its line breaks removed (src/input.cpp:225).  Throws std::runtime_error.
void read_fastx(const std::string& path, const std::function<void(SeqRecord&&)>& on_record);

// Streams the reads of a file as batches: sequences joined by '\n' (any non-ACGT byte ends a run, so
// k-mers never span two reads), at most ~batch_bytes each.
void read_batches(const std::string& path, size_t batch_bytes, const std::function<void(const std::string&)>& on_batch);

// Same contract, but the file is parsed by `threads` workers and handed to on_batch (called on the
// CALLING thread only) through a bounded queue, so parsing overlaps the GPU work done inside
// on_batch.  Plain FASTQ/FASTA files are mmap'ed and cut into chunks at record boundaries; .gz
// input is inflated by one producer thread.  Batch order is unspecified (counting is commutative).
// Replaces gfalibs loadKmers' reader thread + readBatches queue (reference src/input.cpp:95-96,
// src/graph-builder.cpp:41-58).
void read_batches_parallel(const std::string& path, size_t batch_bytes, unsigned threads,
                           const std::function<void(const std::string&)>& on_batch);

// Zero-copy variant for the count path: the parser threads write the sequences straight into buffers the sink hands out
// (pinned host memory in the CLI) and submit full buffers themselves -- no queue, no consumer thread, no extra copy.
//   acquire(thread, &cap)     a writable buffer of cap bytes for parser thread `thread` (may block until one is free)
//   submit(thread, buf, len)  buf holds len bytes of '\n'-separated sequences; the sink owns it again
// Both are called concurrently from up to `threads` threads (thread ids 0..threads-1); .gz input is inflated and parsed by
// one thread (id 0).
//   acquire_big(thread, n)    (optional) a one-off buffer of n bytes for a sequence that does not fit a pool buffer (a
//                             chromosome-scale FASTA record), handed back through submit_big; without it such a sequence is an error
//   abort()                   (optional) called once when a parser thread has failed: wakes threads blocked in acquire so that
//                             they throw instead of waiting for buffers that will never come back
struct BatchSink {
    std::function<char*(unsigned thread, size_t* cap)> acquire;
    std::function<void(unsigned thread, char* buf, size_t len)> submit;
    std::function<char*(unsigned thread, size_t n)> acquire_big;
    std::function<void(unsigned thread, char* buf, size_t len)> submit_big;
    std::function<void()> abort;
};
void read_batches_sink(const std::string& path, unsigned threads, const BatchSink& sink);

}  // namespace kqhost

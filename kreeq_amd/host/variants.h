// Candidate-error search behind `kreeq validate -o vcf` (reference src/variants.cpp); see variants.cpp.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "fastx.h"
#include "kreeq_amd.h"

namespace kqhost {

enum VariantType { VAR_SNV, VAR_INS, VAR_DEL, VAR_COM };        // gfalibs DBGpath::type as src/variants.cpp:283-297 assigns it

struct DbgPath {                                                // gfalibs DBGpath: one alternative path found at a position
    VariantType type = VAR_SNV;
    uint64_t pos = 0;                                           // first base behind the source k-mer, inside its segment (src/variants.cpp:139-140)
    std::string sequence;
    uint32_t ref_len = 0;                                       // COM only: original bases replaced
};
struct VariantSite { size_t seq_index = 0; uint64_t seg_start = 0; std::vector<DbgPath> paths; };

// Where the search gets the graph from.  One resident table (graph_of_handle), or a database that is loaded map range by
// map range (the reference's search loops over map ranges the same way, src/variants.cpp:78-84, with a cache of what it has
// seen, :199-210): branch_scan fills flags[c] for every k-mer start of `joined` (kq_branch_scan's contract), lookup the
// logical entry of every key of `want` (kq_lookup_keys' contract).
struct GraphSource {
    std::function<void(const std::string& joined, uint32_t cov_cutoff, std::vector<uint8_t>& flags)> branch_scan;
    std::function<void(const std::vector<uint64_t>& want, std::vector<kq_entry>& got)> lookup;
};
GraphSource graph_of_handle(kq_handle* h);

// DBG::correctSequences over all sequences: device pre-filter + host searches with batched device lookups
std::vector<VariantSite> find_candidate_errors(const GraphSource& g, int k, const std::vector<SeqRecord>& seqs, int kmer_depth, int max_span,
                                               uint32_t cov_cutoff, const std::function<void(const std::string&)>& log = nullptr);
inline std::vector<VariantSite> find_candidate_errors(kq_handle* h, int k, const std::vector<SeqRecord>& seqs, int kmer_depth, int max_span,
                                                      uint32_t cov_cutoff, const std::function<void(const std::string&)>& log = nullptr) {
    return find_candidate_errors(graph_of_handle(h), k, seqs, kmer_depth, max_span, cov_cutoff, log);
}
// the VCF text of the sites (header + one record per path)
std::vector<std::string> vcf_lines(const std::vector<SeqRecord>& seqs, const std::vector<VariantSite>& sites);

}  // namespace kqhost

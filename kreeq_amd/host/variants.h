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

// DBG::correctSequences over all sequences: device pre-filter + host searches with batched device lookups
std::vector<VariantSite> find_candidate_errors(kq_handle* h, int k, const std::vector<SeqRecord>& seqs, int kmer_depth, int max_span,
                                               uint32_t cov_cutoff, const std::function<void(const std::string&)>& log = nullptr);
// the VCF text of the sites (header + one record per path)
std::vector<std::string> vcf_lines(const std::vector<SeqRecord>& seqs, const std::vector<VariantSite>& sites);

}  // namespace kqhost

// C ABI of the host-side .kreeq (de)serialiser (kreeq_db.cpp) for callers outside the CLI: the multi-GPU driver
// (kreeq_amd/dist.py) has every rank write the map files it owns and rank 0 the high-copy map + .index, which is the
// reference's "separate databases + union" HPC flow (README.md:31-39) without the union step: the shards are
// bucket-disjoint, so the files of all ranks together ARE the one database.  No GPU code here.
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "kreeq_db.h"

using namespace kqhost;

static thread_local std::string g_err;

extern "C" {

const char* kqh_last_error(void) { return g_err.c_str(); }

// map files [map_lo, map_hi) from the logical entries of those maps; the high-copy entries among them are returned in
// hc_out (room for n entries), *n_hc = their number
int kqh_write_maps(const char* dir, int map_count, int map_lo, int map_hi, const kq_entry* entries, uint64_t n, kq_entry* hc_out, uint64_t* n_hc) {
    try {
        std::vector<kq_entry> v(entries, entries + n), hc;
        write_db_maps(dir, map_count, map_lo, map_hi, v, hc);
        if (n_hc) *n_hc = hc.size();
        if (hc_out && !hc.empty()) memcpy(hc_out, hc.data(), hc.size() * sizeof(kq_entry));
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// .map.hc.bin (all high-copy k-mers of the database) + .index
int kqh_write_finish(const char* dir, int k, int map_count, const kq_entry* hc, uint64_t n_hc) {
    try {
        std::vector<kq_entry> v(hc, hc + n_hc);
        write_db_finish(dir, k, map_count, v);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
// logical entries of a database; out == NULL: count only
int kqh_read_db(const char* dir, kq_entry* out, uint64_t cap, uint64_t* n_out, int* k, int* map_count) {
    try {
        std::vector<kq_entry> v;
        DbIndex idx;
        read_db(dir, v, &idx);
        if (n_out) *n_out = v.size();
        if (k) *k = idx.k;
        if (map_count) *map_count = idx.map_count;
        if (out) {
            if (v.size() > cap) { g_err = "entry buffer too small"; return -2; }
            if (!v.empty()) memcpy(out, v.data(), v.size() * sizeof(kq_entry));
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

}  // extern "C"

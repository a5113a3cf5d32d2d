// kreeq_db.h -- .kreeq database directory (de)serialiser, host side.
// Format decoded from the reference's fixture databases:
//   <db>/.index        "<k>\n<mapCount>\n"                     (reference src/kreeq-output.cpp:88-94)
//   <db>/.map.<m>.bin  phmap::parallel_flat_hash_map<u64, DBGkmer>   binary dump, m = key % mapCount
//   <db>/.map.hc.bin   phmap::parallel_flat_hash_map<u64, DBGkmer32> binary dump of ALL high-copy k-mers
// The writer emulates sequential insertion into parallel-hashmap's raw_hash_set (mix hash, submap
// selection, H2 control bytes, triangular group probing, growth_left), so that the reference's
// phmap_load (src/graph-builder.cpp:307-308) can read what we write.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "kreeq_amd.h"

namespace kqhost {

struct DbIndex { int k = 0; int map_count = 0; };

// throws std::runtime_error with a one-line message on malformed input
DbIndex read_index(const std::string& db_dir);
void write_index(const std::string& db_dir, int k, int map_count);

// Logical entries of one database: 8-bit maps (tombstones cov==255 dropped) + high-copy map.
// Appends to `out`.
void read_db(const std::string& db_dir, std::vector<kq_entry>& out, DbIndex* idx = nullptr);

// The same in pieces, for callers that hold only a range of maps at a time (the reference's loadMapRange, src/kreeq.cpp:59-74):
// read_db_hc reads the database's high-copy map once (it is small: the k-mers with cov >= 255); read_db_maps appends the
// logical entries of the maps [map_lo, map_hi) -- their 8-bit maps (tombstones dropped) and the high-copy k-mers of those maps.
void read_db_hc(const std::string& db_dir, std::vector<kq_entry>& hc_out);
void read_db_maps(const std::string& db_dir, const DbIndex& idx, int map_lo, int map_hi, const std::vector<kq_entry>& hc, std::vector<kq_entry>& out);

// Writes all map files.  `entries` = logical entries of the whole table (any order).
void write_db(const std::string& db_dir, int k, int map_count, const std::vector<kq_entry>& entries);

// The same in pieces, for runs that hold only a range of maps in memory at a time: write the map
// files of [map_lo, map_hi), collecting high-copy k-mers, and finish with .map.hc.bin + .index.
void write_db_maps(const std::string& db_dir, int map_count, int map_lo, int map_hi, const std::vector<kq_entry>& entries,
                   std::vector<kq_entry>& hc_out);
void write_db_finish(const std::string& db_dir, int k, int map_count, const std::vector<kq_entry>& hc_entries);

// exposed for tests: phmap's 64-bit mix and the submap index it derives
uint64_t phmap_mix64(uint64_t key);
unsigned phmap_submap(uint64_t hashval);

}  // namespace kqhost

#include "kreeq_db.h"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace kqhost {

namespace {

constexpr uint64_t kDumpVersion = 0xFFFFFFFFFFFFFFF5ull;
constexpr unsigned kSubmaps = 256;        // PM<T>: N = 8 (reference include/kreeq.h:138-144)
constexpr unsigned kGroup = 16;           // SSE2 group width
constexpr uint8_t kEmpty = 0x80, kSentinel = 0xFF;

struct Val8 { uint8_t fw[4], bw[4], cov; };           // DBGkmer   (include/kreeq.h:20-21), 9 B, slot 24 B
struct Val32 { uint32_t fw[4], bw[4], cov; };         // DBGkmer32 (include/kreeq.h:69-70), 36 B, slot 48 B

std::vector<uint8_t> slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> d((size_t)n);
    if (n) f.read((char*)d.data(), n);
    return d;
}

// one phmap dump -> (key, value bytes)
template <class V, class F>
void read_dump(const std::string& path, F&& on_entry) {
    std::vector<uint8_t> d = slurp(path);
    const size_t slot = (8 + sizeof(V) + 7) / 8 * 8;
    size_t off = 0;
    auto need = [&](size_t n) { if (off + n > d.size()) throw std::runtime_error("truncated map file " + path); };
    auto rd64 = [&]() { need(8); uint64_t v; memcpy(&v, d.data() + off, 8); off += 8; return v; };
    uint64_t nsub = rd64();
    for (uint64_t s = 0; s < nsub; ++s) {
        uint64_t ver = rd64(), size = rd64(), cap = rd64();
        if (ver != kDumpVersion) throw std::runtime_error("unexpected phmap dump version in " + path);
        if (size == 0) continue;
        need(cap + kGroup + 1);
        const uint8_t* ctrl = d.data() + off; off += cap + kGroup + 1;
        need(cap * slot);
        const uint8_t* slots = d.data() + off; off += cap * slot;
        (void)rd64();   // growth_left
        uint64_t seen = 0;
        for (uint64_t i = 0; i < cap; ++i) {
            if (ctrl[i] & 0x80) continue;
            uint64_t key; V v;
            memcpy(&key, slots + i * slot, 8);
            memcpy(&v, slots + i * slot + 8, sizeof(V));
            on_entry(key, v);
            ++seen;
        }
        if (seen != size) throw std::runtime_error("inconsistent submap size in " + path);
    }
    if (off != d.size()) throw std::runtime_error("trailing bytes in " + path);
}

inline uint64_t growth_of(uint64_t cap) { return cap - cap / 8; }   // CapacityToGrowth, group width 16

// sequential-insert emulation of one raw_hash_set
template <class V>
struct SubmapImage {
    uint64_t cap = 0, size = 0;
    std::vector<uint8_t> ctrl;
    std::vector<uint8_t> slots;
    static constexpr size_t kSlot = (8 + sizeof(V) + 7) / 8 * 8;

    void build(const std::vector<std::pair<uint64_t, V>>& items) {
        size = items.size();
        if (!size) return;
        cap = 1;
        while (growth_of(cap) < size) cap = cap * 2 + 1;
        ctrl.assign(cap + kGroup + 1, kEmpty);
        ctrl[cap] = kSentinel;
        slots.assign(cap * kSlot, 0);
        for (const auto& it : items) {
            const uint64_t h = phmap_mix64(it.first);
            const uint8_t h2 = (uint8_t)(h & 0x7F);
            uint64_t offset = (h >> 7) & cap, index = 0, pos = 0;
            for (;;) {                                           // find_first_non_full
                bool found = false;
                for (unsigned j = 0; j < kGroup; ++j) {
                    uint8_t c = ctrl[offset + j];                // group load may run into the cloned bytes
                    if (c == kEmpty) { pos = (offset + j) & cap; found = true; break; }
                }
                if (found) break;
                index += kGroup;
                offset = (offset + index) & cap;
            }
            ctrl[pos] = h2;                                      // set_ctrl incl. the mirrored byte
            ctrl[((pos - kGroup) & cap) + 1 + ((kGroup - 1) & cap)] = h2;
            memcpy(slots.data() + pos * kSlot, &it.first, 8);
            memcpy(slots.data() + pos * kSlot + 8, &it.second, sizeof(V));
        }
    }
    void dump(std::ofstream& f) const {
        uint64_t hdr[3] = { kDumpVersion, size, cap };
        f.write((const char*)hdr, sizeof hdr);
        if (!size) return;
        f.write((const char*)ctrl.data(), (std::streamsize)ctrl.size());
        f.write((const char*)slots.data(), (std::streamsize)slots.size());
        uint64_t growth_left = growth_of(cap) - size;
        f.write((const char*)&growth_left, 8);
    }
};

template <class V>
void write_dump(const std::string& path, std::vector<std::pair<uint64_t, V>>& items) {
    std::vector<std::vector<std::pair<uint64_t, V>>> sub(kSubmaps);
    std::sort(items.begin(), items.end(), [](const auto& a, const auto& b) { return a.first < b.first; });   // deterministic bytes
    for (const auto& it : items) sub[phmap_submap(phmap_mix64(it.first))].push_back(it);
    std::ofstream f(path, std::ios::binary | std::ios::trunc);
    if (!f) throw std::runtime_error("cannot write " + path);
    uint64_t n = kSubmaps;
    f.write((const char*)&n, 8);
    for (unsigned s = 0; s < kSubmaps; ++s) {
        SubmapImage<V> img;
        img.build(sub[s]);
        img.dump(f);
    }
    if (!f) throw std::runtime_error("write failed: " + path);
}

}  // namespace

uint64_t phmap_mix64(uint64_t a) {
    const unsigned __int128 p = (unsigned __int128)a * 0xde5fb9d2630458e9ull;
    return (uint64_t)(p >> 64) + (uint64_t)p;
}
unsigned phmap_submap(uint64_t h) { return (unsigned)((h >> 8) ^ (h >> 16) ^ (h >> 24)) & (kSubmaps - 1); }

DbIndex read_index(const std::string& db) {
    std::ifstream f(db + "/.index");
    if (!f) throw std::runtime_error("cannot open " + db + "/.index");
    DbIndex idx;
    std::string l1, l2;
    std::getline(f, l1);
    std::getline(f, l2);
    try { idx.k = std::stoi(l1); } catch (...) { throw std::runtime_error("bad .index in " + db); }
    try { idx.map_count = l2.empty() ? 128 : std::stoi(l2); } catch (...) { idx.map_count = 128; }
    return idx;
}

void write_index(const std::string& db, int k, int map_count) {
    std::ofstream f(db + "/.index", std::ios::trunc);
    if (!f) throw std::runtime_error("cannot write " + db + "/.index");
    f << k << "\n" << map_count << std::endl;                       // src/kreeq-output.cpp:91
}

void read_db_hc(const std::string& db, std::vector<kq_entry>& hc_out) {
    read_dump<Val32>(db + "/.map.hc.bin", [&](uint64_t key, const Val32& v) {
        kq_entry e{};
        e.key = key; e.cov = v.cov; e.hc = 1;
        for (int w = 0; w < 4; ++w) { e.fw[w] = v.fw[w]; e.bw[w] = v.bw[w]; }
        hc_out.push_back(e);
    });
}
void read_db_maps(const std::string& db, const DbIndex& idx, int map_lo, int map_hi, const std::vector<kq_entry>& hc, std::vector<kq_entry>& out) {
    size_t n_tomb = 0, n_hc = 0;
    for (int m = map_lo; m < map_hi; ++m) {
        read_dump<Val8>(db + "/.map." + std::to_string(m) + ".bin", [&](uint64_t key, const Val8& v) {
            if (v.cov == 255) { ++n_tomb; return; }                 // lives in the high-copy map (src/graph-builder.cpp:247)
            kq_entry e{};
            e.key = key; e.cov = v.cov; e.hc = 0;
            for (int w = 0; w < 4; ++w) { e.fw[w] = v.fw[w]; e.bw[w] = v.bw[w]; }
            out.push_back(e);
        });
    }
    for (auto& e : hc) {
        const int m = (int)(e.key % (uint64_t)idx.map_count);
        if (m >= map_lo && m < map_hi) { out.push_back(e); ++n_hc; }
    }
    if (n_tomb > n_hc) throw std::runtime_error("Error: int32 map missing 255 value from int8 map");   // src/kreeq.cpp:162
}

void read_db(const std::string& db, std::vector<kq_entry>& out, DbIndex* idx_out) {
    DbIndex idx = read_index(db);
    if (idx_out) *idx_out = idx;
    size_t n_tomb = 0, n_hc = 0;
    for (int m = 0; m < idx.map_count; ++m) {
        read_dump<Val8>(db + "/.map." + std::to_string(m) + ".bin", [&](uint64_t key, const Val8& v) {
            if (v.cov == 255) { ++n_tomb; return; }                 // lives in the high-copy map (src/graph-builder.cpp:247)
            kq_entry e{};
            e.key = key; e.cov = v.cov; e.hc = 0;
            for (int w = 0; w < 4; ++w) { e.fw[w] = v.fw[w]; e.bw[w] = v.bw[w]; }
            out.push_back(e);
        });
    }
    read_dump<Val32>(db + "/.map.hc.bin", [&](uint64_t key, const Val32& v) {
        kq_entry e{};
        e.key = key; e.cov = v.cov; e.hc = 1;
        for (int w = 0; w < 4; ++w) { e.fw[w] = v.fw[w]; e.bw[w] = v.bw[w]; }
        out.push_back(e);
        ++n_hc;
    });
    if (n_tomb > n_hc) throw std::runtime_error("Error: int32 map missing 255 value from int8 map");   // src/kreeq.cpp:162
}

// maps [map_lo, map_hi) from `entries` (which must only hold k-mers of those maps); high-copy
// k-mers are appended to `hc_out` for the single .map.hc.bin written by write_db_finish
void write_db_maps(const std::string& db, int map_count, int map_lo, int map_hi, const std::vector<kq_entry>& entries,
                   std::vector<kq_entry>& hc_out) {
    ::mkdir(db.c_str(), 0777);
    // bucket the entries by map in two parallel passes (count per chunk, then fill at the prefix offsets)
    const size_t n_maps = (size_t)(map_hi - map_lo), n = entries.size();
    unsigned hw0 = std::thread::hardware_concurrency();
    const size_t nt0 = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(64, hw0 ? hw0 : 1), n / 65536 + 1));
    std::vector<std::vector<uint64_t>> cnt(nt0, std::vector<uint64_t>(n_maps, 0));
    std::vector<uint64_t> hc_cnt(nt0, 0);
    std::atomic<bool> bad{false};
    auto chunk = [&](size_t t) { return std::make_pair(n * t / nt0, n * (t + 1) / nt0); };
    {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt0; ++t) th.emplace_back([&, t] {
            auto [a, b] = chunk(t);
            for (size_t i = a; i < b; ++i) {
                const int m = (int)(entries[i].key % (uint64_t)map_count);
                if (m < map_lo || m >= map_hi) { bad = true; return; }
                ++cnt[t][(size_t)(m - map_lo)];
                hc_cnt[t] += entries[i].hc ? 1 : 0;
            }
        });
        for (auto& x : th) x.join();
    }
    if (bad) throw std::runtime_error("entry outside the map range being written");
    std::vector<std::vector<std::pair<uint64_t, Val8>>> maps(n_maps);
    std::vector<std::vector<uint64_t>> off(nt0, std::vector<uint64_t>(n_maps, 0));
    for (size_t m = 0; m < n_maps; ++m) {
        uint64_t run = 0;
        for (size_t t = 0; t < nt0; ++t) { off[t][m] = run; run += cnt[t][m]; }
        maps[m].resize(run);
    }
    const size_t hc_base = hc_out.size();
    std::vector<uint64_t> hc_off(nt0, 0);
    { uint64_t run = 0; for (size_t t = 0; t < nt0; ++t) { hc_off[t] = run; run += hc_cnt[t]; } hc_out.resize(hc_base + run); }
    {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt0; ++t) th.emplace_back([&, t] {
            auto [a, b] = chunk(t);
            std::vector<uint64_t> pos = off[t];
            uint64_t hpos = hc_base + hc_off[t];
            for (size_t i = a; i < b; ++i) {
                const kq_entry& e = entries[i];
                const size_t m = (size_t)((int)(e.key % (uint64_t)map_count) - map_lo);
                Val8 v8{};
                if (e.hc) {
                    hc_out[hpos++] = e;
                    v8.cov = 255;                                   // tombstone: "look in the 32-bit map" (:193, :233)
                } else {
                    for (int w = 0; w < 4; ++w) { v8.fw[w] = (uint8_t)e.fw[w]; v8.bw[w] = (uint8_t)e.bw[w]; }
                    v8.cov = (uint8_t)e.cov;
                }
                maps[m][pos[m]++] = std::make_pair(e.key, v8);
            }
        });
        for (auto& x : th) x.join();
    }
    // one file per map: independent, so write them with a few threads
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nt = std::max(1u, std::min(64u, hw ? hw : 1u));
    std::atomic<int> next{map_lo};
    std::mutex err_m;
    std::string err;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&] {
            for (int m; (m = next.fetch_add(1)) < map_hi;) {
                try { write_dump<Val8>(db + "/.map." + std::to_string(m) + ".bin", maps[(size_t)(m - map_lo)]); }
                catch (const std::exception& e) { std::lock_guard<std::mutex> l(err_m); if (err.empty()) err = e.what(); }
            }
        });
    for (auto& t : pool) t.join();
    if (!err.empty()) throw std::runtime_error(err);
}
void write_db_finish(const std::string& db, int k, int map_count, const std::vector<kq_entry>& hc_entries) {
    std::vector<std::pair<uint64_t, Val32>> hc;
    for (const kq_entry& e : hc_entries) {
        Val32 v{};
        for (int w = 0; w < 4; ++w) { v.fw[w] = e.fw[w]; v.bw[w] = e.bw[w]; }
        v.cov = e.cov;
        hc.emplace_back(e.key, v);
    }
    write_dump<Val32>(db + "/.map.hc.bin", hc);
    write_index(db, k, map_count);
}

void write_db(const std::string& db, int k, int map_count, const std::vector<kq_entry>& entries) {
    std::vector<kq_entry> hc;
    write_db_maps(db, map_count, 0, map_count, entries, hc);
    write_db_finish(db, k, map_count, hc);
}

}  // namespace kqhost

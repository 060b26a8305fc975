// Records of an alignment file as the library holds them (sc_ingest.cpp reads the file, sc_depth.hip scans them).
#pragma once
#include <sys/mman.h>

#include <algorithm>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace sc_ingest {
struct Rec {                       // the SAM fields the path reads: 1 2 4 5 6 10 (+ the length of 11)
    const char* qname; const char* cigar; const char* seq;
    int qlen, clen, slen, quallen;
    int flag, pos, mapq;
    int ref_end;                   // last reference position covered, ops M D N = X (what `view` and `mpileup` overlap on)
};
}  // namespace sc_ingest

struct sc_aln {
    std::string error;
    // backing store of the record text: the mapped SAM file, or strings decoded from BAM
    void* map = nullptr; size_t map_len = 0;
    std::vector<std::unique_ptr<char[]>> arenas; size_t arena_used = 0, arena_cap = 0;
    std::unordered_map<std::string, std::vector<sc_ingest::Rec>> by_ref;      // records of a reference, in file order
    long n_records = 0;
    // what a scheduler prices a region with, for EVERY reference of the file -- also for those whose records were not
    // kept (sc_aln_open_filtered: a rank of a multi-GPU run keeps the records of its own shard only)
    struct RefStat { long n = 0, bases = 0; };
    std::unordered_map<std::string, RefStat> stats;
    bool keep_all = true;
    std::unordered_set<std::string> keep;                                     // references whose records are kept when !keep_all
    bool wants(const char* name, size_t n) const { return keep_all || keep.count(std::string(name, n)) != 0; }

    char* alloc(size_t n) {
        if (arena_used + n > arena_cap) {
            arena_cap = std::max<size_t>(n, 1 << 24);
            arenas.emplace_back(new char[arena_cap]);
            arena_used = 0;
        }
        char* p = arenas.back().get() + arena_used;
        arena_used += n;
        return p;
    }
    ~sc_aln() { if (map) munmap(map, map_len); }
};


// Host side of the partial-order-graph stage of the StrainCall path.
//
// Builds the graph the reference builds in
// /root/reference/StrainCall/PartialOrderGraph.cpp:67-265 (thread reads by CIGAR,
// canonise insertions/deletions, forward/backward merge, path collapse, level
// numbering) with index-based storage, tombstoned deletion and linear-time pool
// intersections, then flattens it into the level-major arrays the HIP clustering
// kernels consume (sc_kernels.hip).  The insertion MSA (row a7) is delegated to
// the device through `MsaFn`.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

namespace sc {

enum { ST_MAT = 0, ST_MIS = 1, ST_INS = 2, ST_DEL = 3 };

// One (read, label) entry of a node's read pool.  The label is an index into PoGraph::labtab: a pool of 10^5 entries
// holds a handful of distinct labels, and entries are sorted, merged and moved far more often than they are read.
struct PoolEnt {
    int rid;
    int cn;
    int lab;
};

struct GNode {
    int st = ST_MAT;
    std::string lab;
    std::vector<int> in, out, sib;
    std::vector<PoolEnt> pool;
    bool alive = true;
    int level = -1;
    int id = -1;          // final id (rank among alive nodes), set by finalize_ids()
    int stamp_a = 0, stamp_b = 0, stamp_c = 0;
};

struct AlignedRead {
    int pos;              // 0-based offset inside the window
    std::string cigar;
    std::string seq;
    int cn;               // copy number (exact duplicates collapsed upstream)
};

// Parsed CIGAR, '=' and 'X' already mapped to 'M' (PartialOrderGraph.cpp:13-59).
struct CigarOp { char op; int len; };
void parse_cigar(const std::string& c, std::vector<CigarOp>& out);

// Output of the device threading kernels (k_thread_*, row a5): every M-aligned
// read base bucketed by (reference position i, symbol c).  Index = i*8 + c.
struct ThreadTables {
    uint8_t lut[256];                 // byte -> symbol code, 0xFF = not present
    std::vector<char> sym;            // code -> byte
    std::vector<int> count, minrid;   // [glen*8] class size, first read of the class
    std::vector<int> tmin;            // [glen*64] first read going (i-1,cp) -> (i,c) inside an M run
    std::vector<int> smin, emin;      // [glen*8] first read starting / ending in the class
    std::vector<int> off;             // [glen*8+1] pool offsets
    std::vector<int> pool;            // read ids of every class, ascending
};
// Produces the tables for one region (the device implementation lives in sc_api.cpp).
using ThreadFn = std::function<void(const std::string& G, const std::vector<struct AlignedRead>& R,
                                    const std::vector<std::vector<CigarOp>>& cigars, ThreadTables& out)>;

// rows[t] = padded row of seqs[t]; returns the number of MSA columns.
using MsaFn = std::function<int(const std::vector<std::string>& seqs, std::vector<std::string>& rows)>;

class PoGraph {
public:
    PoGraph(const std::string& G, const std::vector<AlignedRead>& R, const MsaFn& msa, const ThreadFn& thread);

    std::vector<GNode> nodes;
    std::vector<std::string> labtab;       // distinct read labels; PoolEnt::lab indexes it
    int n_alive = 0;
    long msa_calls = 0;

    // `-G` dump, /root/reference/StrainCall/PartialOrderGraph.cpp:318-337
    std::string dump() const;
    int reads_cover(int u, int v) const;   // number_of_reads_cover_nodes, cpp:1218-1244
    int root() const { return 0; }

private:
    std::unordered_map<std::string, int> lab_id_;
    std::unordered_map<unsigned long long, int> lab_cat_;
    int intern(const std::string& lab);
    int concat(int a, int b);              // id of labtab[a] + labtab[b]
    const MsaFn& msa_;
    int stamp_ = 0;
    std::vector<int> order_;   // alive node indices in `nodes` order (valid after finalize_ids)

    int new_node(int st, const std::string& lab);
    void add_edge(int u, int w);
    void add_edge_gap(int u, const std::vector<int>& gap);
    void add_edge_gap_to(int u, int v, const std::vector<int>& gap);
    void del_edge(int u, int v);
    bool linking(int u, int v) const;
    void delete_node(int w, bool bridging);

    struct GapEx { int u, v; std::vector<int> gap; };
    void find_insert_from(int u, std::vector<GapEx>& out);
    void find_common_read_pool(int a, int b, std::vector<std::pair<int, int>>& c);
    void add_dash_chain(int a, int b, int l, const std::vector<std::pair<int, int>>& crp);
    void add_edge_level(int i, int l);
    void delete_edge_level(int i);
    void canonize_insert_at_level(int i);
    int node_level_exclude_delete(int w);
    void level_cache_build();
    std::vector<int> level_cache_;
    bool level_cache_on_ = false;
    void find_delete_from(int w, std::vector<GapEx>& out);
    void canonize_delete_at_level(int i);
    void merge_read_pool(int u, int v);
    void merge_node(int u, int v);
    void directional_merge(bool backward);
    void path_collapse();
    void node_level();
    void finalize_ids();
    void thread_reads(const std::string& G, const std::vector<AlignedRead>& R, const ThreadFn& thread);
};

// libstdc++ std::sort permutation (the reference depends on its tie order at
// PartialOrderGraph.cpp:466, NonparametricClustering.cpp:647,675, StrainCall.cpp:1027).
// less(a,b) compares element identities a,b (values stored in idx).
void std_sort_perm(std::vector<int>& idx, const std::function<bool(int, int)>& less);

// ---------------------------------------------------------------------------
// Level-major flattening for the device.
// Host threads the construction of ONE region's graph may use for its bulk copies (class pools, flattening): 1 while many
// regions are in flight (they already fill the rank's CPUs), more for a single deep region (BASELINE configs[3]: 88 M pool
// entries).  Thread-local: set by the worker that builds the graph.
void set_graph_threads(int n);
int graph_threads();

struct FlatGraph {
    // symbol table: code 0..5 = A C G T - = ; further codes in order of appearance
    std::vector<char> sym;            // code -> char
    int K = 6;
    int code_N = -1;                  // code of 'N' if present

    // nodes (final ids)
    int n_nodes = 0;
    std::vector<int> node_lab_off, node_lab_len;   // into labels[]
    std::vector<uint8_t> labels;                   // symbol codes; 0xFF for '^' and '$'
    std::vector<std::string> node_label_str;       // raw label text (for sequences / hashes)
    std::vector<uint8_t> node_is_end;              // label == "$"
    std::vector<int> out_ptr, out_node, out_support;  // CSR in out-order; support filled by the device

    // level walk of NonparametricClustering.cpp:284-334 (graph-determined)
    int n_levels = 0;
    std::vector<int> level_node_ptr, level_nodes;  // nodes popped at each level, in order
    std::vector<int> level_ent_ptr;                // entries (level_reads) per level
    std::vector<int> ent_rid, ent_cn, ent_lab_off, ent_lab_len;
    std::vector<uint8_t> ent_first;                // first occurrence of rid within its level
    std::vector<int> level_read_count;
    std::vector<uint8_t> level_has_end;            // "$" popped in this level
    std::vector<int> level_end_pos;                // index within level_nodes of the "$" pop (or -1)
    // pools in node order for the edge-support kernel
    std::vector<int> pool_ptr, pool_rid, pool_cn;
    bool pools_sorted = true;
    std::string unsupported;                       // non-empty: graph shape the device path does not model

    // Empties every array and keeps its memory: a region slot flattens one region after the other, and 30 MB of freshly
    // mapped pages per region cost as much in page faults as filling them.
    void reset() {
        sym.clear(); K = 6; code_N = -1; n_nodes = 0;
        node_lab_off.clear(); node_lab_len.clear(); labels.clear(); node_label_str.clear(); node_is_end.clear();
        out_ptr.clear(); out_node.clear(); out_support.clear();
        n_levels = 0; level_node_ptr.clear(); level_nodes.clear(); level_ent_ptr.clear();
        ent_rid.clear(); ent_cn.clear(); ent_lab_off.clear(); ent_lab_len.clear(); ent_first.clear();
        level_read_count.clear(); level_has_end.clear(); level_end_pos.clear();
        pool_ptr.clear(); pool_rid.clear(); pool_cn.clear();
        pools_sorted = true; unsupported.clear();
    }
};

void flatten(const PoGraph& g, int n_reads, FlatGraph& f);

}  // namespace sc

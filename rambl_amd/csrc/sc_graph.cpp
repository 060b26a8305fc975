// Host graph builder + flattener.  See sc_graph.hpp.  Each function cites the
// reference lines whose behaviour it reproduces
// (/root/reference/StrainCall/PartialOrderGraph.cpp unless another file is named).
#include "sc_graph.hpp"
#include <chrono>
#include <cstdio>

#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <atomic>
#include <thread>

namespace sc {

static thread_local int tl_graph_threads = 1;
void set_graph_threads(int n) { tl_graph_threads = n < 1 ? 1 : (n > 64 ? 64 : n); }
int graph_threads() { return tl_graph_threads; }
// fn(i) for i in [0, n), on up to `threads` threads taking indices from a shared counter (the items differ in size by
// orders of magnitude: a backbone class of 59 000 reads beside a sibling of three); exceptions of a helper end the process
// as they would on the calling thread
template <class F>
static void parallel_items(int n, int threads, const F& fn) {
    if (threads <= 1 || n < 2) { for (int i = 0; i < n; i++) fn(i); return; }
    std::atomic<int> next{0};
    auto body = [&] { for (int i = next.fetch_add(1, std::memory_order_relaxed); i < n; i = next.fetch_add(1, std::memory_order_relaxed)) fn(i); };
    std::vector<std::thread> ts;
    for (int t = 1; t < threads; t++) ts.emplace_back(body);
    body();
    for (auto& t : ts) t.join();
}

// ---------------------------------------------------------------------------
// std::sort as libstdc++ permutes it (GCC <bits/stl_algo.h>: introsort with
// median-of-three to first, unguarded partition, heap fallback, final insertion
// sort with threshold 16).
namespace {
struct Sorter {
    std::vector<int>& a;
    const std::function<bool(int, int)>& less;
    void linear_insert(int last) {
        int val = a[last];
        int next = last - 1;
        while (less(val, a[next])) { a[last] = a[next]; last = next; --next; }
        a[last] = val;
    }
    void insertion(int first, int last) {
        if (first == last) return;
        for (int i = first + 1; i != last; ++i) {
            if (less(a[i], a[first])) {
                int val = a[i];
                std::move_backward(a.begin() + first, a.begin() + i, a.begin() + i + 1);
                a[first] = val;
            } else linear_insert(i);
        }
    }
    void push_heap(int first, int hole, int top, int value) {
        int parent = (hole - 1) / 2;
        while (hole > top && less(a[first + parent], value)) {
            a[first + hole] = a[first + parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[first + hole] = value;
    }
    void adjust_heap(int first, int hole, int len, int value) {
        const int top = hole;
        int child = hole;
        while (child < (len - 1) / 2) {
            child = 2 * (child + 1);
            if (less(a[first + child], a[first + child - 1])) child--;
            a[first + hole] = a[first + child];
            hole = child;
        }
        if ((len & 1) == 0 && child == (len - 2) / 2) {
            child = 2 * (child + 1);
            a[first + hole] = a[first + child - 1];
            hole = child - 1;
        }
        push_heap(first, hole, top, value);
    }
    void heap_sort(int first, int last) {
        int len = last - first;
        if (len >= 2) {
            int parent = (len - 2) / 2;
            for (;;) {
                int value = a[first + parent];
                adjust_heap(first, parent, len, value);
                if (parent == 0) break;
                parent--;
            }
        }
        while (last - first > 1) {
            --last;
            int value = a[last];
            a[last] = a[first];
            adjust_heap(first, 0, last - first, value);
        }
    }
    void median_to_first(int result, int x, int y, int z) {
        if (less(a[x], a[y])) {
            if (less(a[y], a[z])) std::swap(a[result], a[y]);
            else if (less(a[x], a[z])) std::swap(a[result], a[z]);
            else std::swap(a[result], a[x]);
        } else if (less(a[x], a[z])) std::swap(a[result], a[x]);
        else if (less(a[y], a[z])) std::swap(a[result], a[z]);
        else std::swap(a[result], a[y]);
    }
    int partition(int first, int last, int pivot) {
        for (;;) {
            while (less(a[first], a[pivot])) ++first;
            --last;
            while (less(a[pivot], a[last])) --last;
            if (!(first < last)) return first;
            std::swap(a[first], a[last]);
            ++first;
        }
    }
    void introsort(int first, int last, int depth) {
        while (last - first > 16) {
            if (depth == 0) { heap_sort(first, last); return; }
            --depth;
            int mid = first + (last - first) / 2;
            median_to_first(first, first + 1, mid, last - 1);
            int cut = partition(first + 1, last, first);
            introsort(cut, last, depth);
            last = cut;
        }
    }
};
}  // namespace

void std_sort_perm(std::vector<int>& idx, const std::function<bool(int, int)>& less) {
    int n = (int)idx.size();
    if (n == 0) return;
    Sorter s{idx, less};
    int lg = 0;
    for (unsigned t = (unsigned)n; t > 1; t >>= 1) lg++;
    s.introsort(0, n, lg * 2);
    if (n > 16) {
        s.insertion(0, 16);
        for (int i = 16; i != n; ++i) s.linear_insert(i);
    } else s.insertion(0, n);
}

// ---------------------------------------------------------------------------
void parse_cigar(const std::string& c, std::vector<CigarOp>& out) {   // cpp:13-59
    int v = 0;
    for (char ch : c) {
        switch (ch) {
            case 'M': case 'I': case 'D': case 'N': case 'S': case 'H': case 'P':
                out.push_back({ch, v}); v = 0; break;
            case '=': case 'X':
                out.push_back({'M', v}); v = 0; break;
            default:
                if (ch >= '0' && ch <= '9') v = v * 10 + (ch - '0');
        }
    }
}

int PoGraph::new_node(int st, const std::string& lab) {
    nodes.emplace_back();
    GNode& w = nodes.back();
    w.st = st; w.lab = lab;
    ++n_alive;
    return (int)nodes.size() - 1;
}
void PoGraph::add_edge(int u, int w) { nodes[u].out.push_back(w); nodes[w].in.push_back(u); }
void PoGraph::add_edge_gap(int u, const std::vector<int>& gap) {
    int a = u;
    for (int w : gap) { add_edge(a, w); a = w; }
}
void PoGraph::add_edge_gap_to(int u, int v, const std::vector<int>& gap) {
    add_edge_gap(u, gap);
    add_edge(gap.back(), v);
}
static inline void erase_first(std::vector<int>& v, int x) {
    auto it = std::find(v.begin(), v.end(), x);
    if (it != v.end()) v.erase(it);
}
void PoGraph::del_edge(int u, int v) { erase_first(nodes[u].out, v); erase_first(nodes[v].in, u); }
bool PoGraph::linking(int u, int v) const {
    const auto& o = nodes[u].out;
    return std::find(o.begin(), o.end(), v) != o.end();
}
// cpp:406-444; the O(N) erase + renumbering is replaced by a tombstone, ids are
// assigned once at the end (finalize_ids) -- the relative order is the same.
void PoGraph::delete_node(int w, bool bridging) {
    // note: in/out of w are not modified while we iterate them
    for (int p : nodes[w].in) {
        for (int c : nodes[w].out) {
            if (bridging && !linking(p, c)) add_edge(p, c);
            erase_first(nodes[c].in, w);
        }
        erase_first(nodes[p].out, w);
    }
    nodes[w].alive = false;
    --n_alive;
}

int PoGraph::reads_cover(int u, int v) const {
    const GNode &a = nodes[u], &b = nodes[v];
    int n = 0;
    if (u == 0) {
        for (const auto& e : b.pool) n += e.cn;
    } else if (b.lab == "$") {
        for (const auto& e : a.pool) n += e.cn;
    } else {
        // sum over pairs with equal rid of v's copy number; pools are rid-sorted in
        // practice, fall back to the quadratic form otherwise
        bool sorted = true;
        for (size_t i = 1; i < a.pool.size() && sorted; i++) sorted = a.pool[i - 1].rid <= a.pool[i].rid;
        for (size_t i = 1; i < b.pool.size() && sorted; i++) sorted = b.pool[i - 1].rid <= b.pool[i].rid;
        if (sorted) {
            size_t i = 0, j = 0;
            while (i < a.pool.size() && j < b.pool.size()) {
                if (a.pool[i].rid < b.pool[j].rid) i++;
                else if (a.pool[i].rid > b.pool[j].rid) j++;
                else {
                    int rid = a.pool[i].rid, mu = 0, sv = 0;
                    while (i < a.pool.size() && a.pool[i].rid == rid) { mu++; i++; }
                    while (j < b.pool.size() && b.pool[j].rid == rid) { sv += b.pool[j].cn; j++; }
                    n += mu * sv;
                }
            }
        } else {
            for (const auto& x : a.pool) for (const auto& y : b.pool) if (x.rid == y.rid) n += y.cn;
        }
    }
    return n;
}

// cpp:355-393
void PoGraph::find_insert_from(int u, std::vector<GapEx>& out) {
    std::vector<int> g, st;
    st.push_back(u);
    while (!st.empty()) {
        int v = st.back(); st.pop_back();
        if (v == u) {
            for (int o : nodes[v].out) if (nodes[o].st == ST_INS) st.push_back(o);
        } else if (nodes[v].st == ST_MAT || nodes[v].st == ST_MIS) {
            out.push_back({u, v, g});
            g.clear();
        } else {
            g.push_back(v);
            for (int o : nodes[v].out) st.push_back(o);
        }
    }
}

// cpp:780-829.  Sets of (rid, copies).
void PoGraph::find_common_read_pool(int a, int b, std::vector<std::pair<int, int>>& c) {
    auto build = [&](int x, const std::vector<int>& adj) {
        std::vector<std::pair<int, int>> s;
        s.reserve(nodes[x].pool.size());
        for (const auto& e : nodes[x].pool) s.emplace_back(e.rid, e.cn);
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        std::vector<std::pair<int, int>> rm;
        for (int o : adj)
            if (nodes[o].st == ST_INS || nodes[o].st == ST_DEL)
                for (const auto& e : nodes[o].pool) rm.emplace_back(e.rid, e.cn);
        if (!rm.empty()) {
            std::sort(rm.begin(), rm.end());
            std::vector<std::pair<int, int>> kept;
            kept.reserve(s.size());
            std::set_difference(s.begin(), s.end(), rm.begin(), rm.end(), std::back_inserter(kept));
            s.swap(kept);
        }
        return s;
    };
    auto ar = build(a, nodes[a].out);
    auto br = build(b, nodes[b].in);
    c.clear();
    std::set_intersection(ar.begin(), ar.end(), br.begin(), br.end(), std::back_inserter(c));
}

void PoGraph::add_dash_chain(int a, int b, int l, const std::vector<std::pair<int, int>>& crp) {
    std::vector<int> gap;
    for (int t = 0; t < l; ++t) {
        int w = new_node(ST_INS, "-");
        auto& pool = nodes[w].pool;
        pool.reserve(crp.size());
        const int dash = intern("-");
        for (const auto& rc : crp) pool.push_back({rc.first, rc.second, dash});
        gap.push_back(w);
    }
    add_edge_gap_to(a, b, gap);
}
// cpp:831-923
void PoGraph::add_edge_level(int i, int l) {
    int u = i, v = i + 1;
    std::vector<std::pair<int, int>> crp;
    if (linking(u, v)) { find_common_read_pool(u, v, crp); add_dash_chain(u, v, l, crp); }
    { auto sv = nodes[v].sib;
      for (int s : sv) if (linking(u, s)) { find_common_read_pool(u, s, crp); add_dash_chain(u, s, l, crp); } }
    { auto su = nodes[u].sib;
      for (int s : su) if (linking(s, v)) { find_common_read_pool(s, v, crp); add_dash_chain(s, v, l, crp); } }
    { auto su = nodes[u].sib; auto sv = nodes[v].sib;
      for (int a : su) for (int b : sv)
          if (linking(a, b)) { find_common_read_pool(a, b, crp); add_dash_chain(a, b, l, crp); } }
}
// cpp:925-961
void PoGraph::delete_edge_level(int i) {
    int u = i, v = i + 1;
    if (linking(u, v)) del_edge(u, v);
    for (int s : nodes[v].sib) if (linking(u, s)) del_edge(u, s);
    for (int s : nodes[u].sib) if (linking(s, v)) del_edge(s, v);
    for (int a : nodes[u].sib) for (int b : nodes[v].sib) if (linking(a, b)) del_edge(a, b);
}

// cpp:446-550
void PoGraph::canonize_insert_at_level(int i) {
    std::vector<GapEx> found;
    find_insert_from(i, found);
    { auto sibs = nodes[i].sib; for (int s : sibs) find_insert_from(s, found); }
    if (found.empty()) return;
    int n = (int)found.size();
    std::vector<int> perm(n);
    for (int t = 0; t < n; t++) perm[t] = t;
    std_sort_perm(perm, [&](int a, int b) { return found[a].gap.size() > found[b].gap.size(); });
    std::vector<GapEx> inserts(n);
    for (int t = 0; t < n; t++) inserts[t] = std::move(found[perm[t]]);

    std::vector<std::string> seqs(n);
    int l = 0, k = 1000000000;
    for (int t = 0; t < n; t++) {
        for (int w : inserts[t].gap) seqs[t] += nodes[w].lab;
        l = std::max(l, (int)seqs[t].size());
        k = std::min(k, (int)seqs[t].size());
    }
    if (n == 1 || l - k == 0) {
        add_edge_level(i, l);
        delete_edge_level(i);
        return;
    }
    std::vector<std::string> rows;
    int ncol = msa_(seqs, rows);
    ++msa_calls;
    for (int t = 0; t < n; ++t) {
        if (rows[t] != seqs[t]) {
            int rid = 0, rcn = 0;
            for (int w : inserts[t].gap) {
                rid = nodes[w].pool[0].rid;
                rcn = nodes[w].pool[0].cn;
                delete_node(w, true);
            }
            std::vector<int> ng;
            for (char ch : rows[t]) {
                std::string lab(1, ch);
                int w = new_node(ST_INS, lab);
                nodes[w].pool.push_back({rid, rcn, intern(lab)});
                ng.push_back(w);
            }
            add_edge_gap_to(inserts[t].u, inserts[t].v, ng);
        }
    }
    add_edge_level(i, ncol);
    delete_edge_level(i);
}

// cpp:571-622
// The level at which the walk below first takes w off its list is w's distance from the root over the edges the walk
// follows (out-edges to nodes that are no deletions, and the siblings of those nodes).  While canonize_delete runs,
// these edges do not change -- it only adds deletion nodes and edges into them -- so one breadth-first pass serves every
// query of the phase (723 walks over 400 000 nodes on the unthinned configs[3] region otherwise).
void PoGraph::level_cache_build() {
    level_cache_.assign(nodes.size(), -1);
    std::vector<int> cur{0}, nxt;
    level_cache_[0] = 0;
    int level = 0;
    while (!cur.empty()) {
        level += 1;
        nxt.clear();
        for (int u : cur)
            for (int o : nodes[u].out) {
                if (nodes[o].st == ST_DEL) continue;
                if (level_cache_[o] < 0) { level_cache_[o] = level; nxt.push_back(o); }
                for (int s : nodes[o].sib) if (level_cache_[s] < 0) { level_cache_[s] = level; nxt.push_back(s); }
            }
        cur.swap(nxt);
    }
}
int PoGraph::node_level_exclude_delete(int w) {
    if (level_cache_on_ && w < (int)level_cache_.size() && level_cache_[w] >= 0) return level_cache_[w];
    std::vector<int> level_node, sub;
    int level = 0;
    int stamp = ++stamp_;
    level_node.push_back(0);
    while (!level_node.empty()) {
        int u = level_node.back(); level_node.pop_back();
        if (u == w) break;
        for (int o : nodes[u].out) {
            if (nodes[o].st == ST_DEL) continue;
            sub.push_back(o);
            for (int s : nodes[o].sib) sub.push_back(s);
        }
        if (level_node.empty()) {
            while (!sub.empty()) {
                int v = sub.back(); sub.pop_back();
                if (nodes[v].stamp_a == stamp) continue;
                level_node.push_back(v);
                nodes[v].stamp_a = stamp;
            }
            level += 1;
            stamp = ++stamp_;
        }
    }
    return level;
}
// cpp:624-672
void PoGraph::find_delete_from(int w, std::vector<GapEx>& out) {
    std::vector<int> gap, st, cnt;
    st.push_back(w); cnt.push_back(0);
    while (!st.empty()) {
        int u = st.back(); st.pop_back();
        int c = cnt.back(); cnt.pop_back();
        if (u == w) {
            for (int o : nodes[u].out) if (nodes[o].st == ST_DEL) { st.push_back(o); cnt.push_back(0); }
        } else if (nodes[u].st == ST_DEL) {
            if (c == 0) {
                st.push_back(u); cnt.push_back(1);
                gap.push_back(u);
                for (int o : nodes[u].out) { st.push_back(o); cnt.push_back(0); }
            } else gap.pop_back();
        } else {
            out.push_back({w, u, gap});
        }
    }
}
// cpp:684-740
void PoGraph::canonize_delete_at_level(int i) {
    std::vector<GapEx> deletes;
    find_delete_from(i, deletes);
    { auto sibs = nodes[i].sib; for (int s : sibs) find_delete_from(s, deletes); }
    if (deletes.empty()) return;
    std::vector<std::pair<int, int>> nl;   // node -> level cache, local to this call
    auto level_of = [&](int x) {
        for (auto& p : nl) if (p.first == x) return p.second;
        int lv = node_level_exclude_delete(x);
        nl.emplace_back(x, lv);
        return lv;
    };
    for (auto& d : deletes) {
        int ul = level_of(d.u), vl = level_of(d.v);
        int dl = vl - ul - 1, dd = (int)d.gap.size();
        if (dl - dd > 0) {
            int v0 = d.gap[0];
            int rid = nodes[v0].pool[0].rid, rcn = nodes[v0].pool[0].cn;
            std::vector<int> ng;
            for (int t = dl - dd; t > 0; --t) {
                int w = new_node(ST_DEL, "=");
                nodes[w].pool.push_back({rid, rcn, intern("=")});
                ng.push_back(w);
            }
            add_edge_gap_to(d.u, v0, ng);
            del_edge(d.u, v0);
        }
    }
}

// cpp:963-1005
int PoGraph::intern(const std::string& lab) {
    auto it = lab_id_.find(lab);
    if (it != lab_id_.end()) return it->second;
    const int id = (int)labtab.size();
    labtab.push_back(lab);
    lab_id_.emplace(lab, id);
    return id;
}
int PoGraph::concat(int a, int b) {
    const unsigned long long key = ((unsigned long long)(unsigned)a << 32) | (unsigned)b;
    auto it = lab_cat_.find(key);
    if (it != lab_cat_.end()) return it->second;
    const int id = intern(labtab[(size_t)a] + labtab[(size_t)b]);
    lab_cat_.emplace(key, id);
    return id;
}
void PoGraph::merge_read_pool(int u, int v) {
    auto cmp = [this](const PoolEnt& a, const PoolEnt& b) {
        if (a.rid != b.rid) return a.rid < b.rid;
        if (a.lab != b.lab) return labtab[(size_t)a.lab] < labtab[(size_t)b.lab];
        return a.cn < b.cn;
    };
    auto& pu = nodes[u].pool;
    auto& pv = nodes[v].pool;
    if (!std::is_sorted(pu.begin(), pu.end(), cmp)) std::sort(pu.begin(), pu.end(), cmp);
    if (!std::is_sorted(pv.begin(), pv.end(), cmp)) std::sort(pv.begin(), pv.end(), cmp);
    std::vector<PoolEnt> res;
    res.reserve(pu.size() + pv.size());
    size_t i = 0, j = 0;
    while (i < pu.size() && j < pv.size()) {
        if (pu[i].rid == pv[j].rid) {
            res.push_back({pu[i].rid, pu[i].cn, concat(pu[i].lab, pv[j].lab)});
            i++; j++;
        } else if (pu[i].rid < pv[j].rid) res.push_back(pu[i++]);
        else res.push_back(pv[j++]);
    }
    res.insert(res.end(), pu.begin() + (long)i, pu.end());
    res.insert(res.end(), pv.begin() + (long)j, pv.end());
    pu.swap(res);
}
// cpp:1007-1038
void PoGraph::merge_node(int u, int v) {
    { auto vin = nodes[v].in;
      for (int p : vin) if (!linking(p, u) && p != u) add_edge(p, u); }
    { auto vout = nodes[v].out;
      for (int c : vout) if (!linking(u, c) && u != c) add_edge(u, c); }
    if (linking(u, v) && nodes[u].st == ST_MAT && nodes[v].st == ST_MAT) nodes[u].lab += nodes[v].lab;
    merge_read_pool(u, v);
    delete_node(v, false);
    std::vector<PoolEnt>().swap(nodes[v].pool);
}
// cpp:1040-1094 / 1097-1159
void PoGraph::directional_merge(bool backward) {
    std::vector<int> q;
    size_t qh = 0;
    int st_merged = ++stamp_, st_visit = ++stamp_;
    if (!backward) q.push_back(0);
    else for (int i = 0; i < (int)nodes.size(); i++) if (nodes[i].alive && nodes[i].lab == "$") q.push_back(i);
    std::vector<std::pair<int, int>> to_merge;
    while (qh < q.size()) {
        int w = q[qh++];
        if (nodes[w].stamp_a == st_merged) continue;
        {
            const auto& adj = backward ? nodes[w].in : nodes[w].out;
            for (size_t a = 0; a < adj.size(); a++) {
                int u = adj[a];
                for (size_t b = a + 1; b < adj.size(); b++) {
                    int v = adj[b];
                    if (u == v) continue;
                    if (nodes[u].st == nodes[v].st && nodes[u].lab == nodes[v].lab)
                        if (nodes[u].stamp_a != st_merged && nodes[v].stamp_a != st_merged) {
                            to_merge.emplace_back(u, v);
                            nodes[v].stamp_a = st_merged;
                        }
                }
            }
        }
        for (auto& m : to_merge) merge_node(m.first, m.second);
        to_merge.clear();
        const auto& adj = backward ? nodes[w].in : nodes[w].out;
        for (int c : adj)
            if (nodes[c].stamp_b != st_visit) { q.push_back(c); nodes[c].stamp_b = st_visit; }
    }
}
// cpp:1171-1216
void PoGraph::path_collapse() {
    int level_size = 0;
    std::vector<int> lq, sq;
    size_t lh = 0;
    int stamp = ++stamp_;
    lq.push_back(0);
    while (lh < lq.size()) {
        int u = lq[lh++];
        if (level_size == 1 && nodes[u].out.size() == 1) {
            int v = nodes[u].out[0];
            while (nodes[v].out.size() == 1) {
                merge_node(u, v);
                v = nodes[u].out[0];
            }
        }
        for (int v : nodes[u].out)
            if (nodes[v].stamp_c != stamp) { sq.push_back(v); nodes[v].stamp_c = stamp; }
        if (lh == lq.size()) {
            lq.swap(sq); sq.clear(); lh = 0;
            level_size = (int)lq.size();
            stamp = ++stamp_;
        }
    }
}
// cpp:769-776 + LevelOrderIterator.cpp:3-56
void PoGraph::node_level() {
    std::vector<int> level_node, sub;
    int n = 0, level = 0, stamp = ++stamp_;
    int cur = 0, cur_level = 0;
    for (int o : nodes[0].out) level_node.push_back(o);
    nodes[0].stamp_a = stamp;
    while (n != n_alive) {
        nodes[cur].level = cur_level;
        if (sub.empty()) { level += 1; stamp = ++stamp_; }
        if (!level_node.empty()) {
            int w = level_node.back(); level_node.pop_back();
            cur = w; cur_level = level;
            n += 1;
            for (int o : nodes[w].out) sub.push_back(o);
            if (level_node.empty()) {
                while (!sub.empty()) {
                    int x = sub.back(); sub.pop_back();
                    if (nodes[x].stamp_a == stamp) continue;
                    level_node.push_back(x);
                    nodes[x].stamp_a = stamp;
                }
            }
        } else n += 1;
    }
}
void PoGraph::finalize_ids() {
    order_.clear();
    for (int i = 0; i < (int)nodes.size(); i++)
        if (nodes[i].alive) { nodes[i].id = (int)order_.size(); order_.push_back(i); }
}

// cpp:94-255, the read loop.  The per-base work (class of every M-aligned base,
// pools, first read of every class / transition) comes from the device tables;
// this routine replays node creations and add_edge calls in the order the
// reference makes them -- by read, then by position inside the read -- and walks
// only the reads that contain insertions or deletions (their private chains).
void PoGraph::thread_reads(const std::string& G, const std::vector<AlignedRead>& R, const ThreadFn& thread) {
    const int glen = (int)G.size();
    const int n = (int)R.size();
#ifdef SC_GRAPH_TIMING
    auto tnow_ = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double ttp_ = tnow_();
#define SC_TPHASE(name) do { const double t_ = tnow_(); fprintf(stderr, "    thread_reads %-14s %.2f ms\n", name, t_ - ttp_); ttp_ = t_; } while (0)
#else
#define SC_TPHASE(name) do {} while (0)
#endif
    std::vector<std::vector<CigarOp>> cig(n);
    std::vector<char> complex_read(n, 0);
    for (int r = 0; r < n; r++) {
        parse_cigar(R[r].cigar, cig[r]);
        for (const CigarOp& c : cig[r]) {
            if (c.op == 'I' || c.op == 'D') complex_read[r] = 1;
            else if (c.op != 'M') throw std::runtime_error("CIGAR operation other than M/I/D/=/X in a cropped read");
        }
    }
    SC_TPHASE("cigars");
    ThreadTables T;
    thread(G, R, cig, T);
    SC_TPHASE("thread stage");
    const int INF = 0x7fffffff;
    if ((int)T.sym.size() > 8) throw std::runtime_error("more than 8 distinct symbols in the reads");

    enum { EV_SIB = 0, EV_CHAIN = 1, EV_EDGE = 2 };
    enum { REF_CLASS = 0, REF_NODE = 1, REF_CHAIN = 2 };
    struct NodeRef { int type, a, b; };                 // CLASS: (i, c); NODE: node index; CHAIN: (chain id, position)
    struct Event {
        int rid, op, t, kind;
        int i, c;                                       // EV_SIB
        int chain;                                      // EV_CHAIN
        NodeRef u, v; bool force, check_ne;             // EV_EDGE
    };
    struct Chain { int st; std::string labels; int rid; std::vector<int> nodes; };
    std::vector<Event> ev;
    std::vector<Chain> chains;

    // (op index, offset) of reference position i inside read r, for ordering
    auto locate = [&](int r, int i, int& op, int& t) {
        if (!complex_read[r]) { op = 0; t = i - R[r].pos; return; }
        int ri = R[r].pos;
        for (int k = 0; k < (int)cig[r].size(); k++) {
            const CigarOp& c = cig[r][k];
            if (c.op == 'M') { if (i < ri + c.len) { op = k; t = i - ri; return; } ri += c.len; }
            else if (c.op == 'D') ri += c.len;
        }
        op = (int)cig[r].size(); t = 0;
    };
    auto class_ref = [&](int i, int c) { return NodeRef{REF_CLASS, i, c}; };
    auto ref_code = [&](int i) { return (int)T.lut[(unsigned char)G[i]]; };

    for (int i = 0; i < glen; i++) {
        const int gc = ref_code(i);
        for (int c = 0; c < 8; c++) {
            const int cls = i * 8 + c;
            if (T.count[cls] == 0) continue;
            int op, t;
            if (c != gc) {                               // first read of a "mis" class creates the sibling (:151-163)
                locate(T.minrid[cls], i, op, t);
                Event e{}; e.rid = T.minrid[cls]; e.op = op; e.t = t; e.kind = EV_SIB; e.i = i; e.c = c;
                ev.push_back(e);
            }
            if (T.smin[cls] != INF && c != gc) {         // read starts here: edge from the backbone node before it
                const int r = T.smin[cls];
                Event e{}; e.rid = r; e.op = 0; e.t = 0; e.kind = EV_EDGE;
                e.u = NodeRef{REF_NODE, R[r].pos, 0}; e.v = class_ref(i, c); e.force = false; e.check_ne = false;
                ev.push_back(e);
            }
            if (T.emin[cls] != INF && c != gc) {         // read ends here: edge to the next backbone node (:252-253)
                const int r = T.emin[cls];
                Event e{}; e.rid = r; e.op = 1 << 29; e.t = 0; e.kind = EV_EDGE;
                e.u = class_ref(i, c); e.v = NodeRef{REF_NODE, i + 2, 0}; e.force = false; e.check_ne = true;
                ev.push_back(e);
            }
            if (i > 0) {
                const int gp = ref_code(i - 1);
                for (int cp = 0; cp < 8; cp++) {
                    const int r = T.tmin[i * 64 + cp * 8 + c];
                    if (r == INF || (cp == gp && c == gc)) continue;    // backbone -> backbone is linked from the start
                    locate(r, i, op, t);
                    Event e{}; e.rid = r; e.op = op; e.t = t; e.kind = EV_EDGE;
                    e.u = class_ref(i - 1, cp); e.v = class_ref(i, c); e.force = false; e.check_ne = false;
                    ev.push_back(e);
                }
            }
        }
    }
    // reads with insertions / deletions: their private chains and the edges around them
    const size_t n_class_events = ev.size();
    for (int r = 0; r < n; r++) {
        if (!complex_read[r]) continue;
        const AlignedRead& rd = R[r];
        const int rlen = (int)rd.seq.size();
        int i = rd.pos, j = 0;
        NodeRef u{REF_NODE, rd.pos, 0};                  // nodes[pos]: the backbone node before the first base
        bool u_is_end = false;
        const int nops = (int)cig[r].size();
        for (int k = 0; k < nops; k++) {
            const CigarOp& c = cig[r][k];
            if (c.op == 'M') {
                if (i + c.len > glen || j + c.len > rlen) throw std::runtime_error("read runs past the window");
                const int cfirst = T.lut[(unsigned char)rd.seq[j]];
                if (k > 0 && cig[r][k - 1].op != 'M') {  // first base after a chain: edge from its tail (:141-174)
                    Event e{}; e.rid = r; e.op = k; e.t = 0; e.kind = EV_EDGE;
                    e.u = u; e.v = class_ref(i, cfirst); e.force = false; e.check_ne = false;
                    ev.push_back(e);
                }
                i += c.len; j += c.len;
                u = class_ref(i - 1, T.lut[(unsigned char)rd.seq[j - 1]]);
                u_is_end = false;
            } else {
                Chain ch; ch.rid = r;
                if (c.op == 'I') {
                    if (j + c.len > rlen) throw std::runtime_error("insertion runs past the read");
                    ch.st = ST_INS; ch.labels = rd.seq.substr(j, c.len); j += c.len;
                } else {
                    ch.st = ST_DEL; ch.labels = std::string((size_t)c.len, '='); i += c.len;
                    if (i > glen) throw std::runtime_error("deletion runs past the window");
                }
                const int cid = (int)chains.size();
                chains.push_back(ch);
                Event e{}; e.rid = r; e.op = k; e.t = 0; e.kind = EV_CHAIN; e.chain = cid;
                ev.push_back(e);
                Event l{}; l.rid = r; l.op = k; l.t = 0; l.kind = EV_EDGE;          // add_edge(u, gap) (:190, :217)
                l.u = u; l.v = NodeRef{REF_CHAIN, cid, 0}; l.force = true; l.check_ne = false;
                ev.push_back(l);
                if (c.op == 'D' && i == glen) {                                    // deletion up to "$" (:209-215)
                    Event m{}; m.rid = r; m.op = k; m.t = 1; m.kind = EV_EDGE;
                    m.u = NodeRef{REF_CHAIN, cid, c.len - 1}; m.v = NodeRef{REF_NODE, glen + 1, 0}; m.force = true; m.check_ne = false;
                    ev.push_back(m);
                    u = NodeRef{REF_NODE, glen + 1, 0};
                    u_is_end = true;
                } else {
                    u = NodeRef{REF_CHAIN, cid, c.len - 1};
                    u_is_end = false;
                }
            }
        }
        if (nops > 0 && cig[r][nops - 1].op != 'M') {    // final `if (!linking(u,v) and u!=v) add_edge(u,v)` (:252-253)
            if (i + 1 >= (int)nodes.size()) throw std::runtime_error("read runs past the window");
            Event e{}; e.rid = r; e.op = 1 << 29; e.t = 0; e.kind = EV_EDGE;
            e.u = u; e.v = NodeRef{REF_NODE, i + 1, 0}; e.force = false; e.check_ne = true;
            ev.push_back(e);
            (void)u_is_end;
        }
    }
    SC_TPHASE("events");
    auto ev_less = [](const Event& a, const Event& b) {
        if (a.rid != b.rid) return a.rid < b.rid;
        if (a.op != b.op) return a.op < b.op;
        if (a.t != b.t) return a.t < b.t;
        return a.kind < b.kind;
    };
    // the events of the reads were made in read and operation order: sort the few class events, merge (the same
    // order as one stable sort of everything)
    if (std::is_sorted(ev.begin() + (long)n_class_events, ev.end(), ev_less)) {
        std::stable_sort(ev.begin(), ev.begin() + (long)n_class_events, ev_less);
        std::inplace_merge(ev.begin(), ev.begin() + (long)n_class_events, ev.end(), ev_less);
    } else {
        std::stable_sort(ev.begin(), ev.end(), ev_less);
    }
    SC_TPHASE("event sort");
    // replay
    std::vector<int> class_node((size_t)glen * 8, -1);
    for (int i = 0; i < glen; i++) if (ref_code(i) < 8) class_node[(size_t)i * 8 + ref_code(i)] = i + 1;   // a gene base no read carries has no class
    auto resolve = [&](const NodeRef& x) -> int {
        if (x.type == REF_NODE) return x.a;
        if (x.type == REF_CLASS) return class_node[(size_t)x.a * 8 + x.b];
        return chains[x.a].nodes[x.b];
    };
    for (const Event& e : ev) {
        if (e.kind == EV_SIB) {
            const int w = new_node(ST_MIS, std::string(1, T.sym[e.c]));
            nodes[e.i + 1].sib.push_back(w);
            class_node[(size_t)e.i * 8 + e.c] = w;
        } else if (e.kind == EV_CHAIN) {
            Chain& ch = chains[e.chain];
            int prev = -1;
            for (char lab : ch.labels) {
                const int w = new_node(ch.st, std::string(1, lab));
                nodes[w].pool.push_back({ch.rid, R[ch.rid].cn, intern(std::string(1, lab))});
                if (prev >= 0) add_edge(prev, w);
                ch.nodes.push_back(w);
                prev = w;
            }
        } else {
            const int a = resolve(e.u), b = resolve(e.v);
            if (a < 0 || b < 0) throw std::runtime_error("edge event before its node");
            if (e.force || (!linking(a, b) && !(e.check_ne && a == b))) add_edge(a, b);
        }
    }
    SC_TPHASE("replay");
    // pools of the backbone / sibling nodes, in read order (a million entries: copy numbers from a compact array, written by index)
    std::vector<int> copies((size_t)n);
    for (int r = 0; r < n; r++) copies[(size_t)r] = R[r].cn;
    int sym_lab[8];
    for (int c = 0; c < 8; c++) sym_lab[c] = c < (int)T.sym.size() ? intern(std::string(1, T.sym[c])) : -1;
    for (int cls = 0; cls < glen * 8; cls++)
        if (T.count[cls] != 0 && class_node[cls] < 0) throw std::runtime_error("class without a node");
    // a class has a node of its own: the pools fill independently of each other (on a deep region this is 88 M entries)
    const int fill_threads = (long)T.pool.size() > (1L << 22) ? graph_threads() : 1;
    parallel_items(glen * 8, fill_threads, [&](int cls) {
        if (T.count[cls] == 0) return;
        auto& pool = nodes[class_node[cls]].pool;
        const int lab = sym_lab[cls & 7];
        const size_t at = pool.size(), cnt = (size_t)(T.off[cls + 1] - T.off[cls]);
        pool.resize(at + cnt);
        PoolEnt* dst = pool.data() + at;
        const int* src = T.pool.data() + T.off[cls];
        for (size_t x = 0; x < cnt; x++) dst[x] = PoolEnt{src[x], copies[(size_t)src[x]], lab};
    });
    SC_TPHASE("pools");
#undef SC_TPHASE
}

// cpp:67-265
PoGraph::PoGraph(const std::string& G, const std::vector<AlignedRead>& R, const MsaFn& msa, const ThreadFn& thread) : msa_(msa) {
    const int glen = (int)G.size();
    nodes.reserve((size_t)glen + 2 + R.size() / 2);
    int u = new_node(ST_MAT, "^");
    for (int i = 0; i < glen; i++) {
        int w = new_node(ST_MAT, std::string(1, G[i]));
        add_edge(u, w);
        u = w;
    }
    int E = new_node(ST_MAT, "$");
    add_edge(u, E);
#ifdef SC_GRAPH_TIMING
    auto now_ = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp_ = now_();
#define SC_PHASE(name) do { const double t_ = now_(); fprintf(stderr, "  graph phase %-18s %.2f ms\n", name, t_ - tp_); tp_ = t_; } while (0)
#else
#define SC_PHASE(name) do {} while (0)
#endif
    thread_reads(G, R, thread);
    SC_PHASE("thread_reads");
    // canonize_graph, cpp:754-767
    for (int i = 0; i < (int)nodes.size(); ++i) {            // canonize_insert :553-564
        if (nodes[i].lab == "$") break;
        canonize_insert_at_level(i);
    }
    SC_PHASE("canonize_insert");
    level_cache_build();
    level_cache_on_ = true;
    for (int i = 0; i < (int)nodes.size(); ++i) {            // canonize_delete :742-752
        if (nodes[i].lab == "$") break;
        canonize_delete_at_level(i);
    }
    level_cache_on_ = false;
    std::vector<int>().swap(level_cache_);
    SC_PHASE("canonize_delete");
    directional_merge(false);
    SC_PHASE("forward_merge");
    directional_merge(true);
    SC_PHASE("backward_merge");
    path_collapse();
    SC_PHASE("path_collapse");
    finalize_ids();
    node_level();
    SC_PHASE("ids + levels");
#undef SC_PHASE
}

std::string PoGraph::dump() const {
    std::string s;
    char buf[64];
    for (int i : order_) {
        const GNode& x = nodes[i];
        int rc = 0;
        for (const auto& e : x.pool) rc += e.cn;
        snprintf(buf, sizeof buf, "#\t%d\t%d\t", x.id, x.level);
        s += buf; s += x.lab;
        snprintf(buf, sizeof buf, "\t%d\n", rc);
        s += buf;
    }
    for (int i : order_) {
        const GNode& x = nodes[i];
        for (int o : x.out) {
            snprintf(buf, sizeof buf, "%d\t%d\t%d\n", x.id, nodes[o].id, reads_cover(i, o));
            s += buf;
        }
    }
    return s;
}

// ---------------------------------------------------------------------------
void flatten(const PoGraph& g, int n_reads, FlatGraph& f) {
#ifdef SC_GRAPH_TIMING
    auto now_ = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp_ = now_();
#define SC_PHASE(name) do { const double t_ = now_(); fprintf(stderr, "  flatten phase %-16s %.2f ms\n", name, t_ - tp_); tp_ = t_; } while (0)
#else
#define SC_PHASE(name) do {} while (0)
#endif
    static const char alpha[6] = {'A', 'C', 'G', 'T', '-', '='};
    f.sym.assign(alpha, alpha + 6);
    f.K = 6;
    int code_of[256];
    for (int i = 0; i < 256; i++) code_of[i] = -1;
    for (int i = 0; i < 6; i++) code_of[(unsigned char)alpha[i]] = i;
    auto code = [&](char c) -> uint8_t {
        int& k = code_of[(unsigned char)c];
        if (k < 0) { k = f.K++; f.sym.push_back(c); if (c == 'N') f.code_N = k; }
        return (uint8_t)k;
    };
    std::vector<int> alive;
    for (int i = 0; i < (int)g.nodes.size(); i++) if (g.nodes[i].alive) alive.push_back(i);
    f.n_nodes = (int)alive.size();
    f.node_lab_off.resize(f.n_nodes); f.node_lab_len.resize(f.n_nodes);
    f.node_label_str.resize(f.n_nodes); f.node_is_end.assign(f.n_nodes, 0);
    f.out_ptr.assign(f.n_nodes + 1, 0);
    f.pool_ptr.assign(f.n_nodes + 1, 0);
    size_t n_pool = 0, n_lab = 0, n_out = 0;
    for (int a = 0; a < f.n_nodes; a++) {
        const GNode& x = g.nodes[alive[a]];
        n_pool += x.pool.size(); n_out += x.out.size(); n_lab += x.lab.size();
        (void)n_lab;                                             // read labels are stored once each (below)
    }
    f.pool_rid.resize(n_pool); f.pool_cn.resize(n_pool); f.out_node.reserve(n_out); f.labels.reserve(n_lab);
    for (int a = 0; a < f.n_nodes; a++) {
        const GNode& x = g.nodes[alive[a]];
        f.node_label_str[a] = x.lab;
        f.node_lab_off[a] = (int)f.labels.size();
        f.node_lab_len[a] = (int)x.lab.size();
        bool special = (x.lab == "^" || x.lab == "$");
        f.node_is_end[a] = (x.lab == "$");
        for (char c : x.lab) f.labels.push_back(special ? (uint8_t)0xFF : code(c));
        f.out_ptr[a + 1] = f.out_ptr[a] + (int)x.out.size();
        for (int o : x.out) f.out_node.push_back(g.nodes[o].id);
        f.pool_ptr[a + 1] = f.pool_ptr[a] + (int)x.pool.size();
    }
    const int bulk_threads = (long)n_pool > (1L << 22) ? graph_threads() : 1;
    {
        std::atomic<bool> sorted{true};
        parallel_items(f.n_nodes, bulk_threads, [&](int a) {
            const GNode& x = g.nodes[alive[a]];
            int prev = -1;
            int* pr = f.pool_rid.data() + f.pool_ptr[a]; int* pc = f.pool_cn.data() + f.pool_ptr[a];
            size_t k = 0;
            bool ok = true;
            for (const auto& e : x.pool) {
                pr[k] = e.rid; pc[k] = e.cn; k++;
                if (e.rid < prev) ok = false;
                prev = e.rid;
            }
            if (!ok) sorted.store(false, std::memory_order_relaxed);
        });
        if (!sorted.load()) f.pools_sorted = false;
    }
    f.out_support.assign(f.out_node.size(), 0);
    SC_PHASE("nodes + pools");

    // level walk, NonparametricClustering.cpp:284-334 and :556-575.  A million entries per region: the arrays grow a node's
    // pool at a time and are written by index (this loop was more than half of a region's set-up on the host).
    std::vector<int> lab_off(g.labtab.size(), -1), lab_len(g.labtab.size(), 0);
    for (size_t i = 0; i < g.labtab.size(); i++) lab_len[i] = (int)g.labtab[i].size();
    size_t ne = 0;                                              // entries so far
    int level = 0;
    if (bulk_threads > 1) {
        // A deep region (88 M entries): the same walk in three passes.  (1) the order of the walk alone -- the nodes of every
        // level, where each one's entries go; (2) in parallel over the walk's nodes, the labels each pool carries, in the order
        // it first carries them, merged in walk order into the coding a single pass would have produced (symbol codes are handed
        // out in the order the entries are walked); (3) in parallel over the LEVELS (the first occurrence of a read is a matter
        // of one level), the entries themselves.
        std::vector<int> level_node{0}, sub;
        std::vector<int> visited(f.n_nodes, -1);
        std::vector<size_t> item_off;                           // per walked node: where its entries start (SIZE_MAX: none)
        f.level_node_ptr.push_back(0);
        f.level_ent_ptr.push_back(0);
        while (!level_node.empty()) {
            int end_pos = -1;
            for (size_t qi = 0; qi < level_node.size(); qi++) {
                const int a = level_node[qi];
                const GNode& x = g.nodes[alive[a]];
                f.level_nodes.push_back(a);
                if (a == 0) item_off.push_back(SIZE_MAX);
                else if (f.node_is_end[a]) { end_pos = (int)qi; item_off.push_back(SIZE_MAX); }
                else { item_off.push_back(ne); ne += x.pool.size(); }
                for (int o : x.out) {
                    const int b = g.nodes[o].id;
                    if (visited[b] != level) { sub.push_back(b); visited[b] = level; }
                }
            }
            f.level_node_ptr.push_back((int)f.level_nodes.size());
            f.level_ent_ptr.push_back((int)ne);
            f.level_has_end.push_back(end_pos >= 0);
            f.level_end_pos.push_back(end_pos);
            level += 1;
            level_node.swap(sub);
            sub.clear();
        }
        const int n_items = (int)f.level_nodes.size();
        std::vector<std::vector<int>> firsts((size_t)n_items);
        {
            std::atomic<int> next{0};
            auto scan = [&] {
                std::vector<int> seen(g.labtab.size(), -1);
                for (int it = next.fetch_add(1); it < n_items; it = next.fetch_add(1)) {
                    if (item_off[(size_t)it] == SIZE_MAX) continue;
                    for (const auto& e : g.nodes[alive[f.level_nodes[(size_t)it]]].pool)
                        if (seen[(size_t)e.lab] != it) { seen[(size_t)e.lab] = it; firsts[(size_t)it].push_back(e.lab); }
                }
            };
            std::vector<std::thread> ts;
            for (int t = 1; t < bulk_threads; t++) ts.emplace_back(scan);
            scan();
            for (auto& t : ts) t.join();
        }
        for (int it = 0; it < n_items; it++)
            for (int lab : firsts[(size_t)it])
                if (lab_off[(size_t)lab] < 0) {
                    lab_off[(size_t)lab] = (int)f.labels.size();
                    for (char c : g.labtab[(size_t)lab]) f.labels.push_back(code(c));
                }
        f.ent_rid.resize(ne); f.ent_cn.resize(ne); f.ent_lab_off.resize(ne); f.ent_lab_len.resize(ne); f.ent_first.resize(ne);
        f.level_read_count.assign((size_t)level, 0);
        {
            std::atomic<int> next{0};
            auto fill = [&] {
                std::vector<int> rid_stamp((size_t)std::max(n_reads, 1), -1);
                for (int lv = next.fetch_add(1); lv < level; lv = next.fetch_add(1)) {
                    int lrc = 0;
                    for (int it = f.level_node_ptr[(size_t)lv]; it < f.level_node_ptr[(size_t)lv + 1]; it++) {
                        if (item_off[(size_t)it] == SIZE_MAX) continue;
                        const GNode& x = g.nodes[alive[f.level_nodes[(size_t)it]]];
                        const size_t at = item_off[(size_t)it];
                        int* er = f.ent_rid.data() + at; int* ec = f.ent_cn.data() + at;
                        int* eo = f.ent_lab_off.data() + at; int* el = f.ent_lab_len.data() + at; uint8_t* ef = f.ent_first.data() + at;
                        size_t k = 0;
                        for (const auto& e : x.pool) {
                            er[k] = e.rid; ec[k] = e.cn;
                            eo[k] = lab_off[(size_t)e.lab]; el[k] = lab_len[(size_t)e.lab];
                            if ((size_t)e.rid >= rid_stamp.size()) rid_stamp.resize((size_t)e.rid + 1, -1);
                            ef[k] = rid_stamp[(size_t)e.rid] != lv;
                            rid_stamp[(size_t)e.rid] = lv;
                            lrc += e.cn;
                            k++;
                        }
                    }
                    f.level_read_count[(size_t)lv] = lrc;
                }
            };
            std::vector<std::thread> ts;
            for (int t = 1; t < bulk_threads; t++) ts.emplace_back(fill);
            fill();
            for (auto& t : ts) t.join();
        }
    } else {
    std::vector<int> level_node{0}, sub;
    std::vector<int> visited(f.n_nodes, -1);
    f.level_node_ptr.push_back(0);
    f.level_ent_ptr.push_back(0);
    std::vector<int> rid_stamp((size_t)std::max(n_reads, 1), -1);
    auto grow = [&](size_t n) {
        if (n <= f.ent_rid.size()) return;
        const size_t cap = std::max(n, f.ent_rid.size() + f.ent_rid.size() / 2 + 1024);
        f.ent_rid.resize(cap); f.ent_cn.resize(cap); f.ent_lab_off.resize(cap); f.ent_lab_len.resize(cap);
        f.ent_first.resize(cap);
    };
    grow(n_pool + 16);
    while (!level_node.empty()) {
        int lrc = 0, end_pos = -1;
        for (size_t qi = 0; qi < level_node.size(); qi++) {
            int a = level_node[qi];
            const GNode& x = g.nodes[alive[a]];
            f.level_nodes.push_back(a);
            if (a == 0) {
            } else if (f.node_is_end[a]) {
                end_pos = (int)qi;
            } else {
                grow(ne + x.pool.size());
                int* er = f.ent_rid.data() + ne; int* ec = f.ent_cn.data() + ne;
                int* eo = f.ent_lab_off.data() + ne; int* el = f.ent_lab_len.data() + ne; uint8_t* ef = f.ent_first.data() + ne;
                size_t k = 0;
                for (const auto& e : x.pool) {
                    er[k] = e.rid; ec[k] = e.cn;
                    // a label is coded where an entry first carries it (symbol codes are handed out in the order the
                    // entries are walked) and shared by every later entry
                    int lo = lab_off[(size_t)e.lab];
                    if (lo < 0) {
                        lo = lab_off[(size_t)e.lab] = (int)f.labels.size();
                        for (char c : g.labtab[(size_t)e.lab]) f.labels.push_back(code(c));
                    }
                    eo[k] = lo; el[k] = lab_len[(size_t)e.lab];
                    if ((size_t)e.rid >= rid_stamp.size()) rid_stamp.resize((size_t)e.rid + 1, -1);
                    ef[k] = rid_stamp[(size_t)e.rid] != level;
                    rid_stamp[(size_t)e.rid] = level;
                    lrc += e.cn;
                    k++;
                    // (a single-character strain label against a multi-character read label is modelled on the
                    // device: the reference looks up the never-set key sub_count[(c, "multi")] -> log 0, k_level)
                }
                ne += k;
            }
            for (int o : x.out) {
                int b = g.nodes[o].id;
                if (visited[b] != level) { sub.push_back(b); visited[b] = level; }
            }
        }
        f.level_node_ptr.push_back((int)f.level_nodes.size());
        f.level_ent_ptr.push_back((int)ne);
        f.level_read_count.push_back(lrc);
        f.level_has_end.push_back(end_pos >= 0);
        f.level_end_pos.push_back(end_pos);
        level += 1;
        level_node.swap(sub);
        sub.clear();
    }
    }
    f.ent_rid.resize(ne); f.ent_cn.resize(ne); f.ent_lab_off.resize(ne); f.ent_lab_len.resize(ne); f.ent_first.resize(ne);
    SC_PHASE("level walk");
#undef SC_PHASE
    f.n_levels = level;
    if (f.K > 16 && f.unsupported.empty()) f.unsupported = "more than 16 distinct symbols in node/read labels";
}

}  // namespace sc

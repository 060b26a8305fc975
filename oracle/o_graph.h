/* TEST INFRASTRUCTURE ONLY (oracle).  Literal CPU restatement of the partial
 * order graph of /root/reference/StrainCall/PartialOrderGraph.{hpp,cpp},
 * PartialOrderGraphNode.cpp and LevelOrderIterator.cpp (SURVEY.md section 8 rows
 * a5, a6, a9, a10, a11).  Every function cites the reference lines it follows.
 * Pointer-keyed std::set/std::map in the reference are only used for membership,
 * so they are restated as per-node stamps. */
#ifndef O_GRAPH_H
#define O_GRAPH_H
#include "o_util.h"

enum { ST_MAT = 0, ST_MIS = 1, ST_INS = 2, ST_DEL = 3 }; /* PartialOrderGraph.hpp:82 */

typedef struct { int rid; char *lab; int cn; } RB;       /* ReadBase, hpp:87 */
typedef VEC(RB) RBVec;

typedef struct Node Node;
typedef VEC(Node *) NodeVec;
struct Node {                                            /* hpp:90-124 */
    int id, level, pos;
    int st;
    char *lab;
    NodeVec in, out, sib;
    RBVec pool;
    /* membership stamps replacing std::set<Node*> */
    int stamp_a, stamp_b, stamp_c;
};

typedef struct {                                         /* hpp:231-357 */
    int N;
    NodeVec nodes;
    NodeVec deleted;
    int stamp;                                           /* stamp generator */
} Graph;

typedef struct { char op; int len; } Cig;
typedef VEC(Cig) CigVec;

typedef struct {                                         /* AlignRead, hpp:218 */
    int pos; char *cigar; char *seq; int cn;
} ARead;
typedef VEC(ARead) AReadVec;

/* PartialOrderGraph.cpp:13-59.  '=' and 'X' map to 'M'; digits accumulate.
 * stoi("") would throw in the reference; here an empty count parses as 0. */
static void parse_cigar(const char *cigar, CigVec *out) {
    int v = 0;
    for (const char *p = cigar; *p; ++p) {
        char c = *p;
        if (c == 'M' || c == 'I' || c == 'D' || c == 'N' || c == 'S' || c == 'H' || c == 'P') {
            Cig r = { c, v }; vec_push(*out, r); v = 0;
        } else if (c == '=' || c == 'X') {
            Cig r = { 'M', v }; vec_push(*out, r); v = 0;
        } else if (c >= '0' && c <= '9') {
            v = v * 10 + (c - '0');
        }
    }
}

static int state_eq(const Node *a, int st, const char *lab) { return a->st == st && strcmp(a->lab, lab) == 0; }

static Node *node_new(Graph *g, int st, const char *lab) {  /* PartialOrderGraphNode.cpp:3-9 + add_node :267-271 */
    Node *w = (Node *)xmalloc(sizeof(Node));
    w->id = g->N; w->level = -1; w->pos = -1; w->st = st; w->lab = xstrdup(lab);
    vec_init(w->in); vec_init(w->out); vec_init(w->sib); vec_init(w->pool);
    w->stamp_a = w->stamp_b = w->stamp_c = 0;
    g->N += 1;
    vec_push(g->nodes, w);
    return w;
}
static void pool_push(Node *w, int rid, const char *lab, int cn) {
    RB r = { rid, xstrdup(lab), cn };
    vec_push(w->pool, r);
}
static void nv_erase_first(NodeVec *v, Node *x) {           /* delete_in / delete_out, Node.cpp:16-46 */
    for (int i = 0; i < v->n; i++)
        if (v->v[i] == x) {
            memmove(&v->v[i], &v->v[i + 1], sizeof(Node *) * (size_t)(v->n - i - 1));
            v->n--;
            break;
        }
}
static void add_edge(Node *u, Node *w) { vec_push(u->out, w); vec_push(w->in, u); }   /* cpp:273-278 */
static void add_edge_gap(Node *u, NodeVec *gap) {            /* cpp:281-292 */
    Node *a = u;
    for (int i = 0; i < gap->n; i++) { add_edge(a, gap->v[i]); a = gap->v[i]; }
}
static void add_edge_gap_to(Node *u, Node *v, NodeVec *gap) { /* cpp:294-309 */
    Node *a = u;
    for (int i = 0; i < gap->n; i++) { add_edge(a, gap->v[i]); a = gap->v[i]; }
    add_edge(gap->v[gap->n - 1], v);
}
static void delete_edge(Node *u, Node *v) { nv_erase_first(&u->out, v); nv_erase_first(&v->in, u); } /* cpp:311-316 */
static int linking(Node *u, Node *v) {                       /* cpp:339-348 */
    for (int i = 0; i < u->out.n; i++) if (u->out.v[i] == v) return 1;
    return 0;
}
static Node *find_sibling(Node *v, int st, const char *lab) { /* Node.cpp:66-75 */
    for (int i = 0; i < v->sib.n; i++) if (state_eq(v->sib.v[i], st, lab)) return v->sib.v[i];
    return NULL;
}

/* cpp:406-444 */
static void delete_node(Graph *g, Node *w, int bridging) {
    for (int i = 0; i < w->in.n; i++) {
        Node *p = w->in.v[i];
        for (int j = 0; j < w->out.n; j++) {
            Node *c = w->out.v[j];
            if (!linking(p, c) && bridging) add_edge(p, c);
            nv_erase_first(&c->in, w);
        }
        nv_erase_first(&p->out, w);
    }
    int w_id = w->id;
    int k;
    for (k = 0; k < g->nodes.n; k++) if (g->nodes.v[k]->id == w_id) break;
    if (k < g->nodes.n) {
        memmove(&g->nodes.v[k], &g->nodes.v[k + 1], sizeof(Node *) * (size_t)(g->nodes.n - k - 1));
        g->nodes.n--;
    }
    g->N -= 1;
    for (int i = w_id; i < g->N; ++i) g->nodes.v[i]->id -= 1;
    vec_push(g->deleted, w);
}

static int oracle_fast_support(void) {
    static int on = -1;
    if (on < 0) on = getenv("SC_ORACLE_FAST_SUPPORT") != NULL;
    return on;
}

/* cpp:1218-1244 */
static int number_of_reads_cover_nodes(Graph *g, Node *u, Node *v) {
    int n = 0;
    if (u == g->nodes.v[0]) {
        for (int i = 0; i < v->pool.n; i++) n += v->pool.v[i].cn;
    } else if (strcmp(v->lab, "$") == 0) {
        for (int i = 0; i < u->pool.n; i++) n += u->pool.v[i].cn;
    } else if (oracle_fast_support()) {
        /* the same sum by counting: sum_j (entries of u's pool with v_j's read) * cn_j.  Only for inputs the literal
           double loop cannot finish (pools of 10^5 entries, tests/golden/make_golden_config4.py); the CPU suite checks
           that both forms print the same -G dump and FASTA on the committed cases. */
        static int *mult = NULL;
        static int cap = 0;
        int hi = 0;
        for (int i = 0; i < u->pool.n; i++) if (u->pool.v[i].rid >= hi) hi = u->pool.v[i].rid + 1;
        for (int j = 0; j < v->pool.n; j++) if (v->pool.v[j].rid >= hi) hi = v->pool.v[j].rid + 1;
        if (hi > cap) { free(mult); cap = hi * 2 + 16; mult = (int *)calloc((size_t)cap, sizeof(int)); }
        for (int i = 0; i < u->pool.n; i++) mult[u->pool.v[i].rid]++;
        for (int j = 0; j < v->pool.n; j++) n += mult[v->pool.v[j].rid] * v->pool.v[j].cn;
        for (int i = 0; i < u->pool.n; i++) mult[u->pool.v[i].rid] = 0;
    } else {
        for (int i = 0; i < u->pool.n; i++)
            for (int j = 0; j < v->pool.n; j++)
                if (u->pool.v[i].rid == v->pool.v[j].rid) n += v->pool.v[j].cn;
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* insert canonisation (a6)                                                   */
typedef struct { Node *u, *v; NodeVec gap; } GapEx;           /* hpp:129 */
typedef VEC(GapEx) GapExVec;

static NodeVec nv_copy(const NodeVec *a) {
    NodeVec c; vec_init(c);
    for (int i = 0; i < a->n; i++) vec_push(c, a->v[i]);
    return c;
}

/* cpp:355-393 */
static void find_insert_from(Node *u, GapExVec *inserts) {
    NodeVec g; vec_init(g);
    NodeVec st; vec_init(st);
    vec_push(st, u);
    while (st.n) {
        Node *v = st.v[--st.n];
        if (v == u) {
            for (int i = 0; i < v->out.n; i++)
                if (v->out.v[i]->st == ST_INS) vec_push(st, v->out.v[i]);
        } else if (v->st == ST_MAT || v->st == ST_MIS) {
            GapEx e; e.u = u; e.v = v; e.gap = nv_copy(&g);
            vec_push(*inserts, e);
            vec_clear(g);
        } else {
            vec_push(g, v);
            for (int i = 0; i < v->out.n; i++) vec_push(st, v->out.v[i]);
        }
    }
    vec_free(g); vec_free(st);
}
/* cpp:395-403 */
static void find_insert_at_level(Graph *g, int i, GapExVec *inserts) {
    Node *u = g->nodes.v[i];
    find_insert_from(u, inserts);
    for (int k = 0; k < u->sib.n; k++) find_insert_from(u->sib.v[k], inserts);
}

/* read_pool_t = std::set<tuple<int,int>>: kept as a sorted unique array */
typedef struct { int rid, cn; } RC;
typedef VEC(RC) RCVec;
static int rc_cmp(const void *a, const void *b) {
    const RC *x = (const RC *)a, *y = (const RC *)b;
    if (x->rid != y->rid) return x->rid < y->rid ? -1 : 1;
    if (x->cn != y->cn) return x->cn < y->cn ? -1 : 1;
    return 0;
}
static void rc_make_set(RCVec *s) {
    qsort(s->v, (size_t)s->n, sizeof(RC), rc_cmp);
    int m = 0;
    for (int i = 0; i < s->n; i++)
        if (m == 0 || rc_cmp(&s->v[m - 1], &s->v[i]) != 0) s->v[m++] = s->v[i];
    s->n = m;
}
static void rc_erase(RCVec *s, int rid, int cn) {
    /* std::set::erase(find(key)); a key that is absent is UB in the reference
     * (erase(end())): here it is ignored. */
    for (int i = 0; i < s->n; i++)
        if (s->v[i].rid == rid && s->v[i].cn == cn) {
            memmove(&s->v[i], &s->v[i + 1], sizeof(RC) * (size_t)(s->n - i - 1));
            s->n--;
            return;
        }
}
/* cpp:780-829 */
static void find_common_read_pool(Node *a, Node *b, RCVec *c) {
    RCVec ar, br; vec_init(ar); vec_init(br);
    for (int i = 0; i < a->pool.n; i++) { RC x = { a->pool.v[i].rid, a->pool.v[i].cn }; vec_push(ar, x); }
    rc_make_set(&ar);
    for (int i = 0; i < a->out.n; i++) {
        Node *o = a->out.v[i];
        if (o->st == ST_INS || o->st == ST_DEL)
            for (int k = 0; k < o->pool.n; k++) rc_erase(&ar, o->pool.v[k].rid, o->pool.v[k].cn);
    }
    for (int i = 0; i < b->pool.n; i++) { RC x = { b->pool.v[i].rid, b->pool.v[i].cn }; vec_push(br, x); }
    rc_make_set(&br);
    for (int i = 0; i < b->in.n; i++) {
        Node *o = b->in.v[i];
        if (o->st == ST_INS || o->st == ST_DEL)
            for (int k = 0; k < o->pool.n; k++) rc_erase(&br, o->pool.v[k].rid, o->pool.v[k].cn);
    }
    vec_clear(*c);
    int i = 0, j = 0;
    while (i < ar.n && j < br.n) {
        int d = rc_cmp(&ar.v[i], &br.v[j]);
        if (d == 0) { vec_push(*c, ar.v[i]); i++; j++; }
        else if (d < 0) i++;
        else j++;
    }
    vec_free(ar); vec_free(br);
}

static void add_dash_chain(Graph *g, Node *a, Node *b, int l, RCVec *crp) {
    NodeVec gap; vec_init(gap);
    for (int t = 0; t < l; ++t) {
        Node *w = node_new(g, ST_INS, "-");
        for (int k = 0; k < crp->n; k++) pool_push(w, crp->v[k].rid, "-", crp->v[k].cn);
        vec_push(gap, w);
    }
    add_edge_gap_to(a, b, &gap);
    vec_free(gap);
}
/* cpp:831-923: add a '-' chain of length l on every direct edge level i -> i+1 */
static void add_edge_level(Graph *g, int i, int l) {
    Node *u = g->nodes.v[i], *v = g->nodes.v[i + 1];
    RCVec crp; vec_init(crp);
    if (linking(u, v)) { find_common_read_pool(u, v, &crp); add_dash_chain(g, u, v, l, &crp); }
    for (int k = 0; k < v->sib.n; k++)
        if (linking(u, v->sib.v[k])) { find_common_read_pool(u, v->sib.v[k], &crp); add_dash_chain(g, u, v->sib.v[k], l, &crp); }
    for (int k = 0; k < u->sib.n; k++)
        if (linking(u->sib.v[k], v)) { find_common_read_pool(u->sib.v[k], v, &crp); add_dash_chain(g, u->sib.v[k], v, l, &crp); }
    for (int a = 0; a < u->sib.n; a++)
        for (int b = 0; b < v->sib.n; b++)
            if (linking(u->sib.v[a], v->sib.v[b])) {
                find_common_read_pool(u->sib.v[a], v->sib.v[b], &crp);
                add_dash_chain(g, u->sib.v[a], v->sib.v[b], l, &crp);
            }
    vec_free(crp);
}
/* cpp:925-961 */
static void delete_edge_level(Graph *g, int i) {
    Node *u = g->nodes.v[i], *v = g->nodes.v[i + 1];
    if (linking(u, v)) delete_edge(u, v);
    for (int k = 0; k < v->sib.n; k++) if (linking(u, v->sib.v[k])) delete_edge(u, v->sib.v[k]);
    for (int k = 0; k < u->sib.n; k++) if (linking(u->sib.v[k], v)) delete_edge(u->sib.v[k], v);
    for (int a = 0; a < u->sib.n; a++)
        for (int b = 0; b < v->sib.n; b++)
            if (linking(u->sib.v[a], v->sib.v[b])) delete_edge(u->sib.v[a], v->sib.v[b]);
}

/* forward declaration: sum-of-pairs MSA (o_msa.h).  rows_out[t] = padded row t,
 * returns number of columns. */
static int msa_sp_align(char **seqs, int n, char ***rows_out);

typedef struct { GapEx *v; } GapSortCtx;
static int gap_less(void *ctx, int a, int b) {                /* cpp:466 lambda: size(a) > size(b) */
    GapSortCtx *c = (GapSortCtx *)ctx;
    return c->v[a].gap.n > c->v[b].gap.n;
}

/* cpp:446-550 */
static void canonize_insert_at_level(Graph *g, int i) {
    GapExVec ins0; vec_init(ins0);
    find_insert_at_level(g, i, &ins0);
    if (ins0.n == 0) { vec_free(ins0); return; }
    int n = ins0.n;
    int *perm = (int *)xmalloc(sizeof(int) * (size_t)n);
    for (int t = 0; t < n; t++) perm[t] = t;
    GapSortCtx sc = { ins0.v };
    std_sort_perm(perm, n, gap_less, &sc);
    GapEx *inserts = (GapEx *)xmalloc(sizeof(GapEx) * (size_t)n);
    for (int t = 0; t < n; t++) inserts[t] = ins0.v[perm[t]];
    free(perm);

    char **seqs = (char **)xmalloc(sizeof(char *) * (size_t)n);
    int l = 0, k = 1000000000;
    for (int t = 0; t < n; t++) {
        NodeVec *gp = &inserts[t].gap;
        size_t len = 0;
        for (int q = 0; q < gp->n; q++) len += strlen(gp->v[q]->lab);
        char *s = (char *)xmalloc(len + 1); s[0] = 0;
        for (int q = 0; q < gp->n; q++) strcat(s, gp->v[q]->lab);
        seqs[t] = s;
        if ((int)len > l) l = (int)len;
        if ((int)len < k) k = (int)len;
    }
    int d = l - k;
    if (n == 1 || d == 0) {
        add_edge_level(g, i, l);
        delete_edge_level(g, i);
    } else {
        char **rows = NULL;
        int ncol = msa_sp_align(seqs, n, &rows);
        for (int t = 0; t < n; ++t) {
            const char *res = rows[t];
            if (strcmp(res, seqs[t]) != 0) {
                int rid = 0, rcn = 0;
                NodeVec *gp = &inserts[t].gap;
                for (int q = 0; q < gp->n; q++) {
                    rid = gp->v[q]->pool.v[0].rid;
                    rcn = gp->v[q]->pool.v[0].cn;
                    delete_node(g, gp->v[q], 1);
                }
                NodeVec ng; vec_init(ng);
                for (const char *p = res; *p; ++p) {
                    char lab[2] = { *p, 0 };
                    Node *w = node_new(g, ST_INS, lab);
                    pool_push(w, rid, lab, rcn);
                    vec_push(ng, w);
                }
                add_edge_gap_to(inserts[t].u, inserts[t].v, &ng);
                vec_free(ng);
            }
        }
        l = ncol;
        add_edge_level(g, i, l);
        delete_edge_level(g, i);
        for (int t = 0; t < n; t++) free(rows[t]);
        free(rows);
    }
    for (int t = 0; t < n; t++) { free(seqs[t]); vec_free(inserts[t].gap); }
    free(seqs); free(inserts); vec_free(ins0);
}
/* cpp:553-564 */
static void canonize_insert(Graph *g) {
    for (int i = 0; i < g->N; ++i) {
        if (strcmp(g->nodes.v[i]->lab, "$") == 0) break;
        canonize_insert_at_level(g, i);
    }
}

/* ------------------------------------------------------------------------- */
/* delete canonisation (a9)                                                   */
/* cpp:571-622 */
static int node_level_exclude_delete(Graph *g, Node *w) {
    NodeVec level_node, sub; vec_init(level_node); vec_init(sub);
    int level = 0;
    int stamp = ++g->stamp;          /* visited_node, cleared per level */
    vec_push(level_node, g->nodes.v[0]);
    while (level_node.n) {
        Node *u = level_node.v[--level_node.n];
        if (u == w) break;
        for (int i = 0; i < u->out.n; i++) {
            Node *o = u->out.v[i];
            if (o->st == ST_DEL) continue;
            vec_push(sub, o);
            for (int k = 0; k < o->sib.n; k++) vec_push(sub, o->sib.v[k]);
        }
        if (level_node.n == 0) {
            while (sub.n) {
                Node *v = sub.v[--sub.n];
                if (v->stamp_a == stamp) continue;
                vec_push(level_node, v);
                v->stamp_a = stamp;
            }
            level += 1;
            stamp = ++g->stamp;
        }
    }
    vec_free(level_node); vec_free(sub);
    return level;
}
/* cpp:624-672 */
static void find_delete_from(Node *w, GapExVec *deletes) {
    NodeVec gap, st; vec_init(gap); vec_init(st);
    IntVec cnt; vec_init(cnt);
    vec_push(st, w); vec_push(cnt, 0);
    while (st.n) {
        Node *u = st.v[--st.n];
        int c = cnt.v[--cnt.n];
        if (u == w) {
            for (int i = 0; i < u->out.n; i++)
                if (u->out.v[i]->st == ST_DEL) { vec_push(st, u->out.v[i]); vec_push(cnt, 0); }
        } else if (u->st == ST_DEL) {
            if (c == 0) {
                vec_push(st, u); vec_push(cnt, 1);
                vec_push(gap, u);
                for (int i = 0; i < u->out.n; i++) { vec_push(st, u->out.v[i]); vec_push(cnt, 0); }
            } else {
                gap.n--;
            }
        } else {
            GapEx e; e.u = w; e.v = u; e.gap = nv_copy(&gap);
            vec_push(*deletes, e);
        }
    }
    vec_free(gap); vec_free(st); vec_free(cnt);
}
/* cpp:684-740 (find_delete_at_level :674-682 inlined) */
static void canonize_delete_at_level(Graph *g, int i) {
    GapExVec deletes; vec_init(deletes);
    Node *u0 = g->nodes.v[i];
    find_delete_from(u0, &deletes);
    for (int k = 0; k < u0->sib.n; k++) find_delete_from(u0->sib.v[k], &deletes);
    if (deletes.n == 0) { vec_free(deletes); return; }
    /* map<Node*,int> nl : node -> level, local to this call */
    NodeVec nl_key; IntVec nl_val; vec_init(nl_key); vec_init(nl_val);
    for (int t = 0; t < deletes.n; t++) {
        Node *u = deletes.v[t].u, *v = deletes.v[t].v;
        int ul = -1, vl = -1, f;
        for (f = 0; f < nl_key.n; f++) if (nl_key.v[f] == u) break;
        if (f == nl_key.n) { int x = node_level_exclude_delete(g, u); vec_push(nl_key, u); vec_push(nl_val, x); }
        ul = nl_val.v[f];
        for (f = 0; f < nl_key.n; f++) if (nl_key.v[f] == v) break;
        if (f == nl_key.n) { int x = node_level_exclude_delete(g, v); vec_push(nl_key, v); vec_push(nl_val, x); }
        vl = nl_val.v[f];
        int dl = vl - ul - 1;
        int dd = deletes.v[t].gap.n;
        if (dl - dd > 0) {
            Node *v0 = deletes.v[t].gap.v[0];
            int rid = v0->pool.v[0].rid, rcn = v0->pool.v[0].cn;
            NodeVec ng; vec_init(ng);
            for (int q = dl - dd; q > 0; --q) {
                Node *w = node_new(g, ST_DEL, "=");
                pool_push(w, rid, "=", rcn);
                vec_push(ng, w);
            }
            add_edge_gap_to(u, v0, &ng);
            delete_edge(u, v0);
            vec_free(ng);
        }
    }
    for (int t = 0; t < deletes.n; t++) vec_free(deletes.v[t].gap);
    vec_free(deletes); vec_free(nl_key); vec_free(nl_val);
}
/* cpp:742-752 */
static void canonize_delete(Graph *g) {
    for (int i = 0; i < g->N; i++) {
        if (strcmp(g->nodes.v[i]->lab, "$") == 0) break;
        canonize_delete_at_level(g, i);
    }
}

/* ------------------------------------------------------------------------- */
/* merging (a10)                                                              */
static int rb_cmp(const void *a, const void *b) {             /* tuple<int,string,int> operator< */
    const RB *x = (const RB *)a, *y = (const RB *)b;
    if (x->rid != y->rid) return x->rid < y->rid ? -1 : 1;
    int c = strcmp(x->lab, y->lab);                            /* labels are 7-bit text */
    if (c) return c < 0 ? -1 : 1;
    if (x->cn != y->cn) return x->cn < y->cn ? -1 : 1;
    return 0;
}
/* cpp:963-1005 */
static void merge_read_pool(Node *u, Node *v) {
    int i = 0, j = 0, m = u->pool.n, n = v->pool.n;
    qsort(u->pool.v, (size_t)m, sizeof(RB), rb_cmp);
    qsort(v->pool.v, (size_t)n, sizeof(RB), rb_cmp);
    RBVec res; vec_init(res);
    while (i < m && j < n) {
        RB *a = &u->pool.v[i], *b = &v->pool.v[j];
        if (a->rid == b->rid) {
            RB r = { a->rid, str_cat(a->lab, b->lab), a->cn }; vec_push(res, r); i++; j++;
        } else if (a->rid < b->rid) {
            RB r = { a->rid, xstrdup(a->lab), a->cn }; vec_push(res, r); i++;
        } else {
            RB r = { b->rid, xstrdup(b->lab), b->cn }; vec_push(res, r); j++;
        }
    }
    while (i < m) { RB r = { u->pool.v[i].rid, xstrdup(u->pool.v[i].lab), u->pool.v[i].cn }; vec_push(res, r); i++; }
    while (j < n) { RB r = { v->pool.v[j].rid, xstrdup(v->pool.v[j].lab), v->pool.v[j].cn }; vec_push(res, r); j++; }
    for (int k = 0; k < u->pool.n; k++) free(u->pool.v[k].lab);
    free(u->pool.v);
    u->pool.v = res.v; u->pool.n = res.n; u->pool.cap = res.cap;
}
/* cpp:1007-1038 */
static void merge_node(Graph *g, Node *u, Node *v) {
    for (int i = 0; i < v->in.n; i++) {
        Node *p = v->in.v[i];
        if (!linking(p, u) && p != u) add_edge(p, u);
    }
    for (int i = 0; i < v->out.n; i++) {
        Node *c = v->out.v[i];
        if (!linking(u, c) && u != c) add_edge(u, c);
    }
    if (linking(u, v) && u->st == ST_MAT && v->st == ST_MAT) {
        char *nl = str_cat(u->lab, v->lab);
        free(u->lab);
        u->lab = nl;
    }
    merge_read_pool(u, v);
    delete_node(g, v, 0);
}
/* cpp:1040-1094 (dir=0) and cpp:1097-1159 (dir=1) */
static void directional_merge(Graph *g, int backward) {
    NodeVec q; vec_init(q);
    int qh = 0;
    int st_merged = ++g->stamp;  /* merged_node  (stamp_a) */
    int st_visit = ++g->stamp;   /* visited      (stamp_b) */
    if (!backward) {
        vec_push(q, g->nodes.v[0]);
    } else {
        for (int i = 0; i < g->nodes.n; i++)
            if (strcmp(g->nodes.v[i]->lab, "$") == 0) vec_push(q, g->nodes.v[i]);
    }
    NodeVec mu, mv; vec_init(mu); vec_init(mv);
    while (qh < q.n) {
        Node *w = q.v[qh++];
        if (w->stamp_a == st_merged) continue;
        NodeVec *adj = backward ? &w->in : &w->out;
        for (int a = 0; a < adj->n; a++) {
            Node *u = adj->v[a];
            for (int b = a + 1; b < adj->n; b++) {
                Node *v = adj->v[b];
                if (u == v) continue;
                if (u->st == v->st && strcmp(u->lab, v->lab) == 0)
                    if (u->stamp_a != st_merged && v->stamp_a != st_merged) {
                        vec_push(mu, u); vec_push(mv, v);
                        v->stamp_a = st_merged;
                    }
            }
        }
        for (int t = 0; t < mu.n; t++) merge_node(g, mu.v[t], mv.v[t]);
        adj = backward ? &w->in : &w->out;
        for (int a = 0; a < adj->n; a++) {
            Node *c = adj->v[a];
            if (c->stamp_b != st_visit) { vec_push(q, c); c->stamp_b = st_visit; }
        }
        vec_clear(mu); vec_clear(mv);
    }
    vec_free(q); vec_free(mu); vec_free(mv);
}
/* cpp:754-767 */
static void canonize_graph(Graph *g) {
    canonize_insert(g);
    canonize_delete(g);
    directional_merge(g, 0);
    directional_merge(g, 1);
}

/* cpp:1171-1216 */
static void path_collapse(Graph *g) {
    int level = 0, level_size = 0;
    NodeVec lq, sq; vec_init(lq); vec_init(sq);
    int lh = 0;
    int stamp = ++g->stamp;      /* multi_in (stamp_c), cleared per level */
    vec_push(lq, g->nodes.v[0]);
    while (lh < lq.n) {
        Node *u = lq.v[lh++];
        if (level_size == 1 && u->out.n == 1) {
            Node *v = u->out.v[0];
            while (v->out.n == 1) {
                merge_node(g, u, v);
                v = u->out.v[0];
            }
        }
        for (int i = 0; i < u->out.n; i++) {
            Node *v = u->out.v[i];
            if (v->stamp_c != stamp) { vec_push(sq, v); v->stamp_c = stamp; }
        }
        if (lh == lq.n) {
            vec_clear(lq); lh = 0;
            for (int i = 0; i < sq.n; i++) vec_push(lq, sq.v[i]);
            vec_clear(sq);
            level += 1;
            level_size = lq.n;
            stamp = ++g->stamp;
        }
    }
    (void)level;
    vec_free(lq); vec_free(sq);
}

/* cpp:769-776 driving LevelOrderIterator.cpp:3-56 (stacks, visited set cleared
 * whenever the sub-level stack is found empty on entry to operator++). */
static void node_level(Graph *g) {
    NodeVec level_node, sub; vec_init(level_node); vec_init(sub);
    Node *begin = g->nodes.v[0];
    int n = 0, level = 0;
    int stamp = ++g->stamp;
    Node *cur = begin; int cur_level = 0;
    for (int i = 0; i < begin->out.n; i++) vec_push(level_node, begin->out.v[i]);
    begin->stamp_a = stamp;
    while (n != g->N) {
        cur->level = cur_level;
        /* operator++ */
        if (sub.n == 0) { level += 1; stamp = ++g->stamp; }
        if (level_node.n) {
            Node *w = level_node.v[--level_node.n];
            cur = w; cur_level = level;
            n += 1;
            for (int i = 0; i < w->out.n; i++) vec_push(sub, w->out.v[i]);
            if (level_node.n == 0) {
                while (sub.n) {
                    Node *x = sub.v[--sub.n];
                    if (x->stamp_a == stamp) continue;
                    vec_push(level_node, x);
                    x->stamp_a = stamp;
                }
            }
        } else {
            n += 1;
        }
    }
    vec_free(level_node); vec_free(sub);
}

/* cpp:67-265 */
static Graph *graph_build(const char *G, ARead *R, int nR) {
    Graph *g = (Graph *)xmalloc(sizeof(Graph));
    g->N = 0; vec_init(g->nodes); vec_init(g->deleted); g->stamp = 0;
    int glen = (int)strlen(G);
    Node *B = node_new(g, ST_MAT, "^");
    Node *u = B, *v, *w;
    for (int i = 0; i < glen; i++) {
        char lab[2] = { G[i], 0 };
        w = node_new(g, ST_MAT, lab);
        w->pos = i;
        add_edge(u, w);
        u = w;
    }
    Node *E = node_new(g, ST_MAT, "$");
    add_edge(u, E);

    for (int rid = 0; rid < nR; rid++) {
        ARead *r = &R[rid];
        u = g->nodes.v[r->pos];
        v = g->nodes.v[r->pos + 1];
        int i = r->pos, j = 0, dl = 0;
        const char *rs = r->seq;
        int rlen = (int)strlen(rs);
        CigVec cig; vec_init(cig);
        parse_cigar(r->cigar, &cig);
        for (int c = 0; c < cig.n; c++) {
            char op = cig.v[c].op; int opl = cig.v[c].len;
            if (op == 'S') {
                j += j + opl;                        /* sic, cpp:126 */
                dl = 0;
                continue;
            } else if (op == 'M') {
                for (int k = 0; k < opl + dl; k++, j++) {
                    /* std::string operator[] past the end is UB; treat as NUL */
                    char rc = j < rlen ? rs[j] : 0;
                    char gc = i < glen ? G[i] : 0;
                    int st = (gc == rc) ? ST_MAT : ST_MIS;
                    char lab[2] = { rc, 0 };
                    if (state_eq(v, st, lab)) {
                        if (!linking(u, v)) add_edge(u, v);
                        pool_push(v, rid, lab, r->cn);
                        u = v;
                        v = g->nodes.v[++i + 1];
                    } else {
                        Node *s = find_sibling(v, st, lab);
                        if (!s) {
                            w = node_new(g, st, lab);
                            add_edge(u, w);
                            pool_push(w, rid, lab, r->cn);
                            vec_push(v->sib, w);
                            u = w;
                            v = g->nodes.v[++i + 1];
                        } else {
                            if (!linking(u, s)) add_edge(u, s);
                            pool_push(s, rid, lab, r->cn);
                            u = s;
                            v = g->nodes.v[++i + 1];
                        }
                    }
                }
                dl = 0;
            } else if (op == 'I') {
                NodeVec gap; vec_init(gap);
                for (int k = 0; k < opl; k++, j++) {
                    char lab[2] = { j < rlen ? rs[j] : 0, 0 };
                    w = node_new(g, ST_INS, lab);
                    pool_push(w, rid, lab, r->cn);
                    vec_push(gap, w);
                }
                add_edge_gap(u, &gap);
                if (gap.n) u = gap.v[gap.n - 1];
                dl = 0;
                vec_free(gap);
            } else if (op == 'D') {
                NodeVec gap; vec_init(gap);
                for (int k = 0; k < opl; k++) {
                    w = node_new(g, ST_DEL, "=");
                    pool_push(w, rid, "=", r->cn);
                    vec_push(gap, w);
                    v = g->nodes.v[++i + 1];
                }
                if (strcmp(v->lab, "$") == 0) {
                    add_edge_gap_to(u, v, &gap);
                    u = v;
                    vec_free(gap);
                    continue;
                }
                add_edge_gap(u, &gap);
                if (gap.n) u = gap.v[gap.n - 1];
                dl = 0;
                vec_free(gap);
            }
        }
        if (!linking(u, v) && u != v) add_edge(u, v);
        vec_free(cig);
    }
    canonize_graph(g);
    path_collapse(g);
    node_level(g);
    return g;
}

/* cpp:318-337 */
static void output_edge(Graph *g, FILE *f) {
    for (int i = 0; i < g->nodes.n; i++) {
        Node *x = g->nodes.v[i];
        int rc = 0;
        for (int k = 0; k < x->pool.n; k++) rc += x->pool.v[k].cn;
        fprintf(f, "#\t%d\t%d\t%s\t%d\n", x->id, x->level, x->lab, rc);
    }
    for (int i = 0; i < g->nodes.n; i++) {
        Node *x = g->nodes.v[i];
        for (int k = 0; k < x->out.n; k++)
            fprintf(f, "%d\t%d\t%d\n", x->id, x->out.v[k]->id, number_of_reads_cover_nodes(g, x, x->out.v[k]));
    }
}

static void node_free(Node *x) {
    free(x->lab);
    for (int k = 0; k < x->pool.n; k++) free(x->pool.v[k].lab);
    vec_free(x->pool); vec_free(x->in); vec_free(x->out); vec_free(x->sib);
    free(x);
}
static void graph_free(Graph *g) {
    for (int i = 0; i < g->nodes.n; i++) node_free(g->nodes.v[i]);
    for (int i = 0; i < g->deleted.n; i++) node_free(g->deleted.v[i]);
    vec_free(g->nodes); vec_free(g->deleted);
    free(g);
}
#endif

/* TEST INFRASTRUCTURE ONLY (oracle).  Literal CPU restatement, in x87 long
 * double like the reference, of the strain model and the level-by-level
 * Dirichlet-process clustering: /root/reference/StrainCall/Strain.cpp and
 * NonparametricClustering.cpp. */
#ifndef O_CLUSTER_H
#define O_CLUSTER_H
#include "o_graph.h"

/* ReadPairs = map<int, vector<int>> (PartialOrderGraph.hpp:88) */
typedef struct { IntVec *mates; int n; } ReadPairs;
static int rp_get(ReadPairs *rp, int id, int k) {
    /* operator[] on a missing id / index past the end is UB in the reference
     * (SURVEY.md section 7 "literal quirks"); treated as "no mate" here. */
    if (id < 0 || id >= rp->n || k < 0 || k >= rp->mates[id].n) return -1;
    return rp->mates[id].v[k];
}

static const char ALPHA6[6] = { 'A', 'C', 'G', 'T', '-', '=' };   /* Strain.cpp:7 */
static int alpha_idx(const char *s) {
    if (s[0] && !s[1]) for (int i = 0; i < 6; i++) if (ALPHA6[i] == s[0]) return i;
    return -1;
}

typedef struct { char *a, *b; ld v; } SubEnt;       /* SubstitutionCount entries outside the 6x6 block */
typedef VEC(SubEnt) SubVec;

typedef struct {
    ld Z;
    ld comp[6];                                     /* comp_count over the alphabet; other keys are always 0 */
    ld sub[6][6];                                   /* sub_count over the alphabet */
    SubVec other;                                   /* sub_count, every other (string,string) key */
    ld abundance;
    ld *rll; unsigned char *has; int nreads;        /* read_loglik map<int,DoubleL> */
    NodeVec path;
} Strain;
typedef VEC(Strain) StrainVec;

static ld *sub_ref(Strain *s, const char *a, const char *b) {      /* sub_count[Substitution(a,b)] */
    int i = alpha_idx(a), j = alpha_idx(b);
    if (i >= 0 && j >= 0) return &s->sub[i][j];
    for (int k = 0; k < s->other.n; k++)
        if (strcmp(s->other.v[k].a, a) == 0 && strcmp(s->other.v[k].b, b) == 0) return &s->other.v[k].v;
    SubEnt e = { xstrdup(a), xstrdup(b), 0 };
    vec_push(s->other, e);
    return &s->other.v[s->other.n - 1].v;
}
static void strain_recount(Strain *s) {                            /* Strain.cpp:58-68, 115-124 */
    s->Z = 0;
    for (int i = 0; i < 6; i++) {
        s->comp[i] = 0;
        for (int j = 0; j < 6; j++) s->comp[i] += s->sub[i][j];
        s->Z += s->comp[i];
    }
}
static void strain_init(Strain *s, int N, ld e, int nreads) {      /* Strain.cpp:41-71 */
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) s->sub[i][j] = (i == j) ? N * (1 - e) : N * e;
    vec_init(s->other);
    strain_recount(s);
    s->abundance = 0;
    s->nreads = nreads;
    s->rll = (ld *)xmalloc(sizeof(ld) * (size_t)nreads);
    s->has = (unsigned char *)xmalloc((size_t)nreads);
    memset(s->has, 0, (size_t)nreads);
    for (int i = 0; i < nreads; i++) s->rll[i] = 0;
    vec_init(s->path);
}
static void strain_copy(Strain *d, const Strain *s) {              /* Strain.cpp:73-83 */
    *d = *s;
    vec_init(d->other);
    for (int k = 0; k < s->other.n; k++) {
        SubEnt e = { xstrdup(s->other.v[k].a), xstrdup(s->other.v[k].b), s->other.v[k].v };
        vec_push(d->other, e);
    }
    d->rll = (ld *)xmalloc(sizeof(ld) * (size_t)s->nreads);
    memcpy(d->rll, s->rll, sizeof(ld) * (size_t)s->nreads);
    d->has = (unsigned char *)xmalloc((size_t)s->nreads);
    memcpy(d->has, s->has, (size_t)s->nreads);
    d->path = nv_copy(&s->path);
}
static void strain_free(Strain *s) {
    for (int k = 0; k < s->other.n; k++) { free(s->other.v[k].a); free(s->other.v[k].b); }
    vec_free(s->other);
    free(s->rll); free(s->has); vec_free(s->path);
}
static ld strain_logprob2(Strain *s, const char *a, const char *b) {   /* Strain.cpp:132-135 */
    int i = alpha_idx(a);
    ld c = (i >= 0) ? s->comp[i] : 0;                  /* comp_count[a] inserts 0 for a new key */
    return logl(*sub_ref(s, a, b)) - logl(c);
}
static ld strain_logprob_id(Strain *s, int id) {                       /* Strain.cpp:147-150: read_loglik[id] */
    if (!s->has[id]) { s->has[id] = 1; s->rll[id] = 0; }
    return s->rll[id];
}
static void strain_update_read_loglik(Strain *s, int id, ld v) {       /* Strain.cpp:85-95 */
    if (!s->has[id]) { s->has[id] = 1; s->rll[id] = v; }
    else s->rll[id] += v;
}
typedef struct { char *a, *b; ld v; } ScEnt;          /* one SubstitutionCount map */
typedef VEC(ScEnt) ScMap;
static void sc_add(ScMap *m, const char *a, const char *b, ld v) {
    for (int k = 0; k < m->n; k++)
        if (strcmp(m->v[k].a, a) == 0 && strcmp(m->v[k].b, b) == 0) { m->v[k].v += v; return; }
    ScEnt e = { xstrdup(a), xstrdup(b), v };
    vec_push(*m, e);
}
static void sc_free(ScMap *m) {
    for (int k = 0; k < m->n; k++) { free(m->v[k].a); free(m->v[k].b); }
    vec_free(*m);
}
static void strain_update_model(Strain *s, ld al, ScMap *sc) {         /* Strain.cpp:106-125 */
    s->abundance += al;
    for (int k = 0; k < sc->n; k++) *sub_ref(s, sc->v[k].a, sc->v[k].b) += sc->v[k].v;
    strain_recount(s);
}
static char *strain_seq(Strain *s) {                                   /* Strain.cpp:162-170 */
    size_t len = 0;
    for (int i = 0; i < s->path.n; i++) len += strlen(s->path.v[i]->lab);
    char *r = (char *)xmalloc(len + 1), *p = r;
    for (int i = 0; i < s->path.n; i++) { size_t l = strlen(s->path.v[i]->lab); memcpy(p, s->path.v[i]->lab, l); p += l; }
    *p = 0;
    return r;
}
static char *strain_plain_seq(Strain *s) {                             /* Strain.cpp:211-223 */
    size_t len = 0;
    for (int i = 0; i < s->path.n; i++) len += strlen(s->path.v[i]->lab);
    char *r = (char *)xmalloc(len + 1), *p = r;
    for (int i = 0; i < s->path.n; i++) {
        const char *pl = s->path.v[i]->lab;
        if (strcmp(pl, "^") && strcmp(pl, "$") && strcmp(pl, "-") && strcmp(pl, "=")) {
            size_t l = strlen(pl); memcpy(p, pl, l); p += l;
        }
    }
    *p = 0;
    return r;
}

static void normalize_ld(ld *f, int n) {                               /* NonparametricClustering.cpp:10-15 */
    ld z = 0;
    for (int i = 0; i < n; i++) z += f[i];
    for (int i = 0; i < n; i++) f[i] /= z;
}

/* std::discrete_distribution<int>(p.begin(),p.end())(gen): libstdc++
 * <bits/random.tcc> param_type::_M_initialize + operator(). */
static int discrete_draw(const ld *p, int S, MT *gen, double *prob, double *cp) {
    if (S < 2) return 0;                               /* no random number consumed */
    for (int i = 0; i < S; i++) prob[i] = (double)p[i];
    double sum = 0.0;
    for (int i = 0; i < S; i++) sum += prob[i];
    for (int i = 0; i < S; i++) prob[i] /= sum;
    double acc = 0.0;
    for (int i = 0; i < S; i++) { acc = (i == 0) ? prob[0] : acc + prob[i]; cp[i] = acc; }
    cp[S - 1] = 1.0;
    double u = mt_canonical(gen);
    int lo = 0, len = S;                               /* std::lower_bound */
    while (len > 0) {
        int half = len >> 1, mid = lo + half;
        if (cp[mid] < u) { lo = mid + 1; len = len - half - 1; }
        else len = half;
    }
    return lo;
}

typedef struct {
    Graph *g;
    ReadPairs *rp;
    int trace; int trace_prec;
    FILE *trace_fp;
    long draws;                                        /* statistics */
} ClusterCtx;

/* NonparametricClustering.cpp:17-125 */
static void hard_clustering(ClusterCtx *cx, StrainVec *strains, RBVec *reads, IntVec *new_reads) {
    int S = strains->n;
    ld *abundance = (ld *)xmalloc(sizeof(ld) * (size_t)S);
    ld *p = (ld *)xmalloc(sizeof(ld) * (size_t)S);
    ScMap *substitute = (ScMap *)xmalloc(sizeof(ScMap) * (size_t)S);
    for (int s = 0; s < S; s++) { abundance[s] = 0; vec_init(substitute[s]); }
    for (int ri = 0; ri < reads->n; ri++) {
        RB *r = &reads->v[ri];
        int id = r->rid;
        for (int cn = r->cn; cn > 0; cn--) {
            int uid = rp_get(cx->rp, id, cn - 1);
            for (int s = 0; s < S; s++) p[s] = strains->v[s].abundance;
            normalize_ld(p, S);
            for (int s = 0; s < S; s++) {
                p[s] = logl(p[s]) + strain_logprob_id(&strains->v[s], id);
                if (uid >= 0) p[s] += strain_logprob_id(&strains->v[s], uid);
                p[s] = expl(p[s]);
            }
            normalize_ld(p, S);
            for (int s = 0; s < S; s++) {
                abundance[s] += p[s];
                const char *sl = strains->v[s].path.v[strains->v[s].path.n - 1]->lab;
                const char *rl = r->lab;
                if (strlen(rl) == 1) {
                    sc_add(&substitute[s], sl, rl, p[s]);
                } else if (new_reads->v[ri]) {
                    int i = (int)strlen(sl), j = (int)strlen(rl);
                    while (i > 0 && j > 0) {
                        char a[2] = { sl[--i], 0 }, b[2] = { rl[--j], 0 };
                        sc_add(&substitute[s], a, b, p[s]);
                    }
                } else {
                    int m = (int)strlen(sl), n = (int)strlen(rl), i = 0, j = 0;
                    while (i < m && j < n) {
                        char a[2] = { sl[i++], 0 }, b[2] = { rl[j++], 0 };
                        sc_add(&substitute[s], a, b, p[s]);
                    }
                }
            }
        }
    }
    for (int s = 0; s < S; s++) { strain_update_model(&strains->v[s], abundance[s], &substitute[s]); sc_free(&substitute[s]); }
    free(abundance); free(p); free(substitute);
}

/* NonparametricClustering.cpp:128-244 */
static void np_bayes_clustering(ClusterCtx *cx, StrainVec *strains, RBVec *reads, int n, ld **abundance_out) {
    int m = reads->n, S = strains->n;
    ld *a = (ld *)xmalloc(sizeof(ld) * (size_t)S), *p = (ld *)xmalloc(sizeof(ld) * (size_t)S);
    double *prob = (double *)xmalloc(sizeof(double) * (size_t)S), *cp = (double *)xmalloc(sizeof(double) * (size_t)S);
    ScMap *sc = (ScMap *)xmalloc(sizeof(ScMap) * (size_t)S);
    MT gen; mt_seed(&gen, 1234);
    for (int s = 0; s < S; s++) { a[s] = strains->v[s].abundance; p[s] = 0; vec_init(sc[s]); }
    int read_size = 0;
    for (int j = 0; j < m; j++) read_size += reads->v[j].cn;
    n = n < 40000 / read_size ? n : 40000 / read_size;
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < m; j++) {
            int id = reads->v[j].rid;
            for (int cn = reads->v[j].cn; cn > 0; cn--) {
                for (int s = 0; s < S; s++) p[s] = a[s];
                normalize_ld(p, S);
                for (int s = 0; s < S; s++) {
                    Strain *st = &strains->v[s];
                    p[s] = logl(p[s]) + strain_logprob_id(st, id);
                    int uid = rp_get(cx->rp, id, cn - 1);
                    if (uid >= 0 && st->has[uid]) p[s] += strain_logprob_id(st, uid);
                    p[s] = expl(p[s]);
                }
                int c = discrete_draw(p, S, &gen, prob, cp);
                cx->draws++;
                a[c] += 1;
                Strain *sc_st = &strains->v[c];
                sc_add(&sc[c], sc_st->path.v[sc_st->path.n - 1]->lab, reads->v[j].lab, 1);
            }
        }
    }
    normalize_ld(a, S);
    for (int s = 0; s < S; s++) a[s] *= read_size;
    for (int s = 0; s < S; s++) for (int k = 0; k < sc[s].n; k++) sc[s].v[k].v /= n;
    *abundance_out = (ld *)xmalloc(sizeof(ld) * (size_t)S);
    memcpy(*abundance_out, a, sizeof(ld) * (size_t)S);
    for (int s = 0; s < S; s++) { strain_update_model(&strains->v[s], a[s], &sc[s]); sc_free(&sc[s]); }
    free(a); free(p); free(prob); free(cp); free(sc);
}

/* NonparametricClustering.cpp:246-254 */
static int ld_desc(const void *x, const void *y) {
    ld a = *(const ld *)x, b = *(const ld *)y;
    return a > b ? -1 : (a < b ? 1 : 0);
}
static ld Qx(const ld *a0, int sz, int n) {
    ld *a = (ld *)xmalloc(sizeof(ld) * (size_t)sz);
    memcpy(a, a0, sizeof(ld) * (size_t)sz);
    qsort(a, (size_t)sz, sizeof(ld), ld_desc);
    ld r = (n >= sz) ? a[sz - 1] : a[n];
    free(a);
    return r;
}

/* NonparametricClustering.cpp:584-612.  Reading b[i] past b's end is UB in the
 * reference; a NUL is read here. */
static ld seq_identity(const char *a, const char *b) {
    int iden = 0, len = 0;
    size_t la = strlen(a), lb = strlen(b);
    for (size_t i = 0; i < la; ++i) {
        char x = a[i], y = i < lb ? b[i] : 0;
        if (x == '-' && y == '-') continue;
        else if (x == '=' && y == '=') continue;
        else if (x == '=' && y == '-') continue;
        else if (x == '-' && y == '=') continue;
        else if (x == '^' && y == '^') continue;
        else if (x == y) iden += 1;
        len += 1;
    }
    return (ld)((iden + 0.0) / len);
}

typedef struct { Strain *v; } StrainSortCtx;
static int strain_abund_gt(void *ctx, int a, int b) {
    StrainSortCtx *c = (StrainSortCtx *)ctx;
    return c->v[a].abundance > c->v[b].abundance;
}
/* std::sort(strains, abundance descending) as libstdc++ permutes it */
static void sort_strains(StrainVec *sv) {
    int n = sv->n;
    if (n < 2) return;
    int *perm = (int *)xmalloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; i++) perm[i] = i;
    StrainSortCtx c = { sv->v };
    std_sort_perm(perm, n, strain_abund_gt, &c);
    Strain *tmp = (Strain *)xmalloc(sizeof(Strain) * (size_t)n);
    for (int i = 0; i < n; i++) tmp[i] = sv->v[perm[i]];
    memcpy(sv->v, tmp, sizeof(Strain) * (size_t)n);
    free(tmp); free(perm);
}
/* NonparametricClustering.cpp:645-670 */
static void merge_strains(StrainVec *sv, ld diff) {
    sort_strains(sv);
    int n = sv->n;
    /* No candidate left (every one was pruned): the reference copy-constructs `merged` from strains[0] of an
     * EMPTY vector (:650) -- undefined behaviour that, in the shipped and the rebuilt binary alike, ends in an
     * empty stdout and exit status 0.  The restatement keeps the empty set. */
    if (n == 0) return;
    char **seqs = (char **)xmalloc(sizeof(char *) * (size_t)n);
    for (int i = 0; i < n; i++) seqs[i] = strain_seq(&sv->v[i]);
    IntVec merged; vec_init(merged);
    vec_push(merged, 0);
    for (int i = 1; i < n; i++) {
        int j;
        for (j = 0; j < merged.n; j++) {
            if (seq_identity(seqs[i], seqs[merged.v[j]]) > 1 - diff) {
                sv->v[merged.v[j]].abundance += sv->v[i].abundance;
                break;
            }
        }
        if (j == merged.n) vec_push(merged, i);
    }
    Strain *out = (Strain *)xmalloc(sizeof(Strain) * (size_t)(merged.n ? merged.n : 1));
    unsigned char *keep = (unsigned char *)xmalloc((size_t)n);
    memset(keep, 0, (size_t)n);
    for (int j = 0; j < merged.n; j++) { out[j] = sv->v[merged.v[j]]; keep[merged.v[j]] = 1; }
    for (int i = 0; i < n; i++) { if (!keep[i]) strain_free(&sv->v[i]); free(seqs[i]); }
    free(sv->v);
    sv->v = out; sv->n = merged.n; sv->cap = merged.n ? merged.n : 1;
    free(seqs); free(keep); vec_free(merged);
}
/* NonparametricClustering.cpp:672-702: apart from the sort, only `count` (never
 * read) and zero-insertions into read_loglik are produced. */
static void read_reassign(StrainVec *sv, RCVec *total_reads) {
    sort_strains(sv);
    for (int t = 0; t < total_reads->n; t++)
        for (int s = 0; s < sv->n; s++) (void)strain_logprob_id(&sv->v[s], total_reads->v[t].rid);
}

static void trace_dump(ClusterCtx *cx, const char *when, int level, StrainVec *sv) {
    if (!cx->trace || sv->n == 0) return;
    FILE *f = cx->trace_fp;
    fprintf(f, "------------------------------\n%s\nlevel: %d\n", when, level);
    for (int s = 0; s < sv->n; s++) {
        char *q = strain_seq(&sv->v[s]);
        fprintf(f, "%s\t%.*Lg\n", q, cx->trace_prec, sv->v[s].abundance);
        free(q);
    }
}

static uint64_t str_hash(const char *s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ull; }
    return h;
}

/* NonparametricClustering.cpp:262-582 */
static void streaming_clustering(ClusterCtx *cx, StrainVec *strains_out, int n, ld e, ld tau, ld diff, int nreads) {
    Graph *g = cx->g;
    int level = 0, branching = 0;
    StrainVec level_strains, sub_strains; vec_init(level_strains); vec_init(sub_strains);
    NodeVec level_node, sub_node; vec_init(level_node); vec_init(sub_node);
    int lh = 0;
    int stamp = ++g->stamp;                           /* has_visited (stamp_a) */
    RBVec level_reads; vec_init(level_reads);
    int level_read_count = 0;
    RCVec total_reads; vec_init(total_reads);         /* set<tuple<int,int>>, kept sorted lazily */
    IntVec new_reads; vec_init(new_reads);

    Strain s0; strain_init(&s0, 100, e, nreads);
    vec_push(level_strains, s0);
    vec_push(level_node, g->nodes.v[0]);
    while (lh < level_node.n) {
        trace_dump(cx, "before clustering", level, &level_strains);
        Node *u = level_node.v[lh++];
        if (u == g->nodes.v[0]) {
            vec_push(level_strains.v[0].path, u);
            level_strains.v[0].abundance = 1;
        } else if (strcmp(u->lab, "$") == 0) {
            rc_make_set(&total_reads);
            read_reassign(&level_strains, &total_reads);
            merge_strains(&level_strains, diff);
            for (int s = 0; s < strains_out->n; s++) strain_free(&strains_out->v[s]);
            vec_clear(*strains_out);
            for (int s = 0; s < level_strains.n; s++) { Strain c; strain_copy(&c, &level_strains.v[s]); vec_push(*strains_out, c); }
        } else {
            for (int k = 0; k < u->pool.n; k++) {
                RB r = u->pool.v[k];                  /* labels borrowed from the node */
                vec_push(level_reads, r);
                level_read_count += r.cn;
            }
        }
        for (int k = 0; k < u->out.n; k++) {
            Node *v = u->out.v[k];
            if (v->stamp_a != stamp) { vec_push(sub_node, v); v->stamp_a = stamp; }
        }
        if (lh == level_node.n) {
            if (level_reads.n) {
                vec_clear(new_reads);
                for (int k = 0; k < level_reads.n; k++) vec_push(new_reads, 0);
                for (int s = 0; s < level_strains.n; s++) {
                    Strain *st = &level_strains.v[s];
                    const char *sb0 = st->path.v[st->path.n - 1]->lab;
                    int sbl = (int)strlen(sb0);
                    for (int ri = 0; ri < level_reads.n; ri++) {
                        int rid = level_reads.v[ri].rid;
                        const char *rb = level_reads.v[ri].lab;
                        ld loglik;
                        if (sbl == 1) {
                            const char *sb = sb0;
                            if (strcmp(sb, "N") == 0) sb = rb;
                            loglik = strain_logprob2(st, sb, rb);
                        } else {
                            loglik = 0;
                            int rbl = (int)strlen(rb);
                            if (!st->has[rid]) {
                                int ii = sbl, jj = rbl;
                                while (ii > 0 && jj > 0) {
                                    char a[2] = { sb0[--ii], 0 }, b[2] = { rb[--jj], 0 };
                                    if (a[0] == 'N') a[0] = b[0];
                                    loglik += strain_logprob2(st, a, b);
                                }
                                new_reads.v[ri] = 1;
                            } else {
                                int ii = 0, jj = 0;
                                while (ii < sbl && jj < rbl) {
                                    char a[2] = { sb0[ii++], 0 }, b[2] = { rb[jj++], 0 };
                                    if (a[0] == 'N') a[0] = b[0];
                                    loglik += strain_logprob2(st, a, b);
                                }
                            }
                        }
                        strain_update_read_loglik(st, rid, loglik);
                    }
                }
                for (int k = 0; k < level_reads.n; k++) { RC x = { level_reads.v[k].rid, level_reads.v[k].cn }; vec_push(total_reads, x); }
            }
            if (branching && level_reads.n) {
                int S = level_strains.n;
                /* A_prior / A_posterior are map<string,DoubleL> keyed by strain_seq():
                 * equal strings share one entry and the last writer wins. */
                char **seqs = (char **)xmalloc(sizeof(char *) * (size_t)S);
                uint64_t *hs = (uint64_t *)xmalloc(sizeof(uint64_t) * (size_t)S);
                int *last = (int *)xmalloc(sizeof(int) * (size_t)S);
                ld *prior = (ld *)xmalloc(sizeof(ld) * (size_t)S), *post = (ld *)xmalloc(sizeof(ld) * (size_t)S);
                for (int s = 0; s < S; s++) { seqs[s] = strain_seq(&level_strains.v[s]); hs[s] = str_hash(seqs[s]); }
                for (int s = 0; s < S; s++) {
                    last[s] = s;
                    for (int t = S - 1; t > s; t--)
                        if (hs[t] == hs[s] && strcmp(seqs[t], seqs[s]) == 0) { last[s] = t; break; }
                }
                for (int s = 0; s < S; s++) prior[s] = level_strains.v[last[s]].abundance;
                ld *abundance = NULL;
                np_bayes_clustering(cx, &level_strains, &level_reads, n, &abundance);
                for (int s = 0; s < S; s++) post[s] = level_strains.v[last[s]].abundance;
                ld A_delta_max = 0;
                for (int s = 0; s < S; s++) { ld d = post[s] - prior[s]; if (A_delta_max < d) A_delta_max = d; }
                ld Z = 0;
                for (int s = 0; s < S; s++) Z += abundance[s];
                ld Zt = Z * tau;
                unsigned char *del = (unsigned char *)xmalloc((size_t)S);
                for (int s = 0; s < S; s++) {
                    ld d = post[s] - prior[s];
                    del[s] = (abundance[s] < Zt || d < 0.01 * A_delta_max) ? 1 : 0;
                }
                int w = 0;
                for (int s = 0; s < S; s++) {
                    if (del[s]) strain_free(&level_strains.v[s]);
                    else level_strains.v[w++] = level_strains.v[s];
                }
                level_strains.n = w;
                for (int s = 0; s < S; s++) free(seqs[s]);
                free(seqs); free(hs); free(last); free(prior); free(post); free(abundance); free(del);
            } else if (level_reads.n) {
                hard_clustering(cx, &level_strains, &level_reads, &new_reads);
            }
            trace_dump(cx, "after clustering", level, &level_strains);

            branching = 0;
            for (int si = 0; si < level_strains.n; si++) {
                Strain *s = &level_strains.v[si];
                Node *v = s->path.v[s->path.n - 1];
                ld oz = 0, moc = 0;
                int nout = v->out.n;
                ld *oc = (ld *)xmalloc(sizeof(ld) * (size_t)(nout ? nout : 1));
                for (int k = 0; k < nout; k++) {
                    int oc0 = number_of_reads_cover_nodes(g, v, v->out.v[k]);
                    oc[k] = oc0; oz += oc0;
                    if (moc < oc0) moc = oc0;
                }
                int dd = 0;
                for (int k = 0; k < nout; k++) {
                    Node *o = v->out.v[k];
                    if (strcmp(o->lab, "$") != 0 && oz > 0) {
                        if (oc[k] <= 1. && oc[k] < moc) { dd += 1; continue; }
                        Strain ns; strain_copy(&ns, s);
                        vec_push(ns.path, o);
                        if (oc[k] > 0) ns.abundance = s->abundance * oc[k] / oz;
                        else { double t = (double)tau; ns.abundance = oz * (0.01 < t ? 0.01 : t); }
                        vec_push(sub_strains, ns);
                    } else {
                        Strain ns; strain_copy(&ns, s);
                        vec_push(ns.path, o);
                        ns.abundance = s->abundance;
                        vec_push(sub_strains, ns);
                    }
                }
                if (nout > 1 + dd) branching = 1;
                free(oc);
            }
            /* the reference's literal is 80 (NonparametricClustering.cpp:532-551); SC_ORACLE_MAX_CANDIDATES lets a test
             * raise it (sc_params.max_candidates of the product) so that levels with up to 128 candidates -- the widest
             * variants of the sampler kernel -- have an oracle too */
            static int cap80 = 0;
            if (!cap80) { const char *e = getenv("SC_ORACLE_MAX_CANDIDATES"); cap80 = e ? atoi(e) : 80; if (cap80 < 1) cap80 = 80; }
            if (sub_strains.n > cap80) {
                ld *ssa = (ld *)xmalloc(sizeof(ld) * (size_t)sub_strains.n);
                for (int s = 0; s < sub_strains.n; s++) ssa[s] = sub_strains.v[s].abundance;
                ld Zt0 = Qx(ssa, sub_strains.n, cap80);
                int w = 0;
                for (int s = 0; s < sub_strains.n; s++) {
                    if (sub_strains.v[s].abundance < Zt0) strain_free(&sub_strains.v[s]);
                    else sub_strains.v[w++] = sub_strains.v[s];
                }
                sub_strains.n = w;
                free(ssa);
            }
            level += 1;
            vec_clear(level_node); lh = 0;
            for (int k = 0; k < sub_node.n; k++) vec_push(level_node, sub_node.v[k]);
            vec_clear(sub_node);
            for (int s = 0; s < level_strains.n; s++) strain_free(&level_strains.v[s]);
            vec_clear(level_strains);
            for (int s = 0; s < sub_strains.n; s++) vec_push(level_strains, sub_strains.v[s]);
            vec_clear(sub_strains);
            stamp = ++g->stamp;
            vec_clear(level_reads);
            level_read_count = 0;
        }
    }
    (void)level_read_count;
    for (int s = 0; s < level_strains.n; s++) strain_free(&level_strains.v[s]);
    vec_free(level_strains); vec_free(sub_strains); vec_free(level_node); vec_free(sub_node);
    vec_free(level_reads); vec_free(total_reads); vec_free(new_reads);
}

/* NonparametricClustering.cpp:776-836 */
static void read_assign(ClusterCtx *cx, StrainVec *strains, ARead *reads, int nreads, int n) {
    int read_size = 0, S = strains->n;
    for (int i = 0; i < nreads; i++) read_size += reads[i].cn;
    n = n < 40000 / read_size ? n : 40000 / read_size;
    MT gen; mt_seed(&gen, 1234);
    ld *a = (ld *)xmalloc(sizeof(ld) * (size_t)(S ? S : 1)), *p = (ld *)xmalloc(sizeof(ld) * (size_t)(S ? S : 1));
    double *prob = (double *)xmalloc(sizeof(double) * (size_t)(S ? S : 1)), *cp = (double *)xmalloc(sizeof(double) * (size_t)(S ? S : 1));
    for (int s = 0; s < S; s++) { a[s] = strains->v[s].abundance; p[s] = 0; }
    for (; n > 0; n--) {
        for (int id = 0; id < nreads; id++) {
            for (int cn = reads[id].cn; cn > 0; --cn) {
                int uid = rp_get(cx->rp, id, cn - 1);
                memcpy(p, a, sizeof(ld) * (size_t)S);
                normalize_ld(p, S);
                for (int i = 0; i < S; i++) {
                    Strain *st = &strains->v[i];
                    p[i] = logl(p[i]) + strain_logprob_id(st, id);
                    if (uid >= 0) p[i] += strain_logprob_id(st, uid);
                    p[i] = expl(p[i]);
                }
                int c = discrete_draw(p, S, &gen, prob, cp);
                cx->draws++;
                a[c] += 1;
            }
        }
    }
    normalize_ld(a, S);
    for (int i = 0; i < S; i++) strains->v[i].abundance = a[i];
    free(a); free(p); free(prob); free(cp);
}
#endif

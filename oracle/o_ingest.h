/* TEST INFRASTRUCTURE ONLY (oracle).  Literal CPU restatement of StrainCall's
 * argv parsing, scan windows and read ingest:
 * /root/reference/StrainCall/StrainCall.cpp:58-154 (a1), :673-848 (a2),
 * :157-185 (a3), :276-670 (a4).  Like the reference it shells out to `samtools`
 * (tests put oracle/tools on PATH); temp files go to $TMPDIR via mkstemp instead
 * of `<roi>_<time>_<rand>` in the cwd (SURVEY.md section 7 "temp-file collisions"). */
#ifndef O_INGEST_H
#define O_INGEST_H
#include <unistd.h>
#include "o_cluster.h"

typedef struct {
    char *gene_file, *mapping_file, *roi;
    int window_size, overlap_size;
    float error_rate;
    int mapping_qual, max_ins, read_len;
    int print_help;
    float tau, diff_rate;
    int max_depth, d0, d1, plot_graph;
} ScParam;

static void sc_param_init(ScParam *p) {                               /* StrainCall.cpp:58-95 */
    p->gene_file = xstrdup(""); p->mapping_file = xstrdup(""); p->roi = xstrdup("");
    p->window_size = 500; p->overlap_size = 100; p->error_rate = 0.01f; p->mapping_qual = 3;
    p->max_ins = 10; p->read_len = 80; p->print_help = 0; p->d0 = 0; p->d1 = 0;
    p->tau = 0.02f; p->diff_rate = 0.01f; p->max_depth = 800; p->plot_graph = 0;
}
static int opt_is(const char *op, const char *s, const char *l) {
    /* "-x", "--long" or "-long" */
    return strcmp(op, s) == 0 || (op[0] == '-' && op[1] == '-' && strcmp(op + 2, l) == 0) ||
           (op[0] == '-' && strcmp(op + 1, l) == 0);
}
static void sc_parse_cmd_line(int argc, char **argv, ScParam *p) {     /* StrainCall.cpp:98-154 */
    for (int i = 0, k = 0; i < argc; ++i) {
        const char *op = argv[i];
        if (op[0] == '-') {
            if (opt_is(op, "-h", "help")) p->print_help = 1;
            else if (opt_is(op, "-r", "roi")) { free(p->roi); p->roi = xstrdup(argv[++i]); }
            else if (opt_is(op, "-w", "window")) p->window_size = atoi(argv[++i]);
            else if (opt_is(op, "-e", "error-rate")) p->error_rate = strtof(argv[++i], NULL);
            else if (opt_is(op, "-q", "map-qual")) p->mapping_qual = atoi(argv[++i]);
            else if (opt_is(op, "-o", "overlap")) p->overlap_size = atoi(argv[++i]);
            else if (opt_is(op, "-l", "read-len")) p->read_len = atoi(argv[++i]);
            else if (opt_is(op, "-t", "tau")) p->tau = strtof(argv[++i], NULL);
            else if (opt_is(op, "-d", "diff-rate")) p->diff_rate = strtof(argv[++i], NULL);
            else if (opt_is(op, "-D", "max-depth")) p->max_depth = atoi(argv[++i]);
            else if (opt_is(op, "-I", "max-ins")) p->max_ins = atoi(argv[++i]);
            else if (opt_is(op, "-G", "plot-graph")) p->plot_graph = 1;
        } else {
            if (k == 0) { free(p->gene_file); p->gene_file = xstrdup(op); }
            else { free(p->mapping_file); p->mapping_file = xstrdup(op); }
            k++;
        }
    }
}

static char *tmp_path(void) {
    const char *d = getenv("TMPDIR");
    char *t = (char *)xmalloc(strlen(d ? d : "/tmp") + 32);
    sprintf(t, "%s/sc_oracle_XXXXXX", d ? d : "/tmp");
    int fd = mkstemp(t);
    if (fd >= 0) close(fd);
    return t;
}
static char *read_line(FILE *f) {          /* getline without the newline; NULL at EOF */
    size_t cap = 256, n = 0;
    char *b = (char *)xmalloc(cap);
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '\n') { b[n] = 0; return b; }
        if (n + 2 > cap) { cap *= 2; b = (char *)xrealloc(b, cap); }
        b[n++] = (char)c;
    }
    if (n == 0) { free(b); return NULL; }
    b[n] = 0;
    return b;
}
/* `ss >> f1 >> f2 ...`: whitespace-separated tokens; missing ones are "" */
static int split_ws(char *line, char **f, int maxf) {
    int n = 0;
    char *p = line;
    while (n < maxf) {
        while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f') p++;
        if (!*p) break;
        f[n++] = p;
        while (*p && !(*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n' || *p == '\v' || *p == '\f')) p++;
        if (*p) *p++ = 0;
    }
    for (int i = n; i < maxf; i++) f[i] = (char *)"";
    return n;
}

/* StrainCall.cpp:157-185 */
static char *load_gene_seq(const char *gene_file, const char *gene_roi) {
    char *tmp = tmp_path();
    size_t L = strlen(gene_file) + strlen(gene_roi) + strlen(tmp) + 64;
    char *cmd = (char *)xmalloc(L);
    snprintf(cmd, L, "samtools faidx %s %s 2>/dev/null 1>%s", gene_file, gene_roi, tmp);
    if (system(cmd)) {}
    size_t cap = 4096, n = 0;
    char *seq = (char *)xmalloc(cap); seq[0] = 0;
    FILE *in = fopen(tmp, "r");
    if (in) {
        for (char *line; (line = read_line(in));) {
            if (line[0] != '>') {
                size_t l = strlen(line);
                if (n + l + 1 > cap) { while (n + l + 1 > cap) cap *= 2; seq = (char *)xrealloc(seq, cap); }
                memcpy(seq + n, line, l + 1); n += l;
            }
            free(line);
        }
        fclose(in);
    }
    remove(tmp); free(tmp); free(cmd);
    return seq;
}
static char *gene_roi_name(const char *roi) {                          /* :188-192 */
    const char *c = strchr(roi, ':');
    return c ? xstrndup(roi, (size_t)(c - roi)) : xstrdup(roi);
}
static int gene_roi_start_pos(const char *roi) {                       /* :194-208 */
    const char *c = strchr(roi, ':');
    return atoi(c ? c + 1 : roi);      /* digits up to the first '-' */
}
static int gene_roi_end_pos(const char *roi) {                         /* :210-220: first '-' anywhere in roi */
    const char *c = strchr(roi, '-');
    return atoi(c ? c + 1 : roi);
}
static char *fai_gene_name(const char *gene_file) {                    /* :222-247: last record */
    size_t L = strlen(gene_file) + 8;
    char *p = (char *)xmalloc(L); snprintf(p, L, "%s.fai", gene_file);
    char *gn = xstrdup("");
    FILE *in = fopen(p, "r");
    if (in) {
        for (char *line; (line = read_line(in));) {
            char *f[5]; split_ws(line, f, 5);
            if (f[0][0]) { free(gn); gn = xstrdup(f[0]); }
            free(line);
        }
        fclose(in);
    }
    free(p);
    return gn;
}
static int fai_gene_length(const char *gene_file, const char *name) {  /* :249-273 */
    size_t L = strlen(gene_file) + 8;
    char *p = (char *)xmalloc(L); snprintf(p, L, "%s.fai", gene_file);
    int len = 0;
    FILE *in = fopen(p, "r");
    if (in) {
        for (char *line; (line = read_line(in));) {
            char *f[5]; split_ws(line, f, 5);
            if (strcmp(f[0], name) == 0) len = atoi(f[1]);
            free(line);
        }
        fclose(in);
    }
    free(p);
    return len;
}

static int read_align_end_pos(int p0, CigVec *c) {                     /* :276-289 */
    for (int i = 0; i < c->n; i++) if (c->v[i].op == 'M' || c->v[i].op == 'D') p0 += c->v[i].len;
    return p0 - 1;
}

/* :291-414.  Returns 0 on success, -1 where std::string::substr would throw. */
static int crop_read_within_window(int wp0, int wp1, const char *seq, const char *qual, CigVec *cigars,
                                   int rp0, int rp1, char **crop_seq, char **crop_cigar) {
    int i = 0, j = 0, ki = 0, kj = 0, opl;
    char op;
    CigVec cc; vec_init(cc);
    int it = 0;
    op = cigars->v[it].op; opl = cigars->v[it].len;
    if (op == 'S') { i += opl; ++it; }
    op = cigars->v[it].op; opl = cigars->v[it].len;
    if (rp0 < wp0 && rp0 < wp1) {
        while (rp0 < wp0 && rp0 < wp1) {
            ki = 0;
            op = cigars->v[it].op; opl = cigars->v[it].len;
            if (op == 'M') { for (; ki < opl; ++ki, ++i, ++rp0) if (rp0 == wp0) break; }
            else if (op == 'D') { for (; ki < opl; ++ki, ++rp0) if (rp0 == wp0) break; }
            else if (op == 'I') i += opl;
            ++it;
        }
    } else {
        ++it;
    }
    if (ki < opl) { Cig r = { op, opl - ki }; vec_push(cc, r); }
    for (; it < cigars->n; ++it) vec_push(cc, cigars->v[it]);

    int rit = cigars->n - 1;
    op = cigars->v[rit].op; opl = cigars->v[rit].len;
    if (op == 'S') { j += opl; --rit; cc.n--; }
    op = cigars->v[rit].op; opl = cigars->v[rit].len;
    while (rp1 > wp1 && rp1 > wp0) {
        kj = 0;
        op = cigars->v[rit].op; opl = cigars->v[rit].len;
        if (op == 'M') { for (; kj < opl; ++kj, ++j, --rp1) if (rp1 == wp1) break; }
        else if (op == 'D') { for (; kj < opl; ++kj, --rp1) if (rp1 == wp1) break; }
        else if (op == 'I') j += opl;
        --rit;
        if (kj == opl || op == 'I') cc.n--;
        else cc.v[cc.n - 1].len -= kj;
    }
    size_t sl = strlen(seq), ql = strlen(qual);
    if ((size_t)i > sl || (size_t)j > ql) { vec_free(cc); return -1; }
    size_t cnt = sl - (size_t)i - (size_t)j;            /* size_t wrap like substr's count */
    if (cnt > sl - (size_t)i) cnt = sl - (size_t)i;
    *crop_seq = xstrndup(seq + i, cnt);
    size_t cap = 16 * (size_t)(cc.n + 1);
    char *cg = (char *)xmalloc(cap); cg[0] = 0;
    for (int k = 0; k < cc.n; k++) { char b[32]; snprintf(b, sizeof b, "%d%c", cc.v[k].len, cc.v[k].op); strcat(cg, b); }
    *crop_cigar = cg;
    vec_free(cc);
    return 0;
}
static int number_of_ambiguous_base(const char *r) {                   /* :416-425 */
    int n = 0;
    for (; *r; ++r) if (*r == 'N' || *r == 'n') n++;
    return n;
}
static int max_insert_size(const char *cigar) {                        /* :427-440 */
    CigVec c; vec_init(c);
    parse_cigar(cigar, &c);
    int ins = 0;
    for (int i = 0; i < c.n; i++) if (c.v[i].len > ins && c.v[i].op == 'I') ins = c.v[i].len;
    vec_free(c);
    return ins;
}

/* map<AlignRead, vector<string>> keyed by (pos, cigar, seq, "", 1) */
typedef struct { int pos; char *cigar, *seq; VEC(char *) names; } DupEnt;
static int dup_cmp(const DupEnt *a, int pos, const char *cigar, const char *seq) {
    if (a->pos != pos) return a->pos < pos ? -1 : 1;
    int c = strcmp(a->cigar, cigar);
    if (c) return c;
    return strcmp(a->seq, seq);
}
typedef struct { char *name; int uid; } NameUid;
typedef struct { const char *name; int idx; } NameIdx;
static int nameidx_cmp(const void *a, const void *b) {
    const NameIdx *x = (const NameIdx *)a, *y = (const NameIdx *)b;
    int c = strcmp(x->name, y->name);
    if (c) return c;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static int nameuid_cmp(const void *a, const void *b) { return strcmp(((const NameUid *)a)->name, ((const NameUid *)b)->name); }

/* :480-670.  Output: reads (sorted unique alignments with copy numbers) and
 * read_pairs[uid] = mate uid (or -1) per read name, in name order. */
static void load_mapping_reads(const char *gene_seq, const char *mapping_file, int mq, int rl, int max_ins,
                               int max_depth, const char *gene_roi, AReadVec *reads, ReadPairs *rp) {
    (void)gene_seq;
    char *tmp = tmp_path();
    size_t L = strlen(mapping_file) + strlen(gene_roi) + strlen(tmp) + 96;
    char *cmd = (char *)xmalloc(L);
    snprintf(cmd, L, "samtools view %s -q %d -F 1804 %s 2>/dev/null 1>%s", mapping_file, mq, gene_roi, tmp);
    if (system(cmd)) {}
    MT gen; mt_seed(&gen, 1234);
    int p0 = gene_roi_start_pos(gene_roi), p1 = gene_roi_end_pos(gene_roi);
    int depth = 0;
    FILE *in = fopen(tmp, "r");
    if (in) {
        for (char *line; (line = read_line(in));) {
            char *f[11]; split_ws(line, f, 11);
            CigVec c; vec_init(c);
            parse_cigar(f[5], &c);
            int len = 0, r0 = atoi(f[3]);
            for (int i = 0; i < c.n; i++) if (c.v[i].op == 'M' || c.v[i].op == 'D') len += c.v[i].len;
            int r1 = r0 + len - 1;
            if (p0 <= r0 && p1 > r1) depth += r1 - r0 + 1;
            else if (p0 <= r0 && p1 <= r1) depth += p1 - r0 + 1;
            else if (p0 > r0 && p1 <= r1) depth += p1 - p0 + 1;
            else if (p0 > r0 && p1 > r1) depth += r1 - p0 + 1;
            vec_free(c); free(line);
        }
        fclose(in);
    }
    depth /= p1 - p0 + 1;
    double q = max_depth / (depth + 0.);
    ld rho = (ld)(1.0 < q ? 1.0 : q);

    VEC(DupEnt) dups; vec_init(dups);       /* kept sorted by key */
    in = fopen(tmp, "r");
    if (in) {
        for (char *line; (line = read_line(in));) {
            char *f[11]; split_ws(line, f, 11);
            if ((int)strlen(f[9]) < rl) { free(line); continue; }
            if (number_of_ambiguous_base(f[9]) > 0) { free(line); continue; }
            int flag = atoi(f[1]);
            char *rn;
            if ((flag & 65) == 65) rn = str_cat(f[0], "/1");
            else if ((flag & 129) == 129) rn = str_cat(f[0], "/2");
            else rn = xstrdup(f[0]);
            CigVec c; vec_init(c);
            parse_cigar(f[5], &c);
            int read_p0 = atoi(f[3]);
            int read_p1 = read_align_end_pos(read_p0, &c);
            int relative_pos = read_p0 - p0;
            if (relative_pos < 0) relative_pos = 0;
            char *seq = NULL, *cigar = NULL;
            if (c.n == 0 || crop_read_within_window(p0, p1, f[9], f[10], &c, read_p0, read_p1, &seq, &cigar) != 0) {
                fprintf(stderr, "oracle: read %s cannot be cropped (the reference would throw here)\n", rn);
                exit(3);
            }
            int maxins = max_insert_size(cigar);
            if ((int)strlen(seq) > rl && maxins < max_ins) {
                if ((ld)mt_canonical(&gen) > rho) { free(seq); free(cigar); free(rn); vec_free(c); free(line); continue; }
                int lo = 0, hi = dups.n;
                while (lo < hi) { int mid = (lo + hi) / 2; if (dup_cmp(&dups.v[mid], relative_pos, cigar, seq) < 0) lo = mid + 1; else hi = mid; }
                if (lo < dups.n && dup_cmp(&dups.v[lo], relative_pos, cigar, seq) == 0) {
                    vec_push(dups.v[lo].names, rn);
                    free(seq); free(cigar);
                } else {
                    DupEnt e; e.pos = relative_pos; e.cigar = cigar; e.seq = seq; vec_init(e.names);
                    vec_push(e.names, rn);
                    vec_push(dups, e);      /* grow, then shift into place */
                    for (int k = dups.n - 1; k > lo; k--) dups.v[k] = dups.v[k - 1];
                    dups.v[lo] = e;
                }
            } else { free(seq); free(cigar); free(rn); }
            vec_free(c); free(line);
        }
        fclose(in);
    }
    /* :607-627 */
    VEC(NameUid) nu; vec_init(nu);
    for (int id = 0; id < dups.n; id++) {
        ARead ar = { dups.v[id].pos, dups.v[id].cigar, dups.v[id].seq, dups.v[id].names.n };
        vec_push(*reads, ar);
        for (int t = 0; t < dups.v[id].names.n; t++) { NameUid x = { dups.v[id].names.v[t], id }; vec_push(nu, x); }
    }
    /* tmp_read_uids[name] = id: one entry per distinct name, the later
     * assignment wins; iteration is in std::string order. */
    {
        int n = nu.n;
        NameIdx *ord = (NameIdx *)xmalloc(sizeof(NameIdx) * (size_t)(n ? n : 1));
        for (int i = 0; i < n; i++) { ord[i].name = nu.v[i].name; ord[i].idx = i; }
        qsort(ord, (size_t)n, sizeof(NameIdx), nameidx_cmp);
        VEC(NameUid) uniq; vec_init(uniq);
        for (int i = 0; i < n; i++) {
            if (i + 1 < n && strcmp(ord[i].name, ord[i + 1].name) == 0) continue;
            vec_push(uniq, nu.v[ord[i].idx]);
        }
        free(ord);
        rp->n = dups.n;
        rp->mates = (IntVec *)xmalloc(sizeof(IntVec) * (size_t)(dups.n ? dups.n : 1));
        for (int i = 0; i < dups.n; i++) vec_init(rp->mates[i]);
        /* :629-665 */
        for (int i = 0; i < uniq.n; i++) {
            const char *rn1 = uniq.v[i].name;
            int uid = uniq.v[i].uid;
            size_t l1 = strlen(rn1);
            int paired = 0;
            char *rn2 = NULL;
            if (l1 >= 2 && strcmp(rn1 + l1 - 2, "/1") == 0) { paired = 1; rn2 = xstrdup(rn1); rn2[l1 - 1] = '2'; }
            else if (l1 >= 2 && strcmp(rn1 + l1 - 2, "/2") == 0) { paired = 1; rn2 = xstrdup(rn1); rn2[l1 - 1] = '1'; }
            int mate = -1;
            if (paired) {
                NameUid key = { rn2, 0 };
                NameUid *hit = (NameUid *)bsearch(&key, uniq.v, (size_t)uniq.n, sizeof(NameUid), nameuid_cmp);
                if (hit) mate = hit->uid;
            }
            vec_push(rp->mates[uid], mate);
            free(rn2);
        }
        vec_free(uniq);
    }
    for (int id = 0; id < dups.n; id++) {
        for (int t = 0; t < dups.v[id].names.n; t++) free(dups.v[id].names.v[t]);
        vec_free(dups.v[id].names);
    }
    vec_free(dups); vec_free(nu);
    remove(tmp); free(tmp); free(cmd);
}

/* :673-783 */
static int window_adjust(const char *mapping_file, int mq, const char *gn, int p0, int p1, int z, int Lg, int *d0, int *d1) {
    int P, Q;
    if (p0 - z < 1) z = p0 - 1;
    P = p0 - z;
    Q = p1 + z;
    if (Q > Lg) Q = Lg;
    char *tmp = tmp_path();
    size_t L = strlen(mapping_file) + strlen(gn) + strlen(tmp) + 128;
    char *cmd = (char *)xmalloc(L);
    snprintf(cmd, L, "samtools mpileup -q %d -Q0  -A  -r %s:%d-%d %s 2>/dev/null 1>%s", mq, gn, P, Q, mapping_file, tmp);
    if (system(cmd)) {}
    /* map<int,(has_insert,has_delete)>: sorted unique by position, last write wins */
    typedef struct { int pos; int hi, hd; } MI;
    VEC(MI) mi; vec_init(mi);
    FILE *in = fopen(tmp, "r");
    if (in) {
        for (char *line; (line = read_line(in));) {
            char *f[6]; split_ws(line, f, 6);
            MI m = { atoi(f[1]), strchr(f[4], '+') != NULL, (strchr(f[4], '-') != NULL) || (strchr(f[4], '*') != NULL) };
            int lo = 0, hi = mi.n;
            while (lo < hi) { int mid = (lo + hi) / 2; if (mi.v[mid].pos < m.pos) lo = mid + 1; else hi = mid; }
            if (lo < mi.n && mi.v[lo].pos == m.pos) mi.v[lo] = m;
            else { vec_push(mi, m); for (int k = mi.n - 1; k > lo; k--) mi.v[k] = mi.v[k - 1]; mi.v[lo] = m; }
            free(line);
        }
        fclose(in);
    }
    remove(tmp); free(tmp); free(cmd);
    if (mi.n == 0) { vec_free(mi); return -1; }   /* reference dereferences begin() of an empty map: UB */
    int k0 = -1, k1 = -1;
    for (int k = 0; k < mi.n; k++) { if (mi.v[k].pos == p0) k0 = k; if (mi.v[k].pos == p1) k1 = k; }
    if (k0 < 0) P = mi.v[0].pos;
    else {
        P = p0;
        while (mi.v[k0].hi || mi.v[k0].hd) { --k0; if (k0 < 0) break; --P; }   /* --begin() == end() in libstdc++ */
    }
    if (k1 < 0) Q = mi.v[mi.n - 1].pos;
    else {
        Q = p1;
        while (mi.v[k1].hi || mi.v[k1].hd) { ++k1; if (k1 >= mi.n) break; ++Q; }
    }
    *d0 = p0 - P;
    *d1 = Q - p1;
    vec_free(mi);
    return 0;
}

typedef struct { char *gn; int p0, p1; } ScWindow;
typedef VEC(ScWindow) WindowVec;
/* :798-848 */
static void make_scan_window(ScParam *pa, WindowVec *windows) {
    int p0, p1, d0 = 0, d1 = 0, z = 50, l, L, LL;
    char *gn;
    if (pa->roi[0] == 0) {
        gn = fai_gene_name(pa->gene_file);
        l = 1; L = fai_gene_length(pa->gene_file, gn); LL = L;
    } else {
        gn = gene_roi_name(pa->roi);
        l = gene_roi_start_pos(pa->roi);
        L = gene_roi_end_pos(pa->roi);
        LL = fai_gene_length(pa->gene_file, gn);
    }
    IntVec visited; vec_init(visited);
    for (p0 = l, p1 = l; p1 < L; p0 += pa->window_size - pa->overlap_size) {
        p1 = p0 + pa->window_size - 1;
        if (p1 > L) p1 = L;
        if (window_adjust(pa->mapping_file, pa->mapping_qual, gn, p0, p1, z, LL, &d0, &d1) != 0) {
            fprintf(stderr, "oracle: no pileup for %s:%d-%d (undefined behaviour in the reference)\n", gn, p0, p1);
            exit(4);
        }
        if (p0 == l) pa->d0 = d0;
        int seen = 0;
        for (int k = 0; k < visited.n; k++) if (visited.v[k] == p1 + d1) seen = 1;
        if (seen) continue;
        ScWindow w = { xstrdup(gn), p0 - d0, p1 + d1 };
        vec_push(*windows, w);
        vec_push(visited, p1 + d1);
    }
    pa->d1 = d1;
    vec_free(visited); free(gn);
}
#endif

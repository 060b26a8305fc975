/* TEST INFRASTRUCTURE ONLY -- the ORACLE.
 *
 * A plain-C CPU restatement of the reference's StrainCall path
 * (/root/reference/StrainCall/), used by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg to CHECK the HIP
 * product path.  Nothing under rambl_amd/ imports, links or executes it.
 *
 * Parity status: PINNED.  The reference has no tests or golden vectors of its own
 * (SURVEY.md section 4); this restatement is pinned against outputs of the
 * reference itself run in the build container -- the shipped binary
 * /root/reference/StrainCall/StrainCall and oracle/_ref/StrainCall_ref built from
 * the reference sources by oracle/Makefile -- via the fixtures committed under
 * tests/golden/ (FASTA, -G graph dumps, per-level traces, MSA vectors) and the
 * differential test tests/test_oracle_vs_ref.py.
 *
 * Arithmetic: x87 long double exactly as the reference (DoubleL).  The command
 * line is the reference's (StrainCall.cpp:98-154); like the reference it shells
 * out to `samtools`.  Extra switches, for tests only, come from the environment:
 *   SC_TRACE=1 [SC_TRACE_PREC=n]  per-level strain/abundance trace on stderr
 *                                 (same text as the reference's dormant blocks,
 *                                 NonparametricClustering.cpp:287-298,460-471)
 */
#include "o_util.h"
#include "o_graph.h"
#include "o_msa.h"
#include "o_cluster.h"
#include "o_ingest.h"

static void help(void) {                                               /* StrainCall.cpp:34-55 */
    fputs("StrainCall marker_gene read_mapping\n"
          "           [-r gn:p0-p1] [-w window_size]\n"
          "           [-e error_rate] [-q map_qual]\n"
          "\n"
          "Options\n"
          "-r,--roi           region of interesting, gn is gene name,\n"
          "                   p0 is starting position, p1 is ending position (inclusive)\n"
          "-w,--window        the size of scanning window [500]\n"
          "-o,--overlap       the size of window-window overlap [100]\n"
          "-e,--error-rate    sequencing error rate [0.01]\n"
          "-D,--max-depth     downsample data to the specified depth [800]\n"
          "-q,--map-qual      only include reads with mapping quality >= INT [3]\n"
          "-I,--max-ins       only include reads with insertions <= INT [10]\n"
          "-l,--read-len      only include reads with length >=INT [80]\n"
          "-t,--tau           only include strains with abundance level >=FLT [0.02]\n"
          "-d,--diff-rate     only include strains with difference rate >=FLT [0.01]\n"
          "-G,--plot-graph    print graph\n"
          "-h,--help          print this message\n"
          "\n", stderr);
}

/* StrainCall.cpp:972-1059 */
int oracle_main(int argc, char **argv) {
    ScParam pa; sc_param_init(&pa);
    --argc; ++argv;
    sc_parse_cmd_line(argc, argv, &pa);
    if (pa.print_help || argc == 0) { help(); return 0; }
    WindowVec windows; vec_init(windows);
    make_scan_window(&pa, &windows);
    long total_draws = 0;
    for (int wi = 0; wi < windows.n; wi++) {
        ScWindow *w = &windows.v[wi];
        char roi[1024];
        snprintf(roi, sizeof roi, "%s:%d-%d", w->gn, w->p0, w->p1);
        char *gene_seq = load_gene_seq(pa.gene_file, roi);
        AReadVec reads; vec_init(reads);
        ReadPairs rp; rp.mates = NULL; rp.n = 0;
        load_mapping_reads(gene_seq, pa.mapping_file, pa.mapping_qual, pa.read_len, pa.max_ins, pa.max_depth, roi, &reads, &rp);
        if (getenv("SC_ORACLE_DUMP_READS")) {
            /* test fixture: what crosses from ingest (a1-a4) into the graph stage */
            char path[4096];
            snprintf(path, sizeof path, "%s.%d", getenv("SC_ORACLE_DUMP_READS"), wi);
            FILE *df = fopen(path, "w");
            if (df) {
                fprintf(df, "WINDOW\t%s\t%d\t%d\n", w->gn, w->p0, w->p1);
                fprintf(df, "REF\t%s\n", gene_seq);
                for (int i = 0; i < reads.n; i++) {
                    fprintf(df, "READ\t%d\t%s\t%s\t%d\t", reads.v[i].pos, reads.v[i].cigar, reads.v[i].seq, reads.v[i].cn);
                    for (int k = 0; k < rp.mates[i].n; k++) fprintf(df, "%s%d", k ? "," : "", rp.mates[i].v[k]);
                    fprintf(df, "\n");
                }
                fclose(df);
            }
        }
        if (reads.n == 0) { free(gene_seq); continue; }
        Graph *g = graph_build(gene_seq, reads.v, reads.n);
        if (!pa.plot_graph) {
            ClusterCtx cx; cx.g = g; cx.rp = &rp; cx.draws = 0;
            cx.trace = getenv("SC_TRACE") != NULL;
            cx.trace_prec = getenv("SC_TRACE_PREC") ? atoi(getenv("SC_TRACE_PREC")) : 6;
            cx.trace_fp = stderr;
            StrainVec strains; vec_init(strains);
            streaming_clustering(&cx, &strains, 5000, (ld)pa.error_rate, (ld)pa.tau, (ld)pa.diff_rate, reads.n);
            read_assign(&cx, &strains, reads.v, reads.n, 5000);
            sort_strains(&strains);                                    /* :1027 */
            char *gene_seq0 = load_gene_seq(pa.gene_file, w->gn);      /* :1031 (result unused) */
            free(gene_seq0);
            for (int si = 0; si < strains.n; si++) {
                if (strains.v[si].abundance >= (ld)pa.tau) {
                    char *out = strain_plain_seq(&strains.v[si]);
                    printf(">contig%s%d%d%d\n%s\n", w->gn, w->p0, w->p1, si, out);
                    free(out);
                }
            }
            for (int si = 0; si < strains.n; si++) strain_free(&strains.v[si]);
            vec_free(strains);
            total_draws += cx.draws;
        } else {
            output_edge(g, stdout);
        }
        graph_free(g);
        for (int i = 0; i < reads.n; i++) { free(reads.v[i].cigar); free(reads.v[i].seq); }
        vec_free(reads);
        for (int i = 0; i < rp.n; i++) vec_free(rp.mates[i]);
        free(rp.mates);
        free(gene_seq);
    }
    if (getenv("SC_ORACLE_STATS")) fprintf(stderr, "oracle_draws %ld\n", total_draws);
    fflush(stdout);
    return 0;
}

/* ---- entry points for ctypes-driven unit tests (liboracle.so) -------------- */
/* Sum-of-pairs MSA of `n` NUL-terminated sequences (row a7).  `out` receives the
 * n padded rows, each `*ncol` chars + NUL, packed back to back. Returns 0. */
int oracle_msa_align(const char **seqs, int n, char *out, int out_cap, int *ncol) {
    char **rows = NULL;
    int nc = msa_sp_align((char **)seqs, n, &rows);
    *ncol = nc;
    int rc = 0;
    if ((long)n * (nc + 1) > out_cap) rc = -1;
    for (int t = 0; t < n; t++) {
        if (!rc) memcpy(out + (long)t * (nc + 1), rows[t], (size_t)nc + 1);
        free(rows[t]);
    }
    free(rows);
    return rc;
}
/* libstdc++ std::sort permutation of n keys, descending (`a > b` comparator). */
static int key_gt(void *ctx, int a, int b) { const double *k = (const double *)ctx; return k[a] > k[b]; }
void oracle_sort_desc_perm(const double *keys, int n, int *perm) {
    for (int i = 0; i < n; i++) perm[i] = i;
    std_sort_perm(perm, n, key_gt, (void *)keys);
}
/* first `n` outputs of generate_canonical<double,53>(mt19937(seed)) */
void oracle_mt_canonical(unsigned seed, int n, double *out) {
    MT m; mt_seed(&m, seed);
    for (int i = 0; i < n; i++) out[i] = mt_canonical(&m);
}

#ifndef ORACLE_NO_MAIN
int main(int argc, char **argv) { return oracle_main(argc, argv); }
#endif

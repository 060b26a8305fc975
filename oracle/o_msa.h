/* TEST INFRASTRUCTURE ONLY (oracle).  Literal CPU restatement of the progressive
 * sum-of-pairs MSA, /root/reference/StrainCall/MultipleSequenceAlignmentSP.cpp:10-301
 * with MSA<>::get (MultipleSequenceAlignment.hpp:59-70) and the SimpleDnaScore
 * table (SimpleDnaScore.cpp:15-42, Score.hpp:33-46) -- SURVEY.md rows a7, a8.
 * `s` starts at 0 (behaviour of the shipped binary). */
#ifndef O_MSA_H
#define O_MSA_H
#include "o_util.h"

/* Score.hpp:35 defaults: match 3, mismatch -5, gap_open -4, gap_extend -2.
 * operator[] on a missing key inserts 0 (std::map), SimpleDnaScore.cpp:10-13. */
static double dna_score(char x, char y) {
    static const char alpha[] = "AaCcGgTt+-";
    if (!x || !y || !strchr(alpha, x) || !strchr(alpha, y)) return 0.0;
    const double match = 3, mismatch = -5, gap_open = -4, gap_extend = -2;
    if (x == y) return match;
    if ((x == 'A' && y == 'a') || (x == 'a' && y == 'A')) return match;
    if ((x == 'C' && y == 'c') || (x == 'c' && y == 'C')) return match;
    if ((x == 'G' && y == 'g') || (x == 'g' && y == 'G')) return match;
    if ((x == 'T' && y == 't') || (x == 't' && y == 'T')) return match;
    if ((x == '+' && y == '-') || (x == '-' && y == '+')) return match;
    if (x == '+' || y == '+') return gap_open + gap_extend;
    if (x == '-' || y == '-') return gap_extend;
    return mismatch;
}

/* The MSA is a list of columns; column c holds `s` characters (one per row). */
typedef struct { char **col; int ncol; int s; } MsaCols;

/* MultipleSequenceAlignmentSP.cpp:52-249 */
static void msa_forward(const char *seq, MsaCols *msa, double *SC, int *PI, int *SI, int *SJ, int *PP,
                        int m, int n, int s) {
    const int mat = 0, ins = 1, del = 2;
    int i, j, k;
    double sp;
    SC[0] = 0; PI[0] = mat; SI[0] = 0; SJ[0] = 0;
    for (k = 0; k < s; k++) PP[k] = mat;
    for (i = 0, j = 1; j < n; ++j) {
        sp = 0;
        if (j == 1) for (k = 0; k < s; ++k) sp += dna_score('A', '+');
        else        for (k = 0; k < s; ++k) sp += dna_score('A', '-');
        sp += SC[i * n + j - 1];
        SC[i * n + j] = sp; PI[i * n + j] = ins; SI[i * n + j] = 0; SJ[i * n + j] = -1;
        for (k = 0; k < s; ++k) PP[i * n * s + j * s + k] = ins;
    }
    for (i = 1, j = 0; i < m; ++i) {
        const char *c1 = msa->col[i - 1];
        sp = 0;
        if (i == 1) for (k = 0; k < s; ++k) sp += dna_score(c1[k], '+');
        else        for (k = 0; k < s; ++k) sp += dna_score(c1[k], '-');
        sp += SC[(i - 1) * n + j];
        SC[i * n + j] = sp; PI[i * n + j] = del; SI[i * n + j] = -1; SJ[i * n + j] = 0;
        for (k = 0; k < s; ++k) PP[i * n * s + j * s + k] = (c1[k] == '-') ? mat : del;
    }
    for (i = 1; i < m; ++i) {
        const char *c1 = msa->col[i - 1];
        for (j = 1; j < n; ++j) {
            char b = seq[j - 1];
            double r1 = 0, r2 = 0, r3 = 0;
            for (k = 0; k < s; ++k) {
                if (c1[k] == '-') {
                    if (PP[(i - 1) * n * s + (j - 1) * s + k] == ins) r1 += dna_score('-', b);
                    else r1 += dna_score('+', b);
                } else r1 += dna_score(c1[k], b);
            }
            r1 += SC[(i - 1) * n + (j - 1)];
            for (k = 0; k < s; ++k) {
                if (PP[i * n * s + (j - 1) * s + k] == ins) r2 += dna_score('-', b);
                else r2 += dna_score('+', b);
            }
            r2 += SC[i * n + (j - 1)];
            for (k = 0; k < s; ++k) {
                if (c1[k] != '-') {
                    if (PP[(i - 1) * n * s + j * s + k] == del) r3 += dna_score(c1[k], '-');
                    else r3 += dna_score(c1[k], '+');
                } else r3 += dna_score(c1[k], '-');
            }
            r3 += SC[(i - 1) * n + j];
            /* :202-245: `it3` is never advanced in the PP-writing loops, so every
             * row copies the gap state of row 0. */
            char c0 = c1[0];
            if (r1 >= r2 && r1 >= r3) {
                SC[i * n + j] = r1; PI[i * n + j] = mat; SI[i * n + j] = -1; SJ[i * n + j] = -1;
                for (k = 0; k < s; ++k) PP[i * n * s + j * s + k] = (c0 == '-') ? ins : mat;
            } else if (r2 >= r1 && r2 >= r3) {
                SC[i * n + j] = r2; PI[i * n + j] = ins; SI[i * n + j] = 0; SJ[i * n + j] = -1;
                for (k = 0; k < s; ++k) PP[i * n * s + j * s + k] = ins;
            } else {
                SC[i * n + j] = r3; PI[i * n + j] = del; SI[i * n + j] = -1; SJ[i * n + j] = 0;
                for (k = 0; k < s; ++k) PP[i * n * s + j * s + k] = (c0 == '-') ? mat : del;
            }
        }
    }
}

/* MultipleSequenceAlignmentSP.cpp:252-301 */
static void msa_backward(const char *seq, MsaCols *msa, int *SI, int *SJ, int m, int n, int s) {
    int cap = m + n + 2, cnt = 0;
    char **rev = (char **)xmalloc(sizeof(char *) * (size_t)cap);
    int r1 = msa->ncol - 1;      /* rit1 */
    int r2 = n - 2;              /* rit2 : seq has n-1 chars */
    int x, y;
    for (x = m - 1; x >= 0;) {
        for (y = n - 1; y >= 0;) {
            if (x == 0 && y == 0) { x -= 1; y -= 1; continue; }
            int i = SI[x * n + y], j = SJ[x * n + y];
            char *tmp = (char *)xmalloc((size_t)s + 2);
            if (i == -1 && j == -1) {
                memcpy(tmp, msa->col[r1], (size_t)s); tmp[s] = seq[r2]; --r1; --r2;
            } else if (i == 0 && j == -1) {
                memset(tmp, '-', (size_t)s); tmp[s] = seq[r2]; --r2;
            } else {
                memcpy(tmp, msa->col[r1], (size_t)s); tmp[s] = '-'; --r1;
            }
            tmp[s + 1] = 0;
            if (cnt == cap) { cap *= 2; rev = (char **)xrealloc(rev, sizeof(char *) * (size_t)cap); }
            rev[cnt++] = tmp;
            x += i; y += j;
        }
    }
    for (int c = 0; c < msa->ncol; c++) free(msa->col[c]);
    free(msa->col);
    msa->col = (char **)xmalloc(sizeof(char *) * (size_t)cnt);
    for (int c = 0; c < cnt; c++) msa->col[c] = rev[cnt - 1 - c];
    msa->ncol = cnt;
    msa->s = s + 1;
    free(rev);
}

/* MultipleSequenceAlignmentSP.cpp:10-49 + MSA<>::get.  Returns #columns and
 * rows_out[t] = row t over all columns. */
static int msa_sp_align(char **seqs, int nseq, char ***rows_out) {
    MsaCols msa;
    int l0 = (int)strlen(seqs[0]);
    msa.col = (char **)xmalloc(sizeof(char *) * (size_t)(l0 ? l0 : 1));
    msa.ncol = l0; msa.s = 1;
    for (int c = 0; c < l0; c++) { msa.col[c] = (char *)xmalloc(2); msa.col[c][0] = seqs[0][c]; msa.col[c][1] = 0; }
    int s = 1;
    for (int t = 1; t < nseq; ++t, ++s) {
        int m = msa.ncol + 1, n = (int)strlen(seqs[t]) + 1;
        double *SC = (double *)xmalloc(sizeof(double) * (size_t)(m * n));
        int *PI = (int *)xmalloc(sizeof(int) * (size_t)(m * n));
        int *SI = (int *)xmalloc(sizeof(int) * (size_t)(m * n));
        int *SJ = (int *)xmalloc(sizeof(int) * (size_t)(m * n));
        int *PP = (int *)xmalloc(sizeof(int) * (size_t)(m * n) * (size_t)s);
        msa_forward(seqs[t], &msa, SC, PI, SI, SJ, PP, m, n, s);
        msa_backward(seqs[t], &msa, SI, SJ, m, n, s);
        free(SC); free(PI); free(SI); free(SJ); free(PP);
    }
    char **rows = (char **)xmalloc(sizeof(char *) * (size_t)nseq);
    for (int t = 0; t < nseq; t++) {
        rows[t] = (char *)xmalloc((size_t)msa.ncol + 1);
        for (int c = 0; c < msa.ncol; c++) rows[t][c] = msa.col[c][t];
        rows[t][msa.ncol] = 0;
    }
    int ncol = msa.ncol;
    for (int c = 0; c < msa.ncol; c++) free(msa.col[c]);
    free(msa.col);
    *rows_out = rows;
    return ncol;
}
#endif

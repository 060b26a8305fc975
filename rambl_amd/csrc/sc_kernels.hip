// HIP kernels of the StrainCall path for gfx950 (MI355X, CDNA4, wave64).
//
//   k_thread_*       threading of the reads along the backbone                 (row a5)
//   k_msa            progressive sum-of-pairs MSA of insertion strings          (a7, a8)
//   k_edge_support   number_of_reads_cover_nodes for every edge                (a16)
//   k_level_sample   one sampler level in one launch: rows of new strains, read
//                    log-likelihood update (a13), draw slots and weight rows, and
//                    the Polya-urn chain (a14, np_bayes_clustering; also a18,
//                    read_assign)
//   k_level          one level without the sampler: the update (a13) and the soft
//                    update (a15, hard_clustering)
//   k_level_copy/update   the first two pieces on a grid, for very large levels
//
// These are integer / fp32 / fp64 loops bound by latency or HBM: no MFMA.  The
// sampler is one dependent chain per region; four or eight wavefronts speculate
// over a window of draws and prove every accepted decision equal to the
// sequential one, near-ties go to an fp64 scan and then to a literal evaluation of
// the reference's formula, so every draw equals the reference's draw.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "sc_device.hpp"

namespace sc {

#define SC_GLOBAL __attribute__((address_space(1)))
#define SC_LDS __attribute__((address_space(3)))

constexpr int LDS_TOTAL = 160 * 1024 - 256;   // dynamic part; the rest covers small static __shared__ variables
constexpr int LDS_SMALL = 18 * 1024;           // per-strain scalars (LevelLds)
constexpr int LDS_BIG = LDS_TOTAL - LDS_SMALL;
static_assert(LDS_SMALL >= (int)(sizeof(double) * 2 * MAXS + sizeof(StrainParam) * MAXS + sizeof(unsigned) * MAXS * KMAX + sizeof(int) * 6 * MAXS + 64), "LDS_SMALL");
static_assert(LDS_BIG >= (int)(sizeof(double) * MAXS * 64), "LDS_BIG");    // every level with <= 8 symbols stages its log tables in LDS

// --------------------------------------------------------------------------
// wave64 helpers
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, BOUND);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, BOUND);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes, in lane order
__device__ __forceinline__ double wave_scan_incl(double v) {
    v += dpp_f64<0x111, 0xF, 0xF, true>(v);   // row_shr:1
    v += dpp_f64<0x112, 0xF, 0xF, true>(v);   // row_shr:2
    v += dpp_f64<0x114, 0xF, 0xF, true>(v);   // row_shr:4
    v += dpp_f64<0x118, 0xF, 0xF, true>(v);   // row_shr:8
    v += dpp_f64<0x142, 0xA, 0xF, false>(v);  // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xC, 0xF, false>(v);  // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ double wave_shr1(double v) {   // lane i gets lane i-1, lane 0 gets 0
    return dpp_f64<0x138, 0xF, 0xF, true>(v);             // wave_shr:1
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// --------------------------------------------------------------------------
// a16: support of edge e = (u -> v): PartialOrderGraph.cpp:1218-1244.
// One wavefront per edge; pools are rid-sorted (checked on the host, `sorted`),
// so the multiplicity of a read in u's pool comes from two binary searches.
__global__ __launch_bounds__(256) void k_edge_support(const int* __restrict__ out_ptr, const int* __restrict__ out_node,
                                                      const int* __restrict__ pool_ptr, const int* __restrict__ pool_rid,
                                                      const int* __restrict__ pool_cn, const uint8_t* __restrict__ node_is_end,
                                                      const int* __restrict__ edge_src, int n_edges, int sorted,
                                                      int* __restrict__ support) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int e = wave; e < n_edges; e += nwaves) {
        const int u = edge_src[e], v = out_node[e];
        const int ub = pool_ptr[u], ue = pool_ptr[u + 1], vb = pool_ptr[v], ve = pool_ptr[v + 1];
        long acc = 0;
        if (u == 0) {
            for (int j = vb + lane; j < ve; j += 64) acc += pool_cn[j];
        } else if (node_is_end[v]) {
            for (int i = ub + lane; i < ue; i += 64) acc += pool_cn[i];
        } else {
            for (int j = vb + lane; j < ve; j += 64) {
                const int rid = pool_rid[j];
                int mult = 0;
                if (sorted) {
                    int lo = ub, hi = ue;                 // lower_bound
                    while (lo < hi) { int mid = (lo + hi) >> 1; if (pool_rid[mid] < rid) lo = mid + 1; else hi = mid; }
                    int lo2 = lo, hi2 = ue;               // upper_bound
                    while (lo2 < hi2) { int mid = (lo2 + hi2) >> 1; if (pool_rid[mid] <= rid) lo2 = mid + 1; else hi2 = mid; }
                    mult = lo2 - lo;
                } else {
                    for (int i = ub; i < ue; i++) mult += (pool_rid[i] == rid);
                }
                acc += (long)mult * pool_cn[j];
            }
        }
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) support[e] = (int)acc;
    }
    (void)out_ptr;
}

// --------------------------------------------------------------------------
// Literal fp64 evaluation of one categorical draw, exactly as the reference
// forms it (NonparametricClustering.cpp:171-194 with libstdc++'s
// discrete_distribution): executed by lane 0 of the sampling wave for the rare
// draw whose uniform lies within the safety margin of a boundary.
struct SlowArgs {          // the few JobDev fields the rare tiers need, passed by value
    const int* qent; const int* quid; const int* ent_rid; const double* ll; long ll_stride; const uint8_t* has;
};
__device__ int exact_draw(const SlowArgs job, const int* s_slot, const volatile double* s_a,
                          volatile double* s_p, int S, int rid, int uid, double u) {
    double Z = 0;
    for (int s = 0; s < S; s++) Z += s_a[s];
    for (int s = 0; s < S; s++) {
        double p = s_a[s] / Z;
        const double* row = job.ll + (long)s_slot[s] * job.ll_stride;
        p = log(p) + (job.has[rid] ? row[rid] : 0.0);
        if (uid >= 0 && job.has[uid]) p += row[uid];
        s_p[s] = exp(p);
    }
    if (S < 2) return 0;
    double sum = 0;
    for (int s = 0; s < S; s++) sum += s_p[s];
    double acc = 0;
    for (int s = 0; s < S; s++) {
        double pr = s_p[s] / sum;
        acc = (s == 0) ? pr : acc + pr;
        s_p[s] = acc;
    }
    s_p[S - 1] = 1.0;
    int lo = 0, len = S;     // std::lower_bound
    while (len > 0) {
        int half = len >> 1, mid = lo + half;
        if (s_p[mid] < u) { lo = mid + 1; len = len - half - 1; }
        else len = half;
    }
    return lo;
}

constexpr double DRAW_EPS64 = 1e-10;  // margin (relative to the total weight) of the fp64 scan tier

// Tier 2 and 3 of one draw: fp64 weights and scan with a 1e-10 margin; if the
// uniform is still within the margin of a boundary (or the slot's log-likelihoods
// lie in the underflow range of the reference's exp), the literal evaluation.
// The slot's log-likelihoods are re-read from the rows (the table kept only their
// fp32 weights).  Wave-uniform call.
template <int NPL>
__device__ __noinline__ int slow_draw(const SlowArgs job, const int* s_slot, volatile double* s_a, volatile double* s_p,
                                      double a0, double a1, int S, int q, int e0, double u, int lane) {
    const double a[2] = {a0, a1};
    const int rid = job.ent_rid[e0 + job.qent[q]], uid = job.quid[q];
    const bool hr = job.has[rid] != 0, hu = uid >= 0 && job.has[uid] != 0;
    double x[NPL], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NPL; i++) {
        const int s = lane * NPL + i;
        x[i] = -INFINITY;
        if (s < S) {
            const double* row = job.ll + (long)s_slot[s] * job.ll_stride;
            double v = hr ? row[rid] : 0.0;
            if (hu) v += row[uid];
            x[i] = v;
            m = fmax(m, v);
        }
    }
    for (int d = 1; d < 64; d <<= 1) m = fmax(m, __shfl_xor(m, d));
    const bool flag = !(m >= -600.0);                     // underflow range of the reference's exp(); also NaN / -inf
    double w[NPL], pair = 0;
#pragma unroll
    for (int i = 0; i < NPL; i++) {
        const int s = lane * NPL + i;
        // fp64 weight a_s * exp(loglik - max)
        w[i] = (s < S) ? a[i] * exp(x[i] - m) : 0.0;
        pair += w[i];
    }
    const double incl = wave_scan_incl(pair);
    const double T = readlane_f64(incl, 63);
    const double tgt = u * T;
    const double lo = tgt - DRAW_EPS64 * T, hi = tgt + DRAW_EPS64 * T;
    bool ok = (T > 0.0) && (T < 1.0e300) && !flag;
    int c;
    if (NPL == 1) {
        const unsigned long long mlo = __ballot(incl >= lo), mhi = __ballot(incl >= hi);
        ok = ok && (mlo == mhi) && (mlo != 0ull);
        c = ok ? (int)__builtin_ctzll(mlo) : 0;
    } else {
        const double E = wave_shr1(incl);
        const double c0 = E + w[0], c1 = E + pair;
        const unsigned long long m0lo = __ballot(c0 >= lo), m0hi = __ballot(c0 >= hi);
        const unsigned long long m1lo = __ballot(c1 >= lo), m1hi = __ballot(c1 >= hi);
        ok = ok && (m0lo == m0hi) && (m1lo == m1hi) && (m1lo != 0ull);
        const int l1 = ok ? (int)__builtin_ctzll(m1lo) : 0;
        c = 2 * l1 + (((m0lo >> l1) & 1ull) ? 0 : 1);
    }
    if (!ok) {
#pragma unroll
        for (int i = 0; i < NPL; i++) { int s = lane * NPL + i; if (s < S) s_a[s] = a[i]; }
        __builtin_amdgcn_wave_barrier();
        int cc = 0;
        if (lane == 0) cc = exact_draw(job, s_slot, s_a, s_p, S, rid, uid, u);
        c = __builtin_amdgcn_readfirstlane(cc);
        __builtin_amdgcn_wave_barrier();
        c |= 0x100;                                    // tell the caller the literal tier ran
    }
    return c;
}

// --------------------------------------------------------------------------
// Wide urn chain: a sliding window of 64 draws over the four wavefronts of the
// workgroup (one per SIMD), four lanes per draw (128 draws on eight wavefronts
// while a lane owns at most 8 strains).
//
// Draw t+p of a pass (p = 0..63) belongs to the quad of lanes 4*(p%16)..+3 of
// wave p/16; lane k of the quad walks its quarter of the strains in order with
// the counts as they are in front of draw t (uniform over the draws), one FMA
// per strain: cum_s = sum_{s'<=s} (a0_s' + k_s') * L[q][s'].  The quarters are
// joined inside the quad by DPP (totals -> offsets and T, then the number of
// boundaries below u*T and the distances to the nearest boundary on either side).
//
// Why a speculative decision is final.  Let d_s = cum_s - u*T.  Each of the p
// draws in front of draw t+p adds one to one count c_j, which adds L[c_j] <= 1
// to T and to every cum_s with s >= c_j: d_s moves by L[c_j]*([c_j <= s] - u),
// i.e. up by at most (1-u) and down by at most u per earlier draw.  So a boundary
// below the target (d_s < 0) stays below it if -d_s > (1-u)*p and one at or above
// it (d_s >= 0) stays there if d_s > u*p -- whatever the earlier draws of the
// window turn out to be.  The test adds eps*T on both sides for the fp32 error of
// the chains and sums (< (2*S + 9) * 2^-24 relative to T).  Each pass accepts the
// draws in front of the first one that fails the test (one LDS integer atomic per
// accepted draw; the waves exchange the position through LDS) and the window
// moves on to that draw, which then has p = 0 and margin eps*T only.  A draw that
// fails at p = 0 is within the fp32 error bound of a boundary (or its row is NaN:
// flagged slot): wave 0 sends it through the fp64 scan and, if needed, the literal
// evaluation.
//
// Weight rows are row-major [Q][stride] fp32, stride = 4 * odd.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fma_rn(float a, float b, float c) {       // three-address FMA (no v_fmac + copy)
    float d;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f2v pk_sub(f2v a, f2v b) {
    f2v d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// min over the raw bits as unsigned integers: among floats, the smallest non-negative one (a negative float has
// the sign bit set and compares above every non-negative float)
__device__ __forceinline__ unsigned min3_u32(unsigned m, unsigned x, unsigned y) {
    unsigned d;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(m), "v"(x), "v"(y));
    return d;
}
template <int QP> __device__ __forceinline__ float quad_f32(float v) {     // quad_perm DPP
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), QP, 0xF, 0xF, true));
}
template <int QP> __device__ __forceinline__ int quad_i32(int v) {
    return __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, true);
}
// LDS-only workgroup barrier: does not wait for outstanding global loads / stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int UWIN = 1024;     // uniforms staged in LDS (fp32), refilled in halves
#ifdef SC_CHAIN_PROF
// experiment builds only (make EXTRA=-DSC_CHAIN_PROF): where a pass of the chain spends its cycles, summed over every pass
// of wave 0: [0] counts read + chains, [1] test + s_x write, [2] first barrier + advance, [3] commits, [4] second barrier,
// [5] passes
__device__ unsigned long long g_chain_prof[12];
#define CHAIN_STAMP(k) do { if (wv == 0) { const unsigned long long t_ = clock64(); prof[k] += t_ - tprev; tprev = t_; } } while (0)
#else
#define CHAIN_STAMP(k) do {} while (0)
#endif

template <int NQ, bool ROWS_LDS, int NW, class JD>
__device__ __forceinline__ void urn_chain_q(const JD& job, const LevelHdr& h, const StrainParam* s_sp, LevelResult* __restrict__ R,
                                            const int* s_slot, volatile double* s_a, volatile double* s_p, unsigned* s_kf,
                                            const float* s_a0f, unsigned* s_cnt, int* s_x, float* s_uwin, const float* rows_lds, int stride, int tid) {
    constexpr int SPL = 4 * NQ, SP = 16 * NQ;               // strains per lane, capacity
    constexpr int NPLC = SP > 64 ? 2 : 1;                  // strains per lane in the checked tier
    constexpr float EPSW = (float)(SP + 16) * 1.5e-7f;     // (2*S + 9) * 2^-24 for the chains and sums + 3 * 2^-24 for the weights (exp_weight)
    // (the wavefront's index as a scalar: what only wavefront 0 does -- the uniforms, the checked tiers -- then costs the
    // others a scalar branch instead of a walk through masked-off vector code)
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), k = lane & 3, pos = wv * 16 + (lane >> 2);
    const int S = h.S, Q = h.Q, n = h.n_sweeps, e0 = h.e0;
    const int Sm1 = S - 1;
    const int total = n * Q;
    const SlowArgs sa{job.qent, job.quid, job.ent_rid, job.ll, job.ll_stride, job.has};
    const SC_GLOBAL double* Ustream = (const SC_GLOBAL double*)job.U;
    const SC_GLOBAL float* Uf = (const SC_GLOBAL float*)job.Uf;
    const SC_GLOBAL float* rows_g = (const SC_GLOBAL float*)job.tabLf;
    auto ld4 = [&](int idx) __attribute__((always_inline)) -> f4v {
        return ROWS_LDS ? *(const f4v*)(rows_lds + idx) : *(const SC_GLOBAL f4v*)(rows_g + idx);
    };
    auto ld1 = [&](int idx) __attribute__((always_inline)) -> float { return ROWS_LDS ? rows_lds[idx] : rows_g[idx]; };

    const int cbase = k * SPL;                              // first strain of this lane's quarter
    const float m0 = k > 0 ? 1.0f : 0.0f, m1 = k > 1 ? 1.0f : 0.0f;
    const float posf = (float)pos + 1.0e-37f;
    double a0m[NPLC];                                       // wave 0, checked tier: strains across the lanes
#pragma unroll
    for (int i = 0; i < NPLC; i++) { const int s = lane * NPLC + i; a0m[i] = (s < S) ? s_sp[s].a0 : 0.0; }
    f4v a0q[NQ];
#pragma unroll
    for (int g = 0; g < NQ; g++) a0q[g] = *(const f4v*)(s_a0f + cbase + 4 * g);
    unsigned long long n_exact = 0, n_slow = 0, n_pass = 0;
    const unsigned long long clk0 = clock64(), wall0 = wall_clock64();

    // uniforms: draws [ulo, ulo + UWIN) live in s_uwin[p & (UWIN-1)]; wave 0 refills
    int ulo = 0;
    bool upf = false;
    f4v ux0 = 0.0f, ux1 = 0.0f;
    if (wv == 0) {
#pragma unroll
        for (int j = 0; j < UWIN / 256; j++) *(f4v*)(s_uwin + 256 * j + 4 * lane) = *(const SC_GLOBAL f4v*)(Uf + 256 * j + 4 * lane);
    }
    __syncthreads();

    int t = 0;
    int ro = (pos % Q) * stride;                            // row offset (floats) of this lane's draw
    const int wrap = Q * stride;
    int upos = pos;                                        // (t + pos) & (UWIN - 1)
    f4v L[NQ];
    float uf = 0.0f, llast = 0.0f;
    int sym = 0;                                           // read symbol of this lane's draw (kept behind the row)
    auto issue_loads = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < NQ; g++) L[g] = ld4(ro + cbase + 4 * g);
        llast = ld1(ro + Sm1);
        sym = __float_as_int(ld1(ro + S));
        uf = s_uwin[upos];
    };
    if (total > 0) issue_loads();
#ifdef SC_CHAIN_PROF
    unsigned long long prof[5] = {0, 0, 0, 0, 0}, tprev = clock64();
#endif
#pragma unroll 1
    while (t < total) {
        asm volatile("" ::: "memory");                      // s_kf below must be re-read
        n_pass++;
        // this lane's quarter: cumulative weights with the counts in front of draw t
        float loc[SPL];
        float run = 0.0f;
        u4v kq[NQ];
#pragma unroll
        for (int g = 0; g < NQ; g++) kq[g] = *(const u4v*)(s_kf + cbase + 4 * g);
        const unsigned klast = s_kf[Sm1];                       // (read with the others: one LDS round trip, not two)
        const float a0last = s_a0f[Sm1];
#pragma unroll
        for (int g = 0; g < NQ; g++) {
            const u4v kk = kq[g];
            const f4v av = a0q[g] + f4v{(float)kk.x, (float)kk.y, (float)kk.z, (float)kk.w};
            loc[4 * g + 0] = run = (g == 0) ? av.x * L[g].x : fma_rn(av.x, L[g].x, run);
            loc[4 * g + 1] = run = fma_rn(av.y, L[g].y, run);
            loc[4 * g + 2] = run = fma_rn(av.z, L[g].z, run);
            loc[4 * g + 3] = run = fma_rn(av.w, L[g].w, run);
        }
        const float alast = a0last + (float)klast;
        CHAIN_STAMP(0);
        // quad: offset of this quarter and the total weight (bitwise the same in the four lanes)
        const float i1 = fmaf(quad_f32<0x90>(run), m0, run);       // + previous lane of the quad   ([0,0,1,2])
        const float i2 = fmaf(quad_f32<0x40>(i1), m1, i1);         // + two lanes back              ([0,0,0,1])
        const float off = i2 - run;
        const float T = quad_f32<0xFF>(i2);
        // position of u*T among the boundaries and the distance to the nearest one on either side
        const float tgt = uf * T;
        const float tb = tgt - off;
        const f2v tb2 = {tb, tb};
        unsigned w = 0;
        unsigned up = 0x7f800000u, dn = 0x7f800000u;        // +inf: nearest boundary at / above and below the target
#pragma unroll
        for (int j = 0; j < SPL; j += 2) {
            const f2v lc = {loc[j], loc[j + 1]};
            const f2v d = pk_sub(lc, tb2);                   // cum - target
            const f2v e = pk_sub(tb2, lc);                   // target - cum
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(d.x), 31);   // (w << 1) | sign(d)
            w = __builtin_amdgcn_alignbit(w, __float_as_uint(d.y), 31);
            up = min3_u32(up, __float_as_uint(d.x), __float_as_uint(d.y));
            dn = min3_u32(dn, __float_as_uint(e.x), __float_as_uint(e.y));
        }
        // Strains >= S-1 and the padding all sit at cum = T >= u*T: they are no boundaries.  They never count (a
        // padding entry can round to a tiny negative difference: the count is clamped, and that draw fails the
        // test below), and their distance T - u*T must not fail a draw whose target lies above the last real
        // boundary cum_{S-2}: alt = u*T - cum_{S-2} is then positive and IS the distance to the nearest real
        // boundary; once it clears the lower margin there is nothing above the target to test.
        const float alt = tgt - (T - alast * llast);
        const float epsT = EPSW * T;
        const float lim_up = fmaf(uf, posf, epsT);           // a boundary at / above the target moves down by <= u per earlier draw
        const float lim_dn = fmaf(1.0f - uf, posf, epsT);    // one below it moves up by <= 1 - u
        // NaN (flagged slot) and T == 0 fail the test
        const bool okl = (__uint_as_float(dn) >= lim_dn) && ((__uint_as_float(up) >= lim_up) || (alt >= lim_dn));
        const unsigned long long F = ~__ballot(okl);
        const int fpos = F ? 16 * wv + ((int)__builtin_ctzll(F) >> 2) : 16 * NW;
        if (lane == 0) s_x[wv] = fpos;
        int c = __popc(w);
        c += quad_i32<0xB1>(c);
        c += quad_i32<0x4E>(c);
        c = min(c, Sm1);
        CHAIN_STAMP(1);
        lds_barrier();
        const int rem = total - t;
        int adv = 16 * NW;
#pragma unroll
        for (int j = 0; j < NW / 4; j++) {
            const i4v xf = *(const i4v*)(s_x + 4 * j);
            adv = min(adv, min(min(xf.x, xf.y), min(xf.z, xf.w)));
        }
        adv = adv < rem ? adv : rem;
        CHAIN_STAMP(2);
        if (adv == 0) {
            // draw t itself: fp64 scan with the exact counts, then the literal tier (wave 0)
            if (wv == 0) {
                const double u = Ustream[t];
                double ad[NPLC];
#pragma unroll
                for (int i = 0; i < NPLC; i++) { const int s = lane * NPLC + i; ad[i] = a0m[i] + (double)((s < S) ? s_kf[s] : 0u); }
                const int qi = __builtin_amdgcn_readfirstlane(ro) / stride;
                const int cc = slow_draw<NPLC>(sa, s_slot, s_a, s_p, ad[0], NPLC > 1 ? ad[NPLC - 1] : 0.0, S, qi, e0, u, lane);
                n_slow++;
                n_exact += (cc >> 8) & 1;
                c = cc & 0xFF;
            }
            adv = 1;
        }
        const bool acc = (pos < adv) && (k == 0);
        const int cs = c * KMAX + sym;                       // draws per (strain, read symbol): the substitution counts of :198-206
        const bool accs = acc && (sym < KMAX);
        t += adv;
        if (t >= total) {
            if (acc) __hip_atomic_fetch_add(&s_kf[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (accs) __hip_atomic_fetch_add(&s_cnt[cs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        if (Q >= 16 * NW) {
            const unsigned r1 = (unsigned)(ro + adv * stride);
            const unsigned r2 = r1 - (unsigned)wrap;         // wraps to a huge value while r1 < wrap
            ro = (int)(r1 < r2 ? r1 : r2);
        } else {
            ro = ((ro / stride + adv) % Q) * stride;
        }
        upos = (upos + adv) & (UWIN - 1);
        // uniforms: prefetch the next half window, swap it in when the window has moved past the old one
        if (wv == 0) {
            if (!upf && t >= ulo + UWIN / 4) {
                ux0 = *(const SC_GLOBAL f4v*)(Uf + ulo + UWIN + 4 * lane);
                ux1 = *(const SC_GLOBAL f4v*)(Uf + ulo + UWIN + 256 + 4 * lane);
                upf = true;
            }
            if (t >= ulo + UWIN / 2) {
                *(f4v*)(s_uwin + ((ulo & (UWIN - 1)) + 4 * lane)) = ux0;
                *(f4v*)(s_uwin + ((ulo & (UWIN - 1)) + 256 + 4 * lane)) = ux1;
                ulo += UWIN / 2;
                upf = false;
            }
        }
        issue_loads();                                       // rows of the new window first, then the commit
        if (acc) __hip_atomic_fetch_add(&s_kf[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (accs) __hip_atomic_fetch_add(&s_cnt[cs], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        CHAIN_STAMP(3);
        lds_barrier();                                       // every wave's commits are in s_kf
        CHAIN_STAMP(4);
    }
#ifdef SC_CHAIN_PROF
    if (tid == 0) { for (int i = 0; i < 5; i++) atomicAdd(&g_chain_prof[i], prof[i]); atomicAdd(&g_chain_prof[5], n_pass); }
#endif
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int i = 0; i < NPLC; i++) {
            const int s = lane * NPLC + i;
            if (s < S) { R->abund[s] = a0m[i] + (double)s_kf[s]; R->kdraw[s] = s_kf[s]; }
        }
        if (lane == 0) {
            R->n_draws = (unsigned long long)total; R->n_exact = n_exact; R->n_slow = n_slow; R->n_pass = n_pass;
            R->chain_cycles = clock64() - clk0; R->chain_wall = wall_clock64() - wall0;
        }
    }
}

// What the wavefronts that do NOT run the chain do meanwhile, where they may not simply end (a resident workgroup, and
// the variants behind k_level_any that it shares): an s_barrier counts every live wavefront of the workgroup, so they
// take part in exactly the barriers of urn_chain_q -- the one after the uniforms are staged, the two of every pass (the
// advance of the pass is read from s_x between them, as the chain's wavefronts read it), the one after the loop.
template <int NW>
__device__ __forceinline__ void urn_chain_shadow(const LevelHdr& h, const int* s_x) {
    const int total = h.n_sweeps * h.Q;
    __syncthreads();
    int t = 0;
#pragma unroll 1
    while (t < total) {
        lds_barrier();
        int adv = 16 * NW;
#pragma unroll
        for (int j = 0; j < NW / 4; j++) {
            const i4v xf = *(const volatile i4v*)(s_x + 4 * j);
            adv = min(adv, min(min(xf.x, xf.y), min(xf.z, xf.w)));
        }
        const int rem = total - t;
        adv = adv < rem ? adv : rem;
        if (adv == 0) adv = 1;
        t += adv;
        if (t >= total) break;
        lds_barrier();
    }
    __syncthreads();
}

// --------------------------------------------------------------------------
// The pieces of one level of the walk (NonparametricClustering.cpp:284-458) that every mode shares.  They
// run inside the level's single workgroup (k_level_sample / k_level); for levels with hundreds of thousands
// of (strain, read) items the host runs the first two on a grid instead (k_level_copy, k_level_update) and
// says so in LevelHdr::done.
enum { LV_COPIES_DONE = 1, LV_ITEMS_DONE = 2, LV_HAS_DONE = 4, LV_HARD_DONE = 8 };
// The region's arrays as the batched level kernels see them: a block of device memory that does not change while the
// region is walked, read through the constant address space -- scalar loads the compiler may repeat at will, exactly
// like kernel arguments (which hold only a pointer to it: LevelItem).
typedef const __attribute__((address_space(4))) JobDev KJob;

// grid, phase 0: rows of strains created by the last extension (Strain copy, Strain.cpp:73-83); copies are
// independent (a destination row is a free row, a source row a surviving parent's).  P: device copy.
__global__ __launch_bounds__(256) void k_level_copy(JobDev job, const LevelParams* __restrict__ P) {
    const int c = blockIdx.y;
    const double2* src = reinterpret_cast<const double2*>(job.ll + (long)P->copy_src[c] * job.ll_stride);
    double2* dst = reinterpret_cast<double2*>(job.ll + (long)P->copy_dst[c] * job.ll_stride);
    const int n2 = (job.n_reads + 1) >> 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// grid, phase 1, single-symbol labels: ll[s][rid] (+)= log P(read symbol | strain symbol), NonparametricClustering.cpp:343-391
__global__ __launch_bounds__(256) void k_level_update(JobDev job, LevelHdr h, const LevelParams* __restrict__ P) {
    __shared__ double s_row[MAXS * KMAX];       // lpt[s][label of s][b]
    __shared__ double s_diag[MAXS * KMAX];      // lpt[s][b][b]  (an N in the strain label matches the read symbol)
    __shared__ int s_slot[MAXS], s_lab[MAXS];
    const int tid = threadIdx.x;
    const int S = h.S, K = job.K, e0 = h.e0, Rn = h.e1 - h.e0, codeN = job.code_N;
    for (int s = tid; s < S; s += blockDim.x) { s_slot[s] = P->sp[s].slot; s_lab[s] = job.labels[P->sp[s].lab_off]; }
    __syncthreads();
    for (int i = tid; i < S * KMAX; i += blockDim.x) {
        const int sx = i / KMAX, b = i % KMAX, a = s_lab[sx];
        const double* lp = P->lpt + (long)sx * K * K;                  // compact [K][K] table of the strain
        s_row[i] = (a < K && b < K) ? lp[a * K + b] : 0.0;
        s_diag[i] = (b < K) ? lp[b * K + b] : 0.0;
    }
    __syncthreads();
    const long total = (long)S * Rn;
    const long stride = job.ll_stride;
    for (long idx = (long)blockIdx.x * blockDim.x + tid; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int sx = (int)(idx / Rn), e = e0 + (int)(idx % Rn);
        const int rid = job.ent_rid[e];
        const int b = job.labels[job.ent_lab_off[e]];
        const bool fresh = job.ent_first[e] && !job.has[rid];
        if (sx == 0) job.isnew[e - e0] = fresh ? 1 : 0;
        int a = s_lab[sx];
        const bool wild = (a == codeN);
        if (wild) a = b;
        const double val = (a < K && b < K) ? (wild ? s_diag[sx * KMAX + b] : s_row[sx * KMAX + b]) : __longlong_as_double(0x7ff8000000000000ll);
        double* cell = job.ll + (long)s_slot[sx] * stride + rid;
        *cell = fresh ? val : (*cell + val);               // Strain::update_read_loglik, Strain.cpp:85-95
    }
}

// grid, after phase 1: the level's reads are present in read_loglik from here on (what the level's own workgroup does with
// 512 threads -- 117 rounds of two dependent loads on a level of 60 000 entries)
__global__ __launch_bounds__(256) void k_level_has(JobDev job, LevelHdr h) {
    const int Rn = h.e1 - h.e0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Rn; r += gridDim.x * blockDim.x) job.has[job.ent_rid[h.e0 + r]] = 1;
}

// hard_clustering (NonparametricClustering.cpp:17-125) of a level with millions of (strain, draw slot) pairs on a grid --
// the unthinned configs[3] region has 103 such levels of 26 strains x 110 000 slots, 9-13 ms each inside ONE workgroup.
// The same arithmetic in the same order as level_plain_body, piece by piece: the draw slots (phase_slots); a zero
// log-likelihood for a mate not seen yet; per slot the responsibilities p_s = exp(x_s - max) / sum (strains in order);
// per strain ONE wavefront that adds its responsibilities exactly as the workgroup's wavefronts do (lane j the slots
// j, j + 64, ... in order, then the tree of fixed shape): bit-identical results, 100 CUs instead of one.
__global__ __launch_bounds__(256) void k_hard_slots(JobDev job, LevelHdr h) {
    const int e0 = h.e0, Rn = h.e1 - h.e0;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < Rn; r += gridDim.x * blockDim.x) {
        const int e = e0 + r;
        const int rid = job.ent_rid[e], cn = job.ent_cn[e];
        const int qb = job.ent_qoff[e];
        const int mb = job.mate_ptr[rid], mn = job.mate_ptr[rid + 1] - mb;
        const uint8_t code = (job.ent_lab_len[e] == 1) ? job.labels[job.ent_lab_off[e]] : (uint8_t)0xFF;
        for (int i = 0; i < cn; i++) {
            const int k = cn - 1 - i;
            job.qent[qb + i] = r;
            job.quid[qb + i] = (k < mn) ? job.mate_idx[mb + k] : -1;
            job.qcode[qb + i] = code;
        }
    }
}
__global__ __launch_bounds__(256) void k_hard_mates(JobDev job, LevelHdr h, const LevelParams* __restrict__ P) {
    const long stride = job.ll_stride;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < h.Q; q += gridDim.x * blockDim.x) {
        const int uid = job.quid[q];
        // logprob(uid) inserts a zero log-likelihood for a mate not seen yet (Strain.cpp:147-150)
        if (uid >= 0 && !job.has[uid])
            for (int s = 0; s < h.S; s++) job.ll[(long)P->sp[s].slot * stride + uid] = 0.0;
    }
}
__global__ __launch_bounds__(256) void k_hard_resp(JobDev job, LevelHdr h, const LevelParams* __restrict__ P) {
    __shared__ int s_slot[MAXS];
    __shared__ double s_logpri[MAXS];
    const int S = h.S, e0 = h.e0;
    for (int s = threadIdx.x; s < S; s += blockDim.x) { s_slot[s] = P->sp[s].slot; s_logpri[s] = P->sp[s].logpri; }
    __syncthreads();
    const long stride = job.ll_stride;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < h.Q; q += gridDim.x * blockDim.x) {
        const int rid = job.ent_rid[e0 + job.qent[q]], uid = job.quid[q];
        if (uid >= 0) job.has[uid] = 1;                      // (every check of k_hard_mates is behind us: a kernel boundary)
        double* col = job.tabA + q;
        double m = -INFINITY;
        for (int s = 0; s < S; s++) {
            const double* row = job.ll + (long)s_slot[s] * stride;
            double x = s_logpri[s] + row[rid];
            if (uid >= 0) x += row[uid];
            col[(long)s * job.qcap] = x;
            m = fmax(m, x);
        }
        double norm = 0;
        for (int s = 0; s < S; s++) norm += exp(col[(long)s * job.qcap] - m);
        for (int s = 0; s < S; s++) {
            double* cell = col + (long)s * job.qcap;
            *cell = exp(*cell - m) / norm;
        }
    }
}
// one wavefront per strain (= per workgroup of 64)
__global__ __launch_bounds__(64) void k_hard_sums(JobDev job, LevelHdr h, const LevelParams* __restrict__ P, LevelResult* __restrict__ R) {
    const int s = blockIdx.x, lane = threadIdx.x, K = job.K, K2 = K * K, Q = h.Q;
    const double* prow = job.tabA + (long)s * job.qcap;
    double acc[KMAX + 1];
#pragma unroll
    for (int b = 0; b <= KMAX; b++) acc[b] = 0.0;
    for (int q = lane; q < Q; q += 64) {
        const double p = prow[q];
        const int code = job.qcode[q];
        acc[KMAX] += p;
#pragma unroll
        for (int b = 0; b < KMAX; b++) acc[b] += (code == b) ? p : 0.0;
    }
#pragma unroll
    for (int b = 0; b <= KMAX; b++) {
        double v = acc[b];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        acc[b] = v;
    }
    const int a = (int)job.labels[P->sp[s].lab_off];
    // the strain's [K][K] block of the substitution histogram: zero except the row of its own symbol (every cell written
    // once, by the lane that owns it; the sums live in lane 0)
    double row[KMAX];
#pragma unroll
    for (int b = 0; b < KMAX; b++) row[b] = __shfl(acc[b], 0);
    for (int i = lane; i < K2; i += 64) {
        double v = 0.0;
#pragma unroll
        for (int b = 0; b < KMAX; b++) if (a < K && b < K && i == a * K + b) v = row[b];
        R->subst[(long)s * K2 + i] = v;
    }
    if (lane == 0) R->abund[s] = acc[KMAX];
}

// The per-strain parameters of the level, straight from host-mapped memory into LDS: 16-byte loads over PCIe,
// every thread a few, one round trip.
__device__ __forceinline__ void stage_params(const LevelHdr& h, const LevelParams* __restrict__ P, StrainParam* s_sp, int* s_copy,
                                             double* s_lpt, bool want_lpt, int K2, int tid, int nt) {
    const i4v* src_sp = reinterpret_cast<const i4v*>(P->sp);
    i4v* dst_sp = reinterpret_cast<i4v*>(s_sp);
    for (int i = tid; i < h.S * 2; i += nt) dst_sp[i] = src_sp[i];
    for (int i = tid; i < h.n_copy; i += nt) { s_copy[i] = P->copy_src[i]; s_copy[MAXS + i] = P->copy_dst[i]; }
    if (want_lpt) {
        const double2* s = reinterpret_cast<const double2*>(P->lpt);
        double2* d = reinterpret_cast<double2*>(s_lpt);
        for (int i = tid; i < (h.S * K2 + 1) / 2; i += nt) d[i] = s[i];
    }
}

// phase 0 inside the workgroup
template <class JD>
__device__ __forceinline__ void phase_copies(const JD& job, const LevelHdr& h, const int* s_copy, int tid, int nt) {
    const long stride = job.ll_stride;
    for (int c = 0; c < ((h.done & LV_COPIES_DONE) ? 0 : h.n_copy); c++) {
        const double2* src = reinterpret_cast<const double2*>(job.ll + (long)s_copy[c] * stride);
        double2* dst = reinterpret_cast<double2*>(job.ll + (long)s_copy[MAXS + c] * stride);
        const int n2 = ((h.copy_n < job.n_reads ? h.copy_n : job.n_reads) + 1) >> 1;      // cells past copy_n hold nothing yet
        int i = tid;
        for (; i + 3 * nt < n2; i += 4 * nt) {        // four independent 16-byte loads in flight per thread
            const double2 v0 = src[i], v1 = src[i + nt], v2 = src[i + 2 * nt], v3 = src[i + 3 * nt];
            dst[i] = v0; dst[i + nt] = v1; dst[i + 2 * nt] = v2; dst[i + 3 * nt] = v3;
        }
        for (; i < n2; i += nt) dst[i] = src[i];
    }
    // the copies are independent (a destination is a free row, a source a surviving parent's row): one barrier for all
    __syncthreads();
}

// phase 1: read log-likelihood update, NonparametricClustering.cpp:343-391, then the reads of the level are
// present in read_loglik (`has`).  s_lpt: the strains' log tables in LDS.
template <class JD>
__device__ __forceinline__ void phase_update(const JD& job, const LevelHdr& h, const StrainParam* s_sp, const int* s_lab,
                                             const double* s_lpt, int tid, int nt) {
    const int S = h.S, K = job.K, K2 = K * K, e0 = h.e0, Rn = h.e1 - h.e0;
    const long stride = job.ll_stride;
    const bool plain_labels = !h.has_dups && !h.any_multi && !(h.done & LV_ITEMS_DONE);
    if (!plain_labels && !(h.done & LV_ITEMS_DONE)) {           // (k_level_update has written isnew with its items)
        // (the walk over multi-symbol labels below and in the soft update reads this; the usual level finds it on the way)
        for (int r = tid; r < Rn; r += nt) {
            const int e = e0 + r;
            job.isnew[r] = (job.ent_first[e] && !job.has[job.ent_rid[e]]) ? 1 : 0;
        }
        __syncthreads();
    }
    const int codeN = job.code_N;
    auto item = [&](int s, int r) {
        const int e = e0 + r;
        const int rid = job.ent_rid[e];
        const uint8_t* sb = job.labels + s_sp[s].lab_off;
        const uint8_t* rb = job.labels + job.ent_lab_off[e];
        const int ls = s_sp[s].lab_len, lr = job.ent_lab_len[e];
        const double* lp = s_lpt + s * K2;
        double val;
        if (ls == 1) {
            int a = sb[0], b = rb[0];
            if (lr == 1) {
                if (a == codeN) a = b;
                val = (a < K && b < K) ? lp[a * K + b] : __longlong_as_double(0x7ff8000000000000ll);
            } else {
                // logprob(sb, "multi"): sub_count[(sb, rb)] is created as 0 (std::map operator[]), so the
                // result is log 0 - log comp(sb) = -inf for a symbol of the alphabet; for N (sb becomes rb)
                // or a symbol outside the alphabet comp is created as 0 too: -inf - -inf
                val = (a < 6 && a != codeN) ? -INFINITY : __longlong_as_double(0x7ff8000000000000ll);
            }
        } else {
            val = 0.0;
            if (job.isnew[r]) {
                int ii = ls, jj = lr;
                while (ii > 0 && jj > 0) {
                    int a = sb[--ii], b = rb[--jj];
                    if (a == codeN) a = b;
                    val += lp[a * K + b];
                }
            } else {
                int ii = 0, jj = 0;
                while (ii < ls && jj < lr) {
                    int a = sb[ii++], b = rb[jj++];
                    if (a == codeN) a = b;
                    val += lp[a * K + b];
                }
            }
        }
        double* cell = job.ll + (long)s_sp[s].slot * stride + rid;
        const bool fresh = job.ent_first[e] && !job.has[rid];
        *cell = fresh ? val : (*cell + val);            // Strain::update_read_loglik, Strain.cpp:85-95
    };
    if (h.done & LV_ITEMS_DONE) {
        // k_level_update has applied the items
    } else if (!h.has_dups && !h.any_multi) {
        // single-symbol labels everywhere (the usual level): a thread takes a read of the level and walks the strains
        // eight at a time, so eight independent row cells are in flight per thread and neighbouring threads (reads
        // sorted by position: neighbouring ids) touch neighbouring cells of each row
        // An item is (sixteen strains, read): sixteen independent row cells in flight per thread, the items of a chunk of
        // strains on neighbouring threads (so that a level of 600 reads x 30 strains is three rounds of the workgroup, not
        // two rounds of reads x four chunks one after the other).
        constexpr int U = 16;
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
        const int nch = (S + U - 1) / U;
        const int items = Rn * nch;
        for (int idx = tid; idx < items; idx += nt) {
            const int ch = idx / Rn, r = idx - ch * Rn;
            const int e = e0 + r;
            const int rid = job.ent_rid[e];
            const int b = job.labels[job.ent_lab_off[e]];
            const bool fresh = job.ent_first[e] && !job.has[rid];
            if (ch == 0) job.isnew[r] = fresh ? 1 : 0;
            const int s0 = ch * U;
            long off[U];                                        // (no early exit from the unrolled loops: the arrays stay in registers)
            double old[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int sx = (s0 + k < S) ? s0 + k : S - 1;
                off[k] = (long)s_sp[sx].slot * stride + rid;
                old[k] = fresh ? 0.0 : job.ll[off[k]];
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int sx = s0 + k;
                if (sx < S) {
                    int a = s_lab[sx];
                    if (a == codeN) a = b;
                    const double val = (a < K && b < K) ? s_lpt[sx * K2 + a * K + b] : qnan;
                    job.ll[off[k]] = fresh ? val : (old[k] + val);       // Strain::update_read_loglik, Strain.cpp:85-95
                }
            }
        }
    } else if (!h.has_dups) {
        const long total = (long)S * Rn;
        for (long idx = tid; idx < total; idx += nt) item((int)(idx / Rn), (int)(idx % Rn));
    } else {
        if (tid < S) for (int r = 0; r < Rn; r++) item(tid, r);
    }
    if (h.done & LV_HAS_DONE) { __syncthreads(); return; }      // (k_level_has, behind the grid's update; the barrier is the phase's)
    __syncthreads();
    for (int r = tid; r < Rn; r += nt) job.has[job.ent_rid[e0 + r]] = 1;
    __syncthreads();
}

// phase 2: draw slots q = (entry, copy); the reference walks copies from cn down to 1 (:161-167)
template <bool BARRIER = true, class JD>
__device__ __forceinline__ void phase_slots(const JD& job, const LevelHdr& h, int tid, int nt) {
    const int e0 = h.e0, Rn = h.e1 - h.e0;
    for (int r = tid; r < Rn; r += nt) {
        const int e = e0 + r;
        const int rid = job.ent_rid[e], cn = job.ent_cn[e];
        const int qb = job.ent_qoff[e];
        const int mb = job.mate_ptr[rid], mn = job.mate_ptr[rid + 1] - mb;
        const uint8_t code = (job.ent_lab_len[e] == 1) ? job.labels[job.ent_lab_off[e]] : (uint8_t)0xFF;
        for (int i = 0; i < cn; i++) {
            const int k = cn - 1 - i;
            const int uid = (k < mn) ? job.mate_idx[mb + k] : -1;
            job.qent[qb + i] = r;
            job.quid[qb + i] = uid;
            job.qcode[qb + i] = code;
        }
    }
    if (BARRIER) __syncthreads();
}

// every result of the level is in host memory before the stamp
__device__ __forceinline__ void finish_level(const LevelHdr& h, LevelResult* __restrict__ R, unsigned long long wall0, int tid) {
    __threadfence_system();
    __syncthreads();
    if (tid == 0) {
        R->level_wall = wall_clock64() - wall0;
        R->xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xF);      // HW_REG_XCC_ID[3:0]
        R->xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xF);      // HW_REG_XCC_ID[3:0]
        __hip_atomic_store(&R->seq, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// exp(y), y <= 0, as an fp32 sampler weight: y = k ln2 + r in fp64 (exact to 2^-60), exp(r) by the fp32 hardware
// exponential, scaled by 2^k.  Relative error < 3 * 2^-24 (fp32 rounding of r: 0.35 * 2^-25; of r * log2 e: 0.5 * 2^-24;
// v_exp_f32: 1 ulp; the final rounding), which the window margin EPSW budgets for; a NaN stays a NaN (it sends the
// draw to the checked tiers), anything below 2^-149 is 0 as in the rounded exact value.
__device__ __forceinline__ float exp_weight(double y) {
    if (y < -104.0) return 0.0f;                              // below the last fp32 denormal (and -inf: log 0)
    const double k = rint(y * 1.4426950408889634);
    double r = fma(k, -0.693147180559945286, y);
    r = fma(k, -2.3190468138462996e-17, r);
    const float e = __builtin_amdgcn_exp2f((float)r * 1.44269504f);
    const double kc = fmax(k, -300.0);                        // ldexpf's int: far below the last denormal is still 0
    return (y == y) ? ldexpf(e, (int)kc) : __int_as_float(0x7fc00000);
}

__host__ __device__ inline int chain_w_stride(int S) {
    const int s4 = (S + 1 + 3) & ~3;                          // S weights + the read symbol
    return (s4 & 4) ? s4 : s4 + 4;                            // 4 * odd: conflict-free 16-byte row reads
}

// LDS of the level kernels: per-strain scalars first, then one big region that holds the strains' log tables
// during the update and the sampler's uniforms + weight rows (or the soft update's histogram) afterwards.
struct LevelLds {
    double* s_a; double* s_p; unsigned* s_cnt; int* s_slot; unsigned* s_kf; float* s_a0f; int* s_x; int* s_copy; int* s_lab;
    StrainParam* s_sp; unsigned char* s_big;
};
__device__ __forceinline__ LevelLds level_lds(unsigned char* raw) {
    LevelLds l;
    l.s_a = reinterpret_cast<double*>(raw);                        // [MAXS]
    l.s_p = l.s_a + MAXS;                                          // [MAXS]
    l.s_sp = reinterpret_cast<StrainParam*>(l.s_p + MAXS);         // [MAXS]
    l.s_cnt = reinterpret_cast<unsigned*>(l.s_sp + MAXS);          // [MAXS*KMAX]
    l.s_slot = reinterpret_cast<int*>(l.s_cnt + MAXS * KMAX);      // [MAXS]
    l.s_kf = reinterpret_cast<unsigned*>(l.s_slot + MAXS);         // [MAXS] draws per strain so far
    l.s_a0f = reinterpret_cast<float*>(l.s_kf + MAXS);             // [MAXS] fp32 copy of the starting weights
    l.s_copy = reinterpret_cast<int*>(l.s_a0f + MAXS);             // [2*MAXS]
    l.s_lab = l.s_copy + 2 * MAXS;                                 // [MAXS] first symbol of the strain's node label
    l.s_x = l.s_lab + MAXS;                                        // [8] first failing position per wave
    l.s_big = raw + LDS_SMALL;
    return l;
}

// --------------------------------------------------------------------------
// a13 + a14 / a18: one sampler level in ONE launch -- np_bayes_clustering
// (NonparametricClustering.cpp:128-244) on the level's reads, and read_assign
// (:776-836) on the pseudo-level of all reads.  One workgroup: the per-strain
// parameters arrive from host-mapped memory, new strains get their rows, the
// level's read log-likelihoods are updated (a13), the draw slots and the fp32
// weight rows L[q][s] = exp(ll - max_s ll) are built straight into LDS (HBM when
// they do not fit), then four or eight wavefronts run the urn chain
// (urn_chain_q) and the results go to host-mapped memory, stamped.
// NB = ceil(S / 16): every lane of a quad owns 4 * NB consecutive strains.
constexpr int CHAINW_ROWS_BYTES = LDS_BIG - UWIN * 4;
constexpr int CHAIN_THREADS = 512;   // eight wavefronts build the level
// wavefronts that run the chain = window of 16 * NW draws: all eight while a lane's share of the strains is
// small (the pass is latency-bound and a wider window accepts more draws), four otherwise (the others leave)
constexpr int chain_nw(int nb) { return nb <= 2 ? 8 : 4; }
// the level's scalars out of an item, whatever address space the item is read through (field by field: a struct in the
// constant address space has no copy constructor)
template <class IT>
__device__ __forceinline__ LevelHdr load_hdr(const IT& it) {
    LevelHdr h;
    h.mode = it.h.mode; h.S = it.h.S; h.e0 = it.h.e0; h.e1 = it.h.e1; h.has_dups = it.h.has_dups; h.any_multi = it.h.any_multi;
    h.Q = it.h.Q; h.n_sweeps = it.h.n_sweeps; h.n_copy = it.h.n_copy; h.do_update = it.h.do_update; h.done = it.h.done;
    h.copy_n = it.h.copy_n; h.seq = it.h.seq;
    return h;
}
template <int NB, bool ROWS_LDS, bool STAY, class IT>
__device__ __forceinline__ void level_sample_body(const IT& it, unsigned char* s_raw) {
    const unsigned long long wall0 = wall_clock64();
    const LevelHdr h = load_hdr(it);
    KJob& job = *(KJob*)it.job;
    const LevelParams* __restrict__ P = it.P;
    LevelResult* __restrict__ R = it.R;
    const LevelLds l = level_lds(s_raw);
    float* s_uwin = reinterpret_cast<float*>(l.s_big);           // [UWIN]
    float* s_rows = s_uwin + UWIN;
    const int tid = threadIdx.x;
    int nt = blockDim.x;
    const int S = h.S, Q = h.Q, e0 = h.e0, Rn = h.e1 - h.e0;
    const int stride = chain_w_stride(S);
    const bool upd = h.do_update && Rn > 0;
    stage_params(h, P, l.s_sp, l.s_copy, reinterpret_cast<double*>(l.s_big), upd && !(h.done & LV_ITEMS_DONE), job.K * job.K, tid, nt);
    for (int i = tid; i < MAXS * KMAX; i += nt) l.s_cnt[i] = 0;
    if (tid == 0) { R->error = 0; R->n_draws = 0; R->n_exact = 0; R->n_slow = 0; R->n_pass = 0; R->chain_cycles = 0; R->chain_wall = 0; }
    __syncthreads();
    if (tid == 0) R->phase_ticks[0] = (unsigned)(wall_clock64() - wall0);
    if (tid < MAXS) {
        l.s_slot[tid] = tid < S ? l.s_sp[tid].slot : 0;
        l.s_kf[tid] = 0u;
        l.s_a0f[tid] = tid < S ? (float)l.s_sp[tid].a0 : 0.0f;
        l.s_lab[tid] = tid < S ? (int)job.labels[l.s_sp[tid].lab_off] : 0;
    }
    phase_copies(job, h, l.s_copy, tid, nt);
    if (tid == 0) R->phase_ticks[1] = (unsigned)(wall_clock64() - wall0);
    // the draw slots depend on the level's entries alone: their loads and stores go out in front of the update's, and
    // the update's closing barriers cover them (phase_ticks[3] - [2] is therefore ~0 on a level with an update)
    if (upd) {
        phase_slots<false>(job, h, tid, nt);
        phase_update(job, h, l.s_sp, l.s_lab, reinterpret_cast<const double*>(l.s_big), tid, nt);
    }
    if (tid == 0) R->phase_ticks[2] = (unsigned)(wall_clock64() - wall0);
    if (!upd) phase_slots(job, h, tid, nt);
    if (tid == 0) R->phase_ticks[3] = (unsigned)(wall_clock64() - wall0);

    // what the sampler draws from.  Per draw slot q the fp32 weight row L[q][s] = exp(x_s - max_s x_s),
    // x_s = ll(read) + ll(mate) in fp64, with the read's symbol behind it.  G lanes share a slot, each walks
    // <= 8 strains; the max is joined by shuffles.  A slot whose log-likelihoods lie in the underflow range of
    // the reference's exp() gets a NaN row, which sends its draws to the checked tiers.
    {
        // A lane takes a draw slot q and walks the strains: the 64 slots of a wavefront are neighbouring reads (slots follow
        // the level's entries, entries the read ids), so one load instruction -- one strain's row at 64 nearby read ids --
        // touches a dozen cache lines instead of 64.  (A lane used to take eight strains of one slot: every 8-byte cell came
        // with its own 128-byte line from the L2, 3.3 MB per level of 860 read copies x 30 strains at the 64 bytes a clock a
        // compute unit gets -- 25 us, the longest part of a level outside the chain.)  Up to 32 strains the cells stay in
        // registers between the maximum and the exponentials; beyond, they are read a second time (from the L1 / L2).
        constexpr int SR = (NB <= 2) ? 16 * NB : 16;           // cells held per lane
        const long lstride = job.ll_stride;
        SC_GLOBAL float* rows_g = (SC_GLOBAL float*)job.tabLf;
        auto put4 = [&](long idx, f4v v) __attribute__((always_inline)) {
            if (ROWS_LDS) *(f4v*)(s_rows + idx) = v; else *(SC_GLOBAL f4v*)(rows_g + idx) = v;
        };
        const float qnanf = __int_as_float(0x7fc00000);
        for (int q = tid; q < Q; q += nt) {
            const int rid = job.ent_rid[e0 + job.qent[q]], uid = job.quid[q];
            const bool hr = h.do_update ? true : job.has[rid] != 0;      // the update has just entered the level's reads
            const bool hu = uid >= 0 && job.has[uid] != 0;
            const int c0 = job.qcode[q];
            const float symf = __int_as_float(c0 < KMAX ? c0 : KMAX);    // the read symbol rides behind the weights (a tiny denormal under a zero count: no effect on the sums)
            const long Lf = (long)q * stride;
            auto cell = [&](int sx) __attribute__((always_inline)) -> double {
                const double* row = job.ll + (long)l.s_slot[sx] * lstride;
                double v = hr ? row[rid] : 0.0;
                if (hu) v += row[uid];
                return v;
            };
            double m = -INFINITY;
            double x[SR];
            if (NB <= 2) {
#pragma unroll
                for (int i = 0; i < SR; i++) { x[i] = -INFINITY; if (i < S) { x[i] = cell(i); m = fmax(m, x[i]); } }
            } else {
                for (int s0 = 0; s0 < S; s0 += SR) {
#pragma unroll
                    for (int i = 0; i < SR; i++) if (s0 + i < S) m = fmax(m, cell(s0 + i));
                }
            }
            const bool flag = !(m >= -600.0);                // underflow range of the reference's exp(); also NaN / -inf
            // the row: S weights, the symbol, zeros up to the stride (a multiple of four floats)
            for (int s0 = 0; s0 < stride; s0 += SR) {
                if (NB > 2) {
#pragma unroll
                    for (int i = 0; i < SR; i++) x[i] = (s0 + i < S) ? cell(s0 + i) : -INFINITY;
                }
#pragma unroll
                for (int i = 0; i < SR; i += 4) {
                    if (s0 + i < stride) {
                        f4v w;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int sx = s0 + i + k;
                            const float wk = sx < S ? (flag ? qnanf : exp_weight(x[i + k] - m)) : (sx == S ? symf : 0.0f);
                            if (k == 0) w.x = wk; else if (k == 1) w.y = wk; else if (k == 2) w.z = wk; else w.w = wk;
                        }
                        put4(Lf + s0 + i, w);
                    }
                }
            }
            // the chain reads whole 16-strain blocks: keep what follows the last row finite
            if (q == Q - 1) for (int i = 0; i < 16; i += 4) put4(Lf + stride + i, f4v{0.0f, 0.0f, 0.0f, 0.0f});
        }
    }
    __syncthreads();
    if (tid == 0) R->phase_ticks[4] = (unsigned)(wall_clock64() - wall0);
    constexpr int NW = chain_nw(NB);
    if (tid >= 64 * NW) {
        if (!STAY) return;                                 // a finished wavefront no longer counts at the barriers below
        urn_chain_shadow<NW>(h, l.s_x);                    // ... one that has to stay takes part in them
    } else {
        urn_chain_q<NB, ROWS_LDS, NW>(job, h, l.s_sp, R, l.s_slot, l.s_a, l.s_p, l.s_kf, l.s_a0f, l.s_cnt, l.s_x, s_uwin, s_rows, stride, tid);
    }
    if (!STAY) nt = 64 * NW;
    __syncthreads();
    for (int i = tid; i < S * KMAX; i += nt) R->cnt[i] = l.s_cnt[i];
    finish_level(h, R, wall0, tid);
}

// --------------------------------------------------------------------------
// a13 + a15: one level without the sampler in one launch -- the read log-likelihood update, and for MODE_HARD
// the soft update hard_clustering (NonparametricClustering.cpp:17-125).  Single workgroup.
template <class IT>
__device__ __forceinline__ void level_plain_body(const IT& it, unsigned char* s_raw) {
    const unsigned long long wall0 = wall_clock64();
    const LevelHdr h = load_hdr(it);
    KJob& job = *(KJob*)it.job;
    const LevelParams* __restrict__ P = it.P;
    LevelResult* __restrict__ R = it.R;
    const LevelLds l = level_lds(s_raw);
    double* s_tab = reinterpret_cast<double*>(l.s_big);          //   [S][K][K] log tables, later the substitution histogram
    const int tid = threadIdx.x, nt = blockDim.x;
    const int S = h.S, K = job.K, K2 = K * K, e0 = h.e0, Rn = h.e1 - h.e0;
    const long stride = job.ll_stride;
    const bool upd = h.do_update && Rn > 0;
    stage_params(h, P, l.s_sp, l.s_copy, s_tab, upd && !(h.done & LV_ITEMS_DONE), K2, tid, nt);
    if (tid == 0) { R->error = 0; R->n_draws = 0; R->n_exact = 0; R->n_slow = 0; R->n_pass = 0; R->chain_cycles = 0; R->chain_wall = 0; }
    __syncthreads();
    if (tid == 0) R->phase_ticks[0] = (unsigned)(wall_clock64() - wall0);
    if (tid < S) l.s_lab[tid] = (int)job.labels[l.s_sp[tid].lab_off];
    phase_copies(job, h, l.s_copy, tid, nt);
    if (tid == 0) R->phase_ticks[1] = (unsigned)(wall_clock64() - wall0);
    if (upd) phase_update(job, h, l.s_sp, l.s_lab, s_tab, tid, nt);
    if (tid == 0) R->phase_ticks[2] = (unsigned)(wall_clock64() - wall0);
    if (Rn <= 0 || S <= 0 || h.mode != MODE_HARD || (h.done & LV_HARD_DONE)) { finish_level(h, R, wall0, tid); return; }
    phase_slots(job, h, tid, nt);
    if (tid == 0) R->phase_ticks[3] = (unsigned)(wall_clock64() - wall0);

    // hard_clustering, NonparametricClustering.cpp:17-125
    const int Q = h.Q;
    // logprob(uid) inserts a zero log-likelihood for a mate not seen yet (Strain.cpp:147-150)
    for (int q = tid; q < Q; q += nt) {
        const int uid = job.quid[q];
        if (uid >= 0 && !job.has[uid])
            for (int s = 0; s < S; s++) job.ll[(long)l.s_sp[s].slot * stride + uid] = 0.0;
    }
    __syncthreads();
    for (int q = tid; q < Q; q += nt) { const int uid = job.quid[q]; if (uid >= 0) job.has[uid] = 1; }
    __syncthreads();
    // x[s][q] = log prior + ll(read) + ll(mate): one (strain, slot) pair per thread and step, so the scattered
    // row reads of a slot's strains are all in flight together
    for (long idx = tid; idx < (long)S * Q; idx += nt) {
        const int s = (int)(idx / Q), q = (int)(idx % Q);
        const int rid = job.ent_rid[e0 + job.qent[q]], uid = job.quid[q];
        const double* row = job.ll + (long)l.s_sp[s].slot * stride;
        double x = l.s_sp[s].logpri + row[rid];
        if (uid >= 0) x += row[uid];
        job.tabA[(long)s * job.qcap + q] = x;
    }
    __syncthreads();
    for (int q = tid; q < Q; q += nt) {
        // p_s = exp(x_s) / sum_s exp(x_s) (:186-192).  The reference forms it in long double, where
        // exp(-800) is an ordinary number (a read laid against a long collapsed node it does not match
        // reaches such log-likelihoods for every strain); in fp64 the same quotient needs the maximum
        // taken out first.  A NaN x_s still makes every p NaN, as in the reference.
        const double* col = job.tabA + q;
        double m = -INFINITY;
        for (int s = 0; s < S; s++) m = fmax(m, col[(long)s * job.qcap]);
        double norm = 0;
        for (int s = 0; s < S; s++) norm += exp(col[(long)s * job.qcap] - m);
        for (int s = 0; s < S; s++) {
            double* cell = job.tabA + (long)s * job.qcap + q;
            *cell = exp(*cell - m) / norm;
        }
    }
    for (int i = tid; i < S * K2; i += nt) s_tab[i] = 0.0;
    __syncthreads();
    if (tid == 0) R->phase_ticks[4] = (unsigned)(wall_clock64() - wall0);
    if (!h.any_multi) {
        // One wavefront per strain: lane j adds the responsibilities of the slots j, j + 64, ... (in slot order), per read
        // symbol, then the 64 partial sums are joined by a tree of fixed shape.  The reference adds them one after the other
        // in long double; any fixed order of fp64 additions is as close to that as another (1e-13 relative, the tests allow
        // 1e-9), and two candidates with equal inputs still get bitwise equal sums, which is what their ties rest on.
        const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
        for (int s = wv; s < S; s += nw) {
            const double* prow = job.tabA + (long)s * job.qcap;
            double acc[KMAX + 1];
#pragma unroll
            for (int b = 0; b <= KMAX; b++) acc[b] = 0.0;
            for (int q = lane; q < Q; q += 64) {
                const double p = prow[q];
                const int code = job.qcode[q];
                acc[KMAX] += p;
#pragma unroll
                for (int b = 0; b < KMAX; b++) acc[b] += (code == b) ? p : 0.0;
            }
#pragma unroll
            for (int b = 0; b <= KMAX; b++) {
                double v = acc[b];
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
                acc[b] = v;
            }
            if (lane == 0) {
                R->abund[s] = acc[KMAX];
                const int a = l.s_lab[s];
                if (a < K)
                    for (int b = 0; b < K; b++) s_tab[s * K2 + a * K + b] = acc[b];
            }
        }
    } else {
        if (tid < S) {
            const int s = tid;
            const double* prow = job.tabA + (long)s * job.qcap;
            double* hist = s_tab + s * K2;
            const uint8_t* sb = job.labels + l.s_sp[s].lab_off;
            const int ls = l.s_sp[s].lab_len;
            double acc = 0;
            for (int q = 0; q < Q; q++) {
                const double p = prow[q];
                acc += p;
                const int r = job.qent[q], e = e0 + r;
                const uint8_t* rb = job.labels + job.ent_lab_off[e];
                const int lr = job.ent_lab_len[e];
                if (lr == 1) {
                    if (ls == 1) hist[sb[0] * K + rb[0]] += p;
                } else if (job.isnew[r]) {
                    int i = ls, j = lr;
                    while (i > 0 && j > 0) { int a = sb[--i], b = rb[--j]; hist[a * K + b] += p; }
                } else {
                    int i = 0, j = 0;
                    while (i < ls && j < lr) { int a = sb[i++], b = rb[j++]; hist[a * K + b] += p; }
                }
            }
            R->abund[s] = acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < S * K2; i += nt) R->subst[i] = s_tab[i];
    finish_level(h, R, wall0, tid);
}

// One kernel per kind of level (a single region in flight launches its levels directly) ...
template <int NB, bool ROWS_LDS>
__global__ __launch_bounds__(CHAIN_THREADS) void k_level_sample(LevelBatch batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    level_sample_body<NB, ROWS_LDS, false>(batch.it[blockIdx.x], s_raw);
}
__global__ __launch_bounds__(512) void k_level(LevelBatch batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    level_plain_body(batch.it[blockIdx.x], s_raw);
}
// ... and one kernel for a batch of levels of ANY kind (many regions in flight: with one kind per launch, and a
// stream held until its batch is done, the five or so kinds that wait at any time take turns for the free streams;
// tools/launch_policy_sim.py).  Workgroup b looks at the kind of its item and calls the variant: the variants are
// functions of their own (`noinline`: each keeps its own register allocation -- merged into one body they spilled
// 56 scalars into the sampler's pass loop), reading the item through the constant address space like kernel arguments.
typedef const __attribute__((address_space(4))) LevelItem KItem;
template <int NB, bool ROWS_LDS>
__device__ __noinline__ void level_sample_call(const LevelItem* it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    level_sample_body<NB, ROWS_LDS, true>(*it, s_raw);      // every wavefront stays to the end (shared with the resident workgroups)
}
__device__ __noinline__ void level_plain_call(const LevelItem* it) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    level_plain_body(*it, s_raw);
}
__device__ __forceinline__ void level_dispatch(const LevelItem* it) {
    const int kind = it->kind & 0xFF;                                             // (the upper bits carry the item's LDS need)
    if (kind == 0) { level_plain_call(it); return; }
    const bool wl = (kind - 1) & 1;
#define SC_ANY(NB) case NB: if (wl) level_sample_call<NB, true>(it); else level_sample_call<NB, false>(it); break;
    switch ((kind - 1) / 2 + 1) {
        SC_ANY(1) SC_ANY(2) SC_ANY(3) SC_ANY(4) SC_ANY(5) SC_ANY(6) SC_ANY(7)
        default: if (wl) level_sample_call<8, true>(it); else level_sample_call<8, false>(it);
    }
#undef SC_ANY
}
__global__ __launch_bounds__(CHAIN_THREADS) void k_level_any(LevelBatch batch) {
    __shared__ LevelItem s_item;                                                  // the variants read their item through a generic pointer
    KItem* it = (KItem*)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.x;      // batch is the only argument
    if (threadIdx.x < sizeof(LevelItem) / 4) reinterpret_cast<unsigned*>(&s_item)[threadIdx.x] = reinterpret_cast<const __attribute__((address_space(4))) unsigned*>(it)[threadIdx.x];
    __syncthreads();
    level_dispatch(&s_item);
    (void)batch;
}

// Resident level workers: workgroup b serves slot b of the context (see Mailbox in sc_device.hpp).  Wavefront 0 polls
// the slot's mailbox over PCIe (a relaxed system-scope load, then a nap that grows to ~3 us), the other wavefronts wait
// at the workgroup barrier.  A new level: one system-scope acquire (the arrays of a new region arrive by DMA while this
// workgroup stays on its CU: its L1 must not serve lines of the region before) and a scalar-cache invalidate (the
// region's JobDev block is read through the constant address space), then the item goes to LDS and the variant it names
// runs exactly as it does behind k_level_any.  Every wavefront reaches the exit: `stop`, or a heartbeat that stands still.
__global__ __launch_bounds__(CHAIN_THREADS) void k_level_resident(ResidentArgs a) {
    __shared__ __attribute__((aligned(16))) LevelItem s_item;
    __shared__ int s_cmd;
    Mailbox* mb = a.mail + blockIdx.x;
    const int tid = threadIdx.x;
    unsigned last = 0, served = 0;
    if (tid == 0) {
        last = __hip_atomic_load(&mb->ack, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;) {
        if (tid == 0) {
            int cmd = 0;
            unsigned naps = 0;
            unsigned long long t_hb = wall_clock64();
            unsigned hb0 = __hip_atomic_load(&a.ctl->heartbeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (;;) {
                const unsigned sq = __hip_atomic_load(&mb->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (sq != last) { cmd = 1; break; }
                if ((naps & 15u) == 15u) {
                    if (__hip_atomic_load(&a.ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
                    const unsigned hb = __hip_atomic_load(&a.ctl->heartbeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    const unsigned long long now = wall_clock64();
                    if (hb != hb0) { hb0 = hb; t_hb = now; }
                    else if (now - t_hb > a.idle_ticks) break;           // nobody is there any more
                }
                naps++;
                if (naps < 32) __builtin_amdgcn_s_sleep(4);
                else if (naps < 256) __builtin_amdgcn_s_sleep(32);
                else __builtin_amdgcn_s_sleep(127);
            }
            if (cmd) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
                __builtin_amdgcn_s_dcache_inv();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const volatile u4v* src = reinterpret_cast<const volatile u4v*>(&mb->item);      // five 16-byte reads over PCIe
                u4v* dst = reinterpret_cast<u4v*>(&s_item);
                static_assert(sizeof(LevelItem) == 80 && alignof(Mailbox) >= 8, "mailbox item");
#pragma unroll
                for (int i = 0; i < 5; i++) dst[i] = src[i];
                last = s_item.h.seq;
                served++;
                // an item that does not name one worker's pair of blocks was not written by the library: leave rather than follow its pointers
                const long ip = s_item.P - a.P_base, ir = s_item.R - a.R_base;
                if (ip < 0 || ip >= a.n_blocks || ir != ip || a.P_base + ip != s_item.P || s_item.job == nullptr) cmd = 2;
            }
            s_cmd = cmd;
        }
        __syncthreads();
        if (s_cmd != 1) break;
        level_dispatch(&s_item);
        __syncthreads();                                   // the level is stamped; s_item and s_cmd may be rewritten
    }
    if (tid == 0) {
        __hip_atomic_store(&mb->levels, served, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&mb->state, s_cmd == 2 ? 3u : 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// --------------------------------------------------------------------------
// a7/a8: progressive sum-of-pairs MSA, MultipleSequenceAlignmentSP.cpp:10-301,
// scored with SimpleDnaScore (SimpleDnaScore.cpp:15-42, Score.hpp:35).
//
// The per-row gap memory PP of the reference is uniform over the rows of a cell
// everywhere except in column j = 0 (its writers never advance the row iterator,
// :208-217/:235-245), so the three sum-of-pairs candidates of a cell reduce to
// dot products of the column's character counts with the score table.  The DP of
// one progressive step runs on one wavefront: lane j owns DP column j and the
// (i-1,j-1)/(i,j-1) dependencies arrive from lane j-1 through a lane shift, one
// anti-diagonal per iteration.  All values are small integers (exact in int32).
__device__ __forceinline__ int cls_of(char c) {
    switch (c) {
        case 'A': return 0; case 'a': return 1; case 'C': return 2; case 'c': return 3;
        case 'G': return 4; case 'g': return 5; case 'T': return 6; case 't': return 7;
        case '+': return 8; case '-': return 9; default: return 10;
    }
}
__device__ __forceinline__ int dna_score_cls(int x, int y) {
    if (x > 9 || y > 9) return 0;                 // std::map operator[] on a missing key
    if (x == y) return 3;
    if (x < 8 && y < 8 && (x >> 1) == (y >> 1)) return 3;
    if ((x == 8 && y == 9) || (x == 9 && y == 8)) return 3;
    if (x == 8 || y == 8) return -6;              // gap_open + gap_extend
    if (x == 9 || y == 9) return -2;              // gap_extend
    return -5;
}


constexpr int MSA_CM = 1024;                       // widest alignment (columns) kept in LDS
constexpr size_t MSA_LDS = 2 * MSA_CM * 10 * sizeof(unsigned short) + 2 * MSA_CM + (MSA_CM + 1) * 64 + 2 * (MSA_CM + 64) * sizeof(int);

// WIDE: the alignment may grow past MSA_CM columns (hundreds of distinct insertion strings at one site:
// the reference's scoring tends to open new columns); the same state then lives in HBM scratch sized for
// the sum of the sequence lengths instead of LDS.
template <bool WIDE>
__global__ __launch_bounds__(256) void k_msa(MsaDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char m_raw[];
    const int CM = WIDE ? d.cmax : MSA_CM;                                        // capacity in columns
    unsigned short* s_cnt = WIDE ? reinterpret_cast<unsigned short*>(d.counts)    // [2][CM][10] class counts per column
                                 : reinterpret_cast<unsigned short*>(m_raw);
    char* s_c0 = reinterpret_cast<char*>(s_cnt + 2 * (size_t)CM * 10);            // [2][CM] row-0 character per column
    unsigned char* s_mv = WIDE ? d.moves : reinterpret_cast<unsigned char*>(s_c0 + 2 * CM);    // [(CM+1)][64] traceback moves
    int* s_trace = WIDE ? d.trace : reinterpret_cast<int*>(s_mv + (CM + 1) * 64);              // [2*(CM+64)]
    __shared__ int s_ncol, s_newn, s_err;
    const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63;
    const int n = d.n;
    int cur = 0;
    if (tid == 0) { s_ncol = d.seq_off[1] - d.seq_off[0]; s_err = (s_ncol > CM ? 2 : 0) | (n > 65535 ? 8 : 0); }
    __syncthreads();
    if (!s_err) {   // first sequence: one column per character
        const int l0 = s_ncol;
        for (int c = tid; c < l0; c += nt) {
            const char ch = d.seqs[d.seq_off[0] + c];
            d.cols[0][(long)c * n + 0] = ch;
            for (int k = 0; k < 10; k++) s_cnt[c * 10 + k] = 0;
            const int cl = cls_of(ch);
            if (cl < 10) s_cnt[c * 10 + cl] = 1;
            s_c0[c] = ch;
        }
    }
    __syncthreads();
    for (int t = 1; t < n && !s_err; t++) {
        const int s = t;                                  // rows already aligned
        const int ncol = s_ncol;
        const int m = ncol + 1;
        const char* seq = d.seqs + d.seq_off[t];
        const int len = d.seq_off[t + 1] - d.seq_off[t];
        const int nn = len + 1;
        const int mvs = WIDE ? d.mv_stride : 64;          // DP columns the traceback table holds per row
        if (nn > mvs || ncol + len > CM || ncol + len > d.cmax) {
            if (tid == 0) s_err = (nn > mvs ? 1 : 0) | (ncol + len > CM ? 2 : 0) | (ncol + len > d.cmax ? 4 : 0);
            __syncthreads();
            break;
        }
        const char* colc = d.cols[cur];
        const unsigned short* cntc = s_cnt + (size_t)cur * CM * 10;
        const char* c0c = s_c0 + (size_t)cur * CM;
        // ---- forward, wave 0: lane l = DP column j = 64 * chunk + l, time step tau handles row i = tau - l.  A sequence of
        // more than 63 bases takes several chunks of 64 columns, one after the other; the last column of a chunk leaves its
        // cells in `edge` (row by row), where lane 0 of the next chunk finds its left and diagonal neighbours
        if (tid < 64)
        for (int chunk = 0; chunk * 64 < nn; chunk++) {
            const int j = chunk * 64 + lane;
            const bool more = (chunk + 1) * 64 < nn;          // another chunk follows: record the last column
            int* edge = WIDE ? d.edge : nullptr;              // [2][m]: score, state of the cells (i, 64 * chunk - 1)
            const int b = (j >= 1 && j < nn) ? cls_of(seq[j - 1]) : 10;
            int scb[9];
#pragma unroll
            for (int c = 0; c < 9; c++) scb[c] = dna_score_cls(c, b);
            const int sb_minus = dna_score_cls(9, b), sb_plus = dna_score_cls(8, b);
            int sc_up = 0;          // SC[i-1][j]
            int st_up = 0;          // state of cell (i-1, j): 0 mat, 1 ins, 2 del
            // row 0: SC[0][j] = s*(-6) + (j-1)*s*(-2), state ins for j >= 1, mat at j = 0
            if (j >= 1) { sc_up = s * (-6) + (j - 1) * s * (-2); st_up = 1; }
            int sc_left_prev = 0, st_left_prev = 0;   // cell (i-1, j-1) as delivered last step
            if (chunk > 0 && lane == 0) { sc_left_prev = s * (-6) + (j - 2) * s * (-2); st_left_prev = 1; }     // cell (0, j-1)
            const int width = (nn - chunk * 64 < 64) ? nn - chunk * 64 : 64;      // DP columns of this chunk
            for (int tau = 1; tau < m + width - 1; tau++) {
                const int i = tau - lane;
                // values of cell (i, j-1) computed by lane l-1 in the previous step (lane 0 of a later chunk: by the chunk before)
                int sc_l = __shfl_up(sc_up, 1);            // lane l-1's current (i, j-1) sits in its sc_up
                int st_l = __shfl_up(st_up, 1);
                if (WIDE && chunk > 0 && lane == 0 && i < m) { sc_l = edge[i]; st_l = edge[d.cmax + 1 + i]; }
                int sc_new = sc_up, st_new = st_up;
                if (i >= 1 && i < m && j < nn) {
                    const unsigned short* cnt = cntc + (i - 1) * 10;
                    if (j == 0) {
                        int sp = 0;
                        const int y = (i == 1) ? 8 : 9;
#pragma unroll
                        for (int c = 0; c < 10; c++) sp += (int)cnt[c] * dna_score_cls(c, y);
                        sc_new = sc_up + sp;
                        st_new = 3;                       // per-row state, never ins and never uniform-del
                    } else {
                        const int nd = cnt[9];
                        const char c0 = c0c[i - 1];
                        int r1 = 0, r3 = 0;
                        const int y3 = (st_up == 2) ? 9 : 8;
#pragma unroll
                        for (int c = 0; c < 9; c++) {
                            const int k = cnt[c];
                            r1 += k * scb[c];             // diagonal: cell (i-1, j-1)
                            r3 += k * dna_score_cls(c, y3);   // delete: cell (i-1, j)
                        }
                        r1 += nd * (st_left_prev == 1 ? sb_minus : sb_plus) + sc_left_prev;
                        r3 += nd * 3 + sc_up;             // score('-','-')
                        const int r2 = s * (st_l == 1 ? sb_minus : sb_plus) + sc_l;   // insert: cell (i, j-1)
                        unsigned char mv;
                        if (r1 >= r2 && r1 >= r3) { sc_new = r1; st_new = (c0 == '-') ? 1 : 0; mv = 0; }
                        else if (r2 >= r1 && r2 >= r3) { sc_new = r2; st_new = 1; mv = 1; }
                        else { sc_new = r3; st_new = (c0 == '-') ? 0 : 2; mv = 2; }
                        s_mv[(size_t)i * mvs + j] = mv;
                    }
                    if (WIDE && more && lane == 63) { edge[i] = sc_new; edge[d.cmax + 1 + i] = st_new; }     // read 63 steps ago by this chunk's lane 0
                }
                // what lane j-1 held BEFORE this step is cell (i-1, j-1) for the next step
                sc_left_prev = sc_l; st_left_prev = st_l;
                if (i >= 1 && i < m && j < nn) { sc_up = sc_new; st_up = st_new; }
            }
        }
        __syncthreads();
        // ---- traceback, MultipleSequenceAlignmentSP.cpp:252-301 (thread 0)
        if (tid == 0) {
            int x = m - 1, y = nn - 1, cnt = 0;
            int r1 = ncol - 1, r2 = len - 1;
            while (!(x == 0 && y == 0)) {
                int mv;
                if (x == 0) mv = 1; else if (y == 0) mv = 2; else mv = s_mv[(size_t)x * mvs + y];
                if (mv == 0) { s_trace[2 * cnt] = r1; s_trace[2 * cnt + 1] = r2; --r1; --r2; --x; --y; }
                else if (mv == 1) { s_trace[2 * cnt] = -1; s_trace[2 * cnt + 1] = r2; --r2; --y; }
                else { s_trace[2 * cnt] = r1; s_trace[2 * cnt + 1] = -1; --r1; --x; }
                cnt++;
            }
            s_newn = cnt;
        }
        __syncthreads();
        // ---- rebuild columns (reversed traceback order); counts follow incrementally
        const int newn = s_newn;
        char* coln = d.cols[cur ^ 1];
        unsigned short* cntn = s_cnt + (size_t)(cur ^ 1) * CM * 10;
        char* c0n = s_c0 + (size_t)(cur ^ 1) * CM;
        for (long idx = tid; idx < (long)newn * (s + 1); idx += nt) {
            const int c = (int)(idx / (s + 1)), k = (int)(idx % (s + 1));
            const int src = s_trace[2 * (newn - 1 - c)], sj = s_trace[2 * (newn - 1 - c) + 1];
            char ch;
            if (k < s) ch = (src >= 0) ? colc[(long)src * n + k] : '-';
            else ch = (sj >= 0) ? seq[sj] : '-';
            coln[(long)c * n + k] = ch;
        }
        for (int c = tid; c < newn; c += nt) {
            const int src = s_trace[2 * (newn - 1 - c)], sj = s_trace[2 * (newn - 1 - c) + 1];
            const int cl = cls_of((sj >= 0) ? seq[sj] : '-');
            for (int k = 0; k < 10; k++) {
                int v = (src >= 0) ? (int)cntc[src * 10 + k] : (k == 9 ? s : 0);
                if (k == cl) v += 1;
                cntn[c * 10 + k] = (unsigned short)v;
            }
            c0n[c] = (src >= 0) ? c0c[src] : '-';
        }
        if (tid == 0) s_ncol = newn;
        cur ^= 1;
        __syncthreads();
    }
    if (tid == 0) { *d.ncol_out = s_ncol; *d.err_out = s_err | (cur << 8); }
}

// --------------------------------------------------------------------------
// a5: threading of the reads along the backbone (PartialOrderGraph.cpp:94-255,
// the per-base M loop :129-177).  A read base aligned to reference position i
// with symbol c lands in node class (i, c): the backbone node if c is the
// reference base, else the "mis" sibling for that symbol.  The kernels bucket all
// M-aligned bases of a packed read batch into those classes:
//   k_thread_count  one wavefront per read: class sizes, first read of every class
//                   (it creates the sibling), first read of every class-to-class
//                   transition / read start / read end (it adds the edge)
//   k_thread_scan   exclusive scan of the class sizes
//   k_thread_fill   read ids into the class pools
//   k_thread_sort   each pool into read order (the order the reference appends in)
// The host stitches nodes and edges from these tables in first-touch order and
// only walks the reads that contain insertions or deletions (sc_graph.cpp).

template <bool FILL>
__global__ __launch_bounds__(256) void k_thread_walk(ThreadDev d) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int r = wave; r < d.n_reads; r += nwaves) {
        const int c0 = d.cig_off[r], c1 = d.cig_off[r + 1];
        const int s0 = d.seq_off[r], slen = d.seq_off[r + 1] - s0;
        int i = d.pos[r], j = 0;
        bool prev_m = false;
        for (int k = c0; k < c1; k++) {
            const char op = d.cig_op[k];
            const int len = d.cig_len[k];
            if (op == 'M') {
                if (i + len > d.glen || j + len > slen) { if (lane == 0) atomicOr(d.err, 1); break; }
                for (int t = lane; t < len; t += 64) {
                    const int c = d.lut[(unsigned char)d.seq[s0 + j + t]];
                    const int cls = (i + t) * 8 + c;
                    if (!FILL) {
                        atomicAdd(&d.count[cls], 1);
                        atomicMin(&d.minrid[cls], r);
                        if (t > 0 || prev_m) {
                            const int cp = d.lut[(unsigned char)d.seq[s0 + j + t - 1]];
                            atomicMin(&d.tmin[(i + t) * 64 + cp * 8 + c], r);
                        } else if (k == c0) {
                            atomicMin(&d.smin[cls], r);
                        }
                        if (t == len - 1 && k == c1 - 1) atomicMin(&d.emin[cls], r);
                    } else {
                        const int p = atomicAdd(&d.cursor[cls], 1);
                        d.pool[d.off[cls] + p] = r;
                    }
                }
                i += len; j += len; prev_m = true;
            } else if (op == 'I') { j += len; prev_m = false; }
            else if (op == 'D') { i += len; prev_m = false; }
            else { if (lane == 0) atomicOr(d.err, 2); break; }
        }
    }
}

__global__ __launch_bounds__(1024) void k_thread_scan(const int* __restrict__ count, int* __restrict__ off, int n) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = tid * per, e = min(n, b + per);
    int sum = 0;
    for (int k = b; k < e; k++) sum += count[k];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int k = 0; k < 1024; k++) { const int v = part[k]; part[k] = acc; acc += v; } off[n] = acc; }
    __syncthreads();
    int acc = part[tid];
    for (int k = b; k < e; k++) { off[k] = acc; acc += count[k]; }
}

// every pool into ascending read order (ids are distinct inside a class).  Reads are sorted by
// start position, so the reads of one class span a short id range: a bitmap of that range in LDS
// gives every read its rank with two popcounts.  A class whose reads span more ids than a wavefront's
// bitmap holds (deep coverage: 59 000 reads over every position of configs[3]) goes on the list of
// k_thread_sort_big, which gives it a whole workgroup and a bitmap of half a million ids.  (Until round 3 such a class was
// ranked by comparing every read with every other: 9.3 s of the 19 s of the unthinned configs[3] region.)
constexpr int SORT_WORDS = 512;
__global__ __launch_bounds__(256) void k_thread_sort(const int* __restrict__ off, const int* __restrict__ in, int* __restrict__ out, int ncls,
                                                     int* __restrict__ big) {
    constexpr int WORDS = SORT_WORDS;                // 16 384 ids per wavefront
    __shared__ unsigned bits[4][WORDS];
    __shared__ int wpre[4][WORDS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int c = wave; c < ncls; c += nwaves) {
        const int b = off[c], n = off[c + 1] - b;
        if (n <= 0) continue;
        if (n == 1) { if (lane == 0) out[b] = in[b]; continue; }
        int lo = 0x7fffffff, hi = -1;
        for (int x = lane; x < n; x += 64) { const int v = in[b + x]; lo = min(lo, v); hi = max(hi, v); }
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if (hi - lo < WORDS * 32) {
            const int nw = ((hi - lo) >> 5) + 1;
            for (int k = lane; k < nw; k += 64) bits[w][k] = 0u;
            __builtin_amdgcn_wave_barrier();
            for (int x = lane; x < n; x += 64) { const int v = in[b + x] - lo; atomicOr(&bits[w][v >> 5], 1u << (v & 31)); }
            __builtin_amdgcn_wave_barrier();
            // exclusive prefix of the word popcounts (nw <= 512: eight words per lane)
            int run = 0;
            for (int k0 = 0; k0 < nw; k0 += 64) {
                const int k = k0 + lane;
                const int pc = (k < nw) ? __popc(bits[w][k]) : 0;
                int incl = pc;
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
                if (k < nw) wpre[w][k] = run + incl - pc;
                run += __shfl(incl, 63);
            }
            __builtin_amdgcn_wave_barrier();
            for (int x = lane; x < n; x += 64) {
                const int rid = in[b + x], v = rid - lo;
                const int rank = wpre[w][v >> 5] + __popc(bits[w][v >> 5] & ((1u << (v & 31)) - 1u));
                out[b + rank] = rid;
            }
            __builtin_amdgcn_wave_barrier();
        } else if (lane == 0) {
            big[1 + atomicAdd(&big[0], 1)] = c;
        }
    }
}

// The wide classes: one workgroup per class and pass over [lo, hi] in stretches of BIG_WORDS * 32 ids -- bits of the
// stretch's reads, exclusive prefix of the word popcounts (sixteen words per thread, then a scan over the threads), every
// read of the stretch to its rank.  Linear in the pool for the ranges that occur (one stretch up to 524 288 ids).
constexpr int BIG_WORDS = 16384;
__global__ __launch_bounds__(1024) void k_thread_sort_big(const int* __restrict__ off, const int* __restrict__ in, int* __restrict__ out,
                                                          const int* __restrict__ big, int words) {
    extern __shared__ unsigned s_big_raw[];
    unsigned* bits = s_big_raw;                                  // [BIG_WORDS]
    int* wpre = reinterpret_cast<int*>(s_big_raw + BIG_WORDS);   // [BIG_WORDS]
    __shared__ int s_part[1024];
    __shared__ int s_lo, s_hi, s_base;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int nbig = big[0];
    for (int bi = blockIdx.x; bi < nbig; bi += gridDim.x) {
        const int c = big[1 + bi];
        const int b = off[c], n = off[c + 1] - b;
        if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; s_base = 0; }
        __syncthreads();
        int lo = 0x7fffffff, hi = -1;
        for (int x = tid; x < n; x += nt) { const int v = in[b + x]; lo = min(lo, v); hi = max(hi, v); }
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
        if ((tid & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
        __syncthreads();
        lo = s_lo; hi = s_hi;
        const long span = (long)words * 32;                      // ids per stretch (words <= BIG_WORDS; smaller only in tests)
        for (long c0 = lo; c0 <= hi; c0 += span) {
            const long c1 = c0 + span;                           // this stretch: ids [c0, c1)
            const int nw = (int)(((c1 <= hi ? c1 - 1 : (long)hi) - c0) >> 5) + 1;
            const int base = s_base;                             // reads of the class in the stretches before this one (written after the last barrier of a stretch, read before its first)
            for (int k = tid; k < nw; k += nt) bits[k] = 0u;
            __syncthreads();
            for (int x = tid; x < n; x += nt) {
                const long v = (long)in[b + x] - c0;
                if (v >= 0 && v < span) atomicOr(&bits[v >> 5], 1u << (v & 31));
            }
            __syncthreads();
            // exclusive prefix of the word popcounts: a thread's sixteen words, then the threads
            constexpr int PER = BIG_WORDS / 1024;
            int mine = 0;
#pragma unroll
            for (int j = 0; j < PER; j++) { const int k = tid * PER + j; if (k < nw) mine += __popc(bits[k]); }
            s_part[tid] = mine;
            __syncthreads();
            if (tid < 64) {
                // 1024 partial sums: sixteen per lane of the first wavefront, then across the lanes
                int loc[16], tot = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) { loc[j] = tot; tot += s_part[tid * 16 + j]; }
                int incl = tot;
                for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (tid >= o) incl += t; }
                const int excl = incl - tot;
#pragma unroll
                for (int j = 0; j < 16; j++) s_part[tid * 16 + j] = excl + loc[j];
            }
            __syncthreads();
            {
                int run = s_part[tid];
#pragma unroll
                for (int j = 0; j < PER; j++) { const int k = tid * PER + j; if (k < nw) { wpre[k] = run; run += __popc(bits[k]); } }
            }
            __syncthreads();
            for (int x = tid; x < n; x += nt) {
                const int rid = in[b + x];
                const long v = (long)rid - c0;
                if (v >= 0 && v < span) out[b + base + wpre[v >> 5] + __popc(bits[v >> 5] & ((1u << (v & 31)) - 1u))] = rid;
            }
            if (tid == nt - 1) { const int k = nw - 1; s_base = base + wpre[k] + __popc(bits[k]); }
            __syncthreads();
        }
    }
}

// --------------------------------------------------------------------------
// host-callable launchers (called from sc_api.cpp)
void launch_edge_support(hipStream_t st, const int* out_ptr, const int* out_node, const int* pool_ptr, const int* pool_rid,
                         const int* pool_cn, const uint8_t* node_is_end, const int* edge_src, int n_edges, int sorted,
                         int* support) {
    if (n_edges <= 0) return;
    int waves_per_block = 4;
    int blocks = (n_edges + waves_per_block - 1) / waves_per_block;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_edge_support, dim3(blocks), dim3(256), 0, st, out_ptr, out_node, pool_ptr, pool_rid, pool_cn,
                       node_is_end, edge_src, n_edges, sorted, support);
}
constexpr size_t LEVEL_LDS = LDS_TOTAL;        // the log tables / histogram of S strains over K symbols: S * K * K doubles
constexpr size_t CHAIN_LDS = LDS_TOTAL;
template <int NB, bool L> static int set_sample_attr() {
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_level_sample<NB, L>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS);
}
int init_kernels() {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_level), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LEVEL_LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_any), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_level_resident), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msa<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MSA_LDS);
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_thread_sort_big), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_WORDS * 8);
    rc |= set_sample_attr<1, true>(); rc |= set_sample_attr<1, false>();
    rc |= set_sample_attr<2, true>(); rc |= set_sample_attr<2, false>();
    rc |= set_sample_attr<3, true>(); rc |= set_sample_attr<3, false>();
    rc |= set_sample_attr<4, true>(); rc |= set_sample_attr<4, false>();
    rc |= set_sample_attr<5, true>(); rc |= set_sample_attr<5, false>();
    rc |= set_sample_attr<6, true>(); rc |= set_sample_attr<6, false>();
    rc |= set_sample_attr<7, true>(); rc |= set_sample_attr<7, false>();
    rc |= set_sample_attr<8, true>(); rc |= set_sample_attr<8, false>();
    return rc;
}
// doubles of LDS a level can spend on the strains' log tables / the soft update's histogram: S * K * K must fit
int level_table_capacity() { return LDS_BIG / (int)sizeof(double); }
// A level of ordinary size is ONE launch (launch_level_batch).  Only when a level has so many (strain, read) items or
// such long rows that a single workgroup would crawl (unthinned deep coverage) do the row copies and the
// single-symbol update run on a grid first; `Pd` is then a device copy of the parameters the caller has put
// in front of these launches on the same stream.  Returns the LV_* bits to pass on in LevelHdr::done.
constexpr long GRID_ITEMS = 1L << 17;          // (strain, read) items of a level
constexpr long GRID_COPY_WORDS = 1L << 19;     // doubles copied for new strains
constexpr long GRID_HARD = 1L << 18;           // (strain, draw slot) pairs of a hard update
static bool hard_on_grid(const LevelHdr& h) {
    // (behind the grid's update only: the pieces of a level keep their order)
    return h.mode == MODE_HARD && h.do_update && (long)h.S * (h.e1 - h.e0) > GRID_ITEMS && !h.has_dups && !h.any_multi &&
           (long)h.S * h.Q > GRID_HARD && h.e1 > h.e0;
}
bool level_wants_grid(const JobDev& job, const LevelHdr& h) {
    const long items = (long)h.S * (h.e1 - h.e0);
    return (h.do_update && items > GRID_ITEMS && !h.has_dups && !h.any_multi) || (long)h.n_copy * job.n_reads > GRID_COPY_WORDS;
}
int launch_level_grid(hipStream_t st, const JobDev& job, const LevelHdr& h, const LevelParams* Pd, LevelResult* R) {
    const int S = h.S, Rn = h.e1 - h.e0;
    int done = 0;
    const bool update_on_grid = h.do_update && (long)S * Rn > GRID_ITEMS && !h.has_dups && !h.any_multi;
    // The rows of the level's new candidates come first, whoever makes them: a candidate's row must be its parent's row of
    // BEFORE this level's update.  (Until round 3 the update could go to the grid while a few small copies stayed with the
    // level's own kernel, which then copied the parent's already updated row over the child's: the child was scored with its
    // parent's symbol at this level.  Levels of more than 131 072 (candidate, read) items with fewer than 2^19 / reads new
    // candidates -- tests/golden/wide_cap120 found it at 121 candidates x 1 085 reads.)
    if (h.n_copy > 0 && ((long)h.n_copy * job.n_reads > GRID_COPY_WORDS || update_on_grid)) {
        const int n2 = (job.n_reads + 1) >> 1;
        int bx = (n2 + 1023) / 1024;
        bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
        hipLaunchKernelGGL(k_level_copy, dim3(bx, h.n_copy), dim3(256), 0, st, job, Pd);
        done |= LV_COPIES_DONE;
    }
    const long items = (long)S * Rn;
    if (update_on_grid) {
        int g = (int)((items + 511) / 512);
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        hipLaunchKernelGGL(k_level_update, dim3(g), dim3(256), 0, st, job, h, Pd);
        done |= LV_ITEMS_DONE;
        int gh = (Rn + 255) / 256;
        hipLaunchKernelGGL(k_level_has, dim3(gh < 1 ? 1 : (gh > 1024 ? 1024 : gh)), dim3(256), 0, st, job, h);
        done |= LV_HAS_DONE;
        if (hard_on_grid(h)) {
            int gq = (h.Q + 255) / 256;
            gq = gq < 1 ? 1 : (gq > 2048 ? 2048 : gq);
            hipLaunchKernelGGL(k_hard_slots, dim3(gh < 1 ? 1 : (gh > 1024 ? 1024 : gh)), dim3(256), 0, st, job, h);
            hipLaunchKernelGGL(k_hard_mates, dim3(gq), dim3(256), 0, st, job, h, Pd);
            hipLaunchKernelGGL(k_hard_resp, dim3(gq), dim3(256), 0, st, job, h, Pd);
            hipLaunchKernelGGL(k_hard_sums, dim3(S), dim3(64), 0, st, job, h, Pd, R);
            done |= LV_HARD_DONE;
        }
    }
    return done;
}
// One launch = the current level of `n` regions whose levels need the same kernel (workgroup b = batch.it[b]).
// kind 0: no sampler (k_level); kind 1 + 2 * (NB - 1) + L: the sampler variant for NB = ceil(S / 16) register
// blocks, L = weight rows fit in LDS.
int level_kind(const LevelHdr& h) {
    const int S = h.S, Q = h.Q, Rn = h.e1 - h.e0;
    const bool chain = h.mode == MODE_SAMPLE && h.n_sweeps > 0 && S > 1 && Rn > 0;
    if (!chain) return 0;
    static const bool rows_lds_allowed = !(getenv("SC_ROWS_LDS") && atoi(getenv("SC_ROWS_LDS")) == 0);     // measurements: rows in HBM everywhere
    const bool wl = rows_lds_allowed && ((long)Q * chain_w_stride(S) + 16) * 4 <= (long)CHAINW_ROWS_BYTES;
    int nb = (S + 15) / 16;
    nb = nb < 1 ? 1 : (nb > 8 ? 8 : nb);
    return 1 + 2 * (nb - 1) + (wl ? 1 : 0);
}
// LDS a level needs, in KB: the per-strain scalars, then whichever is larger of the strains' log tables (S * K * K
// doubles, staged for the update; the soft update's histogram has the same shape) and the sampler's uniforms + weight
// rows.  A launch asks for the largest need among its items instead of the whole CU's LDS, so that two (small sampler
// levels) to four (levels without sampler) workgroups share a CU once more levels are in flight than the GPU has CUs.
int level_lds_kb(const LevelHdr& h, int K) {
    const int kind = level_kind(h);
    const long S = h.S, Rn = h.e1 - h.e0;
    long big = 0;
    if ((h.do_update && Rn > 0) || h.mode == MODE_HARD) big = (long)sizeof(double) * S * K * K;
    if (kind != 0) {
        long rows = (long)UWIN * 4;
        if ((kind - 1) & 1) rows += ((long)h.Q * chain_w_stride(h.S) + 16) * 4;
        if (rows > big) big = rows;
    }
    long need = LDS_SMALL + big + 64;
    if (need > LDS_TOTAL) need = LDS_TOTAL;
    return (int)((need + 1023) / 1024);
}
static size_t batch_lds(const LevelBatch& b, int n) {
    static const bool exact = !(getenv("SC_LDS_EXACT") && atoi(getenv("SC_LDS_EXACT")) == 0);
    if (!exact) return LDS_TOTAL;
    int kb = 0;
    for (int i = 0; i < n; i++) { const int k = (b.it[i].kind >> 8) & 0xFFFF; kb = k > kb ? k : kb; }
    size_t bytes = (size_t)kb * 1024;
    if (kb == 0 || bytes > (size_t)LDS_TOTAL) bytes = LDS_TOTAL;
    return bytes;
}
// every item carries its kind (LevelItem::kind, low byte) and its LDS need in KB (the bits above)
void launch_level_any(hipStream_t st, const LevelBatch& b, int n) {
    hipLaunchKernelGGL(k_level_any, dim3(n), dim3(CHAIN_THREADS), batch_lds(b, n), st, b);
}
void launch_level_batch(hipStream_t st, int kind, const LevelBatch& b, int n) {
    const size_t lds = batch_lds(b, n);
    if (kind == 0) {
        hipLaunchKernelGGL(k_level, dim3(n), dim3(512), lds, st, b);
        return;
    }
    const bool wl = (kind - 1) & 1;
#define SC_SAMPLE(NB) case NB: if (wl) hipLaunchKernelGGL((k_level_sample<NB, true>), dim3(n), dim3(CHAIN_THREADS), lds, st, b); \
                               else hipLaunchKernelGGL((k_level_sample<NB, false>), dim3(n), dim3(CHAIN_THREADS), lds, st, b); break;
    switch ((kind - 1) / 2 + 1) {
        SC_SAMPLE(1) SC_SAMPLE(2) SC_SAMPLE(3) SC_SAMPLE(4) SC_SAMPLE(5) SC_SAMPLE(6) SC_SAMPLE(7)
        default: if (wl) hipLaunchKernelGGL((k_level_sample<8, true>), dim3(n), dim3(CHAIN_THREADS), lds, st, b);
                 else hipLaunchKernelGGL((k_level_sample<8, false>), dim3(n), dim3(CHAIN_THREADS), lds, st, b);
    }
#undef SC_SAMPLE
}
// one grid of `slots` resident workgroups, each with the whole LDS of its CU (every variant must fit)
void launch_resident(hipStream_t st, const ResidentArgs& a, int slots) {
    hipLaunchKernelGGL(k_level_resident, dim3(slots), dim3(CHAIN_THREADS), CHAIN_LDS, st, a);
}
void launch_msa(hipStream_t st, const MsaDev& d) {
    if (d.cmax > MSA_CM || d.mv_stride > 64) hipLaunchKernelGGL(k_msa<true>, dim3(1), dim3(256), 0, st, d);      // state in HBM scratch
    else hipLaunchKernelGGL(k_msa<false>, dim3(1), dim3(256), MSA_LDS, st, d);
}
// a5 in four launches; `pool_sorted` receives the class pools in read order.
void launch_thread(hipStream_t st, const ThreadDev& d, int* pool_sorted) {
    const int ncls = d.glen * 8;
    int blocks = (d.n_reads + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_thread_walk<false>), dim3(blocks), dim3(256), 0, st, d);
    hipLaunchKernelGGL(k_thread_scan, dim3(1), dim3(1024), 0, st, d.count, d.off, ncls);
    hipLaunchKernelGGL((k_thread_walk<true>), dim3(blocks), dim3(256), 0, st, d);
    int sblocks = (ncls + 3) / 4;
    if (sblocks > 2048) sblocks = 2048;
    hipLaunchKernelGGL(k_thread_sort, dim3(sblocks), dim3(256), 0, st, d.off, d.pool, pool_sorted, ncls, d.big);
    // (no class of a region with fewer reads than a wavefront's bitmap has ids can be wide)
    static const int big_words = [] {                          // SC_SORT_BIG_WORDS: a short stretch, so that a test reaches the second one
        const char* e = getenv("SC_SORT_BIG_WORDS");
        const int w = e ? atoi(e) : BIG_WORDS;
        return w < 64 ? 64 : (w > BIG_WORDS ? BIG_WORDS : w);
    }();
    if (d.n_reads > SORT_WORDS * 32)
        hipLaunchKernelGGL(k_thread_sort_big, dim3(512), dim3(1024), BIG_WORDS * 8, st, d.off, d.pool, pool_sorted, d.big, big_words);
}

}  // namespace sc
#ifdef SC_CHAIN_PROF
extern "C" int sc_debug_chain_prof(unsigned long long* out) {       // experiment builds only
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sc::g_chain_prof), sizeof(unsigned long long) * 12);
}
#endif

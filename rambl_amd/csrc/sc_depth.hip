// Stage 1 of rambl.py on the device: per-base depth summed over the samples' alignment files, merged into intervals
// with their mean depth -- /root/reference/scripts/coverage_all_samples.py:21-186, which pipes
//   samtools depth <bams>  |  awk (sum the per-file columns)  |  sort  |  bedtools merge -c 4 -o mean -d 10
// through temp files.  Here:
//   k_depth_mark      every aligned run [s, e] of every read (CIGAR M = X; deletions and skips do not count, as in
//                     samtools depth) adds +1 at s and -1 at e + 1 of one difference array over all references
//   k_depth_segments  one wavefront per reference streams its part of the array once (16-byte loads, four positions
//                     per lane), turns it into depths by a running prefix sum, and cuts the covered positions into
//                     intervals wherever more than `max_gap` uncovered positions lie between two covered ones
//                     (bedtools merge -d on the one-base records [p, p + 1)); per interval: start, end, sum of the
//                     depths and number of covered positions (the mean is sum / n)
// The segment kernel is the HBM-bound one: 4 bytes per reference base in, a few intervals per reference out.
// References never share reads, so every reference starts at depth 0 and there is no carry between wavefronts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/straincall_hip.h"
#include "sc_ingest.hpp"

namespace {

struct Interval { int ref, start, end, n; long long sum; };
constexpr int FIXED = 2;           // intervals of a reference that have their own output slots

__global__ __launch_bounds__(256) void k_depth_mark(const unsigned* __restrict__ run_s, const unsigned* __restrict__ run_e, long n_runs,
                                                    int* __restrict__ diff) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_runs; i += (long)gridDim.x * blockDim.x) {
        atomicAdd(&diff[run_s[i]], 1);
        atomicAdd(&diff[run_e[i] + 1u], -1);
    }
}

// wave64 prefix operations on the DPP lanes (no LDS round trips): row shifts inside the rows of 16, then the two
// row broadcasts of gfx9
template <int CTRL, int ROW_MASK, bool BOUND> __device__ __forceinline__ int dpp_i32(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, BOUND);
}
__device__ __forceinline__ int wave_excl_scan_i32(int v, int& total) {
    int x = v;
    x += dpp_i32<0x111, 0xF, true>(0, x);       // row_shr:1
    x += dpp_i32<0x112, 0xF, true>(0, x);       // row_shr:2
    x += dpp_i32<0x114, 0xF, true>(0, x);       // row_shr:4
    x += dpp_i32<0x118, 0xF, true>(0, x);       // row_shr:8
    x += dpp_i32<0x142, 0xA, false>(0, x);      // row_bcast:15 -> rows 1, 3
    x += dpp_i32<0x143, 0xC, false>(0, x);      // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}
__device__ __forceinline__ int wave_excl_max_i32(int v, int lowest, int& all) {
    int x = v;
    x = max(x, dpp_i32<0x111, 0xF, false>(lowest, x));
    x = max(x, dpp_i32<0x112, 0xF, false>(lowest, x));
    x = max(x, dpp_i32<0x114, 0xF, false>(lowest, x));
    x = max(x, dpp_i32<0x118, 0xF, false>(lowest, x));
    x = max(x, dpp_i32<0x142, 0xA, false>(lowest, x));
    x = max(x, dpp_i32<0x143, 0xC, false>(lowest, x));
    all = __builtin_amdgcn_readlane(x, 63);
    return dpp_i32<0x138, 0xF, false>(lowest, x);   // wave_shr:1: lane i gets lane i - 1, lane 0 keeps `lowest`
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return __shfl(v, 0);
}

// PPL positions per lane and step: 4 (one 16-byte load) while no interval can break inside four consecutive
// positions (max_gap >= 3), else 1.
template <int PPL>
__global__ __launch_bounds__(256) void k_depth_segments(const int* __restrict__ diff, const unsigned* __restrict__ ref_off,
                                                        const int* __restrict__ ref_len, int n_refs, int max_gap,
                                                        Interval* __restrict__ out_ref, int* __restrict__ ref_count,
                                                        Interval* __restrict__ out_more, int cap_more, int* __restrict__ n_more) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int NONE = -(1 << 30);
    for (int ref = wave; ref < n_refs; ref += nwaves) {
        const unsigned base = ref_off[ref];
        const int len = ref_len[ref];
        int carry = 0;                 // depth in front of the step's first position
        int last_cov = NONE;           // last covered position so far (wave-uniform)
        int seg_start = NONE, seg_end = NONE;      // the open interval
        long long acc_sum = 0;         // its depth sum and covered positions, lane-private parts
        int acc_n = 0;
        // the first FIXED intervals of a reference go to its own slots (one shared counter for every interval of every
        // reference is one address for tens of thousands of wavefronts: the atomics serialise); the rare rest to a shared list
        int n_emitted = 0;
        auto emit = [&](int s, int e, long long sum, int n) {
            if (lane == 0) {
                if (n_emitted < FIXED) out_ref[(size_t)ref * FIXED + n_emitted] = Interval{ref, s, e, n, sum};
                else {
                    const int k = atomicAdd(n_more, 1);
                    if (k < cap_more) out_more[k] = Interval{ref, s, e, n, sum};
                }
            }
            n_emitted++;
        };
        // the cells of the step after this one are loaded while this one is worked on
        auto fetch = [&](int c) __attribute__((always_inline)) -> int4 {
            const int q = c + PPL * lane;
            if (q >= len) return make_int4(0, 0, 0, 0);
            if (PPL == 4) return *reinterpret_cast<const int4*>(diff + base + q);
            return make_int4(diff[base + q], 0, 0, 0);
        };
        int4 nxt = fetch(0);
        for (int c0 = 0; c0 < len; c0 += 64 * PPL) {
            const int p0 = c0 + PPL * lane;
            const int4 v = nxt;
            nxt = fetch(c0 + 64 * PPL);
            int d[PPL];
            d[0] = v.x;
            if (PPL == 4) { d[1 % PPL] = v.y; d[2 % PPL] = v.z; d[3 % PPL] = v.w; }
            int run = 0, loc[PPL];
#pragma unroll
            for (int k = 0; k < PPL; k++) { run += d[k]; loc[k] = run; }
            int total;
            const int excl = wave_excl_scan_i32(run, total) + carry;
            carry += total;
            int first = -1, last = -1, pn = 0;
            long long psum = 0;
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const int dep = excl + loc[k];
                const bool cov = (p0 + k < len) && dep > 0;
                if (cov) { if (first < 0) first = k; last = k; pn++; psum += dep; }
            }
            const bool has = pn > 0;
            int chunk_last;
            const int prev_last = max(last_cov, wave_excl_max_i32(has ? p0 + last : NONE, NONE, chunk_last));
            const bool starts = has && (p0 + first - prev_last > max_gap + 1);        // [prev, prev+1) and [p, p+1) merge iff p - prev - 1 <= max_gap
            const unsigned long long B = __ballot(starts);
            if (B == 0ull) {
                acc_sum += psum; acc_n += pn;
                if (chunk_last > seg_end) seg_end = chunk_last;
            } else {
                // one or more intervals start inside this step (rare): lanes in front of the first start still belong to the open one
                unsigned long long rest = B;
                int from = 0;                                  // first lane not yet assigned
                while (true) {
                    const int j = rest ? (int)__builtin_ctzll(rest) : 64;      // next start lane (64: none left)
                    const bool mine = lane >= from && lane < j;
                    const long long s_part = wave_sum_i64(acc_sum + (mine ? psum : 0));
                    const long long n_part = wave_sum_i64((long long)acc_n + (mine ? pn : 0));
                    int e_part = mine && has ? p0 + last : NONE;
                    for (int o = 32; o > 0; o >>= 1) e_part = max(e_part, __shfl_xor(e_part, o));
                    e_part = max(e_part, seg_end);
                    if (j == 64) {
                        // the interval that stays open: keep its sums in lane 0's private part
                        acc_sum = lane == 0 ? s_part : 0; acc_n = lane == 0 ? (int)n_part : 0; seg_end = e_part;
                        break;
                    }
                    if (seg_start != NONE) emit(seg_start, e_part, s_part, (int)n_part);
                    acc_sum = 0; acc_n = 0;
                    seg_start = __shfl(p0 + first, j); seg_end = NONE;
                    from = j;
                    rest &= rest - 1;
                }
            }
            if (chunk_last > last_cov) last_cov = chunk_last;
        }
        if (seg_start != NONE) {
            const long long s_all = wave_sum_i64(acc_sum);
            const long long n_all = wave_sum_i64((long long)acc_n);
            emit(seg_start, seg_end, s_all, (int)n_all);
        }
        if (lane == 0) ref_count[ref] = n_emitted < FIXED ? n_emitted : FIXED;
    }
}

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, std::max<size_t>(n, 16)) == hipSuccess; }
};

}  // namespace

extern "C" {

int sc_depth_scan_runs(int device, const int* ref_len, int n_refs, const int* run_ref, const int* run_start, const int* run_end,
                       long n_runs, int max_gap, int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap,
                       int* n_intervals, sc_depth_stats* stats) {
    if (!ref_len || n_refs < 0 || n_runs < 0 || (n_runs > 0 && (!run_ref || !run_start || !run_end)) || max_gap < 0 || !n_intervals) return SC_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SC_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SC_ERR_HIP;
    // one coordinate space over all references: each starts at a multiple of 4 cells, with room for the -1 behind its last base
    std::vector<unsigned> off((size_t)n_refs + 1, 0);
    unsigned long long cells = 0;
    for (int r = 0; r < n_refs; r++) {
        if (ref_len[r] < 0) return SC_ERR_ARG;
        off[(size_t)r] = (unsigned)cells;
        cells += ((unsigned long long)ref_len[r] + 1 + 3) & ~3ull;
        if (cells > 0xFFFFFFF0ull) return SC_ERR_CAPACITY;
    }
    off[(size_t)n_refs] = (unsigned)cells;
    std::vector<unsigned> rs((size_t)n_runs), re((size_t)n_runs);
    for (long i = 0; i < n_runs; i++) {
        const int r = run_ref[i];
        if (r < 0 || r >= n_refs || run_start[i] < 1 || run_end[i] < run_start[i] || run_end[i] > ref_len[r]) return SC_ERR_ARG;      // 1-based, inclusive
        rs[(size_t)i] = off[(size_t)r] + (unsigned)(run_start[i] - 1);
        re[(size_t)i] = off[(size_t)r] + (unsigned)(run_end[i] - 1);
    }
    DevMem d_diff, d_rs, d_re, d_off, d_len, d_out, d_n, d_fix, d_cnt;
    const size_t diff_bytes = sizeof(int) * ((size_t)cells + 8);
    if (!d_diff.alloc(diff_bytes) || !d_rs.alloc(sizeof(unsigned) * (size_t)n_runs) || !d_re.alloc(sizeof(unsigned) * (size_t)n_runs) ||
        !d_off.alloc(sizeof(unsigned) * ((size_t)n_refs + 1)) || !d_len.alloc(sizeof(int) * (size_t)n_refs) ||
        !d_out.alloc(sizeof(Interval) * (size_t)std::max(cap, 1)) || !d_n.alloc(sizeof(int)) ||
        !d_fix.alloc(sizeof(Interval) * (size_t)std::max(n_refs, 1) * FIXED) || !d_cnt.alloc(sizeof(int) * (size_t)std::max(n_refs, 1)))
        return SC_ERR_HIP;
    hipStream_t st = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = SC_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = SC_ERR_HIP; };
    chk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto& e : ev) chk(hipEventCreate(&e));
    if (rc == SC_OK) {
        if (n_runs > 0) {
            chk(hipMemcpyAsync(d_rs.p, rs.data(), sizeof(unsigned) * (size_t)n_runs, hipMemcpyHostToDevice, st));
            chk(hipMemcpyAsync(d_re.p, re.data(), sizeof(unsigned) * (size_t)n_runs, hipMemcpyHostToDevice, st));
        }
        chk(hipMemcpyAsync(d_off.p, off.data(), sizeof(unsigned) * ((size_t)n_refs + 1), hipMemcpyHostToDevice, st));
        if (n_refs > 0) chk(hipMemcpyAsync(d_len.p, ref_len, sizeof(int) * (size_t)n_refs, hipMemcpyHostToDevice, st));
        chk(hipMemsetAsync(d_n.p, 0, sizeof(int), st));
        chk(hipEventRecord(ev[0], st));
        chk(hipMemsetAsync(d_diff.p, 0, diff_bytes, st));
        if (n_runs > 0) {
            const int blocks = (int)std::min<long>((n_runs + 255) / 256, 1 << 16);
            hipLaunchKernelGGL(k_depth_mark, dim3(blocks), dim3(256), 0, st, (const unsigned*)d_rs.p, (const unsigned*)d_re.p, n_runs, (int*)d_diff.p);
        }
        chk(hipEventRecord(ev[1], st));
        if (n_refs > 0) {
            const int blocks = std::min((n_refs + 3) / 4, 1 << 16);         // four wavefronts per workgroup, one reference per wavefront
            chk(hipEventRecord(ev[2], st));
            if (max_gap >= 3)
                hipLaunchKernelGGL(k_depth_segments<4>, dim3(blocks), dim3(256), 0, st, (const int*)d_diff.p, (const unsigned*)d_off.p,
                                   (const int*)d_len.p, n_refs, max_gap, (Interval*)d_fix.p, (int*)d_cnt.p, (Interval*)d_out.p, cap, (int*)d_n.p);
            else
                hipLaunchKernelGGL(k_depth_segments<1>, dim3(blocks), dim3(256), 0, st, (const int*)d_diff.p, (const unsigned*)d_off.p,
                                   (const int*)d_len.p, n_refs, max_gap, (Interval*)d_fix.p, (int*)d_cnt.p, (Interval*)d_out.p, cap, (int*)d_n.p);
            chk(hipEventRecord(ev[3], st));
        }
        int n_more = 0;
        std::vector<int> cnt((size_t)n_refs, 0);
        std::vector<Interval> fix((size_t)n_refs * FIXED);
        chk(hipMemcpyAsync(&n_more, d_n.p, sizeof(int), hipMemcpyDeviceToHost, st));
        if (n_refs > 0) {
            chk(hipMemcpyAsync(cnt.data(), d_cnt.p, sizeof(int) * (size_t)n_refs, hipMemcpyDeviceToHost, st));
            chk(hipMemcpyAsync(fix.data(), d_fix.p, sizeof(Interval) * (size_t)n_refs * FIXED, hipMemcpyDeviceToHost, st));
        }
        chk(hipStreamSynchronize(st));
        chk(hipGetLastError());
        long n = n_more;
        for (int r = 0; r < n_refs; r++) n += cnt[(size_t)r];
        *n_intervals = (int)std::min<long>(n, 0x7fffffffL);
        if (rc == SC_OK && n > cap) rc = SC_ERR_CAPACITY;
        if (rc == SC_OK && n > 0) {
            if (!iv_ref || !iv_start || !iv_end || !iv_sum || !iv_n) rc = SC_ERR_ARG;
            else {
                std::vector<Interval> iv;
                iv.reserve((size_t)n);
                for (int r = 0; r < n_refs; r++) for (int k = 0; k < cnt[(size_t)r]; k++) iv.push_back(fix[(size_t)r * FIXED + k]);
                if (n_more > 0) {
                    const size_t at = iv.size();
                    iv.resize(at + (size_t)n_more);
                    chk(hipMemcpy(iv.data() + at, d_out.p, sizeof(Interval) * (size_t)n_more, hipMemcpyDeviceToHost));
                }
                std::sort(iv.begin(), iv.end(), [](const Interval& a, const Interval& b) { return a.ref != b.ref ? a.ref < b.ref : a.start < b.start; });
                for (long i = 0; i < n; i++) {
                    iv_ref[i] = iv[(size_t)i].ref; iv_start[i] = iv[(size_t)i].start + 1; iv_end[i] = iv[(size_t)i].end + 1;      // 1-based, inclusive
                    iv_sum[i] = (long)iv[(size_t)i].sum; iv_n[i] = iv[(size_t)i].n;
                }
            }
        }
        if (stats && rc == SC_OK) {
            float ms = 0;
            std::memset(stats, 0, sizeof *stats);
            stats->cells = (long)cells; stats->runs = n_runs;
            if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) stats->mark_ms = ms;
            if (n_refs > 0 && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) stats->segments_ms = ms;
        }
    }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

int sc_depth_scan(int device, sc_aln* const* alns, int n_alns, const char* const* ref_names, const int* ref_len, int n_refs, int max_gap,
                  int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap, int* n_intervals, sc_depth_stats* stats) {
    if (!alns || n_alns < 0 || !ref_names || !ref_len) return SC_ERR_ARG;
    std::vector<int> rr, rs, re;
    for (int f = 0; f < n_alns; f++) {
        if (!alns[f]) return SC_ERR_ARG;
        for (int r = 0; r < n_refs; r++) {
            auto it = alns[f]->by_ref.find(ref_names[r]);
            if (it == alns[f]->by_ref.end()) continue;
            for (const sc_ingest::Rec& rec : it->second) {
                if (rec.flag & 0x704) continue;                      // samtools depth: unmapped, secondary, QC fail, duplicate
                long v = 0;
                int p = rec.pos;
                for (int i = 0; i < rec.clen; i++) {
                    const char ch = rec.cigar[i];
                    if (ch >= '0' && ch <= '9') { v = v * 10 + (ch - '0'); continue; }
                    if (ch == 'M' || ch == '=' || ch == 'X') {
                        const int a = std::max(p, 1), b = std::min(p + (int)v - 1, ref_len[r]);
                        if (a <= b) { rr.push_back(r); rs.push_back(a); re.push_back(b); }
                        p += (int)v;
                    } else if (ch == 'D' || ch == 'N') {
                        p += (int)v;                                  // a deleted / skipped base is no depth
                    }
                    v = 0;
                }
            }
        }
    }
    return sc_depth_scan_runs(device, ref_len, n_refs, rr.data(), rs.data(), re.data(), (long)rr.size(), max_gap, iv_ref, iv_start, iv_end,
                              iv_sum, iv_n, cap, n_intervals, stats);
}

}  // extern "C"

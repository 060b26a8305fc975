// Stage 1 of rambl.py on the device: per-base depth summed over the samples' alignment files, merged into intervals
// with their mean depth -- /root/reference/scripts/coverage_all_samples.py:21-186, which pipes
//   samtools depth <bams>  |  awk (sum the per-file columns)  |  sort  |  bedtools merge -c 4 -o mean -d 10
// through temp files.  Here ONE kernel, k_depth_fused, one wavefront per reference:
//   * the aligned runs [s, e] of the reference's reads (CIGAR M = X; deletions and skips do not count, as in samtools
//     depth) arrive bucketed by reference, 8 bytes each; the wavefront adds +1 at s and -1 at e + 1 of a difference array
//     that lives in ITS OWN LDS tile (2 048 cells: a 16S gene fits one tile; a longer reference is walked tile by tile,
//     the runs of a tile found by binary search in its start-sorted runs, a run that began in an earlier tile clamped
//     to the tile's first cell) -- no array in HBM, no global atomics, nothing to clear;
//   * then it turns the tile into depths by a running prefix sum (16-byte LDS reads, four positions per lane, DPP scans)
//     and cuts the covered positions into intervals wherever more than `max_gap` uncovered positions lie between two
//     covered ones (bedtools merge -d on the one-base records [p, p + 1)); per interval: start, end, sum of the depths
//     and number of covered positions (the mean is sum / n).
// Round 2 had two kernels (k_depth_mark: two global atomics per run into a 4-byte-per-base array in HBM, 0.76 ms on 10^8
// bases; k_depth_segments streaming that array, 0.09 ms).  The array was the traffic: 8 bytes per run in and a few
// intervals per reference out is all that has to cross HBM.
// References never share reads, so every reference starts at depth 0 and there is no carry between wavefronts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/straincall_hip.h"
#include "sc_ingest.hpp"

namespace {

struct Interval { int ref, start, end, n; long long sum; };
constexpr int FIXED = 2;           // intervals of a reference that have their own output slots

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

constexpr int TILE = 2048;          // cells of a wavefront's LDS tile

// PPL positions per lane and step: 4 (one 16-byte LDS read) while no interval can break inside four consecutive
// positions (max_gap >= 3), else 1.
template <int PPL>
__global__ __launch_bounds__(256) void k_depth_fused(const uint2* __restrict__ runs, const unsigned* __restrict__ run_ptr,
                                                     const int* __restrict__ ref_len, int n_refs, int max_gap, int max_run,
                                                     Interval* __restrict__ out_ref, int* __restrict__ ref_count,
                                                     Interval* __restrict__ out_more, int cap_more, int* __restrict__ n_more) {
    __shared__ __attribute__((aligned(16))) int s_tile[4][TILE + 8];
    const int lane = threadIdx.x & 63;
    int* tile = s_tile[threadIdx.x >> 6];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    constexpr int NONE = -(1 << 30);
    for (int ref = wave; ref < n_refs; ref += nwaves) {
        const int len = ref_len[ref];
        const unsigned rp0 = run_ptr[ref], rp1 = run_ptr[ref + 1];
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
        for (int c0 = 0; c0 < len; c0 += TILE) {
            const int tl = min(TILE, len - c0);                       // positions of this tile
            // ---- the tile's difference array, in LDS
            for (int i = 4 * lane; i < tl + 4; i += 256) *reinterpret_cast<int4*>(tile + i) = make_int4(0, 0, 0, 0);
            unsigned ra = rp0, rb = rp1;
            if (len > TILE) {
                // runs are sorted by start here (the host sorts the runs of a reference longer than a tile): those that can
                // touch [c0, c0 + tl) start at c0 - max_run or later and before c0 + tl
                const unsigned lo_key = c0 > max_run ? (unsigned)(c0 - max_run) : 0u, hi_key = (unsigned)(c0 + tl);
                unsigned lo = rp0, hi = rp1;
                while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (runs[mid].x < lo_key) lo = mid + 1; else hi = mid; }
                ra = lo; hi = rp1;
                while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (runs[mid].x < hi_key) lo = mid + 1; else hi = mid; }
                rb = lo;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the zeroes are in before the first add (one wavefront: LDS keeps its order)
            for (unsigned r = ra + lane; r < rb; r += 64) {
                const uint2 se = runs[r];
                const int s0 = (int)se.x, e0 = (int)se.y;
                if (e0 >= c0 && s0 < c0 + tl) {
                    atomicAdd(&tile[(s0 > c0 ? s0 : c0) - c0], 1);   // a run that began before the tile counts from its first cell
                    if (e0 + 1 < c0 + tl) atomicAdd(&tile[e0 + 1 - c0], -1);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- depths and intervals
            int carry = 0;                 // depth in front of the step's first position (the clamped starts carry it into a tile)
            for (int q0 = 0; q0 < tl; q0 += 64 * PPL) {
                const int q = q0 + PPL * lane;
                const int p0 = c0 + q;
                int d[PPL];
                if (PPL == 4) {
                    const int4 v = q < tl ? *reinterpret_cast<const int4*>(tile + q) : make_int4(0, 0, 0, 0);
                    d[0] = v.x; d[1 % PPL] = v.y; d[2 % PPL] = v.z; d[3 % PPL] = v.w;
                } else {
                    d[0] = q < tl ? tile[q] : 0;
                }
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
                    const bool cov = (q + k < tl) && dep > 0;
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
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // every read of the tile is back before the next tile's zeroes
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
struct PinMem {
    void* p = nullptr;
    ~PinMem() { if (p) (void)hipHostFree(p); }
    bool alloc(size_t n) { return hipHostMalloc(&p, std::max<size_t>(n, 16), hipHostMallocDefault) == hipSuccess; }
};
double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int host_threads() {
    long quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long q = 0, per = 0;
        if (fscanf(f, "%ld %ld", &q, &per) == 2 && q > 0 && per > 0) quota = (q + per - 1) / per;
        fclose(f);
    }
    long n = (long)std::thread::hardware_concurrency();
    if (quota > 0 && quota < n) n = quota;
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) { const long k = atol(e); if (k > 1) n = std::max<long>(n / k, 1); }
    if (const char* e = getenv("SC_INGEST_THREADS")) n = atol(e);
    return (int)std::min<long>(std::max<long>(n, 1), 32);
}

// The device part: runs bucketed by reference (run_ptr[n_refs + 1]), 0-based inclusive (start, end) pairs; the runs of a
// reference longer than a tile sorted by start.  `runs` is page-locked.
int depth_scan_bucketed(int device, const int* ref_len, int n_refs, const unsigned* run_ptr, const uint2* runs, long n_runs, int max_run,
                        int max_gap, int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap, int* n_intervals,
                        sc_depth_stats* stats) {
    DevMem d_runs, d_ptr, d_len, d_out, d_n, d_fix, d_cnt;
    if (!d_runs.alloc(sizeof(uint2) * (size_t)n_runs) || !d_ptr.alloc(sizeof(unsigned) * ((size_t)n_refs + 1)) ||
        !d_len.alloc(sizeof(int) * (size_t)n_refs) || !d_out.alloc(sizeof(Interval) * (size_t)std::max(cap, 1)) || !d_n.alloc(sizeof(int)) ||
        !d_fix.alloc(sizeof(Interval) * (size_t)std::max(n_refs, 1) * FIXED) || !d_cnt.alloc(sizeof(int) * (size_t)std::max(n_refs, 1)))
        return SC_ERR_HIP;
    hipStream_t st = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = SC_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == SC_OK) rc = SC_ERR_HIP; };
    chk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto& e : ev) chk(hipEventCreate(&e));
    if (rc == SC_OK) {
        chk(hipEventRecord(ev[0], st));
        if (n_runs > 0) chk(hipMemcpyAsync(d_runs.p, runs, sizeof(uint2) * (size_t)n_runs, hipMemcpyHostToDevice, st));
        chk(hipMemcpyAsync(d_ptr.p, run_ptr, sizeof(unsigned) * ((size_t)n_refs + 1), hipMemcpyHostToDevice, st));
        if (n_refs > 0) chk(hipMemcpyAsync(d_len.p, ref_len, sizeof(int) * (size_t)n_refs, hipMemcpyHostToDevice, st));
        chk(hipMemsetAsync(d_n.p, 0, sizeof(int), st));
        chk(hipEventRecord(ev[1], st));
        if (n_refs > 0) {
            const int blocks = std::min((n_refs + 3) / 4, 1 << 16);         // four wavefronts per workgroup, one reference per wavefront
            chk(hipEventRecord(ev[2], st));
            if (max_gap >= 3)
                hipLaunchKernelGGL(k_depth_fused<4>, dim3(blocks), dim3(256), 0, st, (const uint2*)d_runs.p, (const unsigned*)d_ptr.p,
                                   (const int*)d_len.p, n_refs, max_gap, max_run, (Interval*)d_fix.p, (int*)d_cnt.p, (Interval*)d_out.p, cap, (int*)d_n.p);
            else
                hipLaunchKernelGGL(k_depth_fused<1>, dim3(blocks), dim3(256), 0, st, (const uint2*)d_runs.p, (const unsigned*)d_ptr.p,
                                   (const int*)d_len.p, n_refs, max_gap, max_run, (Interval*)d_fix.p, (int*)d_cnt.p, (Interval*)d_out.p, cap, (int*)d_n.p);
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
            long cells = 0;
            for (int r = 0; r < n_refs; r++) cells += ref_len[r];
            stats->cells = cells; stats->runs = n_runs;
            if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) stats->upload_ms = ms;
            if (n_refs > 0 && hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) stats->kernel_ms = ms;
        }
    }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    if (st) (void)hipStreamDestroy(st);
    return rc;
}

// runs of a reference longer than a tile in start order (the kernel finds a tile's runs by binary search)
void sort_long_refs(const int* ref_len, int n_refs, const unsigned* run_ptr, uint2* runs) {
    for (int r = 0; r < n_refs; r++)
        if (ref_len[r] > TILE)
            std::sort(runs + run_ptr[r], runs + run_ptr[r + 1], [](const uint2& a, const uint2& b) { return a.x < b.x; });
}

}  // namespace

extern "C" {

int sc_depth_scan_runs(int device, const int* ref_len, int n_refs, const int* run_ref, const int* run_start, const int* run_end,
                       long n_runs, int max_gap, int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap,
                       int* n_intervals, sc_depth_stats* stats) {
    if (!ref_len || n_refs < 0 || n_runs < 0 || (n_runs > 0 && (!run_ref || !run_start || !run_end)) || max_gap < 0 || !n_intervals) return SC_ERR_ARG;
    if (n_runs > 0xFFFFFFF0L) return SC_ERR_CAPACITY;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SC_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SC_ERR_HIP;
    if (stats) std::memset(stats, 0, sizeof *stats);
    const double t0 = wall_ms();
    for (int r = 0; r < n_refs; r++) if (ref_len[r] < 0) return SC_ERR_ARG;
    // bucket the runs by reference (a counting sort: the order inside a reference is kept)
    std::vector<unsigned> ptr((size_t)n_refs + 1, 0);
    for (long i = 0; i < n_runs; i++) {
        const int r = run_ref[i];
        if (r < 0 || r >= n_refs || run_start[i] < 1 || run_end[i] < run_start[i] || run_end[i] > ref_len[r]) return SC_ERR_ARG;      // 1-based, inclusive
        ptr[(size_t)r + 1]++;
    }
    for (int r = 0; r < n_refs; r++) ptr[(size_t)r + 1] += ptr[(size_t)r];
    PinMem pin;
    if (!pin.alloc(sizeof(uint2) * (size_t)n_runs)) return SC_ERR_HIP;
    uint2* runs = (uint2*)pin.p;
    std::vector<unsigned> cur(ptr.begin(), ptr.end() - 1);
    int max_run = 1;
    for (long i = 0; i < n_runs; i++) {
        runs[cur[(size_t)run_ref[i]]++] = make_uint2((unsigned)(run_start[i] - 1), (unsigned)(run_end[i] - 1));
        max_run = std::max(max_run, run_end[i] - run_start[i] + 1);
    }
    sort_long_refs(ref_len, n_refs, ptr.data(), runs);
    if (stats) stats->prepare_ms = wall_ms() - t0;
    return depth_scan_bucketed(device, ref_len, n_refs, ptr.data(), runs, n_runs, max_run, max_gap, iv_ref, iv_start, iv_end, iv_sum, iv_n,
                               cap, n_intervals, stats);
}

int sc_depth_scan(int device, sc_aln* const* alns, int n_alns, const char* const* ref_names, const int* ref_len, int n_refs, int max_gap,
                  int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap, int* n_intervals, sc_depth_stats* stats) {
    if (!alns || n_alns < 0 || !ref_names || !ref_len || n_refs < 0 || max_gap < 0 || !n_intervals) return SC_ERR_ARG;
    for (int f = 0; f < n_alns; f++) if (!alns[f]) return SC_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SC_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SC_ERR_HIP;
    if (stats) std::memset(stats, 0, sizeof *stats);
    // ---- the aligned runs of every reference, over all files: references are independent, so the host threads of this
    // rank take them in turns (the CIGAR walk of 10^7 records is the longest part of the stage once the device needs 0.1 ms)
    const double t0 = wall_ms();
    std::vector<std::vector<uint2>> per_ref((size_t)n_refs);
    std::atomic<int> next{0};
    std::atomic<int> max_run_all{1};
    auto work = [&]() {
        int max_run = 1;
        for (int r = next.fetch_add(1); r < n_refs; r = next.fetch_add(1)) {
            std::vector<uint2>& out = per_ref[(size_t)r];
            for (int f = 0; f < n_alns; f++) {
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
                            if (a <= b) { out.push_back(make_uint2((unsigned)(a - 1), (unsigned)(b - 1))); max_run = std::max(max_run, b - a + 1); }
                            p += (int)v;
                        } else if (ch == 'D' || ch == 'N') {
                            p += (int)v;                                  // a deleted / skipped base is no depth
                        }
                        v = 0;
                    }
                }
            }
        }
        int m = max_run_all.load();
        while (max_run > m && !max_run_all.compare_exchange_weak(m, max_run)) {}
    };
    {
        const int nt = std::max(1, std::min(host_threads(), n_refs / 64 + 1));
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto& th : pool) th.join();
    }
    if (stats) stats->extract_ms = wall_ms() - t0;
    const double t1 = wall_ms();
    std::vector<unsigned> ptr((size_t)n_refs + 1, 0);
    unsigned long long total = 0;
    for (int r = 0; r < n_refs; r++) { if (ref_len[r] < 0) return SC_ERR_ARG; total += per_ref[(size_t)r].size(); if (total > 0xFFFFFFF0ull) return SC_ERR_CAPACITY; ptr[(size_t)r + 1] = (unsigned)total; }
    PinMem pin;
    if (!pin.alloc(sizeof(uint2) * (size_t)total)) return SC_ERR_HIP;
    uint2* runs = (uint2*)pin.p;
    for (int r = 0; r < n_refs; r++)
        if (!per_ref[(size_t)r].empty()) std::memcpy(runs + ptr[(size_t)r], per_ref[(size_t)r].data(), sizeof(uint2) * per_ref[(size_t)r].size());
    sort_long_refs(ref_len, n_refs, ptr.data(), runs);
    if (stats) stats->prepare_ms = wall_ms() - t1;
    return depth_scan_bucketed(device, ref_len, n_refs, ptr.data(), runs, (long)total, max_run_all.load(), max_gap, iv_ref, iv_start, iv_end, iv_sum,
                               iv_n, cap, n_intervals, stats);
}

}  // extern "C"

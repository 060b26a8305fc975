// Host driver + C-ABI of libstraincall_hip.so.  See include/straincall_hip.h.
//
// Per region (a worker thread): build the partial order graph (sc_graph.cpp; the per-base threading of the reads and
// the insertion MSA run on the device: k_thread_*, k_msa), flatten it level-major, upload it once, compute every edge
// support on the device (k_edge_support), then walk the levels of
// /root/reference/StrainCall/NonparametricClustering.cpp:262-582.  The walk keeps only the scalar bookkeeping of the
// candidate strains on the host, in long double as the reference has it (substitution models, abundances, pruning /
// extension decisions); the per-read work of a level -- rows of new candidates, log-likelihood update, soft update or
// Polya-urn sampler -- is ONE kernel launch (k_level / k_level_sample), its parameters read from host-mapped memory,
// its results and a completion stamp written back to it; the per-strain read log-likelihood rows never leave HBM.
//
// Per context: the level server (Ctx::serve_levels), the one thread that launches level kernels and watches the stamps.
// Levels of different regions that need the same kernel leave as one grid on one of a few shared launch streams, so
// that a hundred regions in flight need no more hardware queues than the GPU runs side by side.
#include <hip/hip_runtime.h>
#include <malloc.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/straincall_hip.h"
#include "sc_device.hpp"
#include "sc_fiber.hpp"
#include "sc_graph.hpp"

namespace sc {

// launchers defined in sc_kernels.hip
void launch_edge_support(hipStream_t st, const int* out_ptr, const int* out_node, const int* pool_ptr, const int* pool_rid,
                         const int* pool_cn, const uint8_t* node_is_end, const int* edge_src, int n_edges, int sorted,
                         int* support);
bool level_wants_grid(const JobDev& job, const LevelHdr& h);
int launch_level_grid(hipStream_t st, const JobDev& job, const LevelHdr& h, const LevelParams* Pd, LevelResult* R);
int level_kind(const LevelHdr& h);
int level_lds_kb(const LevelHdr& h, int K);
int level_table_capacity();
void launch_level_batch(hipStream_t st, int kind, const LevelBatch& b, int n);
void launch_level_any(hipStream_t st, const LevelBatch& b, int n);
void launch_resident(hipStream_t st, const ResidentArgs& a, int slots);
void launch_msa(hipStream_t st, const MsaDev& d);
void launch_thread(hipStream_t st, const ThreadDev& d, int* pool_sorted);
int init_kernels();

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct ScError : std::runtime_error {
    int code;
    ScError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw HipError(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// growable device buffer
// A device buffer that only grows.  While regions are in flight nothing is handed back to the driver (hipFree waits for
// the device): an outgrown buffer is kept until the worker goes; growth is geometric, so that is at most as much again --
// nothing next to 288 GB.  (Measured: no difference to freeing at once; the stalls under load came from pageable copies,
// see PinnedArena.  SC_DEVBUF_KEEP=0 restores the old behaviour.)
struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    std::vector<void*> outgrown;
    void* ensure(size_t n) {
        if (n > cap) {
            static const bool keep = !(getenv("SC_DEVBUF_KEEP") && atoi(getenv("SC_DEVBUF_KEEP")) == 0);
            if (p) { if (keep) outgrown.push_back(p); else (void)hipFree(p); }
            size_t want = keep ? std::max<size_t>(n + n / 2 + 4096, 2 * cap) : n + n / 4 + 256;
            HIPCHK(hipMalloc(&p, want));
            cap = want;
        }
        return p;
    }
    ~DevBuf() { if (p) (void)hipFree(p); for (void* q : outgrown) (void)hipFree(q); }
};
// Pinned staging for a region's transfers.  A copy between the device and ordinary (pageable) host memory makes the
// runtime pin those pages for the copy and let them go afterwards; with regions in flight that costs far more than the
// copy -- registering and releasing user pages suspends every queue of the process (level kernels of ALL regions lasting
// ~30 ms at once, a few times per region).  So every sizeable transfer goes through page-locked memory the worker owns:
// grow-only chunks, handed out by a bump pointer, reused by the next region.
struct PinnedArena {
    struct Chunk { char* p; size_t cap, used; };
    struct Back { void* dst; const void* src; size_t n; };      // device-to-host copies still to be moved to their vectors
    std::vector<Chunk> chunks;
    std::vector<Back> back;
    bool on = true;                   // false: pass the copies through (a single region in flight suspends nobody)
    void* take(size_t n) {
        n = (n + 255) & ~(size_t)255;
        for (Chunk& c : chunks) if (c.cap - c.used >= n) { void* r = c.p + c.used; c.used += n; return r; }
        size_t cap = std::max<size_t>(n, (size_t)8 << 20);
        if (!chunks.empty()) cap = std::max(cap, 2 * chunks.back().cap);
        char* q = nullptr;
        HIPCHK(hipHostMalloc((void**)&q, cap, hipHostMallocDefault));
        chunks.push_back(Chunk{q, cap, n});
        return q;
    }
    void reset() { for (Chunk& c : chunks) c.used = 0; back.clear(); }
    void h2d(void* dst, const void* src, size_t n, hipStream_t st) {
        if (n == 0) return;
        if (!on) { HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st)); return; }
        void* q = take(n);
        memcpy(q, src, n);
        HIPCHK(hipMemcpyAsync(dst, q, n, hipMemcpyHostToDevice, st));
    }
    void d2h(void* dst, const void* src, size_t n, hipStream_t st) {          // complete after the stream's synchronisation + land()
        if (n == 0) return;
        if (!on) { HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, st)); return; }
        void* q = take(n);
        HIPCHK(hipMemcpyAsync(q, src, n, hipMemcpyDeviceToHost, st));
        back.push_back(Back{dst, q, n});
    }
    void land() { for (const Back& b : back) memcpy(b.dst, b.src, b.n); back.clear(); }
    ~PinnedArena() { for (Chunk& c : chunks) (void)hipHostFree(c.p); }
};
template <class T> static T* upload(PinnedArena& ar, DevBuf& b, const std::vector<T>& v, hipStream_t st) {
    T* d = (T*)b.ensure(std::max<size_t>(v.size(), 1) * sizeof(T));
    ar.h2d(d, v.data(), v.size() * sizeof(T), st);
    return d;
}

// mt19937(1234) -> generate_canonical<double,53>: the stream every sampler call
// of the reference starts from (NonparametricClustering.cpp:142,785)
static std::vector<double> uniform_stream(unsigned seed, int n) {
    std::vector<uint32_t> x(624);
    x[0] = seed;
    for (int i = 1; i < 624; i++) x[i] = 1812433253u * (x[i - 1] ^ (x[i - 1] >> 30)) + (uint32_t)i;
    int p = 624;
    auto next = [&]() -> uint32_t {
        if (p >= 624) {
            const uint32_t UP = 0x80000000u, LO = 0x7fffffffu;
            for (int k = 0; k < 624 - 397; ++k) { uint32_t y = (x[k] & UP) | (x[k + 1] & LO); x[k] = x[k + 397] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
            for (int k = 624 - 397; k < 623; ++k) { uint32_t y = (x[k] & UP) | (x[k + 1] & LO); x[k] = x[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
            uint32_t y = (x[623] & UP) | (x[0] & LO);
            x[623] = x[396] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
            p = 0;
        }
        uint32_t z = x[p++];
        z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);
        return z;
    };
    std::vector<double> u(n);
    for (int i = 0; i < n; i++) {
        double sum = 0.0, tmp = 1.0;
        for (int k = 2; k != 0; --k) { sum += (double)next() * tmp; tmp *= 4294967296.0; }
        double r = sum / tmp;
        if (r >= 1.0) r = std::nextafter(1.0, 0.0);
        u[i] = r;
    }
    return u;
}

// ---------------------------------------------------------------------------
struct Job {
    int handle = 0;
    // inputs
    std::string ref;
    std::vector<AlignedRead> reads;
    std::vector<int> mate_off, mate_idx;
    sc_params params{};
    // outputs
    std::vector<std::string> seqs;
    std::vector<double> abund;
    std::string graph_dump, trace;
    std::vector<int> edge_support;
    std::vector<int> thr_count, thr_first, thr_pool;
    std::string thr_sym;
    sc_stats stats{};
    double t_submit = 0;
    int status = 0;       // 0 queued/running, 1 done
    int rc = SC_OK;
    std::string err;
};

typedef long double ld;               // the reference keeps abundances and counts in DoubleL = long double (x87 80-bit)
// Host bookkeeping of one candidate (Strain, PartialOrderGraph.hpp:362-402).  The substitution model is the bulky
// part and lives in a pool: a candidate that survives a level keeps its model where it is (no copy when the candidate
// lists are filtered, sorted or extended), only the second and later children of a parent get a copy.
struct Model {
    // The tables are [ks][ks] with ks = the symbols of the region (6 for a gene of A C G T and its reads): a region in flight
    // comes back to its models once per level, after a few hundred other regions have used the core's caches, so what a
    // level touches is kept small and contiguous (rows of 16 entries spread the 36 live ones over four times the lines).
    int ks = KMAX;
    ld sub[KK];                       // sub_count over the symbol table, stride ks
    ld comp[6]; ld Z;
    ld lsub[KK];                      // logl(sub), valid where `stale` is clear
    double lpc[KK];                   // log sub(a,b) - log comp(a) as the device reads it; rows in `dirty` are stale
    uint16_t stale[KMAX];             // per row: entries whose count changed since lsub was formed
    unsigned dirty;
    Model() = default;
    Model(const Model& o) { *this = o; }
    Model& operator=(const Model& o) {            // the live [ks][ks] part only
        ks = o.ks; Z = o.Z; dirty = o.dirty;
        const size_t n = (size_t)ks * ks;
        std::memcpy(sub, o.sub, sizeof(ld) * n); std::memcpy(lsub, o.lsub, sizeof(ld) * n); std::memcpy(lpc, o.lpc, sizeof(double) * n);
        std::memcpy(comp, o.comp, sizeof comp); std::memcpy(stale, o.stale, sizeof stale);
        return *this;
    }
};
struct HStrain {
    ld abundance;
    int model;                        // index into the model pool
    int slot;                         // row of the device read_loglik matrix
    int tail;                         // path arena index
    int node;                         // last node of the path
    uint64_t hash; int seqlen;        // rolling hash / length of strain_seq()
};
struct PathRec { int node, parent; };

struct Worker;
// A level waiting for its launch: the worker's slot, the level's scalars and which kernel it needs.
struct LevelRequest { Worker* w; LevelItem item; int kind; bool timed; };
// Launch streams are shared by all regions in flight.  A stream carries one batch at a time (`busy` = regions of that
// batch whose stamp has not been seen yet), so kernels of different regions never queue behind each other: a level
// that finds every stream busy waits in `pending` and leaves with the next batch of its kind.
struct LaunchStream { hipStream_t st = nullptr; int busy = 0; int unretired = 0; };
enum { GEN_STOPPED = 0, GEN_RUNNING = 1, GEN_STOPPING = 2 };
constexpr int KIND_POSTED = -1;        // LevelRequest::kind of a level already in its slot's mailbox: the server only watches its stamp
struct Ctx {
    int device = 0;
    // page-locked staging arenas, shared: a region holds one only while it is set up, so a handful serves any number in flight
    std::mutex amu;
    std::vector<PinnedArena*> arenas, free_arenas;
    int arena_limit = 0;              // 0: staging off (one region in flight, or SC_PINNED_STAGING=0)
    PinnedArena* lease_arena(PinnedArena* passthrough);
    void release_arena(PinnedArena* a);
    std::string last_error;
    std::mutex mu;                    // guards queue, jobs, idle, stop, last_error, job status
    std::condition_variable cv_done;
    std::deque<std::shared_ptr<Job>> queue;
    std::map<int, std::shared_ptr<Job>> jobs;
    int next_handle = 1;
    bool stop = false;
    // Regions in flight are fibers (sc_fiber.hpp): `workers` are the slots (stream_count of them, each with its device
    // buffers and its host-mapped parameter / result blocks), `pool` the few host threads that run whichever of them
    // is ready -- sized from the CPU quota of this rank, not from the number of regions in flight.
    std::vector<std::unique_ptr<Worker>> workers;
    std::vector<Worker*> idle;                // slots without a region, parked
    std::unique_ptr<FiberPool> pool;
    std::atomic<int> fibers_left{0};
    // Regions being set up (graph construction: tens of milliseconds of one CPU each) at any one time: a part of the
    // executor threads only, so that the others stay free for the continuations of the regions in flight.
    double t_created = 0; int n_fast = 0, n_long = 0;
    std::atomic<long>* wake_hist = nullptr;    // SC_SERVER_LOG: wake latencies, bucket b = below 2^b us
    std::mutex smu;                   // set-up places: regions that found none park in line (they used to go round the scheduler)
    int setups = 0;
    std::deque<Worker*> setup_waiters;
    int setup_limit = 1;
    bool split_exec = false;                  // the pool has threads of its own for the set-ups
    void setup_enter(Worker* w);
    void setup_leave();
    LevelParams* P_all = nullptr;             // host-mapped blocks of all slots (one allocation each)
    LevelResult* R_all = nullptr;
    LevelParams* Pd_all = nullptr;
    std::vector<LaunchStream> lstreams;
    std::vector<hipStream_t> setup_streams;   // uploads, graph kernels: shared round-robin by the workers
    // the level server: one thread launches every level and sees every completion stamp (serve_levels)
    sc::SpinLock plk;                         // guards pending (a few nanoseconds per level from every executor: never a sleeping lock)
    std::deque<LevelRequest> pending;         // requests the server has not taken yet
    std::atomic<int> n_pending{0};
    std::mutex dmu;                           // the server sleeps here (dcv) while nothing is pending or in flight
    std::condition_variable dcv;
    std::atomic<bool> server_asleep{false};
    std::atomic<bool> server_stop{false};
    std::thread server;
    void submit_level(const LevelRequest& rq);
    void serve_levels();
    // Resident contexts without a level server (SC_POLL_EXEC, the default): a worker whose level is in its mailbox raises
    // its flag and parks; the continuation threads look at the flagged workers' stamps between two fibers and while they
    // spin for one (FiberPool::set_poll), and the thread that sees a stamp makes the region ready.  The CPU the server spent
    // going round the stamps is an executor's.
    bool poll_exec = false;
    std::unique_ptr<std::atomic<uint8_t>[]> polled;      // [workers]: 1 = parked until its level's stamp arrives
    bool poll_stamps();                                  // true: some worker is still waiting for its stamp
    void poll_health();                                  // the heart thread, once a second: flagged workers whose workgroup has gone
    double* dU = nullptr;             // uniform stream on the device
    float* dUf = nullptr;             // fp32 copy
    // Resident level workers (k_level_resident): while regions are in flight one workgroup per slot stays on its CU and
    // takes the slot's levels from a mailbox in host-mapped memory; no launch per level.  A "generation" of the grid lives
    // from the first level posted after an idle period until no region is in flight any more (so that a device
    // synchronisation by the caller never waits on it), or until the context goes.
    bool resident = false;
    int res_slots = 0;                // mailboxes = workgroups of the grid: regions that can WALK at a time
    // A region needs a mailbox only while it walks its levels; its set-up (graph, uploads) happens on a worker of its own
    // before that.  A context has more workers than mailboxes, so the next regions are set up while every workgroup is busy,
    // and a workgroup that finishes a region finds the next one ready (mailboxes are handed from region to region).
    std::mutex mmu;
    std::vector<int> free_mail;               // mailboxes nobody walks on
    std::deque<Worker*> mail_waiters;         // regions whose set-up is done, parked until a mailbox falls free
    std::vector<unsigned> mail_seq, mail_done;     // per mailbox: last stamp posted / seen completed
    int acquire_mailbox(Worker* w);
    void release_mailbox(int m);
    Mailbox* mail_h = nullptr; Mailbox* mail_d = nullptr;
    ResidentCtl* ctl_h = nullptr; ResidentCtl* ctl_d = nullptr;
    hipStream_t rstream = nullptr;
    std::mutex gen_mu;
    std::atomic<int> gen_state{0};    // GEN_*; written under gen_mu
    std::atomic<int> regions_active{0};
    long generations = 0;
    std::thread heart;                // keeps ResidentCtl::heartbeat moving while the context lives
    std::atomic<bool> heart_stop{false};
    void resident_ensure(Worker* w);
    void resident_idle();
    void resident_shutdown();
};

// CPUs this rank may use: the cgroup quota when there is one (a GPU box hands out a share of its host), divided among the
// ranks that share the host (one process per GPU: LOCAL_WORLD_SIZE, set by torch.distributed.run and by bench.py).
static double cpu_budget_host() {
    double n = (double)std::thread::hardware_concurrency();
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64] = {0};
        double period = 0;
        if (fscanf(f, "%63s %lf", quota, &period) == 2 && std::strcmp(quota, "max") != 0 && period > 0) {
            const double q = atof(quota) / period;
            if (q > 0 && (n <= 0 || q < n)) n = q;
        }
        fclose(f);
    }
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0 && k < n) n = k; }
    return n > 1 ? n : 1;
}
// The CPUs next to a GPU: `local_cpulist` of its PCI device (the cores of the socket its root port hangs on), within what
// the process may use.  A GPU box is a two-socket host whose scheduler moves a rank's threads over both; the level
// mailboxes, the completion stamps and the host-mapped parameter blocks are read and written across PCIe by both sides
// several hundred thousand times a second, and from the far socket every one of those crosses the socket link as well.
static bool gpu_local_cpus(int device, cpu_set_t* out) {
    const char* e = getenv("SC_NUMA_BIND");
    if (e && atoi(e) == 0) return false;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) return false;
    for (char* c = bdf; *c; c++) *c = (char)tolower((unsigned char)*c);
    FILE* f = fopen((std::string("/sys/bus/pci/devices/") + bdf + "/local_cpulist").c_str(), "r");
    if (!f) return false;
    char line[4096] = {0};
    const bool got = fgets(line, sizeof line, f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed, local;
    CPU_ZERO(&local);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
    for (const char* c = line; *c && *c != '\n';) {                       // "0-63,128-191"
        char* end = nullptr;
        const long a = strtol(c, &end, 10);
        if (end == c) break;
        long b = a;
        c = end;
        if (*c == '-') { b = strtol(c + 1, &end, 10); c = end; }
        for (long k = a; k <= b && k < CPU_SETSIZE; k++) if (k >= 0 && CPU_ISSET((int)k, &allowed)) CPU_SET((int)k, &local);
        if (*c == ',') c++;
    }
    if (CPU_COUNT(&local) == 0) return false;
    *out = local;
    return true;
}
static int local_world_size() {
    const char* e = getenv("LOCAL_WORLD_SIZE");
    const int k = e ? atoi(e) : 1;
    return k > 1 ? k : 1;
}
}  // namespace sc
// Host threads a context with `stream_count` regions in flight starts on a rank that shares its host with
// `local_world - 1` others (0: read LOCAL_WORLD_SIZE), given `cpus` CPUs for the host (0: the cgroup quota / affinity
// mask): out[0] executor threads (they run the regions' fibers), out[1] the level server, out[2] ingest threads of
// sc_aln_open.  Pure arithmetic (no device): tests/test_stage5.py checks that 8 ranks on 16 CPUs stay within them.
// Binds the calling thread -- and every thread it starts afterwards -- to the CPUs next to GPU `device` (what a launcher does
// with `numactl --cpunodebind` per rank).  Returns how many CPUs that is; 0 when the topology is not known, the device does not
// exist or SC_NUMA_BIND=0: nothing is changed then.  sc_ctx_create does the same for the threads and the host memory of the
// context itself and leaves its caller where it was.
extern "C" int sc_host_bind(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return 0;
    cpu_set_t local;
    if (!sc::gpu_local_cpus(device, &local)) return 0;
    if (sched_setaffinity(0, sizeof local, &local) != 0) return 0;
    return CPU_COUNT(&local);
}
extern "C" int sc_host_plan(int stream_count, int local_world, double cpus, int* out) {
    if (!out || stream_count < 1) return SC_ERR_ARG;
    double n = cpus > 0 ? cpus : sc::cpu_budget_host();
    n /= local_world > 0 ? local_world : sc::local_world_size();
    if (n < 1) n = 1;
    int exec = (int)n - 1;                     // one CPU for the level server
    if (exec < 1) exec = 1;
    if (exec > stream_count) exec = stream_count;
    if (exec > 32) exec = 32;
    out[0] = exec; out[1] = stream_count > 1 ? 1 : 0;
    out[2] = (int)std::min<double>(std::max<double>(n, 1), 32);
    return SC_OK;
}
namespace sc {

struct Worker {
    Ctx* ctx;
    Fiber* fib = nullptr;
    int slot = 0;                     // worker index
    hipStream_t st = nullptr;         // a setup stream of the context (not owned), or a private one (own_stream)
    bool own_stream = false;
    hipEvent_t sync_ev = nullptr;     // marks "everything this worker has put on `st` so far" (sync_stream)
    // hand-shake with the level server: 1 = a level is on its way / in flight, 2 = its stamp was seen, 3 = failed
    std::atomic<int> level_state{0};
    unsigned level_want = 0;          // stamp of that level
    int cur_stream = -1;              // its launch stream (server's bookkeeping)
    std::string level_err;
    double t_seen = 0, wake_acc[2] = {0, 0};
    int mslot = -1;                   // the mailbox (= workgroup of the resident grid) this region walks on, -1 while it has none
    double t_posted = 0;              // when the level went into the slot's mailbox (resident workers)
    double t_batch_launched = 0;      // diagnostics: when the level's batch was launched, and its size
    int batch_n = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;     // the pair of the level being launched (from ev_pool when timing)
    std::vector<hipEvent_t> ev_pool;  // want_timing: one pair per sampler level of the region, read when the region is done
    size_t ev_used = 0;
    LevelParams* Ph = nullptr;        // host-mapped: written here, read by the level's kernel over PCIe
    LevelParams* Pm = nullptr;        //   its device address
    LevelParams* Pd = nullptr;        // device copy, only for the grid kernels of very large levels
    LevelResult* Rh = nullptr;        // host-mapped, written by the kernel, stamped last
    LevelResult* Rd = nullptr;
    bool own_blocks = false;          // Ph / Pd / Rh are this worker's own allocations (sc_msa_align's private worker)
    unsigned seq = 0;                 // stamp of the last level launched
    DevBuf b_ent_rid, b_ent_cn, b_ent_lab_off, b_ent_lab_len, b_ent_first, b_ent_qoff, b_labels, b_mate_ptr, b_mate_idx,
        b_ll, b_has, b_isnew, b_tabA, b_tabLf, b_qcode, b_qent, b_quid, b_out_ptr, b_out_node, b_pool_ptr, b_pool_rid,
        b_pool_cn, b_isend, b_esrc, b_support, b_jobdev;
    DevBuf m_seqs, m_off, m_cols0, m_cols1, m_counts, m_moves, m_trace, m_out, m_edge;
    PinnedArena* stage = nullptr;     // page-locked staging of the region's uploads / downloads: leased from the context
                                      // for the region's set-up (Ctx::lease_arena), handed back when its copies have landed
    PinnedArena passthrough;          // on = false
    DevBuf t_ref, t_pos, t_seqoff, t_seq, t_cigoff, t_cigop, t_ciglen, t_lut, t_tabs, t_pool, t_pool2;
    FlatGraph flat;                   // the level-major arrays of the region being set up / walked
    std::vector<int> ent_qoff_buf;
    bool setup_held = false;          // this region holds one of the context's set-up places
    std::vector<ld> cnt_scratch;      // [MAXS][KMAX] draws per (strain, read symbol) of the level just sampled

    void init();
    void run();
    void process(Job& job);
    void wait_level();
    void sync_stream();
    int msa_device(const std::vector<std::string>& seqs, std::vector<std::string>& rows);
    void thread_device(const std::string& G, const std::vector<AlignedRead>& R, const std::vector<std::vector<CigarOp>>& cig,
                       ThreadTables& T);
    void cluster(Job& job, const PoGraph& g, FlatGraph& f);
};

void Worker::init() {
    HIPCHK(hipSetDevice(ctx->device));
    if (!st) { HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); own_stream = true; }
    HIPCHK(hipEventCreateWithFlags(&sync_ev, hipEventDisableTiming));
    if (!Ph) {
        own_blocks = true;
        HIPCHK(hipHostMalloc((void**)&Ph, sizeof(LevelParams), hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(hipMalloc((void**)&Pd, sizeof(LevelParams)));
        HIPCHK(hipHostMalloc((void**)&Rh, sizeof(LevelResult), hipHostMallocMapped | hipHostMallocCoherent));
    }
    HIPCHK(hipHostGetDevicePointer((void**)&Pm, Ph, 0));
    HIPCHK(hipHostGetDevicePointer((void**)&Rd, Rh, 0));
    std::memset(Rh, 0, sizeof(LevelResult));
    cnt_scratch.assign((size_t)MAXS * KMAX, 0);
}

// Everything this worker has put on its set-up stream is done.  The stream is shared with other regions, so the wait is
// for an event recorded now, not for the stream to drain; a fiber lets the other ready regions run meanwhile.
void Worker::sync_stream() {
    if (!FiberPool::in_fiber() || ctx->workers.size() <= 1) { HIPCHK(hipStreamSynchronize(st)); return; }
    HIPCHK(hipEventRecord(sync_ev, st));
    const double t0 = now_ms();
    for (unsigned spins = 0;; spins++) {
        const hipError_t e = hipEventQuery(sync_ev);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) throw HipError(std::string("set-up stream: ") + hipGetErrorString(e));
        // a set-up stream that makes no progress for minutes is stuck (e.g. queued behind something that waits for this
        // region): an error for this region, not a hang of the process
        if ((spins & 0x3FFu) == 0x3FFu && now_ms() - t0 > 180000.0) throw HipError("set-up stream: no progress for 3 minutes");
        // the copies / kernels waited for take from 0.1 to a few milliseconds: while nothing else is ready, this thread sleeps a
        // little instead of going round the scheduler (its lock is the one the level server makes regions ready under)
        if (ctx->pool->ready_now() == 0) std::this_thread::sleep_for(std::chrono::microseconds(40));
        FiberPool::yield();
    }
}

// The level server.  Workers hand their next level to this thread and park; it is the only thread that launches level
// kernels and the only one that watches the completion stamps, so a finished level is seen within a microsecond
// however many regions are in flight, and nobody polls or contends for the launch path.
//   * A level's kernel stores its stamp into host memory after everything else it reports (system-scope release):
//     completion is seen without a stream synchronisation; the region's fiber is made ready and an executor thread
//     picks it up (no futex round trip per level).
//   * Launch streams are shared by all regions.  A stream carries one batch at a time, so kernels of different
//     regions never queue behind each other; while a stream is free, every waiting level (up to MAXB) leaves as one
//     grid, workgroup b = region b of the batch: the kernel of their kind when they all need the same one, k_level_any
//     (which calls the variant each item names) otherwise.  SC_ANY_KIND=0 keeps one kind per launch (measurements).
//   * A launch the runtime rejects fails the levels of its batch at once; a stream that drains while stamps of its
//     batch are still missing (a kernel that ended without stamping) fails them at the periodic check.
PinnedArena* Ctx::lease_arena(PinnedArena* passthrough) {
    if (arena_limit <= 0) { passthrough->on = false; return passthrough; }
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(amu);
            if (!free_arenas.empty()) { PinnedArena* a = free_arenas.back(); free_arenas.pop_back(); a->reset(); return a; }
            if ((int)arenas.size() < arena_limit) { PinnedArena* a = new PinnedArena(); arenas.push_back(a); return a; }
        }
        // every arena is with a region that is being set up: let those regions run
        if (FiberPool::in_fiber()) FiberPool::yield(); else std::this_thread::yield();
    }
}
void Ctx::release_arena(PinnedArena* a) {
    std::lock_guard<std::mutex> lk(amu);
    free_arenas.push_back(a);
}
void Ctx::serve_levels() {
    (void)hipSetDevice(device);
    std::deque<LevelRequest> waiting;          // taken from `pending`, not launched yet
    std::vector<Worker*> flying;               // launched, stamp not seen yet
    std::string dead;                          // non-empty: a launch stream has failed, every level fails from now on
    auto finish = [this](Worker* w, int state, const std::string& err) {
        w->level_err = err;
        w->t_seen = now_ms();
        w->level_state.store(state, std::memory_order_release);
        pool->make_ready(w->fib);
    };
    auto stamped = [](Worker* w) { return __atomic_load_n(&w->Rh->seq, __ATOMIC_ACQUIRE) == w->level_want; };
    // a resident workgroup that has left (heartbeat limit, or a fault that ended the grid) or never started will not stamp
    auto check_resident = [&]() {
        if (!resident) return;
        for (size_t i = 0; i < flying.size();) {
            Worker* w = flying[i];
            if (w->mslot < 0) { ++i; continue; }
            const unsigned ms = __atomic_load_n(&mail_h[w->mslot].state, __ATOMIC_ACQUIRE);
            const bool never = ms == 0u && now_ms() - w->t_posted > 20000.0;       // more slots than the GPU holds resident
            if (w->cur_stream >= 0 || (ms < 2u && !never) || stamped(w)) { ++i; continue; }
            flying[i] = flying.back(); flying.pop_back();
            finish(w, 3, never ? "the slot's resident level worker has not started within 20 s (more slots than the GPU holds resident workgroups?)"
                         : ms == 3u ? "the slot's resident level worker received an item that was not its own"
                                    : "the slot's resident level worker has left before the level was done");
        }
    };
    unsigned loops = 0;
    unsigned idle_spins = 0;
    double t_check = now_ms();
    const bool sweep_log = getenv("SC_SERVER_LOG") != nullptr;          // diagnostics: how long one round of the loop takes while levels fly
    double sweep_t0 = 0, sweep_sum = 0, sweep_max = 0; long sweep_n = 0, sweep_fly = 0;
    const bool any_kind = !(getenv("SC_ANY_KIND") && atoi(getenv("SC_ANY_KIND")) == 0);
    for (;;) {
        if (n_pending.load(std::memory_order_seq_cst) > 0) {
            int took = 0;
            plk.lock();
            while (!pending.empty()) {
                if (pending.front().kind == KIND_POSTED) { pending.front().w->cur_stream = -1; flying.push_back(pending.front().w); }
                else waiting.push_back(pending.front());
                pending.pop_front();
                took++;
            }
            plk.unlock();
            n_pending.fetch_sub(took, std::memory_order_seq_cst);
        } else if (waiting.empty() && flying.empty()) {
            // nothing to watch: sleep until a region hands a level in (announce first, then look again: submit_level looks at
            // the flag after it has counted its request)
            std::unique_lock<std::mutex> lk(dmu);
            server_asleep.store(true, std::memory_order_seq_cst);
            if (n_pending.load(std::memory_order_seq_cst) == 0 && !server_stop.load(std::memory_order_seq_cst)) dcv.wait_for(lk, std::chrono::milliseconds(50));
            server_asleep.store(false, std::memory_order_seq_cst);
            if (server_stop.load(std::memory_order_seq_cst)) break;
            continue;
        }
        if ((++loops & 0xFFFFu) == 0 && resident && now_ms() - t_check > 2000.0) { t_check = now_ms(); check_resident(); }
        bool progressed = false;
        if (sweep_log) {
            const double t = now_ms();
            if (sweep_t0 > 0 && !flying.empty()) { const double d = t - sweep_t0; sweep_sum += d; sweep_max = std::max(sweep_max, d); sweep_n++; sweep_fly += (long)flying.size(); }
            sweep_t0 = t;
        }
        // completions
        for (size_t i = 0; i < flying.size();) {
            Worker* w = flying[i];
            if (stamped(w)) {
                if (w->cur_stream >= 0) lstreams[(size_t)w->cur_stream].busy--;
                w->cur_stream = -1;
                flying[i] = flying.back(); flying.pop_back();
                finish(w, 2, "");
                progressed = true;
            } else {
                ++i;
            }
        }
        // launches
        while (!waiting.empty()) {
            if (!dead.empty()) { finish(waiting.front().w, 3, dead); waiting.pop_front(); progressed = true; continue; }
            int fs = -1;
            for (size_t i = 0; i < lstreams.size(); i++) if (lstreams[i].busy == 0) { fs = (int)i; break; }
            if (fs < 0) break;
            const int kind = waiting.front().kind;
            LevelBatch batch;
            Worker* who[MAXB];
            int n = 0;
            bool timed = false, mixed = false;
            for (auto it = waiting.begin(); it != waiting.end() && n < MAXB;) {
                if (it->kind != kind) { if (!any_kind) { ++it; continue; } mixed = true; }
                batch.it[n] = it->item;
                who[n++] = it->w;
                timed = timed || it->timed;
                it = waiting.erase(it);
            }
            hipStream_t st = lstreams[(size_t)fs].st;
            (void)hipGetLastError();
            if (timed) for (int i = 0; i < n; i++) (void)hipEventRecord(who[i]->ev0, st);
            if (mixed) launch_level_any(st, batch, n);          // every waiting level, whatever variant it needs
            else launch_level_batch(st, kind, batch, n);
            const hipError_t le = hipGetLastError();
            if (le != hipSuccess) {
                // the runtime did not take the launch: nothing of this batch will ever stamp
                const std::string msg = std::string("level kernel launch: ") + hipGetErrorString(le);
                for (int i = 0; i < n; i++) finish(who[i], 3, msg);
                progressed = true;
                continue;
            }
            if (timed) for (int i = 0; i < n; i++) (void)hipEventRecord(who[i]->ev1, st);
            lstreams[(size_t)fs].busy = n;
            lstreams[(size_t)fs].unretired++;
            const double tl = now_ms();
            for (int i = 0; i < n; i++) { who[i]->cur_stream = fs; who[i]->t_batch_launched = tl; who[i]->batch_n = n; flying.push_back(who[i]); }
            progressed = true;
        }
        if (progressed) { idle_spins = 0; continue; }
        if ((idle_spins & 0xFFu) == 0) {
            // idle: let the runtime retire finished launches of one free stream (it does so only when asked; left alone
            // they pile up for whoever synchronises the device next, ~10 us each)
            for (auto& ls : lstreams)
                if (ls.busy == 0 && ls.unretired > 0) { if (hipStreamQuery(ls.st) == hipSuccess) ls.unretired = 0; break; }
        }
        __builtin_ia32_pause();
        if ((++idle_spins & 0xFFFFFu) == 0) {
            // nothing has moved for a while: has a stream died under its batch, or drained without every stamp of it?
            for (size_t si = 0; si < lstreams.size(); si++) {
                LaunchStream& ls = lstreams[si];
                if (ls.busy == 0) continue;
                const hipError_t e = hipStreamQuery(ls.st);
                if (e == hipErrorNotReady) continue;
                if (e != hipSuccess) { dead = std::string("level kernel: ") + hipGetErrorString(e); break; }
                // the stream is empty: every kernel of the batch has ended, so a stamp that is still missing now (read
                // again after the query) will never come
                for (size_t i = 0; i < flying.size();) {
                    Worker* w = flying[i];
                    if (w->cur_stream != (int)si || stamped(w)) { ++i; continue; }
                    ls.busy--;
                    w->cur_stream = -1;
                    flying[i] = flying.back(); flying.pop_back();
                    finish(w, 3, "a level kernel ended without its completion stamp");
                }
            }
            check_resident();
            if (!dead.empty()) {
                for (Worker* w : flying) { if (w->cur_stream >= 0) lstreams[(size_t)w->cur_stream].busy = 0; w->cur_stream = -1; finish(w, 3, dead); }
                flying.clear();
            }
        }
    }
    if (sweep_log && sweep_n) fprintf(stderr, "level server: %ld rounds with levels flying, %.2f us each (longest %.1f us), %.1f levels flying on average\n",
                                      sweep_n, 1e3 * sweep_sum / sweep_n, 1e3 * sweep_max, (double)sweep_fly / sweep_n);
    for (Worker* w : flying) finish(w, 3, "context destroyed");
    for (auto& rq : waiting) finish(rq.w, 3, "context destroyed");
}
void Ctx::setup_enter(Worker* w) {
    {
        std::lock_guard<std::mutex> lk(smu);
        if (setups < setup_limit) { setups++; return; }
        setup_waiters.push_back(w);
    }
    FiberPool::park();                         // setup_leave hands its place over and makes this fiber ready
}
void Ctx::setup_leave() {
    Worker* next = nullptr;
    {
        std::lock_guard<std::mutex> lk(smu);
        if (!setup_waiters.empty()) { next = setup_waiters.front(); setup_waiters.pop_front(); }
        else setups--;
    }
    if (next) pool->make_ready(next->fib, split_exec);          // (a set-up: for the pool's set-up threads)
}
// The region's set-up is done: a mailbox to walk its levels on.  Parks until one falls free.
int Ctx::acquire_mailbox(Worker* w) {
    {
        std::lock_guard<std::mutex> lk(mmu);
        if (!free_mail.empty()) { const int m = free_mail.back(); free_mail.pop_back(); return m; }
        w->mslot = -1;
        mail_waiters.push_back(w);
    }
    FiberPool::park();                         // release_mailbox hands one over (w->mslot) and makes the fiber ready
    return w->mslot;
}
void Ctx::release_mailbox(int m) {
    Worker* next = nullptr;
    {
        std::lock_guard<std::mutex> lk(mmu);
        if (!mail_waiters.empty()) { next = mail_waiters.front(); mail_waiters.pop_front(); next->mslot = m; }
        else free_mail.push_back(m);
    }
    if (next) pool->make_ready(next->fib);
}
// A level is about to be posted to slot w->slot: make sure a generation of the resident grid is there to take it.
void Ctx::resident_ensure(Worker* w) {
    // the ordinary case, once per level from every executor, takes no lock: the generation runs and the mailbox's workgroup is there
    if (gen_state.load(std::memory_order_acquire) == GEN_RUNNING && __atomic_load_n(&mail_h[w->mslot].state, __ATOMIC_ACQUIRE) < 2u) return;
    const double t0 = now_ms();
    for (;;) {
        if (now_ms() - t0 > 60000.0) throw HipError("resident level workers: the previous grid has not left after a minute");
        {
            std::lock_guard<std::mutex> lk(gen_mu);
            if (gen_state == GEN_RUNNING) {
                if (__atomic_load_n(&mail_h[w->mslot].state, __ATOMIC_ACQUIRE) < 2u) return;
                // the slot's workgroup has left although regions are in flight (the heartbeat limit): end this generation
                __atomic_store_n(&ctl_h->stop, 1u, __ATOMIC_RELEASE);
                gen_state = GEN_STOPPING;
            }
            if (gen_state == GEN_STOPPING) {
                const hipError_t e = hipStreamQuery(rstream);
                if (e == hipSuccess) gen_state = GEN_STOPPED;
                else if (e != hipErrorNotReady) throw HipError(std::string("resident level workers: ") + hipGetErrorString(e));
            }
            if (gen_state == GEN_STOPPED) {
                __atomic_store_n(&ctl_h->stop, 0u, __ATOMIC_RELEASE);
                // a workgroup starts from the last stamp its mailbox has seen completed: what is in the mailbox beyond that is new
                for (int i = 0; i < res_slots; i++) { mail_h[i].ack = __atomic_load_n(&mail_done[(size_t)i], __ATOMIC_ACQUIRE); mail_h[i].state = 0; }
                __atomic_thread_fence(__ATOMIC_RELEASE);
                ResidentArgs ra{mail_d, ctl_d, 300000000ull, workers[0]->Pm, workers[0]->Rd, (int)workers.size()};     // 3 s of 100 MHz ticks without a heartbeat
                (void)hipGetLastError();
                launch_resident(rstream, ra, res_slots);
                const hipError_t le = hipGetLastError();
                if (le != hipSuccess) throw HipError(std::string("resident level workers, launch: ") + hipGetErrorString(le));
                gen_state = GEN_RUNNING;
                generations++;
                return;
            }
        }
        if (FiberPool::in_fiber() && workers.size() > 1) FiberPool::yield(); else std::this_thread::yield();
    }
}
// No region is in flight any more: the generation ends, so that nothing of this context stays on the GPU while the
// caller does something else with it (a device synchronisation would wait for the grid).
void Ctx::resident_idle() {
    std::lock_guard<std::mutex> lk(gen_mu);
    if (gen_state == GEN_RUNNING && regions_active.load(std::memory_order_acquire) == 0) {
        __atomic_store_n(&ctl_h->stop, 1u, __ATOMIC_RELEASE);
        gen_state = GEN_STOPPING;
    }
}
void Ctx::resident_shutdown() {
    if (!resident) return;
    {
        std::lock_guard<std::mutex> lk(gen_mu);
        if (ctl_h) __atomic_store_n(&ctl_h->stop, 1u, __ATOMIC_RELEASE);
        if (gen_state == GEN_RUNNING) gen_state = GEN_STOPPING;
    }
    if (rstream) {
        // the grid leaves within a few naps of its pollers; bounded, so that a workgroup that does not leave cannot hold the host
        const double t0 = now_ms();
        while (hipStreamQuery(rstream) == hipErrorNotReady && now_ms() - t0 < 10000.0) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    gen_state = GEN_STOPPED;
}
bool Ctx::poll_stamps() {
    const size_t n = workers.size();
    bool waiting = false;
    for (size_t i = 0; i < n; i++) {
        if (polled[i].load(std::memory_order_acquire) != 1) continue;
        Worker* w = workers[i].get();
        if (__atomic_load_n(&w->Rh->seq, __ATOMIC_ACQUIRE) != w->level_want) { waiting = true; continue; }
        uint8_t one = 1;
        if (!polled[i].compare_exchange_strong(one, 0, std::memory_order_acq_rel)) continue;      // another thread saw it first
        // The flag is this thread's now -- but is it still the flag of the level whose stamp was seen?  Another thread may
        // have seen that stamp first, the region may have run on and be parked for its NEXT level by the time the exchange
        // above succeeded (it then took the new level's flag).  Look again, with the flag in hand: nothing changes under it.
        if (__atomic_load_n(&w->Rh->seq, __ATOMIC_ACQUIRE) != w->level_want) {
            polled[i].store(1, std::memory_order_seq_cst);
            waiting = true;
            continue;
        }
        w->t_seen = now_ms();
        w->level_state.store(2, std::memory_order_release);
        pool->make_ready(w->fib);
    }
    return waiting;
}
void Ctx::poll_health() {
    const size_t n = workers.size();
    for (size_t i = 0; i < n; i++) {
        if (polled[i].load(std::memory_order_acquire) != 1) continue;
        Worker* w = workers[i].get();
        const int m = w->mslot;
        if (m < 0) continue;
        const unsigned ms = __atomic_load_n(&mail_h[m].state, __ATOMIC_ACQUIRE);
        const bool never = ms == 0u && now_ms() - w->t_posted > 20000.0;       // more slots than the GPU holds resident
        if ((ms < 2u && !never) || __atomic_load_n(&w->Rh->seq, __ATOMIC_ACQUIRE) == w->level_want) continue;
        uint8_t one = 1;
        if (!polled[i].compare_exchange_strong(one, 0, std::memory_order_acq_rel)) continue;
        {   // (as in poll_stamps: with the flag in hand, is it still the level that was judged?)
            const int m2 = w->mslot;
            const unsigned ms2 = m2 >= 0 ? __atomic_load_n(&mail_h[m2].state, __ATOMIC_ACQUIRE) : 0u;
            const bool never2 = ms2 == 0u && now_ms() - w->t_posted > 20000.0;
            if (m2 != m || (ms2 < 2u && !never2) || __atomic_load_n(&w->Rh->seq, __ATOMIC_ACQUIRE) == w->level_want) {
                polled[i].store(1, std::memory_order_seq_cst);
                continue;
            }
        }
        w->level_err = never ? "the slot's resident level worker has not started within 20 s (more slots than the GPU holds resident workgroups?)"
                     : ms == 3u ? "the slot's resident level worker received an item that was not its own"
                                : "the slot's resident level worker has left before the level was done";
        w->t_seen = now_ms();
        w->level_state.store(3, std::memory_order_release);
        pool->make_ready(w->fib);
    }
}
void Ctx::submit_level(const LevelRequest& rq) {
    rq.w->level_state.store(1, std::memory_order_release);
    plk.lock();
    pending.push_back(rq);
    plk.unlock();
    n_pending.fetch_add(1, std::memory_order_seq_cst);
    if (server_asleep.load(std::memory_order_seq_cst)) {
        { std::lock_guard<std::mutex> lk(dmu); }
        dcv.notify_one();
    }
}
// The region's fiber parks until the server has seen the level's stamp (or failed the level).  The server makes the
// fiber ready exactly once per request, so the fiber parks exactly once per request -- also when the level is already
// done by the time it gets here (it then comes straight back).
void Worker::wait_level() {
    const double t_park = now_ms();
    FiberPool::park();
    const double t_back = now_ms();
    wake_acc[0] += t_park - t_posted;           // handing the level to the server
    wake_acc[1] += t_back - t_seen;             // the server has seen the stamp -> this fiber runs again
    if (ctx->wake_hist) { const double us = 1e3 * (t_back - t_seen); int b = 0; while (b < 23 && us >= (double)(1 << b)) b++; ctx->wake_hist[b].fetch_add(1, std::memory_order_relaxed); }
    const int state = level_state.load(std::memory_order_acquire);
    if (state == 3) throw HipError(level_err);
    if (state != 2) throw HipError("a region was resumed before its level was done");
}

// a7 on the device.  Returns the number of columns.
int Worker::msa_device(const std::vector<std::string>& seqs, std::vector<std::string>& rows) {
    const int n = (int)seqs.size();
    std::vector<int> off(n + 1, 0);
    std::string packed;
    for (int i = 0; i < n; i++) { packed += seqs[i]; off[i + 1] = (int)packed.size(); }
    const int cmax = (int)packed.size() + 1;
    size_t longest = 0;                                   // (the first sequence only seeds the columns: any length)
    for (int i = 1; i < n; i++) longest = std::max(longest, seqs[i].size());
    MsaDev d;
    char* dseq = (char*)m_seqs.ensure(packed.size() + 1);
    int* doff = (int*)m_off.ensure(sizeof(int) * (n + 1));
    stage->h2d(dseq, packed.data(), packed.size(), st);               // page-locked staging while other regions are in flight
    stage->h2d(doff, off.data(), sizeof(int) * (n + 1), st);
    d.seqs = dseq; d.seq_off = doff; d.n = n; d.cmax = cmax;
    d.cols[0] = (char*)m_cols0.ensure((size_t)cmax * n);
    d.cols[1] = (char*)m_cols1.ensure((size_t)cmax * n);
    d.counts = (int*)m_counts.ensure(sizeof(int) * 11 * (size_t)cmax);
    d.mv_stride = (int)((longest + 1 + 63) / 64) * 64;
    d.moves = (uint8_t*)m_moves.ensure((size_t)(cmax + 1) * (size_t)d.mv_stride);
    d.edge = (int*)m_edge.ensure(sizeof(int) * 2 * (size_t)(cmax + 1));
    d.trace = (int*)m_trace.ensure(sizeof(int) * 2 * (size_t)(cmax + 64));
    int* dout = (int*)m_out.ensure(sizeof(int) * 2);
    d.ncol_out = dout; d.err_out = dout + 1;
    launch_msa(st, d);
    int out[2];
    stage->d2h(out, dout, sizeof(out), st);
    sync_stream();
    stage->land();
    if (out[1] & 0xFF) {
        size_t longest = 0;
        for (auto& q : seqs) longest = std::max(longest, q.size());
        throw ScError(SC_ERR_UNSUPPORTED, "MSA kernel capacity exceeded (" + std::string((out[1] & 1) ? "a sequence longer than 63; " : "") +
                      std::string((out[1] & 2) ? "more than 1024 columns; " : "") + std::string((out[1] & 4) ? "column buffer; " : "") +
                      std::string((out[1] & 8) ? "more than 65535 sequences; " : "") + std::to_string(n) + " sequences, longest " +
                      std::to_string(longest) + ", columns so far " + std::to_string(out[0]) + ")");
    }
    const int ncol = out[0], cur = out[1] >> 8;
    std::vector<char> cols((size_t)ncol * n);
    if (ncol > 0) {
        stage->d2h(cols.data(), d.cols[cur], (size_t)ncol * n, st);
        sync_stream();
        stage->land();
    }
    rows.assign(n, std::string((size_t)ncol, '-'));
    for (int c = 0; c < ncol; c++)
        for (int k = 0; k < n; k++) rows[k][c] = cols[(size_t)c * n + k];
    return ncol;
}


// a5 on the device: packs the read batch, runs k_thread_* and returns the class tables.
void Worker::thread_device(const std::string& G, const std::vector<AlignedRead>& R, const std::vector<std::vector<CigarOp>>& cig,
                           ThreadTables& T) {
    const int glen = (int)G.size(), n = (int)R.size();
#ifdef SC_GRAPH_TIMING
    double tdp_ = now_ms();
#define SC_DPHASE(name) do { HIPCHK(hipStreamSynchronize(st)); const double t_ = now_ms(); fprintf(stderr, "      thread_device %-12s %.2f ms\n", name, t_ - tdp_); tdp_ = t_; } while (0)
#else
#define SC_DPHASE(name) do {} while (0)
#endif
    // symbol table of the READS: A C G T first, then every other byte that occurs in a read, in byte order.  A base of the
    // gene that no read carries (an IUPAC code of a 16S reference) keeps the code 0xFF: no read base equals it, so every
    // read base aligned there lands in a sibling class, as `G[i]==r[j]` decides in the reference (PartialOrderGraph.cpp:133)
    bool present[256] = {false};
    for (const auto& r : R) for (unsigned char c : r.seq) present[c] = true;
    std::memset(T.lut, 0xFF, sizeof T.lut);
    T.sym.clear();
    for (char c : {'A', 'C', 'G', 'T'}) { T.lut[(unsigned char)c] = (uint8_t)T.sym.size(); T.sym.push_back(c); }
    for (int c = 0; c < 256; c++)
        if (present[c] && T.lut[c] == 0xFF) {
            if (T.sym.size() >= 8) throw ScError(SC_ERR_UNSUPPORTED, "more than 8 distinct symbols in the reads");
            T.lut[c] = (uint8_t)T.sym.size(); T.sym.push_back((char)c);
        }
    std::vector<int> pos(n), seq_off(n + 1, 0), cig_off(n + 1, 0), cig_len;
    std::string seq, cig_op;
    long m_bases = 0;
    for (int r = 0; r < n; r++) {
        pos[r] = R[r].pos;
        seq += R[r].seq; seq_off[r + 1] = (int)seq.size();
        for (const CigarOp& c : cig[r]) { cig_op.push_back(c.op); cig_len.push_back(c.len); if (c.op == 'M') m_bases += c.len; }
        cig_off[r + 1] = (int)cig_op.size();
    }
    SC_DPHASE("pack");
    const int ncls = glen * 8;
    ThreadDev d{};
    d.glen = glen; d.n_reads = n;
    char* dref = (char*)t_ref.ensure((size_t)glen + 1);
    stage->h2d(dref, G.data(), (size_t)glen, st);
    d.ref = dref;
    d.pos = upload(*stage, t_pos, pos, st);
    d.seq_off = upload(*stage, t_seqoff, seq_off, st);
    char* dseq = (char*)t_seq.ensure(seq.size() + 1);
    stage->h2d(dseq, seq.data(), seq.size(), st);
    d.seq = dseq;
    d.cig_off = upload(*stage, t_cigoff, cig_off, st);
    char* dop = (char*)t_cigop.ensure(cig_op.size() + 1);
    stage->h2d(dop, cig_op.data(), cig_op.size(), st);
    d.cig_op = dop;
    d.cig_len = upload(*stage, t_ciglen, cig_len, st);
    uint8_t* dlut = (uint8_t*)t_lut.ensure(256);
    stage->h2d(dlut, T.lut, 256, st);
    d.lut = dlut;
    // tables: count | minrid | smin | emin (ncls each) | tmin (8*ncls) | off (ncls+1) | cursor (ncls) | err | big (1 + ncls)
    const size_t words = (size_t)ncls * 4 + (size_t)ncls * 8 + (size_t)ncls + 1 + (size_t)ncls + 1 + 1 + (size_t)ncls;
    int* tabs = (int*)t_tabs.ensure(sizeof(int) * words);
    d.count = tabs; d.minrid = tabs + ncls; d.smin = tabs + 2 * (size_t)ncls; d.emin = tabs + 3 * (size_t)ncls;
    d.tmin = tabs + 4 * (size_t)ncls; d.off = tabs + 12 * (size_t)ncls; d.cursor = d.off + ncls + 1; d.err = d.cursor + ncls; d.big = d.err + 1;
    HIPCHK(hipMemsetAsync(d.count, 0, sizeof(int) * (size_t)ncls, st));
    HIPCHK(hipMemsetAsync(d.minrid, 0x7f, sizeof(int) * (size_t)ncls * 11, st));          // minrid, smin, emin, tmin = 0x7f7f7f7f
    HIPCHK(hipMemsetAsync(d.off, 0, sizeof(int) * ((size_t)ncls * 2 + 3), st));           // off, cursor, err, the count of big classes
    d.pool = (int*)t_pool.ensure(sizeof(int) * (size_t)std::max<long>(m_bases, 1));
    int* pool_sorted = (int*)t_pool2.ensure(sizeof(int) * (size_t)std::max<long>(m_bases, 1));
    SC_DPHASE("uploads");
    launch_thread(st, d, pool_sorted);
    SC_DPHASE("kernels");
    T.count.resize(ncls); T.minrid.resize(ncls); T.smin.resize(ncls); T.emin.resize(ncls);
    T.tmin.resize((size_t)ncls * 8); T.off.resize((size_t)ncls + 1); T.pool.resize((size_t)m_bases);
    int err = 0;
    stage->d2h(T.count.data(), d.count, sizeof(int) * (size_t)ncls, st);
    stage->d2h(T.minrid.data(), d.minrid, sizeof(int) * (size_t)ncls, st);
    stage->d2h(T.smin.data(), d.smin, sizeof(int) * (size_t)ncls, st);
    stage->d2h(T.emin.data(), d.emin, sizeof(int) * (size_t)ncls, st);
    stage->d2h(T.tmin.data(), d.tmin, sizeof(int) * (size_t)ncls * 8, st);
    stage->d2h(T.off.data(), d.off, sizeof(int) * ((size_t)ncls + 1), st);
    stage->d2h(T.pool.data(), pool_sorted, sizeof(int) * (size_t)std::max<long>(m_bases, 0), st);
    stage->d2h(&err, d.err, sizeof(int), st);
    const double t_sync0 = now_ms();
    sync_stream();
    stage->land();
    if (getenv("SC_SYNC_LOG")) fprintf(stderr, "sync thread_device %.3f ms\n", now_ms() - t_sync0);
    SC_DPHASE("downloads");
#undef SC_DPHASE
    if (err) throw ScError(SC_ERR_ARG, "a read runs outside the window or past its own bases");
    const int INF = 0x7fffffff;
    auto fix = [&](std::vector<int>& v) { for (int& x : v) if (x == 0x7f7f7f7f) x = INF; };
    fix(T.minrid); fix(T.smin); fix(T.emin); fix(T.tmin);
}

// ... and the same for the rows of `rows` only: a row whose counts did not change keeps its sum (the same additions in the
// same order give the same long double), and a level changes one row of a candidate -- its model is cold in the caches
// by the time the region comes back to it, so the lines it touches count
static void recount_rows(Model& s, unsigned rows) {
    rows &= 0x3Fu;
    if (!rows) return;
    for (int i = 0; i < 6; i++) {
        if (!(rows & (1u << i))) continue;
        s.comp[i] = 0;
        for (int j = 0; j < 6; j++) s.comp[i] += s.sub[i * s.ks + j];
    }
    s.Z = 0;
    for (int i = 0; i < 6; i++) s.Z += s.comp[i];
}
static void recount(Model& s) {                                           // Strain.cpp:115-124
    s.Z = 0;
    for (int i = 0; i < 6; i++) {
        s.comp[i] = 0;
        for (int j = 0; j < 6; j++) s.comp[i] += s.sub[i * s.ks + j];
        s.Z += s.comp[i];
    }
}
}  // namespace sc
// The reference adds 1 to a candidate's weight once per draw, in x87 long double (NonparametricClustering.cpp:195): k
// separate roundings, not one.  Inside a binade every a + j is exact (1 is a multiple of the unit in the last place
// while a < 2^64), so the only additions that round are the ones that cross into the next binade: the same k additions
// in O(log k) steps, bit for bit (tests/native/add_ones_check.cpp compares it with the literal loop).
extern "C" long double sc_add_ones(long double a, unsigned long k) {
    if (!std::isfinite((double)a) && !(a == a && a - a == 0)) return a + (long double)k;       // inf / NaN stay what they are
    while (k > 0) {
        if (!(a >= 1)) { a += 1; k--; continue; }            // below 1 (or negative): the literal addition, at most a few times
        {
            // the usual case without a call into libm (this runs once per candidate and level): all k additions stay below
            // the next power of two -- read off the x87 representation (sign + 15-bit exponent above a 64-bit mantissa)
            union { long double v; struct { uint64_t mant; uint16_t se; } b; } top;
            top.v = a;
            top.b.se = (uint16_t)((top.b.se & 0x7fffu) + 1u);      // 2^e for a in [2^(e-1), 2^e)
            top.b.mant = 0x8000000000000000ull;
            // (a < 2^64: an ulp of at most 1, so the difference and the sum are exact)
            if ((top.b.se & 0x7fffu) <= 16383u + 64u && top.v - a > (long double)k) return a + (long double)k;
        }
        int e;
        (void)frexpl(a, &e);                                 // a in [2^(e-1), 2^e)
        const long double top = ldexpl(1.0L, e);
        const long double room = top - a;                    // exact (Sterbenz)
        if (!(room >= 1) && !(room > 0)) { a += 1; k--; continue; }
        const long double jr = ceill(room) - 1;              // additions that stay below the next power of two
        if (jr >= (long double)k) return a + (long double)k;
        const unsigned long j = (unsigned long)jr;
        a += (long double)j; k -= j;                         // exact
        a += 1; k--;                                         // the crossing one rounds like the reference's
    }
    return a;
}
namespace sc {
static uint64_t hash_extend(uint64_t h, const std::string& lab) {
    for (unsigned char c : lab) { h ^= c; h *= 1099511628211ull; }
    return h;
}
static void fmt_g17(std::string& out, double v) {
    char b[64];
    snprintf(b, sizeof b, "%.17g", v);
    out += b;
}

void Worker::cluster(Job& job, const PoGraph& g, FlatGraph& f) {
    const int n_reads = (int)job.reads.size();
    const double t_cluster0 = now_ms();
    ev_used = 0;
    const sc_params& pa = job.params;
    const ld e = (ld)pa.error_rate, tau = (ld)pa.tau, diff = (ld)pa.diff_rate;     // float widened, StrainCall.cpp:58-154
    const int K = f.K;

    // ---- pseudo level holding every read once, for read_assign (NonparametricClustering.cpp:776-836)
    const int final_e0 = (int)f.ent_rid.size();
    long total_copies = 0;
    {
        int qo = 0;
        for (int i = 0; i < n_reads; i++) {
            f.ent_rid.push_back(i); f.ent_cn.push_back(job.reads[i].cn); f.ent_lab_off.push_back(0);
            f.ent_lab_len.push_back(0); f.ent_first.push_back(1);
            qo += job.reads[i].cn;
        }
        total_copies = qo;
    }
    // prefix of copy numbers inside each level
    std::vector<int>& ent_qoff = ent_qoff_buf;                           // (the slot's: reused from region to region)
    ent_qoff.assign(f.ent_rid.size(), 0);
    int max_level_entries = n_reads, max_level_q = 0;
    for (int l = 0; l < f.n_levels; l++) {
        int qo = 0;
        for (int x = f.level_ent_ptr[l]; x < f.level_ent_ptr[l + 1]; x++) { ent_qoff[x] = qo; qo += f.ent_cn[x]; }
        max_level_entries = std::max(max_level_entries, f.level_ent_ptr[l + 1] - f.level_ent_ptr[l]);
        max_level_q = std::max(max_level_q, qo);
    }
    { int qo = 0; for (int i = 0; i < n_reads; i++) { ent_qoff[final_e0 + i] = qo; qo += job.reads[i].cn; } }
    const long qcap = std::max<long>(std::max<long>(max_level_q, total_copies), 1);
    // cells of a read_loglik row that can hold a value when level l starts: the reads of the levels before it and their
    // mates (the soft update enters a mate the first time it is asked for, Strain.cpp:147-150) -- a prefix of the read ids
    std::vector<int> level_hi((size_t)f.n_levels + 1, 0);
    for (int l = 0; l < f.n_levels; l++) {
        int hi = level_hi[(size_t)l];
        for (int x = f.level_ent_ptr[l]; x < f.level_ent_ptr[l + 1]; x++) {
            const int rid = f.ent_rid[x];
            hi = std::max(hi, rid + 1);
            for (int k = job.mate_off[(size_t)rid]; k < job.mate_off[(size_t)rid + 1]; k++) hi = std::max(hi, job.mate_idx[(size_t)k] + 1);
        }
        level_hi[(size_t)l + 1] = hi;
    }

    // ---- upload the static arrays
    JobDev jd{};
    jd.ent_rid = upload(*stage, b_ent_rid, f.ent_rid, st);
    jd.ent_cn = upload(*stage, b_ent_cn, f.ent_cn, st);
    jd.ent_lab_off = upload(*stage, b_ent_lab_off, f.ent_lab_off, st);
    jd.ent_lab_len = upload(*stage, b_ent_lab_len, f.ent_lab_len, st);
    jd.ent_first = upload(*stage, b_ent_first, f.ent_first, st);
    jd.ent_qoff = upload(*stage, b_ent_qoff, ent_qoff, st);
    jd.labels = upload(*stage, b_labels, f.labels, st);
    jd.mate_ptr = upload(*stage, b_mate_ptr, job.mate_off, st);
    jd.mate_idx = upload(*stage, b_mate_idx, job.mate_idx, st);
    jd.n_reads = n_reads; jd.K = K; jd.code_N = f.code_N;
    jd.ll_stride = ((long)n_reads + 3) & ~3L;
    jd.ll = (double*)b_ll.ensure(sizeof(double) * (size_t)jd.ll_stride * MAXS);
    jd.has = (uint8_t*)b_has.ensure((size_t)n_reads + 8);
    HIPCHK(hipMemsetAsync(jd.has, 0, (size_t)n_reads + 8, st));
    jd.U = ctx->dU;
    jd.Uf = ctx->dUf;
    jd.isnew = (uint8_t*)b_isnew.ensure((size_t)max_level_entries + 8);
    jd.qcap = qcap;
    jd.tabA = (double*)b_tabA.ensure(sizeof(double) * (size_t)qcap * MAXS);
    jd.tabLf = (float*)b_tabLf.ensure(sizeof(float) * (size_t)(std::min<long>(qcap, MAX_DRAWS) + 4) * 136);
    jd.qcode = (uint8_t*)b_qcode.ensure((size_t)qcap + 8);
    jd.qent = (int*)b_qent.ensure(sizeof(int) * (size_t)qcap);
    jd.quid = (int*)b_quid.ensure(sizeof(int) * (size_t)qcap);
    // the batched level kernels find the region through a pointer: the block travels once, with the uploads
    const JobDev* jd_dev = (const JobDev*)b_jobdev.ensure(sizeof(JobDev));
    stage->h2d((void*)jd_dev, &jd, sizeof jd, st);

    // ---- a16: every edge support on the device
    {
        std::vector<int> esrc(f.out_node.size());
        for (int a = 0; a < f.n_nodes; a++) for (int x = f.out_ptr[a]; x < f.out_ptr[a + 1]; x++) esrc[x] = a;
        int* d_out_ptr = upload(*stage, b_out_ptr, f.out_ptr, st);
        int* d_out_node = upload(*stage, b_out_node, f.out_node, st);
        int* d_pool_ptr = upload(*stage, b_pool_ptr, f.pool_ptr, st);
        int* d_pool_rid = upload(*stage, b_pool_rid, f.pool_rid, st);
        int* d_pool_cn = upload(*stage, b_pool_cn, f.pool_cn, st);
        uint8_t* d_isend = upload(*stage, b_isend, f.node_is_end, st);
        int* d_esrc = upload(*stage, b_esrc, esrc, st);
        int* d_sup = (int*)b_support.ensure(sizeof(int) * std::max<size_t>(esrc.size(), 1));
        launch_edge_support(st, d_out_ptr, d_out_node, d_pool_ptr, d_pool_rid, d_pool_cn, d_isend, d_esrc, (int)esrc.size(),
                            f.pools_sorted ? 1 : 0, d_sup);
        if (!esrc.empty())
            stage->d2h(f.out_support.data(), d_sup, sizeof(int) * esrc.size(), st);
        const double t_sync0 = now_ms();
        sync_stream();
        stage->land();
        if (stage != &passthrough) ctx->release_arena(stage);      // every transfer of the set-up is done
        stage = &passthrough;
        if (getenv("SC_SYNC_LOG")) fprintf(stderr, "sync uploads+edge_support %.3f ms (since cluster start %.3f)\n", now_ms() - t_sync0, now_ms() - t_cluster0);
        job.edge_support = f.out_support;
    }

    // ---- level walk: from here on the region's host work is a few microseconds per level
    if (setup_held) { ctx->setup_leave(); setup_held = false; }
    job.stats.setup_ms = now_ms() - t_cluster0;
    struct MailHold {                  // the mailbox the region walks on (resident workers): taken now, handed on when the walk ends
        Worker* w;
        void drop() { if (w->mslot >= 0) { const int m = w->mslot; w->mslot = -1; w->ctx->release_mailbox(m); } }
        ~MailHold() { drop(); }
    } mail_hold{this};
    if (ctx->resident) {
        const double t_m0 = now_ms();
        mslot = ctx->acquire_mailbox(this);
        job.stats.mailbox_ms = now_ms() - t_m0;
        __atomic_store_n(&Rh->seq, 0u, __ATOMIC_RELEASE);       // (stamps are the mailbox's from here on: never 0)
    }
    std::vector<HStrain> level_strains, sub_strains;
    std::vector<Model> models;                                           // pool; free entries in free_models
    std::vector<int> free_models;
    auto model_new = [&]() { if (!free_models.empty()) { const int m = free_models.back(); free_models.pop_back(); return m; }
                             models.emplace_back(); return (int)models.size() - 1; };
    auto drop = [&](const HStrain& s, std::vector<int>& free_slots_) { free_slots_.push_back(s.slot); free_models.push_back(s.model); };
    std::vector<PathRec> arena;
    std::vector<int> free_slots;
    for (int i = MAXS - 1; i >= 0; i--) free_slots.push_back(i);
    std::vector<HStrain> final_strains;
    std::string& tr = job.trace;
    const bool want_trace = pa.want_trace != 0;

    auto strain_seq = [&](const HStrain& s) {
        std::vector<int> rev;
        for (int t = s.tail; t >= 0; t = arena[t].parent) rev.push_back(arena[t].node);
        std::string q;
        for (auto it = rev.rbegin(); it != rev.rend(); ++it) q += f.node_label_str[*it];
        return q;
    };
    auto trace_dump = [&](const char* when, int level, const std::vector<HStrain>& sv) {
        if (!want_trace || sv.empty()) return;
        tr += "------------------------------\n"; tr += when; tr += "\nlevel: "; tr += std::to_string(level); tr += "\n";
        for (const auto& s : sv) { tr += strain_seq(s); tr += "\t"; fmt_g17(tr, (double)s.abundance); tr += "\n"; }
    };
    auto sort_strains = [&](std::vector<HStrain>& sv) {                  // std::sort, abundance descending
        std::vector<int> perm(sv.size());
        for (size_t i = 0; i < sv.size(); i++) perm[i] = (int)i;
        std_sort_perm(perm, [&](int a, int b) { return sv[a].abundance > sv[b].abundance; });
        std::vector<HStrain> t;
        t.reserve(sv.size());
        for (int i : perm) t.push_back(sv[i]);
        sv.swap(t);
    };
    auto seq_identity = [](const std::string& a, const std::string& b) {  // NonparametricClustering.cpp:584-612
        int iden = 0, len = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            const char x = a[i], y = i < b.size() ? b[i] : 0;
            if (x == '-' && y == '-') continue;
            else if (x == '=' && y == '=') continue;
            else if (x == '=' && y == '-') continue;
            else if (x == '-' && y == '=') continue;
            else if (x == '^' && y == '^') continue;
            else if (x == y) iden += 1;
            len += 1;
        }
        return (ld)((iden + 0.0) / len);
    };

    {   // level_strains.push_back(Strain(100,e)), NonparametricClustering.cpp:281; Strain.cpp:41-71
        HStrain s{};
        s.model = model_new();
        Model& m = models[(size_t)s.model];
        m.ks = K;
        for (int i = 0; i < K * K; i++) m.sub[i] = 0;
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) m.sub[i * K + j] = (i == j) ? 100 * (1 - e) : 100 * e;
        recount(m);
        m.dirty = 0xFFFFu;
        for (int a = 0; a < KMAX; a++) m.stale[a] = 0xFFFFu;
        s.abundance = 0; s.slot = free_slots.back(); free_slots.pop_back();
        s.tail = -1; s.node = -1; s.hash = 1469598103934665603ull; s.seqlen = 0;
        level_strains.push_back(s);
    }
    bool branching = false;
    std::vector<std::pair<int, int>> pending_copies;     // (src slot, dst slot) for the next launch
    double sampler_ms = 0;
    long sampler_launches = 0, level_launches = 0, draws = 0, exact = 0, slow = 0, sampler_copies = 0, sampler_strains = 0, passes = 0;
    unsigned long long chain_cycles = 0, chain_wall = 0, level_ticks = 0, sampler_ticks = 0;

    // where the host's time between two levels goes (sc_stats.host_us): [0] parameters of the level (log tables, the
    // host-mapped block), [1] results of the level into the candidates' models, pruning, [2] extension of the candidates
    int la_cache[MAXS];
    double host_acc[3] = {0, 0, 0};
    double t_mark = now_ms();
    auto lap = [&](int k) { const double t = now_ms(); host_acc[k] += t - t_mark; t_mark = t; };
    double t_last_done = now_ms();
    FILE* level_log = getenv("SC_LEVEL_LOG") ? fopen((std::string(getenv("SC_LEVEL_LOG")) + "." + std::to_string(slot)).c_str(), "a") : nullptr;   // diagnostics only
    int cur_level = 0;
    auto run_level = [&](int mode, int e0, int e1, int Q, int n_sweeps, bool do_update, const std::vector<HStrain>& sv,
                         bool has_dups, bool any_multi) {
        LevelParams& P = *Ph;
        const int S = (int)sv.size();
        LevelHdr H{};
        H.mode = mode; H.S = S; H.e0 = e0; H.e1 = e1; H.has_dups = has_dups; H.any_multi = any_multi; H.Q = Q;
        H.n_sweeps = n_sweeps; H.do_update = do_update ? 1 : 0;
        H.n_copy = (int)pending_copies.size();
        H.copy_n = do_update ? level_hi[(size_t)cur_level] : n_reads;
        for (int c = 0; c < H.n_copy; c++) { P.copy_src[c] = pending_copies[c].first; P.copy_dst[c] = pending_copies[c].second; }
        pending_copies.clear();
        {   // every candidate owns its row of the read log-likelihood matrix
            uint64_t seen[2] = {0, 0};
            for (int s = 0; s < S; s++) {
                const int r = sv[s].slot;
                if (r < 0 || r >= MAXS || (seen[r >> 6] >> (r & 63)) & 1) throw ScError(SC_ERR_INTERNAL, "two candidates share a read_loglik row");
                seen[r >> 6] |= 1ull << (r & 63);
            }
        }
        // The region comes back to cold caches (hundreds of other regions have used this core since its last level): what the
        // loops below touch per candidate -- its node's label, its model's header and log table -- is asked for up front, all
        // candidates at once, instead of one miss after the other.
        for (int s = 0; s < S; s++) {
            const int nd = sv[s].node;
            if (nd >= 0) { __builtin_prefetch(&f.node_lab_off[(size_t)nd]); __builtin_prefetch(&f.node_lab_len[(size_t)nd]); }
            if (do_update) {
                const Model& hm = models[(size_t)sv[s].model];
                __builtin_prefetch(&hm.dirty); __builtin_prefetch(&hm.stale[0]);
                for (int o = 0; o < K * K; o += 8) __builtin_prefetch(&hm.lpc[o]);
            }
        }
        ld za = 0;
        for (int s = 0; s < S; s++) za += sv[s].abundance;                 // normalize(), :10-15
        for (int s = 0; s < S; s++) {
            StrainParam& sp = P.sp[s];
            sp.slot = sv[s].slot;
            sp.lab_off = sv[s].node >= 0 ? f.node_lab_off[sv[s].node] : 0;
            sp.lab_len = sv[s].node >= 0 ? f.node_lab_len[sv[s].node] : 0;
            la_cache[s] = (sp.lab_len == 1) ? (int)f.labels[(size_t)sp.lab_off] : -1;       // the candidate's symbol, for the update after the level
            sp.pad = 0;
            sp.a0 = (double)sv[s].abundance;
            sp.logpri = (mode == MODE_HARD) ? (double)logl(sv[s].abundance / za) : 0.0;      // the sampler takes a0 itself
            if (!do_update) continue;
            // log table of the strain: only the rows its counts changed in since the last level are redone, and in
            // them only the logarithms of the counts that changed
            Model& hm = models[(size_t)sv[s].model];
            for (int a = 0; a < K; a++) {
                if (!(hm.dirty & (1u << a))) continue;
                const ld lc = logl(a < 6 ? hm.comp[a] : (ld)0);                             // log comp_count[a], Strain.cpp:132-135
                for (int b = 0; b < K; b++) {
                    if (hm.stale[a] & (1u << b)) hm.lsub[a * K + b] = logl(hm.sub[a * K + b]);
                    hm.lpc[a * K + b] = (double)(hm.lsub[a * K + b] - lc);
                }
                hm.stale[a] = 0;
            }
            hm.dirty = 0;
            double* dst = P.lpt + (size_t)s * K * K;                                          // compact [K][K]
            std::memcpy(dst, hm.lpc, sizeof(double) * (size_t)K * K);
        }
        const bool chain = (mode == MODE_SAMPLE) && S > 1 && n_sweeps > 0;
        const bool timed = chain && pa.want_timing && !ctx->resident;      // (no launch to bracket with events when the workers are resident)
        if (level_wants_grid(jd, H)) {
            // a very large level: row copies / the single-symbol update on a grid, from a device copy of the parameters
            const size_t bytes = offsetof(LevelParams, lpt) + sizeof(double) * (size_t)S * K * K;
            HIPCHK(hipMemcpyAsync(Pd, Ph, bytes, hipMemcpyHostToDevice, st));
            H.done = launch_level_grid(st, jd, H, Pd, Rd);
            sync_stream();                               // the level's kernel runs on another stream
        }
        H.seq = (ctx->resident && mslot >= 0) ? ++ctx->mail_seq[(size_t)mslot] : ++seq;     // (a mailbox keeps its own count: regions take turns on it)
        level_want = H.seq;
        if (timed) {
            // a fresh pair of events per sampler launch; their times are read after the walk, not between levels
            if (ev_used + 2 > ev_pool.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
                ev_pool.push_back(a); ev_pool.push_back(b);
            }
            ev0 = ev_pool[ev_used]; ev1 = ev_pool[ev_used + 1];
            ev_used += 2;
        }
        const double t_launched = level_log ? now_ms() : 0.0;
        lap(0);
        if (ctx->resident) {
            // the slot's resident workgroup takes the level from its mailbox: the item, then its stamp (release)
            ctx->resident_ensure(this);
            Mailbox& mb = ctx->mail_h[mslot];
            mb.item = LevelItem{jd_dev, H, level_kind(H) | (level_lds_kb(H, K) << 8), Pm, Rd};
            __atomic_store_n(&mb.seq, H.seq, __ATOMIC_RELEASE);
            t_posted = now_ms();
            t_batch_launched = level_log ? t_posted : 0.0; batch_n = 1;
            if (ctx->workers.size() == 1) {
                unsigned spins = 0;
                while (__atomic_load_n(&Rh->seq, __ATOMIC_ACQUIRE) != H.seq) {
                    __builtin_ia32_pause();
                    if ((++spins & 0xFFFFFu) != 0) continue;
                    const unsigned ms = __atomic_load_n(&mb.state, __ATOMIC_ACQUIRE);
                    if (ms >= 2u && __atomic_load_n(&Rh->seq, __ATOMIC_ACQUIRE) != H.seq)
                        throw HipError("the resident level worker has left before the level was done");
                    if (ms == 0u && now_ms() - t_posted > 20000.0) throw HipError("the resident level worker has not started within 20 s");
                }
            } else {
                if (ctx->poll_exec) {
                    level_state.store(1, std::memory_order_release);
                    ctx->polled[(size_t)slot].store(1, std::memory_order_seq_cst);      // (level_want and the mailbox are written: an executor may look)
                    ctx->pool->ensure_poller();
                } else {
                    ctx->submit_level(LevelRequest{this, LevelItem{}, KIND_POSTED, false});
                }
                wait_level();
            }
        } else if (ctx->workers.size() == 1) {
            // a single region in flight: nobody to batch with, so the worker launches its level itself and watches the stamp
            LevelBatch batch;
            batch.it[0] = LevelItem{jd_dev, H, level_kind(H) | (level_lds_kb(H, K) << 8), Pm, Rd};
            hipStream_t ls = ctx->lstreams[0].st;
            if (timed) HIPCHK(hipEventRecord(ev0, ls));
            (void)hipGetLastError();
            launch_level_batch(ls, level_kind(H), batch, 1);
            { const hipError_t le = hipGetLastError(); if (le != hipSuccess) throw HipError(std::string("level kernel launch: ") + hipGetErrorString(le)); }
            if (timed) HIPCHK(hipEventRecord(ev1, ls));
            t_batch_launched = level_log ? now_ms() : 0.0; batch_n = 1;
            // while this level runs: let the runtime retire the launches behind it (it does so only when asked, and a
            // region leaves ~1 500 of them for whoever synchronises the device next: ~10 us each)
            (void)hipStreamQuery(ls);
            unsigned spins = 0;
            while (__atomic_load_n(&Rh->seq, __ATOMIC_ACQUIRE) != H.seq) {
                __builtin_ia32_pause();
                if ((++spins & 0x3FFFFu) == 0) {             // every few milliseconds: has the stream died?
                    const hipError_t e = hipStreamQuery(ls);
                    if (e == hipSuccess) { if (__atomic_load_n(&Rh->seq, __ATOMIC_ACQUIRE) != H.seq) throw HipError("a level kernel ended without its completion stamp"); }
                    else if (e != hipErrorNotReady) throw HipError(std::string("level kernel: ") + hipGetErrorString(e));
                }
            }
        } else {
            t_posted = now_ms();
            ctx->submit_level(LevelRequest{this, LevelItem{jd_dev, H, level_kind(H) | (level_lds_kb(H, K) << 8), Pm, Rd}, level_kind(H), timed});
            wait_level();
        }
        if (ctx->resident && mslot >= 0) __atomic_store_n(&ctx->mail_done[(size_t)mslot], H.seq, __ATOMIC_RELEASE);
        t_mark = now_ms();                                   // (the wait for the level is not host work)
        level_launches++;
        if (chain) {
            sampler_launches++; sampler_copies += Q;
            draws += (long)Rh->n_draws; exact += (long)Rh->n_exact; slow += (long)Rh->n_slow; sampler_strains += S; passes += (long)Rh->n_pass;
            chain_cycles += Rh->chain_cycles; chain_wall += Rh->chain_wall;
        }
        level_ticks += Rh->level_wall;
        if (chain) sampler_ticks += Rh->level_wall;
        if (level_log) {
            const double t_done = now_ms();
            fprintf(level_log, "h %d mode %d S %d Q %d n %d level_us %.1f chain_us %.1f cyc %llu passes %llu slow %llu xcc %d ncopy %d multi %d "
                    "ph %.1f %.1f %.1f %.1f %.1f host_us %.1f wait_us %.1f pend_us %.1f batch %d\n", job.handle, mode, S, Q,
                    n_sweeps, Rh->level_wall * 0.01, chain ? Rh->chain_wall * 0.01 : 0.0, chain ? (unsigned long long)Rh->chain_cycles : 0ull,
                    chain ? (unsigned long long)Rh->n_pass : 0ull, chain ? (unsigned long long)Rh->n_slow : 0ull, Rh->xcc, H.n_copy, (int)any_multi,
                    Rh->phase_ticks[0] * 0.01, Rh->phase_ticks[1] * 0.01, Rh->phase_ticks[2] * 0.01, Rh->phase_ticks[3] * 0.01, Rh->phase_ticks[4] * 0.01,
                    1e3 * (t_launched - t_last_done), 1e3 * (t_done - t_launched), 1e3 * (t_batch_launched - t_launched), batch_n);
            t_last_done = t_done;
        }
        job.stats.xcd_levels[Rh->xcc & 7]++;
        job.stats.kind_levels[std::min(std::max(level_kind(H), 0), 16)]++;
    };

    struct Cand { int parent; int node; ld abundance; };
    std::vector<Cand> cands;                                  // buffers of the walk, reused from level to level
    std::vector<int> first_child;
    std::vector<HStrain> kept_buf;
    for (int level = 0; level < f.n_levels; level++) {
        const int n0 = f.level_node_ptr[level], n1 = f.level_node_ptr[level + 1];
        for (int x = n0; x < n1; x++) {
            trace_dump("before clustering", level, level_strains);
            const int u = f.level_nodes[x];
            if (u == 0) {
                if (level_strains.empty()) throw ScError(SC_ERR_INTERNAL, "no strain at the root");
                HStrain& s = level_strains[0];
                arena.push_back({0, s.tail});
                s.tail = (int)arena.size() - 1; s.node = 0;
                s.hash = hash_extend(s.hash, f.node_label_str[0]); s.seqlen += (int)f.node_label_str[0].size();
                s.abundance = 1;
            } else if (f.node_is_end[u]) {
                // read_reassign (only its sort has an effect, :672-702), merge_strains (:645-670)
                // Every candidate pruned before the end of the gene: the reference runs into undefined behaviour here
                // (merged(1, strains[0]) of an empty vector, :650) and in practice prints nothing and exits 0; so does
                // this path (no contig for the region).
                if (level_strains.empty()) { final_strains.clear(); continue; }
                sort_strains(level_strains);
                sort_strains(level_strains);
                std::vector<std::string> seqs;
                for (auto& s : level_strains) seqs.push_back(strain_seq(s));
                std::vector<int> merged{0};
                for (int i = 1; i < (int)level_strains.size(); i++) {
                    size_t j;
                    for (j = 0; j < merged.size(); j++)
                        if (seq_identity(seqs[i], seqs[merged[j]]) > 1 - diff) {
                            level_strains[merged[j]].abundance += level_strains[i].abundance;
                            break;
                        }
                    if (j == merged.size()) merged.push_back(i);
                }
                std::vector<HStrain> kept;
                std::vector<char> keep(level_strains.size(), 0);
                for (int j : merged) { kept.push_back(level_strains[j]); keep[j] = 1; }
                for (size_t i = 0; i < level_strains.size(); i++) if (!keep[i]) drop(level_strains[i], free_slots);
                level_strains.swap(kept);
                final_strains = level_strains;
            }
        }
        const int e0 = f.level_ent_ptr[level], e1 = f.level_ent_ptr[level + 1];
        const int Rn = e1 - e0;
        cur_level = level;
        if (Rn > 0 && !level_strains.empty()) {
            const int S = (int)level_strains.size();
            if (S > MAXS) throw ScError(SC_ERR_CAPACITY, "more than 128 candidate strains at one level");
            if ((long)S * K * K > (long)level_table_capacity())
                throw ScError(SC_ERR_UNSUPPORTED, std::to_string(S) + " candidate strains over " + std::to_string(K) +
                              " distinct symbols: their log tables do not fit the level kernel's LDS");
            bool has_dups = false, any_multi = false;
            for (int x = e0; x < e1; x++) { if (!f.ent_first[x]) has_dups = true; if (f.ent_lab_len[x] != 1) any_multi = true; }
            for (auto& s : level_strains) if (f.node_lab_len[s.node] != 1) any_multi = true;
            const int Q = f.level_read_count[level];
            if (branching) {
                // np_bayes_clustering, :128-244 (+ pruning :404-454)
                const int n = std::min(pa.sweeps_cap, pa.draw_budget / Q);
                int last[MAXS];
                for (int s = 0; s < S; s++) {
                    last[s] = s;
                    for (int t = S - 1; t > s; t--)
                        if (level_strains[t].hash == level_strains[s].hash && level_strains[t].seqlen == level_strains[s].seqlen) { last[s] = t; break; }
                }
                ld prior[MAXS], post[MAXS], a[MAXS];
                for (int s = 0; s < S; s++) prior[s] = level_strains[last[s]].abundance;
                run_level(MODE_SAMPLE, e0, e1, Q, n, true, level_strains, has_dups, any_multi);
                ld (*cnt)[KMAX] = reinterpret_cast<ld (*)[KMAX]>(cnt_scratch.data());     // (not thread_local: the fiber changes threads)
                for (int s = 0; s < S; s++) {                      // (cold caches: see run_level)
                    const Model& m_ = models[(size_t)level_strains[s].model];
                    __builtin_prefetch(&Rh->cnt[s * KMAX]);
                    if ((s & 15) == 0) __builtin_prefetch(&Rh->kdraw[s]);
                    __builtin_prefetch(&m_.comp[0]); __builtin_prefetch(&m_.comp[4]); __builtin_prefetch(&m_.stale[0]);
                    const int la = la_cache[s];
                    if (la >= 0 && la < K) { __builtin_prefetch(&m_.sub[la * K]); __builtin_prefetch(&m_.sub[la * K + 4]); }
                }
                for (int s = 0; s < S; s++) for (int b = 0; b < K; b++) cnt[s][b] = 0;
                if (S == 1 || n <= 0) {
                    // a single weight consumes no random numbers (libstdc++ discrete_distribution)
                    a[0] = level_strains[0].abundance;
                    for (int s = 1; s < S; s++) a[s] = level_strains[s].abundance;
                    if (n > 0) {
                        const long tot = (long)n * Q;
                        a[0] = sc_add_ones(a[0], (unsigned long)tot);
                        for (int x = e0; x < e1; x++)
                            if (f.ent_lab_len[x] == 1) cnt[0][f.labels[f.ent_lab_off[x]]] += (ld)n * f.ent_cn[x];
                    }
                } else {
                    for (int s = 0; s < S; s++) {
                        a[s] = sc_add_ones(level_strains[s].abundance, Rh->kdraw[s]);     // a[c] += 1 per draw, :195 (one rounding per draw)
                        for (int b = 0; b < K; b++) cnt[s][b] = (ld)Rh->cnt[s * KMAX + b];
                    }
                }
                ld z = 0;
                for (int s = 0; s < S; s++) z += a[s];
                for (int s = 0; s < S; s++) a[s] /= z;
                for (int s = 0; s < S; s++) a[s] *= Q;
                for (int s = 0; s < S; s++) {
                    HStrain& st_ = level_strains[s];
                    Model& m_ = models[(size_t)st_.model];
                    st_.abundance += a[s];                                   // update_model, Strain.cpp:106-125
                    unsigned changed = 0;
                    {
                        const int la = la_cache[s];                          // the symbol of the candidate's node (single-symbol labels only)
                        if (la >= 0 && la < K) {
                            for (int b = 0; b < K; b++)
                                if (cnt[s][b] > 0) { m_.sub[la * K + b] += cnt[s][b] / n; m_.stale[la] |= (uint16_t)(1u << b); }
                            m_.dirty |= 1u << la;
                            changed = 1u << la;
                        }
                    }
                    recount_rows(m_, changed);
                }
                for (int s = 0; s < S; s++) post[s] = level_strains[last[s]].abundance;
                ld A_delta_max = 0;
                for (int s = 0; s < S; s++) { ld d = post[s] - prior[s]; if (A_delta_max < d) A_delta_max = d; }
                ld Z = 0;
                for (int s = 0; s < S; s++) Z += a[s];
                const ld Zt = Z * tau;
                std::vector<HStrain>& kept = kept_buf;               // (the walk's own: no allocation per level)
                kept.clear();
                for (int s = 0; s < S; s++) {
                    const ld d = post[s] - prior[s];
                    if (a[s] < Zt || d < 0.01 * A_delta_max) drop(level_strains[s], free_slots);
                    else kept.push_back(level_strains[s]);
                }
                level_strains.swap(kept);
            } else {
                // hard_clustering, :17-125
                run_level(MODE_HARD, e0, e1, Q, 0, true, level_strains, has_dups, any_multi);
                for (int s = 0; s < S; s++) {                      // (cold caches: see run_level)
                    const Model& m_ = models[(size_t)level_strains[s].model];
                    for (int o = 0; o < K * K; o += 8) __builtin_prefetch(&Rh->subst[(size_t)s * K * K + o]);
                    for (int o = 0; o < K * K; o += 4) __builtin_prefetch(&m_.sub[o]);
                    __builtin_prefetch(&m_.comp[0]); __builtin_prefetch(&m_.comp[4]); __builtin_prefetch(&m_.stale[0]);
                    if ((s & 7) == 0) __builtin_prefetch(&Rh->abund[s]);
                }
                for (int s = 0; s < S; s++) {
                    HStrain& st_ = level_strains[s];
                    Model& m_ = models[(size_t)st_.model];
                    st_.abundance += (ld)Rh->abund[s];
                    const double* sub_d = Rh->subst + (size_t)s * K * K;      // compact [K][K]
                    unsigned changed = 0;
                    for (int a = 0; a < K; a++)
                        for (int b = 0; b < K; b++) {
                            const double d = sub_d[a * K + b];
                            if (d != 0.0) {                                   // (x + 0.0 == x for every x the counts can hold: they are never -0)
                                m_.sub[a * K + b] += (ld)d;
                                m_.dirty |= 1u << a; m_.stale[a] |= (uint16_t)(1u << b); changed |= 1u << a;
                            }
                        }
                    recount_rows(m_, changed);
                }
            }
        }
        trace_dump("after clustering", level, level_strains);
        lap(1);

        // ---- candidate extension, :473-551
        branching = false;
        cands.clear();
        for (const HStrain& s : level_strains) __builtin_prefetch(&f.out_ptr[(size_t)s.node]);
        for (const HStrain& s : level_strains) {
            const int ob = f.out_ptr[(size_t)s.node];
            __builtin_prefetch(&f.out_node[(size_t)ob]); __builtin_prefetch(&f.out_support[(size_t)ob]);
        }
        for (int si = 0; si < (int)level_strains.size(); si++) {
            const HStrain& s = level_strains[si];
            const int v = s.node;
            const int ob = f.out_ptr[v], oe = f.out_ptr[v + 1];
            ld oz = 0, moc = 0;
            for (int x = ob; x < oe; x++) { const ld oc0 = f.out_support[x]; oz += oc0; if (moc < oc0) moc = oc0; }
            int dd = 0;
            for (int x = ob; x < oe; x++) {
                const int o = f.out_node[x];
                const ld oc = f.out_support[x];
                if (!f.node_is_end[o] && oz > 0) {
                    if (oc <= 1. && oc < moc) { dd += 1; continue; }
                    ld ab;
                    if (oc > 0) ab = s.abundance * oc / oz;
                    else ab = oz * std::min(0.01, (double)tau);
                    cands.push_back({si, o, ab});
                } else {
                    cands.push_back({si, o, s.abundance});
                }
            }
            if (oe - ob > 1 + dd) branching = true;
        }
        if ((int)cands.size() > pa.max_candidates) {                          // :532-551, Qx :246-254
            std::vector<ld> ssa;
            for (auto& c : cands) ssa.push_back(c.abundance);
            std::sort(ssa.begin(), ssa.end(), [](ld x, ld y) { return x > y; });
            const ld Zt0 = (pa.max_candidates >= (int)ssa.size()) ? ssa.back() : ssa[pa.max_candidates];
            std::vector<Cand> kept;
            for (auto& c : cands) if (!(c.abundance < Zt0)) kept.push_back(c);
            cands.swap(kept);
        }
        if ((int)cands.size() > MAXS) throw ScError(SC_ERR_CAPACITY, "more than 128 candidate strains at one level");
        // materialise: the first surviving child of a parent inherits its row, the others copy it
        first_child.assign(level_strains.size(), -1);
        for (int c = 0; c < (int)cands.size(); c++) if (first_child[cands[c].parent] < 0) first_child[cands[c].parent] = c;
        for (size_t p = 0; p < level_strains.size(); p++) if (first_child[p] < 0) drop(level_strains[p], free_slots);
        sub_strains.clear();
        for (int c = 0; c < (int)cands.size(); c++) {
            const HStrain& par = level_strains[cands[c].parent];
            HStrain ns = par;
            if (first_child[cands[c].parent] == c) { ns.slot = par.slot; ns.model = par.model; }
            else {
                if (free_slots.empty()) throw ScError(SC_ERR_CAPACITY, "out of read_loglik rows");
                ns.slot = free_slots.back(); free_slots.pop_back();
                pending_copies.push_back({par.slot, ns.slot});
                ns.model = model_new();                                  // (may move the pool: take the source by index)
                models[(size_t)ns.model] = models[(size_t)par.model];
            }
            arena.push_back({cands[c].node, par.tail});
            ns.tail = (int)arena.size() - 1; ns.node = cands[c].node;
            ns.hash = hash_extend(par.hash, f.node_label_str[cands[c].node]);
            ns.seqlen = par.seqlen + (int)f.node_label_str[cands[c].node].size();
            ns.abundance = cands[c].abundance;
            sub_strains.push_back(ns);
        }
        level_strains.swap(sub_strains);
        sub_strains.clear();
        lap(2);
    }

    // ---- read_assign, :776-836, then the final sort, StrainCall.cpp:1027
    std::vector<HStrain>& fs = final_strains;
    const int S = (int)fs.size();
    if (S > 0) {
        const int Q = (int)total_copies;
        const int n = std::min(pa.sweeps_cap, pa.draw_budget / std::max(Q, 1));
        std::vector<ld> a(S);
        if (S == 1 || n <= 0) {
            for (int s = 0; s < S; s++) a[s] = fs[s].abundance;
            if (n > 0) a[0] = sc_add_ones(a[0], (unsigned long)((long)n * Q));
        } else {
            pending_copies.clear();
            run_level(MODE_SAMPLE, final_e0, final_e0 + n_reads, Q, n, false, fs, false, true);
            for (int s = 0; s < S; s++) {
                a[s] = sc_add_ones(fs[s].abundance, Rh->kdraw[s]);                   // :823, one rounding per draw
            }
        }
        // The last level is done: the mailbox goes to the next region.  What is left of this one -- its sequences, and giving
        // back what the walk has allocated (the graph, the candidates' models: milliseconds of free()) -- is a long stretch, and
        // those belong on the pool's set-up threads: on a continuation thread it would hold up ~100 levels of other regions.
        mail_hold.drop();
        if (ctx->split_exec && FiberPool::in_fiber()) FiberPool::yield();
        ld z = 0;
        for (int s = 0; s < S; s++) z += a[s];
        for (int s = 0; s < S; s++) fs[s].abundance = a[s] / z;
        sort_strains(fs);
        for (auto& s : fs) {
            std::string q;
            std::vector<int> rev;
            for (int t = s.tail; t >= 0; t = arena[t].parent) rev.push_back(arena[t].node);
            for (auto it = rev.rbegin(); it != rev.rend(); ++it) {
                const std::string& pl = f.node_label_str[*it];                 // Strain::plain_seq, Strain.cpp:211-223
                if (pl != "^" && pl != "$" && pl != "-" && pl != "=") q += pl;
            }
            job.seqs.push_back(q);
            job.abund.push_back((double)s.abundance);
        }
    }
    if (level_log) fclose(level_log);
    for (size_t k = 0; k + 1 < ev_used; k += 2) {
        float ms = 0;
        HIPCHK(hipEventSynchronize(ev_pool[k + 1]));
        HIPCHK(hipEventElapsedTime(&ms, ev_pool[k], ev_pool[k + 1]));
        sampler_ms += ms;
    }
    ev_used = 0;
    for (int k = 0; k < 3; k++) job.stats.host_us[k] = 1e3 * host_acc[k];
    for (int k = 0; k < 2; k++) { job.stats.wake_us[k] = 1e3 * wake_acc[k]; wake_acc[k] = 0; }
    job.stats.sampler_kernel_ms = sampler_ms;
    job.stats.sampler_launches = sampler_launches;
    job.stats.sampler_read_copies = sampler_copies;
    job.stats.level_launches = level_launches;
    job.stats.draws = draws;
    job.stats.exact_draws = exact;
    job.stats.slow_draws = slow;
    job.stats.chain_passes = passes;
    job.stats.chain_cycles = (long)chain_cycles;
    job.stats.chain_wall_ticks = (long)chain_wall;
    job.stats.sampler_strains = sampler_strains;
    job.stats.level_kernel_ticks = (long)level_ticks;
    job.stats.sampler_level_ticks = (long)sampler_ticks;
}

void Worker::process(Job& job) {
    const double t0 = now_ms();
    // page-locked staging for the set-up of this region; cluster() hands it back once the last copy has landed
    struct Lease {
        Worker* w;
        ~Lease() { if (w->stage && w->stage != &w->passthrough) w->ctx->release_arena(w->stage); w->stage = nullptr; }
    } lease{this};
    struct Setup {                     // one of the context's set-up places, held until the level walk starts (cluster())
        Worker* w;
        ~Setup() { if (w->setup_held) { w->ctx->setup_leave(); w->setup_held = false; } }
    } setup{this};
    job.stats.queue_ms = t0 - job.t_submit;
    if (ctx->split_exec) FiberPool::yield();           // the set-up belongs on one of the pool's set-up threads
    ctx->setup_enter(this);
    setup_held = true;
    job.stats.place_ms = now_ms() - t0;
    stage = ctx->lease_arena(&passthrough);
    MsaFn msa = [this](const std::vector<std::string>& seqs, std::vector<std::string>& rows) { return msa_device(seqs, rows); };
    ThreadFn thr = [this, &job](const std::string& G, const std::vector<AlignedRead>& R, const std::vector<std::vector<CigarOp>>& cg,
                          ThreadTables& T) {
        thread_device(G, R, cg, T);
        if (job.params.graph_only || job.params.want_graph) {      // kept for sc_roi_thread_tables
            job.thr_count = T.count; job.thr_first = T.minrid; job.thr_pool = T.pool; job.thr_sym.assign(T.sym.begin(), T.sym.end());
        }
    };
    // a context with one region in flight gives the region's bulk copies (class pools, flattening: 88 M entries on the
    // unthinned configs[3] region) the rank's other CPUs; with many in flight those already run other regions
    sc::set_graph_threads(ctx->workers.size() == 1 ? std::min(8, std::max(1, (int)(sc::cpu_budget_host() / sc::local_world_size()))) : 1);
    PoGraph g(job.ref, job.reads, msa, thr);
    job.stats.msa_calls = g.msa_calls;
    if (job.params.graph_only || job.params.want_graph) job.graph_dump = g.dump();      // -G text, PartialOrderGraph.cpp:318-337
    FlatGraph& f = flat;               // (the slot's arrays, reused from region to region)
    f.reset();
    flatten(g, (int)job.reads.size(), f);
    job.stats.n_nodes = f.n_nodes; job.stats.n_levels = f.n_levels; job.stats.n_unique_reads = (int)job.reads.size();
    long copies = 0;
    for (auto& r : job.reads) copies += r.cn;
    job.stats.n_read_copies = copies;
    const double t1 = now_ms();
    job.stats.graph_ms = t1 - t0;
    if (!job.params.graph_only) {
        if (!f.unsupported.empty()) throw ScError(SC_ERR_UNSUPPORTED, f.unsupported);
        cluster(job, g, f);
    }
    job.stats.cluster_ms = now_ms() - t1;
}

// The body of a slot's fiber: takes regions off the context's queue until the context stops; parks in `idle` while there
// is none (sc_roi_submit makes one idle slot ready per region it queues).
void Worker::run() {
    for (;;) {
        std::shared_ptr<Job> job;
        bool parked = false;
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            if (!ctx->queue.empty()) { job = ctx->queue.front(); ctx->queue.pop_front(); }
            else if (ctx->stop) break;
            else { ctx->idle.push_back(this); parked = true; }
        }
        if (parked) { FiberPool::park(); continue; }
        try {
            if (!st || !Ph) throw HipError("the slot's stream or host-mapped blocks were not created");
            process(*job);
            job->rc = SC_OK;
        } catch (const ScError& ex) { job->rc = ex.code; job->err = ex.what(); }
        catch (const HipError& ex) { job->rc = SC_ERR_HIP; job->err = ex.what(); }
        catch (const std::exception& ex) { job->rc = SC_ERR_INTERNAL; job->err = ex.what(); }
        catch (...) { job->rc = SC_ERR_INTERNAL; job->err = "unknown exception"; }
        // the last region in flight takes the resident grid with it -- before the caller learns that the region is done,
        // so that whoever waits for the region and then synchronises the device finds the grid on its way out
        if (ctx->regions_active.fetch_sub(1, std::memory_order_acq_rel) == 1 && ctx->resident) ctx->resident_idle();
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            job->status = 1;
            if (job->rc != SC_OK) ctx->last_error = job->err;
        }
        ctx->cv_done.notify_all();
    }
    ctx->fibers_left.fetch_sub(1, std::memory_order_release);
}

}  // namespace sc

using namespace sc;
struct sc_ctx { Ctx c; };

extern "C" {

int sc_ctx_create(int device, int stream_count, sc_ctx** out) {
    if (!out) return SC_ERR_ARG;
    *out = nullptr;
    // 16 hardware queues run side by side on this GPU (more are time-sliced: measured); the launch and setup streams below
    // want one each.  Only effective if HIP is not initialised yet in this process (rambl_amd/__init__.py sets it too).
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SC_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SC_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SC_ERR_NO_DEVICE;   // kernels are built for gfx950 only
    if (hipSetDevice(device) != hipSuccess) return SC_ERR_HIP;
    if (init_kernels() != 0) return SC_ERR_HIP;
    // the context's threads (they inherit the mask of the thread that starts them) and the host memory it allocates and first
    // touches here: next to the GPU; the caller's own mask comes back when this function returns
    struct Near {
        cpu_set_t before; bool moved = false;
        explicit Near(int dev) {
            cpu_set_t local;
            if (sched_getaffinity(0, sizeof before, &before) == 0 && sc::gpu_local_cpus(dev, &local)) moved = sched_setaffinity(0, sizeof local, &local) == 0;
        }
        ~Near() { if (moved) (void)sched_setaffinity(0, sizeof before, &before); }
    } near{device};
    {
        // A region builds its graph out of ~10^5 small allocations and a few of tens of megabytes; handed back to the system
        // and mapped again for every region they cost their size in page faults.  Keep freed memory in the process.
        static std::once_flag once;
        std::call_once(once, [] {
            const char* e = getenv("SC_MALLOC_TUNE");
            if (e && atoi(e) == 0) return;
            mallopt(M_MMAP_THRESHOLD, 32 << 20);
            mallopt(M_TRIM_THRESHOLD, 1 << 30);
            mallopt(M_TOP_PAD, 64 << 20);
        });
    }
    sc_ctx* h = new sc_ctx();
    Ctx* ctx = &h->c;
    ctx->device = device;
    std::vector<double> u = uniform_stream(1234u, MAX_DRAWS + 2048);   // padded: the chain stages windows of 1024
    if (hipMalloc((void**)&ctx->dU, sizeof(double) * u.size()) != hipSuccess ||
        hipMemcpy(ctx->dU, u.data(), sizeof(double) * u.size(), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return SC_ERR_HIP;
    }
    std::vector<float> uf(u.begin(), u.end());
    if (hipMalloc((void**)&ctx->dUf, sizeof(float) * uf.size()) != hipSuccess ||
        hipMemcpy(ctx->dUf, uf.data(), sizeof(float) * uf.size(), hipMemcpyHostToDevice) != hipSuccess) {
        delete h;
        return SC_ERR_HIP;
    }
    if (stream_count < 1) stream_count = 1;
    if (stream_count > 512) stream_count = 512;
    {
        // resident level workers: one workgroup per slot holds a CU (and all of its LDS) while regions are in flight, so the
        // slots stop short of the 256 CUs -- the set-up kernels of the regions (read threading, MSA, edge support) need CUs too
        // Default: resident workers when several regions are in flight (no launch per level, no stream held by the slowest
        // level of a batch: +30-60 % reads/s at 128-224 in flight); a launch per level for a single region (its level
        // kernels are then kernels of their own, which the compiler allocates ~5 % faster than the same code behind a call).
        const char* e = getenv("SC_RESIDENT");
        ctx->resident = e ? atoi(e) != 0 : stream_count > 1;
        const char* rs = getenv("SC_RESIDENT_SLOTS");
        // Measured on MI355X: 224 resident workgroups (8 wavefronts each, 1 792 in all) start, the ones beyond do not (232: their
        // regions wait for ever, or the grid faults) -- the kernel keeps its variants as functions, their stack frames live in
        // scratch memory, and the queue's scratch holds 7 wavefronts per CU.  32 CUs stay free for the set-up kernels.
        int cap = rs ? atoi(rs) : std::max(prop.multiProcessorCount - 32, 1);
        cap = cap < 1 ? 1 : (cap > prop.multiProcessorCount ? prop.multiProcessorCount : cap);
        ctx->res_slots = ctx->resident ? std::min(stream_count, cap) : stream_count;
        // workers = regions walking (one mailbox each) + regions being set up meanwhile: those the caller asks for beyond the
        // mailboxes (stream_count above the cap) or SC_SETUP_WORKERS.  None by default: on a 16-CPU share of a host the set-ups
        // are bounded by the CPUs, not by the workers that wait for one (measured: 0 / 28 / 56 extra, no difference beyond noise)
        if (ctx->resident && ctx->res_slots > 1) {
            int extra = 0;
            if (const char* ex = getenv("SC_SETUP_WORKERS")) extra = std::max(0, atoi(ex));
            stream_count = std::min(std::max(stream_count, ctx->res_slots + extra), 512);
        }
    }
    if (ctx->resident) {
        // The resident grid stays in its hardware queue for as long as regions are in flight: nothing else may ever be
        // queued behind it (a set-up kernel of a region behind the grid that waits for that region's levels would never
        // start).  Streams share hardware queues once there are more streams than queues, and queues are kept per
        // priority: the grid's stream is the only one of its priority, and this context creates few other streams.
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&ctx->rstream, hipStreamNonBlocking, hi) != hipSuccess) { sc_ctx_destroy(h); return SC_ERR_HIP; }
    }
    {
        const char* e = getenv("SC_LAUNCH_STREAMS");
        int nl = e ? atoi(e) : 11;
        nl = nl < 1 ? 1 : (nl > 30 ? 30 : nl);
        if (ctx->resident) nl = 1;                 // levels are not launched: one stream for the rare grid kernels of huge levels
        if (nl > stream_count) nl = stream_count;
        const int nset = stream_count >= 8 ? 4 : (stream_count > 1 ? 2 : 1);      // 11 + 4 + the null stream = 16 hardware queues
        ctx->lstreams.resize((size_t)nl);
        ctx->setup_streams.resize((size_t)nset);
        bool ok = true;
        for (auto& ls : ctx->lstreams) ok = ok && hipStreamCreateWithFlags(&ls.st, hipStreamNonBlocking) == hipSuccess;
        for (auto& ss : ctx->setup_streams) ok = ok && hipStreamCreateWithFlags(&ss, hipStreamNonBlocking) == hipSuccess;
        if (!ok) { sc_ctx_destroy(h); return SC_ERR_HIP; }
    }
    int plan[3] = {1, 0, 1};
    (void)sc_host_plan(stream_count, 0, 0.0, plan);
    {
        // resident contexts of several regions: the executors watch the stamps themselves, and the level server's CPU is one more
        // executor's (SC_POLL_EXEC=0: the level server as for a launch per level)
        const char* pe = getenv("SC_POLL_EXEC");
        ctx->poll_exec = ctx->resident && stream_count > 1 && !(pe && atoi(pe) == 0);
        if (ctx->poll_exec && plan[1] > 0) plan[0] = std::min(plan[0] + 1, std::min(stream_count, 32));
    }
    if (const char* e = getenv("SC_EXEC_THREADS")) { const int k = atoi(e); if (k >= 1) plan[0] = std::min(k, stream_count); }
    {
        // page-locked staging of the regions' transfers: only worth it while other regions are in flight (it is their queues
        // that a pageable copy suspends); as many arenas as regions can be set up at once on this rank's executor threads
        const char* ps = getenv("SC_PINNED_STAGING");
        const bool want = ps ? atoi(ps) != 0 : stream_count > 1;
        ctx->arena_limit = want ? std::max(plan[0] + 1, 2) : 0;
        ctx->setup_limit = std::max(1, (plan[0] + 1) / 2);
        if (const char* e = getenv("SC_SETUP_LIMIT")) ctx->setup_limit = std::max(1, atoi(e));
    }
    // the slots' host-mapped parameter / result blocks: one allocation each for all slots
    const size_t ns = (size_t)stream_count;
    if (hipHostMalloc((void**)&ctx->P_all, ns * sizeof(LevelParams), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipHostMalloc((void**)&ctx->R_all, ns * sizeof(LevelResult), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
        hipMalloc((void**)&ctx->Pd_all, ns * sizeof(LevelParams)) != hipSuccess) {
        sc_ctx_destroy(h);
        return SC_ERR_HIP;
    }
    try {
        for (int i = 0; i < stream_count; i++) {
            auto w = std::make_unique<Worker>();
            w->ctx = ctx;
            w->slot = i;
            w->st = ctx->setup_streams[(size_t)i % ctx->setup_streams.size()];
            w->Ph = ctx->P_all + i; w->Rh = ctx->R_all + i; w->Pd = ctx->Pd_all + i;
            ctx->workers.push_back(std::move(w));
        }
        for (auto& w : ctx->workers) w->init();
    } catch (const std::exception& ex) {
        ctx->last_error = ex.what();
        sc_ctx_destroy(h);
        return SC_ERR_HIP;
    }
    if (ctx->resident) {
        const size_t nm = (size_t)ctx->res_slots;
        ctx->mail_seq.assign(nm, 0u); ctx->mail_done.assign(nm, 0u);
        for (int m = ctx->res_slots - 1; m >= 0; m--) ctx->free_mail.push_back(m);
        if (hipHostMalloc((void**)&ctx->mail_h, nm * sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostMalloc((void**)&ctx->ctl_h, sizeof(ResidentCtl), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer((void**)&ctx->mail_d, ctx->mail_h, 0) != hipSuccess ||
            hipHostGetDevicePointer((void**)&ctx->ctl_d, ctx->ctl_h, 0) != hipSuccess) {
            sc_ctx_destroy(h);
            return SC_ERR_HIP;
        }
        std::memset(ctx->mail_h, 0, nm * sizeof(Mailbox));
        std::memset(ctx->ctl_h, 0, sizeof(ResidentCtl));
        ctx->heart = std::thread([ctx] {
            unsigned beats = 0;
            while (!ctx->heart_stop.load(std::memory_order_acquire)) {
                __atomic_fetch_add(&ctx->ctl_h->heartbeat, 1u, __ATOMIC_RELEASE);
                if (ctx->poll_exec && ctx->pool && (++beats % 50u) == 0) ctx->poll_health();
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    }
    const int dev = device;
    // half of the executor threads take the regions' set-ups first (graph construction: tens of milliseconds each), the other
    // half never do: the continuation of a region whose level has come back is a few tens of microseconds and must not wait
    int n_long = plan[0] >= 2 ? plan[0] / 2 : 0;
    if (const char* e = getenv("SC_EXEC_LONG")) n_long = std::max(0, std::min(atoi(e), plan[0] - 1));
    ctx->split_exec = n_long > 0;
    if (!getenv("SC_SETUP_LIMIT") && n_long > 0) ctx->setup_limit = 2 * n_long;      // a set-up waits for the GPU part of its time
    ctx->pool.reset(new FiberPool(plan[0], [dev] { (void)hipSetDevice(dev); }, n_long));
    if (getenv("SC_SERVER_LOG")) { ctx->pool->set_diag(true); ctx->wake_hist = new std::atomic<long>[24](); ctx->t_created = now_ms(); ctx->n_fast = plan[0] - n_long; ctx->n_long = n_long; }
    if (ctx->poll_exec) {
        ctx->polled.reset(new std::atomic<uint8_t>[ctx->workers.size()]());
        ctx->pool->set_poll([ctx] { return ctx->poll_stamps(); });
    } else if (stream_count > 1) {
        ctx->server = std::thread([ctx] { ctx->serve_levels(); });
    }
    ctx->fibers_left.store(stream_count, std::memory_order_release);
    for (auto& w : ctx->workers) {
        Worker* p = w.get();
        p->fib = ctx->pool->create([p] { p->run(); });
        if (!p->fib) { ctx->last_error = "cannot map a fiber stack"; ctx->fibers_left.fetch_sub(1); continue; }
        ctx->pool->make_ready(p->fib);
    }
    *out = h;
    return SC_OK;
}

void sc_ctx_destroy(sc_ctx* h) {
    if (!h) return;
    Ctx* ctx = &h->c;
    if (ctx->pool) {
        std::vector<Worker*> wake;
        { std::lock_guard<std::mutex> lk(ctx->mu); ctx->stop = true; wake.swap(ctx->idle); }
        for (Worker* w : wake) ctx->pool->make_ready(w->fib);
        // regions still queued or in flight are finished first (as the worker threads of earlier versions did)
        while (ctx->fibers_left.load(std::memory_order_acquire) > 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    { std::lock_guard<std::mutex> lk(ctx->dmu); ctx->server_stop.store(true, std::memory_order_seq_cst); }
    ctx->dcv.notify_all();
    if (ctx->server.joinable()) ctx->server.join();
    if (ctx->pool) ctx->pool->shutdown();
    if (ctx->wake_hist) {
        // diagnostics: how busy the two kinds of executor were (time stamp counter against the wall clock of the context's
        // life), how long their stretches inside fibers were, how long a region whose level had come back waited for one
        fprintf(stderr, "executors: %d continuation + %d set-up threads over %.1f ms; inside fibers %.1f / %.1f Mticks\n", ctx->n_fast, ctx->n_long,
                now_ms() - ctx->t_created, ctx->pool->busy_ticks(false) * 1e-6, ctx->pool->busy_ticks(true) * 1e-6);
        for (int l = 0; l < 2; l++) {
            fprintf(stderr, "  stretches on %s threads (log2 ticks: count):", l ? "set-up" : "continuation");
            for (int b = 0; b < 40; b++) if (ctx->pool->stretch_count(l, b)) fprintf(stderr, " %d:%ld", b, ctx->pool->stretch_count(l, b));
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "  wake latency (below 2^b us: count):");
        for (int b = 0; b < 24; b++) if (ctx->wake_hist[b].load()) fprintf(stderr, " %d:%ld", b, ctx->wake_hist[b].load());
        fprintf(stderr, "\n");
        delete[] ctx->wake_hist; ctx->wake_hist = nullptr;
    }
    (void)hipSetDevice(ctx->device);
    ctx->resident_shutdown();
    ctx->heart_stop.store(true, std::memory_order_release);
    if (ctx->heart.joinable()) ctx->heart.join();
    if (ctx->rstream) (void)hipStreamDestroy(ctx->rstream);
    if (ctx->mail_h) (void)hipHostFree(ctx->mail_h);
    if (ctx->ctl_h) (void)hipHostFree(ctx->ctl_h);
    for (auto& w : ctx->workers) {
        for (hipEvent_t e : w->ev_pool) (void)hipEventDestroy(e);
        if (w->sync_ev) (void)hipEventDestroy(w->sync_ev);
        if (w->st && w->own_stream) (void)hipStreamDestroy(w->st);
    }
    ctx->workers.clear();
    if (ctx->P_all) (void)hipHostFree(ctx->P_all);
    if (ctx->R_all) (void)hipHostFree(ctx->R_all);
    if (ctx->Pd_all) (void)hipFree(ctx->Pd_all);
    for (auto& ls : ctx->lstreams) if (ls.st) (void)hipStreamDestroy(ls.st);
    for (auto& ss : ctx->setup_streams) if (ss) (void)hipStreamDestroy(ss);

    for (PinnedArena* a : ctx->arenas) delete a;
    ctx->arenas.clear(); ctx->free_arenas.clear();
    if (ctx->dU) (void)hipFree(ctx->dU);
    if (ctx->dUf) (void)hipFree(ctx->dUf);
    delete h;
}

const char* sc_last_error(sc_ctx* h) {
    if (!h) return "";
    std::lock_guard<std::mutex> lk(h->c.mu);
    static thread_local std::string copy;
    copy = h->c.last_error;
    return copy.c_str();
}

const char* sc_roi_error(sc_ctx* h, int handle) {
    if (!h) return "";
    std::lock_guard<std::mutex> lk(h->c.mu);
    auto it = h->c.jobs.find(handle);
    return (it == h->c.jobs.end() || it->second->status != 1) ? "" : it->second->err.c_str();
}

int sc_roi_submit(sc_ctx* h, const char* ref_bases, int ref_len, const int* read_pos, const char* cigar_text,
                  const int* cigar_off, const char* seq_text, const int* seq_off, const int* read_copies,
                  const int* mate_idx, const int* mate_off, int n_reads, const sc_params* params, int* handle_out) {
    if (!h || !ref_bases || ref_len < 0 || n_reads < 0 || !params || !handle_out) return SC_ERR_ARG;
    if (n_reads > 0 && (!read_pos || !cigar_text || !cigar_off || !seq_text || !seq_off || !read_copies || !mate_off)) return SC_ERR_ARG;
    // the device buffers are sized for the reference's literals (uniform stream of MAX_DRAWS values, MAXS rows)
    if (params->draw_budget < 1 || params->draw_budget > MAX_DRAWS || params->sweeps_cap < 0 || params->max_candidates < 1 ||
        params->max_candidates > MAXS)
        return SC_ERR_ARG;
    if (n_reads > 0 && (cigar_off[0] < 0 || seq_off[0] < 0 || mate_off[0] != 0)) return SC_ERR_ARG;
    for (int i = 0; i < n_reads; i++)
        if (read_copies[i] < 1 || cigar_off[i + 1] < cigar_off[i] || seq_off[i + 1] < seq_off[i] || mate_off[i + 1] < mate_off[i])
            return SC_ERR_ARG;
    auto job = std::make_shared<Job>();
    job->t_submit = now_ms();
    job->ref.assign(ref_bases, (size_t)ref_len);
    job->reads.resize((size_t)n_reads);
    for (int i = 0; i < n_reads; i++) {
        if (read_pos[i] < 0 || read_pos[i] > ref_len) return SC_ERR_ARG;
        job->reads[i].pos = read_pos[i];
        job->reads[i].cigar.assign(cigar_text + cigar_off[i], (size_t)(cigar_off[i + 1] - cigar_off[i]));
        job->reads[i].seq.assign(seq_text + seq_off[i], (size_t)(seq_off[i + 1] - seq_off[i]));
        job->reads[i].cn = read_copies[i];
    }
    job->mate_off.assign(mate_off ? mate_off : nullptr, mate_off ? mate_off + n_reads + 1 : nullptr);
    if (job->mate_off.empty()) job->mate_off.assign(1, 0);
    const int nm = job->mate_off.back();
    if (nm > 0 && !mate_idx) return SC_ERR_ARG;
    job->mate_idx.assign(mate_idx, mate_idx + nm);
    for (int v : job->mate_idx) if (v < -1 || v >= n_reads) return SC_ERR_ARG;
    job->params = *params;
    Ctx* ctx = &h->c;
    Worker* wake = nullptr;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        job->handle = ctx->next_handle++;
        ctx->jobs[job->handle] = job;
        ctx->queue.push_back(job);
        ctx->regions_active.fetch_add(1, std::memory_order_acq_rel);
        *handle_out = job->handle;
        if (!ctx->idle.empty()) { wake = ctx->idle.back(); ctx->idle.pop_back(); }
    }
    if (wake) ctx->pool->make_ready(wake->fib);
    return SC_OK;
}

static std::shared_ptr<Job> find_job(sc_ctx* h, int handle) {
    std::lock_guard<std::mutex> lk(h->c.mu);
    auto it = h->c.jobs.find(handle);
    return it == h->c.jobs.end() ? nullptr : it->second;
}

int sc_roi_wait(sc_ctx* h, int handle) {
    if (!h) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job) return SC_ERR_ARG;
    std::unique_lock<std::mutex> lk(h->c.mu);
    h->c.cv_done.wait(lk, [&] { return job->status == 1; });
    if (job->rc != SC_OK) h->c.last_error = job->err;
    return job->rc;
}

int sc_roi_result(sc_ctx* h, int handle, char* seq_buf, long seq_cap, int* seq_off, double* abundance, int max_strains,
                  int* n_strains) {
    if (!h || !n_strains) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job || job->status != 1) return SC_ERR_ARG;
    if (job->rc != SC_OK) return job->rc;
    const int n = (int)job->seqs.size();
    *n_strains = n;
    long tot = 0;
    for (auto& s : job->seqs) tot += (long)s.size();
    if (n > max_strains || tot > seq_cap || !seq_buf || !seq_off || !abundance) return SC_ERR_CAPACITY;
    long o = 0;
    for (int i = 0; i < n; i++) {
        seq_off[i] = (int)o;
        std::memcpy(seq_buf + o, job->seqs[i].data(), job->seqs[i].size());
        o += (long)job->seqs[i].size();
        abundance[i] = job->abund[i];
    }
    seq_off[n] = (int)o;
    return SC_OK;
}

static int copy_text(const std::string& s, char* buf, long cap, long* len_out) {
    if (len_out) *len_out = (long)s.size();
    if (!buf || cap < (long)s.size()) return SC_ERR_CAPACITY;
    std::memcpy(buf, s.data(), s.size());
    return SC_OK;
}
int sc_roi_graph_dump(sc_ctx* h, int handle, char* buf, long cap, long* len_out) {
    if (!h) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job || job->status != 1) return SC_ERR_ARG;
    if (job->rc != SC_OK && job->graph_dump.empty()) return job->rc;
    return copy_text(job->graph_dump, buf, cap, len_out);
}
int sc_roi_trace(sc_ctx* h, int handle, char* buf, long cap, long* len_out) {
    if (!h) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job || job->status != 1) return SC_ERR_ARG;
    if (job->rc != SC_OK) return job->rc;
    return copy_text(job->trace, buf, cap, len_out);
}
int sc_roi_stats(sc_ctx* h, int handle, sc_stats* out) {
    if (!h || !out) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job || job->status != 1) return SC_ERR_ARG;
    *out = job->stats;
    return SC_OK;
}
int sc_roi_edge_support(sc_ctx* h, int handle, int* support, int cap, int* n_edges) {
    if (!h || !n_edges) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job || job->status != 1) return SC_ERR_ARG;
    if (job->rc != SC_OK) return job->rc;
    *n_edges = (int)job->edge_support.size();
    if (!support || cap < *n_edges) return SC_ERR_CAPACITY;
    std::memcpy(support, job->edge_support.data(), sizeof(int) * job->edge_support.size());
    return SC_OK;
}
int sc_roi_thread_tables(sc_ctx* h, int handle, int* count, int* first_read, int cls_cap, int* pool, long pool_cap,
                         char* symbols, int* n_cls, long* n_pool) {
    if (!h || !n_cls || !n_pool) return SC_ERR_ARG;
    auto job = find_job(h, handle);
    if (!job || job->status != 1) return SC_ERR_ARG;
    *n_cls = (int)job->thr_count.size();
    *n_pool = (long)job->thr_pool.size();
    if (!count || !first_read || !pool || !symbols || cls_cap < *n_cls || pool_cap < *n_pool) return SC_ERR_CAPACITY;
    std::memcpy(count, job->thr_count.data(), sizeof(int) * job->thr_count.size());
    std::memcpy(first_read, job->thr_first.data(), sizeof(int) * job->thr_first.size());
    std::memcpy(pool, job->thr_pool.data(), sizeof(int) * job->thr_pool.size());
    std::memset(symbols, 0, 8);
    std::memcpy(symbols, job->thr_sym.data(), std::min<size_t>(8, job->thr_sym.size()));
    return SC_OK;
}
int sc_roi_release(sc_ctx* h, int handle) {
    if (!h) return SC_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->c.mu);
    auto it = h->c.jobs.find(handle);
    if (it == h->c.jobs.end() || it->second->status != 1) return SC_ERR_ARG;
    h->c.jobs.erase(it);
    return SC_OK;
}

int sc_msa_align(sc_ctx* h, const char* seq_text, const int* seq_off, int n, char* rows_out, long cap, int* ncol_out) {
    if (!h || !seq_text || !seq_off || n < 1 || !ncol_out) return SC_ERR_ARG;
    Ctx* ctx = &h->c;
    // runs on a private worker object (own stream) so it can be called while regions are in flight
    try {
        HIPCHK(hipSetDevice(ctx->device));
        Worker w;
        w.ctx = ctx;
        w.init();
        w.stage = &w.passthrough;
        w.passthrough.on = false;
        std::vector<std::string> seqs((size_t)n), rows;
        for (int i = 0; i < n; i++) seqs[i].assign(seq_text + seq_off[i], (size_t)(seq_off[i + 1] - seq_off[i]));
        int ncol;
        if (n == 1) { ncol = (int)seqs[0].size(); rows = seqs; }
        else ncol = w.msa_device(seqs, rows);
        *ncol_out = ncol;
        int rc = SC_OK;
        if (!rows_out || (long)n * (ncol + 1) > cap) rc = SC_ERR_CAPACITY;
        else for (int i = 0; i < n; i++) { std::memcpy(rows_out + (long)i * (ncol + 1), rows[i].data(), (size_t)ncol); rows_out[(long)i * (ncol + 1) + ncol] = 0; }
        (void)hipHostFree(w.Ph); (void)hipFree(w.Pd); (void)hipHostFree(w.Rh);
        (void)hipEventDestroy(w.sync_ev);
        (void)hipStreamDestroy(w.st);
        return rc;
    } catch (const ScError& ex) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->last_error = ex.what(); return ex.code; }
    catch (const std::exception& ex) { std::lock_guard<std::mutex> lk(ctx->mu); ctx->last_error = ex.what(); return SC_ERR_HIP; }
}

}  // extern "C"

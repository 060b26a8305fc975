// Device-side data layout shared by the host driver (sc_api.cpp) and the
// HIP kernels (sc_kernels.hip).  All pointers are device pointers unless the
// field name ends in _h (host-mapped pinned memory written/read by kernels).
#pragma once
#include <cstdint>

namespace sc {

constexpr int MAXS = 128;        // candidate strains alive at one level (reference keeps <= ~81)
constexpr int KMAX = 16;         // symbols: A C G T - = plus up to ten others (N and the IUPAC codes of a 16S reference)
constexpr int KK = KMAX * KMAX;  // capacity; the tables of a level are compact [K][K] with K = symbols of the window
constexpr int MAX_DRAWS = 40000; // NonparametricClustering.cpp:160 / :781 draw budget

enum LevelMode { MODE_HARD = 0, MODE_SAMPLE = 1 };

// Static per-ROI arrays (one set per job, resident in HBM for the whole walk).
struct JobDev {
    // level entries (NonparametricClustering.cpp:318-322 level_reads), level-major
    const int* ent_rid;
    const int* ent_cn;
    const int* ent_lab_off;
    const int* ent_lab_len;
    const uint8_t* ent_first;
    const int* ent_qoff;         // prefix sum of copy numbers inside the entry's level
    const uint8_t* labels;       // symbol codes
    // ReadPairs (PartialOrderGraph.hpp:88) as CSR: mate of copy k of read r = mate_idx[mate_ptr[r]+k]
    const int* mate_ptr;
    const int* mate_idx;
    int n_reads;
    int K;
    int code_N;
    // per-strain read log-likelihood rows (Strain::read_loglik), ll[slot*ll_stride + rid]
    double* ll;
    long ll_stride;
    uint8_t* has;                // read present in read_loglik (identical for every strain)
    const double* U;             // generate_canonical<double,53>(mt19937(1234)) stream, MAX_DRAWS values
    const float* Uf;             // the same stream rounded to fp32 (first tier of the sampler)
    // scratch
    uint8_t* isnew;              // [max entries per level]
    double* tabA;                // [MAXS][qcap]  responsibilities of the soft update (strain-major)
    float* tabLf;                // [qcap][stride]  sampler weights, fp32, when they do not fit in LDS
    uint8_t* qcode;              // [qcap] read-label symbol code of a draw slot (0xFF: not a single symbol)
    int* qent;                   // [qcap] entry index of a draw slot
    int* quid;                   // [qcap] mate read id of a draw slot (-1 none)
    long qcap;
};

// Per-level parameters.  The scalars travel as kernel arguments (LevelHdr); the per-strain part lives
// in host-mapped pinned memory that the kernel of the level reads directly over PCIe, once, into LDS
// (no copy in front of the launch).  Only the first h.S entries of `sp` / rows of `lpt` and the first
// h.n_copy copy pairs are read.
struct LevelHdr {
    int mode;
    int S;                       // strains at this level
    int e0, e1;                  // entry range
    int has_dups;                // an rid occurs twice in this level
    int any_multi;               // some label (strain node or entry) has more than one symbol
    int Q;                       // read_size = sum of copy numbers
    int n_sweeps;                // min(5000, 40000/Q)
    int n_copy;                  // row copies to perform first
    int do_update;               // apply the level's read log-likelihood update (not for read_assign)
    int done;                    // LV_* pieces already run by grid kernels (very large levels only)
    int copy_n;                  // leading cells of a row that can hold a value yet (reads and mates met so far): what a row copy moves
    unsigned seq;                // completion stamp the kernel stores into LevelResult::seq when it is done
};
struct StrainParam {
    int slot;                    // ll row of the strain
    int lab_off, lab_len;        // node label of the strain (into labels)
    int pad;
    double a0;                   // abundance before clustering
    double logpri;               // log(a_s / sum a) for the hard update
};
struct LevelParams {
    int copy_src[MAXS], copy_dst[MAXS];
    StrainParam sp[MAXS];
    double lpt[MAXS * KK];       // log sub(a,b) - log comp(a): compact [S][K][K], K = JobDev::K
};

// One launch serves the current level of up to MAXB regions: workgroup b takes batch.it[b].  What the level needs
// to find its region travels in the kernel-argument segment: a pointer to the region's JobDev (device memory, filled
// once per region, read through the constant address space), the level's scalars, the host-mapped parameter /
// result blocks.  MAXB keeps the segment below 4 KB.
struct LevelParams;
struct LevelResult;
struct LevelItem {
    const JobDev* job;           // device memory; constant while the region is walked
    LevelHdr h;
    int kind;                    // level_kind(h): which variant of the level kernel serves it (k_level_any looks here)
    const LevelParams* P;        // host-mapped
    LevelResult* R;              // host-mapped
};
constexpr int MAXB = 48;
struct LevelBatch { LevelItem it[MAXB]; };
static_assert(sizeof(LevelItem) == 80 && sizeof(LevelBatch) <= 4096, "kernel-argument segment");

// Resident level workers (k_level_resident): one workgroup per slot of the context stays on its CU while regions are in
// flight and takes the slot's levels from a mailbox in host-mapped memory instead of being launched once per level.
// The host writes `item`, then `seq` (release); the workgroup polls `seq`, runs the level, stamps LevelResult::seq as the
// launched kernels do and polls again.  It ends when the context raises `stop` (no region in flight any more, or the
// context is destroyed) or when the host's heartbeat stands still for `idle_ticks` (a host that has died or hangs).
struct alignas(128) Mailbox {
    LevelItem item;              // the level to run
    unsigned seq;                // host: LevelHdr::seq of `item`, stored last
    unsigned ack;                // host: the slot's last stamp before this generation of the grid was launched
    unsigned state;              // kernel: 1 resident, 2 gone, 3 gone after an item that did not name this slot's blocks
    unsigned levels;             // kernel: levels served by this generation (diagnostics)
    unsigned pad[8];
};
static_assert(sizeof(Mailbox) == 128, "one mailbox per 128 bytes");
struct ResidentCtl {
    unsigned stop;               // host: nonzero -> every workgroup leaves after its current level
    unsigned heartbeat;          // host: keeps changing while the context's server thread is alive
    unsigned pad[30];
};
struct LevelParams;
struct LevelResult;
struct ResidentArgs {
    Mailbox* mail;               // host-mapped, [slots]
    const ResidentCtl* ctl;      // host-mapped
    unsigned long long idle_ticks;   // 100 MHz ticks without a heartbeat change after which a workgroup gives up
    const LevelParams* P_base;   // the workers' host-mapped parameter / result blocks: an item must name one of them
    LevelResult* R_base;
    int n_blocks;
};

// Per-level results, written by the kernel into host-mapped pinned memory; `seq` last (system-scope release).
struct LevelResult {
    double abund[MAXS];          // HARD: sum of responsibilities; SAMPLE: urn weights a[] after the sweeps
    double subst[MAXS * KK];     // HARD: responsibility-weighted substitution counts, compact [S][K][K]
    unsigned cnt[MAXS * KMAX];   // SAMPLE: draws per (strain, read symbol)
    unsigned kdraw[MAXS];        // SAMPLE: draws per strain (abund = a0 + kdraw, exactly)
    unsigned long long n_draws;
    unsigned long long n_slow;   // draws that went through the fp64 scan tier
    unsigned long long n_pass;   // window passes of the sampler chain
    unsigned long long chain_cycles, chain_wall;   // shader cycles / 100 MHz ticks spent in the urn chain
    unsigned long long n_exact;  // draws resolved by the literal fp64 path
    unsigned long long level_wall;                 // 100 MHz ticks from the start of the level's kernel to its end
    unsigned phase_ticks[6];     // diagnostics: 100 MHz ticks at the end of staging / copies / update / slots / table / chain
    int error;
    int xcc;                     // XCD the level's workgroup ran on (HW_REG_XCC_ID)
    unsigned seq;                // == LevelHdr::seq once every other field is in place
};

// Buffers of the read-threading kernels (k_thread_*).
struct ThreadDev {
    const char* ref; int glen;
    int n_reads;
    const int* pos; const int* seq_off; const char* seq;
    const int* cig_off; const char* cig_op; const int* cig_len;
    const uint8_t* lut;         // byte -> symbol code 0..7
    int* count; int* minrid;    // [glen*8]
    int* tmin;                  // [glen*64] first read taking class (i-1,cp) -> (i,c)
    int* smin; int* emin;       // [glen*8] first read that starts / ends in the class
    int* off; int* cursor;      // [glen*8 + 1]
    int* pool;                  // [total M bases]
    int* err;                   // nonzero: a read runs outside the window / its bases
    int* big;                   // [1 + glen*8] classes whose pool spans too many read ids for a wavefront's bitmap: count, then the list
};

// Buffers of one progressive MSA call (k_msa).
struct MsaDev {
    const char* seqs;       // packed sequences
    const int* seq_off;     // n + 1 offsets
    int n;
    int cmax;               // capacity in columns
    char* cols[2];          // [cmax][n] column-major MSA, double buffered
    int* counts;            // [cmax][11] character-class counts of the current columns
    uint8_t* moves;         // [(cmax+1)][mv_stride] traceback moves: 0 diag, 1 insert(left), 2 delete(up)
    int mv_stride;          // DP columns per row of `moves`: 64 while every sequence has at most 63 bases
    int* edge;              // [2][cmax+1] last DP column of a 64-column chunk, for sequences longer than 63
    int* trace;             // [2*cmax] traceback script
    int* ncol_out;
    int* err_out;
};

}  // namespace sc

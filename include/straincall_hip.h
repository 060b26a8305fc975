/* C-ABI of libstraincall_hip.so -- the MI355X (gfx950) implementation of the
 * StrainCall hot path of homopolymer/RAMBL.
 *
 * The reference exposes this path only as a process (StrainCall argv -> FASTA on
 * stdout, /root/reference/StrainCall/StrainCall.cpp:972-1059, launched per region
 * by /root/reference/scripts/rambl.py:169-201).  Inside that process the path is
 * the sequence
 *
 *     PartialOrderGraph(gene_seq, reads)          StrainCall.cpp:1017
 *     pog->infer_strains(strains, read_pairs, 5000, e, tau, diff)   :1021
 *     pog->read_assign(strains, reads, read_pairs, 5000)            :1024
 *     sort by abundance, print                     :1027-1046
 *     pog->output_edge(cout)   (with -G)           :1050
 *
 * and this library is the drop-in for exactly that sequence: the caller (the
 * Python entry point rambl_amd/cli.py, or a cgo/JNI/ctypes binding written by a
 * maintainer) hands over what load_gene_seq and
 * load_mapping_reads produced (StrainCall.cpp:157-185, :480-670) as plain
 * arrays and receives strains, abundances, the -G dump and the per-level trace.
 *
 * Conventions: the caller owns every buffer; every function returns 0 on success
 * or a negative SC_ERR_* code and never throws; a context is bound to one HIP
 * device; sc_roi_submit may be called from one host thread at a time; up to
 * `stream_count` regions are in flight at once, as fibers on a few host threads
 * sized from the CPU quota of the rank (sc_host_plan), not one thread per region.
 * There is no CPU fallback: if no gfx950 device is present sc_ctx_create fails.
 */
#ifndef STRAINCALL_HIP_H
#define STRAINCALL_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SC_OK 0
#define SC_ERR_NO_DEVICE (-1)     /* no HIP device / wrong architecture */
#define SC_ERR_HIP (-2)           /* a HIP runtime call failed (see sc_last_error) */
#define SC_ERR_ARG (-3)           /* invalid argument / unknown handle */
#define SC_ERR_UNSUPPORTED (-4)   /* graph shape outside the device model (message in sc_last_error) */
#define SC_ERR_CAPACITY (-5)      /* caller buffer too small, or more than 128 live candidates */
#define SC_ERR_INTERNAL (-6)

typedef struct sc_ctx sc_ctx;

/* sc_parameter fields that reach the path (StrainCall.cpp:58-95) plus the three
 * literals of the reference's call sites (:1021,:1024 n=5000; budget 40000 at
 * NonparametricClustering.cpp:160,781; candidate cap 80 at :532). */
typedef struct sc_params {
    float error_rate;      /* -e */
    float tau;             /* -t */
    float diff_rate;       /* -d */
    int sweeps_cap;        /* 5000 */
    int draw_budget;       /* 40000 */
    int max_candidates;    /* 80 */
    int graph_only;        /* -G: build + dump the graph, no clustering */
    int want_trace;        /* keep the per-level strain/abundance trace */
    int want_timing;       /* time every sampler launch with HIP events on the region's stream (sc_stats) */
    int want_graph;        /* keep the -G text (sc_roi_graph_dump) and the threading tables of a full run too */
} sc_params;

typedef struct sc_stats {
    double graph_ms;          /* host graph build + flatten + upload */
    double cluster_ms;        /* level walk incl. kernels */
    double sampler_kernel_ms; /* sum of HIP-event times of the SAMPLE-mode launches */
    long sampler_launches;
    long sampler_read_copies; /* sum over SAMPLE launches of the read copies (draw slots) of the level */
    long level_launches;
    long draws;               /* categorical draws made on the device */
    long slow_draws;          /* of which needed the fp64 scan tier */
    long exact_draws;         /* of which resolved by the literal fp64 path */
    long sampler_strains;     /* sum over SAMPLE launches of the candidate count */
    long chain_passes;        /* window passes of the sampler chain (draws / passes = draws accepted per pass) */
    long chain_cycles;        /* shader cycles spent inside the urn chains */
    long chain_wall_ticks;    /* the same in 100 MHz ticks */
    long level_kernel_ticks;  /* 100 MHz ticks inside the level kernels (start of the kernel to its completion stamp) */
    long sampler_level_ticks; /* the part of it spent in sampler levels (one workgroup of k_level_sample each) */
    long xcd_levels[8];       /* level kernels that ran on each of the 8 XCDs (HW_REG_XCC_ID) */
    long msa_calls;
    int n_nodes, n_levels, n_unique_reads;
    long n_read_copies;
    double setup_ms;          /* the part of cluster_ms before the first level: uploads of the level-major arrays, edge support */
    double queue_ms;          /* from sc_roi_submit until a slot took the region */
    double place_ms;          /* the part of graph_ms spent waiting for one of the context's set-up places */
    double mailbox_ms;        /* set-up done, waiting for a resident workgroup to fall free (the part of cluster_ms after setup_ms) */
    double host_us[3];        /* host work between the levels, summed over the walk: [0] the level's parameters (log tables into the
                               * host-mapped block), [1] its results into the candidates' models + pruning, [2] candidate extension */
    double wake_us[2];        /* with several regions in flight, summed over the walk: [0] handing the levels to the level server, [1] from
                               * the server seeing a level's stamp until the region's fiber runs again */
    long kind_levels[17];     /* levels served by each variant of the level kernel: [0] no sampler (k_level); [1 + 2 * (NB - 1) + L]
                               * the sampler for NB = ceil(candidates / 16) register blocks, L = 1 weight rows in LDS, 0 in HBM */
} sc_stats;

/* Replaces process start-up; `device` is a HIP ordinal, `stream_count` the number of regions that walk their levels at a
 * time (with several, on resident level workers: at most 224 of them, and the context sets up to a quarter more regions up
 * meanwhile, so that a workgroup that finishes a region finds the next one ready).  Fails with SC_ERR_NO_DEVICE when there
 * is no GPU. */
int sc_ctx_create(int device, int stream_count, sc_ctx** ctx_out);
void sc_ctx_destroy(sc_ctx* ctx);
const char* sc_last_error(sc_ctx* ctx);
/* The message of one region (sc_last_error holds the context's latest, which may be another region's when several
 * fail side by side).  Valid until sc_roi_release. */
const char* sc_roi_error(sc_ctx* ctx, int handle);
/* Binds the calling thread, and the threads it starts afterwards, to the CPUs next to GPU `device` (local_cpulist of its
 * PCI device; what `numactl --cpunodebind` does for a rank).  Returns the number of CPUs, 0 when nothing was changed (topology
 * unknown, no such device, SC_NUMA_BIND=0).  sc_ctx_create places the context's own threads and host memory the same way
 * without moving its caller.  (No counterpart in the reference: rambl.py:190-194 leaves placement to the OS.) */
int sc_host_bind(int device);

/* Host threads a context with `stream_count` regions in flight starts: out[0] executor threads (they run the regions'
 * fibers), out[1] the level server (0 or 1; a context on resident level workers starts one more executor instead: its
 * executors watch the levels' completion stamps themselves), out[2] threads sc_aln_open inflates BGZF members on.  `cpus` = CPUs of the
 * host share (0: the cgroup quota / affinity mask of the process), divided by `local_world` ranks sharing it (0: the
 * environment's LOCAL_WORLD_SIZE, as torch.distributed.run sets it) -- rambl.py's Pool(cores) (scripts/rambl.py:190-194)
 * gives every region a process; here eight ranks on one host must fit its cores.  No device needed. */
int sc_host_plan(int stream_count, int local_world, double cpus, int* out);

/* Replaces `new PartialOrderGraph(gene_seq, reads)` + infer_strains + read_assign
 * + the sort (StrainCall.cpp:1017-1027).
 *   ref_bases[ref_len]            window substring of the gene (load_gene_seq)
 *   read_pos[i]                   0-based offset in the window (AlignRead<0>)
 *   cigar_text/cigar_off[n+1]     cropped CIGAR strings (AlignRead<1>)
 *   seq_text/seq_off[n+1]         cropped read bases (AlignRead<2>)
 *   read_copies[i]                copy number (AlignRead<4>)
 *   mate_idx/mate_off[n+1]        ReadPairs: mate uid (or -1) per copy of read i
 * Reads must be in the order load_mapping_reads emits them (StrainCall.cpp:610). */
int sc_roi_submit(sc_ctx* ctx, const char* ref_bases, int ref_len, const int* read_pos, const char* cigar_text,
                  const int* cigar_off, const char* seq_text, const int* seq_off, const int* read_copies,
                  const int* mate_idx, const int* mate_off, int n_reads, const sc_params* params, int* handle_out);
int sc_roi_wait(sc_ctx* ctx, int handle);

/* Strains in output order (abundance descending, libstdc++ sort order for ties,
 * StrainCall.cpp:1027).  seq_buf receives the ungapped sequences back to back
 * (Strain::plain_seq), seq_off[n_strains+1] their offsets. */
int sc_roi_result(sc_ctx* ctx, int handle, char* seq_buf, long seq_cap, int* seq_off, double* abundance,
                  int max_strains, int* n_strains);
/* Replaces pog->output_edge(cout) (PartialOrderGraph.cpp:318-337). */
int sc_roi_graph_dump(sc_ctx* ctx, int handle, char* buf, long cap, long* len_out);
/* Text of the reference's dormant debug blocks (NonparametricClustering.cpp:287-298,
 * :460-471), abundances printed with %.17g. */
int sc_roi_trace(sc_ctx* ctx, int handle, char* buf, long cap, long* len_out);
int sc_roi_stats(sc_ctx* ctx, int handle, sc_stats* out);
int sc_roi_release(sc_ctx* ctx, int handle);

/* Row a7 on its own: MultipleSequenceAlignmentSP<...>::align + MSA<>::get
 * (MultipleSequenceAlignmentSP.cpp:10-49, MultipleSequenceAlignment.hpp:59-70).
 * rows_out receives n rows of (*ncol_out) characters + NUL, back to back. */
int sc_msa_align(sc_ctx* ctx, const char* seq_text, const int* seq_off, int n, char* rows_out, long cap, int* ncol_out);

/* Row a16 on its own: number_of_reads_cover_nodes for every edge of a region
 * (PartialOrderGraph.cpp:1218-1244), in the edge order of the -G dump. */
int sc_roi_edge_support(sc_ctx* ctx, int handle, int* support, int cap, int* n_edges);

/* Row a5 on its own: the class tables of the read-threading kernels for a region
 * (the per-base M loop of PartialOrderGraph::build, PartialOrderGraph.cpp:129-177):
 * count[i*8+c] = reads whose base aligned to window position i is symbol c,
 * first_read[i*8+c] = the first such read (it creates the node), pool = the read ids
 * of every class back to back in class order, ascending inside a class.
 * symbols[8] receives the symbol of every code (0 = unused). */
int sc_roi_thread_tables(sc_ctx* ctx, int handle, int* count, int* first_read, int cls_cap, int* pool, long pool_cap,
                         char* symbols, int* n_cls, long* n_pool);

/* ---- rows a2-a4 on the host: the alignment file and a window's reads (rambl_amd/csrc/sc_ingest.cpp) ----------
 *
 * The reference shells out to samtools for every window (StrainCall.cpp:496 `view -q mq -F 1804 region`, :696
 * `mpileup -q mq -Q0 -A -r region`) and parses the text.  Here the file (SAM text, or BAM read natively: BGZF +
 * the record layout of the SAM specification, section 4) is read once and indexed by reference name. */
typedef struct sc_aln sc_aln;
typedef struct sc_reads sc_reads;

int sc_aln_open(const char* path, sc_aln** out);     /* on a parse error *out still carries the message */
/* The same, keeping the records of the named references only (n_names = 0: of none; n_names < 0: of all) while
 * sc_aln_ref_stats still answers for every reference of the file: a rank of a multi-GPU run opens the file once without
 * records to price the regions, and once with the names of its own shard (rambl_amd/stage5.py). */
int sc_aln_open_filtered(const char* path, const char* const* names, int n_names, sc_aln** out);
void sc_aln_close(sc_aln* aln);
const char* sc_aln_error(sc_aln* aln);
long sc_aln_records(sc_aln* aln);
/* alignments of one reference and the reference bases they cover: what a scheduler needs to price a region */
int sc_aln_ref_stats(sc_aln* aln, const char* gene, long* n_records, long* aligned_bases);

/* What window_adjust reads out of the pileup of gene:P-Q (StrainCall.cpp:702-736): for position P+i, whether any
 * read covers it, and whether column 5 of its pileup line would hold a '+' (insertion) / a '-' or '*' (deletion) --
 * including the mapping-quality character after '^' that the reference mistakes for one.  Arrays of Q-P+1 bytes. */
int sc_aln_pileup_flags(sc_aln* aln, const char* gene, int P, int Q, int mq, unsigned char* covered,
                        unsigned char* has_ins, unsigned char* has_del);

/* load_mapping_reads (StrainCall.cpp:480-670) for the window gene:p0-p1: view filter, depth -> keep probability,
 * length / N / insertion filters, crop to the window, mt19937(1234) thinning (drawn only for reads that pass),
 * exact-duplicate collapse in (position, CIGAR, bases) order, mate table. */
int sc_aln_load_reads(sc_aln* aln, const char* gene, int p0, int p1, int mq, int rl, int max_ins, int max_depth,
                      sc_reads** out);
/* The packed arrays sc_roi_submit takes, owned by `reads` (valid until sc_reads_free); n_input = alignments the
 * view returned, depth = the integer mean depth the keep probability came from. */
int sc_reads_get(sc_reads* reads, int* n_reads, const int** pos, const char** cigar_text, const int** cigar_off,
                 const char** seq_text, const int** seq_off, const int** copies, const int** mate_idx,
                 const int** mate_off, long* n_input, int* depth);
void sc_reads_free(sc_reads* reads);

/* ---- rambl.py stage 1 on the device (rambl_amd/csrc/sc_depth.hip) --------------------------------------------
 *
 * /root/reference/scripts/coverage_all_samples.py:21-186 pipes `samtools depth <bams>` (per-base depth per file,
 * deleted and skipped bases not counted, flags 0x704 excluded) through awk (sum over the files), sort and
 * `bedtools merge -c 4 -o mean -d 10` (one-base records [p, p+1) merged while at most 10 uncovered bases separate
 * them; mean of the merged depths).  sc_depth_scan does the same from alignment files the library has read (one
 * kernel: the per-base depth never exists in HBM, a wavefront builds it for its reference in LDS):
 * intervals come back sorted by (reference index, start), 1-based inclusive, with the sum of the depths of their
 * covered positions and their number (mean = sum / n).  Returns SC_ERR_CAPACITY (with *n_intervals set) when `cap`
 * is too small.  samtools' per-file depth cap (8000) is not applied; parity at that tool boundary is unpinned. */
typedef struct sc_depth_stats {
    long cells;            /* reference bases scanned */
    long runs;             /* aligned runs (CIGAR M = X operations) */
    double extract_ms;     /* host: CIGAR walk of the records -> runs, on the rank's host threads (sc_depth_scan only) */
    double prepare_ms;     /* host: runs bucketed by reference into page-locked memory (+ start order inside long references) */
    double upload_ms;      /* HIP events: runs (8 bytes each) and the reference tables to the device */
    double kernel_ms;      /* HIP events: k_depth_fused (difference array in LDS -> depths -> intervals) */
} sc_depth_stats;
int sc_depth_scan(int device, sc_aln* const* alns, int n_alns, const char* const* ref_names, const int* ref_len, int n_refs,
                  int max_gap, int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap, int* n_intervals,
                  sc_depth_stats* stats);
/* The same from bare runs: run i covers positions run_start[i]..run_end[i] (1-based, inclusive) of reference run_ref[i]. */
int sc_depth_scan_runs(int device, const int* ref_len, int n_refs, const int* run_ref, const int* run_start, const int* run_end,
                       long n_runs, int max_gap, int* iv_ref, int* iv_start, int* iv_end, long* iv_sum, int* iv_n, int cap,
                       int* n_intervals, sc_depth_stats* stats);

#ifdef __cplusplus
}
#endif
#endif

/* mapad_amd.h — C ABI of the MI355X-native mapAD read-mapping hot path (libmapad_amd.so).
 *
 * Drop-in boundary for `mapad map` (mpieva/mapAD v0.45.0).  The reference has no FFI today; its internal seam for this
 * path is the generic call k_mismatch_search::<SDM, MB>() made from run_inner (src/map/mapping.rs:153-271) and from
 * Worker::run (src/distributed/worker.rs:80-198), i.e. exactly the worker side of its dispatcher/worker split.  Each
 * entry point below names the reference interface it replaces.  INTEGRATION.md shows the Rust `extern "C"` block a
 * maintainer would add.
 *
 * Conventions: plain pointers and sizes, POD structs, caller-owned inputs, library-owned results released with the
 * matching *_free.  Every function returns 0 on success or a negative mapad_status_t; no exceptions cross the boundary.
 * A context is bound to one GPU and is not re-entrant; use one context per GPU / per host thread.
 * There is NO CPU fallback: without a usable gfx950 device mapad_ctx_create() fails with MAPAD_ERR_NO_DEVICE.
 */
#ifndef MAPAD_AMD_H
#define MAPAD_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mapad_status {
    MAPAD_OK = 0,
    MAPAD_ERR_INVALID = -1,      /* bad argument                                  (errors.rs: Error::InvalidInput) */
    MAPAD_ERR_IO = -2,           /* file could not be read / written              (errors.rs: Error::Io)           */
    MAPAD_ERR_INDEX_VERSION = -3,/* on-disk index version != 5                    (versioned_index.rs:31-40)       */
    MAPAD_ERR_PARSE = -4,        /* malformed index / FASTA                       (errors.rs: Error::ParseError)   */
    MAPAD_ERR_NO_DEVICE = -5,    /* no gfx950 GPU / HIP runtime failure (there is no CPU path)                     */
    MAPAD_ERR_DEVICE = -6,       /* a HIP call failed                                                              */
    MAPAD_ERR_NOMEM = -7,
    MAPAD_ERR_READ_TOO_LONG = -8,/* read longer than MAPAD_MAX_READ_LEN = i16::MAX   (record.rs:144-150)             */
    MAPAD_ERR_UNSUPPORTED = -9   /* input beyond a documented limit of this entry point; another entry point takes it */
} mapad_status_t;

#define MAPAD_MAX_READ_LEN 32767

/* ---- parameters: AlignmentParameters + the two plugin enums (src/map/mod.rs:21-31,
 *      sequence_difference_models.rs:67-72, mismatch_bounds.rs:26-30) ------------------------------------------------ */
enum { MAPAD_MODEL_SIMPLE_ADNA = 0, MAPAD_MODEL_VINDIJA_PWM = 1, MAPAD_MODEL_TEST = 2 };
enum { MAPAD_LIBRARY_SINGLE_STRANDED = 0, MAPAD_LIBRARY_DOUBLE_STRANDED = 1 };
enum { MAPAD_BOUND_DISCRETE = 0, MAPAD_BOUND_CONTINUOUS = 1, MAPAD_BOUND_TEST = 2 };

typedef struct mapad_params {
    int32_t model_kind;
    int32_t library_prep;
    float five_prime_overhang, three_prime_overhang; /* double_stranded: five_prime_overhang is the overhang */
    float ds_deamination_rate, ss_deamination_rate;
    float divergence;                                /* already divided by 3 (main.rs:452)                   */
    int32_t ignore_base_quality;
    float deam_score, mm_score, match_score;         /* TestDifferenceModel                                   */
    int32_t bound_kind;
    float poisson_threshold, base_error_rate;        /* Discrete                                              */
    float cutoff, exponent;                          /* Continuous (cutoff already negated, main.rs:467-470)  */
    float threshold, repr_mm_bound;                  /* TestBound                                             */
    float penalty_gap_open, penalty_gap_extend;
    int32_t gap_dist_ends, max_num_gaps_open;
    int32_t stack_limit_abort;
    uint32_t stack_limit, edit_tree_limit;           /* 0 = reference constants 2 000 000 / 10 000 000        */
    uint64_t chunk_size;                             /* --batch_size, default 250 000                         */
} mapad_params_t;

/* build_alignment_parameters (src/main.rs:418-499): CLI-level values -> derived parameters.
 * poisson_prob < 0 selects the Continuous bound with (as_cutoff, as_cutoff_exponent). */
int mapad_params_from_cli(mapad_params_t* out, int library_prep, float five_prime_overhang, float three_prime_overhang,
                          float ds_deamination_rate, float ss_deamination_rate, float divergence, float poisson_prob,
                          float as_cutoff, float as_cutoff_exponent, float indel_rate, float gap_extension_penalty,
                          int gap_dist_ends, int max_num_gaps_open, int ignore_base_quality, int no_search_limit_recovery,
                          uint64_t chunk_size);

/* trait SequenceDifferenceModel (sequence_difference_models.rs:14-62) */
float mapad_sdm_get(const mapad_params_t* p, uint64_t i, uint64_t read_length, uint8_t from, uint8_t to, uint8_t base_quality);
float mapad_sdm_representative_mismatch_penalty(const mapad_params_t* p);
float mapad_sdm_min_penalty(const mapad_params_t* p, uint64_t i, uint64_t read_length, uint8_t to, uint8_t base_quality, int only_mismatches);
int32_t mapad_sdm_alignment_start(const mapad_params_t* p, uint64_t pattern_length);
/* trait MismatchBound (mismatch_bounds.rs:10-20) */
int mapad_mb_reject(const mapad_params_t* p, float value, uint64_t read_length);
int mapad_mb_reject_iterative(const mapad_params_t* p, float value, float reference);
float mapad_mb_remaining_frac_of_repr_mm(const mapad_params_t* p, float value, uint64_t read_length);

/* ---- index: RtFmdIndex + SampledSuffixArray + FastaIdPositions + OriginalSymbols (src/index/mod.rs) ------------------ */
typedef struct mapad_index mapad_index_t;

/* `mapad index` (src/index/indexing.rs:29-212) on an in-memory FASTA-like input: n_contigs sequences (any case, IUPAC).
 * Ambiguous bases in runs shorter than 20 are replaced by a random compatible base drawn like the reference draws it — rand 0.9
 * StdRng::seed_from_u64(seed) + slice.choose(): ChaCha12 keyed through PCG32, index by Canon's method, restated in host_index.hpp and
 * checked against the one draw the reference's integration test pins —, longer runs become 'X'; originals are kept. */
int mapad_index_build(const char* const* names, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n_contigs,
                      uint64_t seed, mapad_index_t** out);
/* The same products (src/index/indexing.rs:163-195: suffix array -> BWT, SA sample, Less, rank structure) with the suffix sorting done on
 * the MI355X `device_id` (prefix doubling over radix sorts, mapad_amd/csrc/index_gpu.hip); byte-identical to mapad_index_build for every
 * input.  Texts up to 2^40 symbols; MAPAD_ERR_NO_DEVICE without a GPU (mapad_index_build is the host path of this offline step). */
int mapad_index_build_gpu(const char* const* names, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n_contigs,
                          uint64_t seed, int device_id, mapad_index_t** out);
/* load_index_from_path + load_suffix_array/.tpi/.tos (src/index/mod.rs:212-239): reads the 7 files <prefix>.{tbw,tle,toc,trt,tsa,tpi,tos} */
int mapad_index_open(const char* prefix, mapad_index_t** out);
/* writers of indexing.rs:110-208 (snappy frame stream of bincode 1.3, version byte 5) */
int mapad_index_save(const mapad_index_t* idx, const char* prefix);
void mapad_index_free(mapad_index_t* idx);
uint64_t mapad_index_text_len(const mapad_index_t* idx);           /* n = 2|G| + 2 */
int mapad_index_copy_bwt(const mapad_index_t* idx, uint8_t* out);  /* n rank bytes ($=0 A=1 C=2 G=3 T=4 X=5) */
uint32_t mapad_index_n_contigs(const mapad_index_t* idx);
int mapad_index_contig(const mapad_index_t* idx, uint32_t i, const char** name, uint64_t* start, uint64_t* end);
/* SampledSuffixArray pieces, for cross-checks: sizes then copies */
uint64_t mapad_index_sa_sample_len(const mapad_index_t* idx);
uint64_t mapad_index_sa_extra_len(const mapad_index_t* idx);
int mapad_index_copy_sa(const mapad_index_t* idx, uint64_t* sample, uint64_t* extra_rows, uint64_t* extra_vals);

/* the rank structure as the GPU sees it: 64-byte blocks (8 x u64 per 96 BWT rows; layout in mapad_amd/csrc/fmd_device.hpp),
 * less[8] (reference ranks) and the two sentinel rows */
int mapad_index_device_view(const mapad_index_t* idx, const uint64_t** blocks, uint64_t* n_blocks, uint64_t less[8], uint64_t sentinel[2]);
/* SampledSuffixArray::get (src/index/mod.rs:160-187) */
int mapad_index_sa_get(const mapad_index_t* idx, uint64_t row, uint64_t* out);
/* the same for n rows on one host thread (UINT64_MAX for rows past the text): the CPU side of mapad_sa_locate() */
int mapad_index_sa_get_batch(const mapad_index_t* idx, const uint64_t* rows, uint64_t n, uint64_t* out);

/* ---- mapping context: one GPU, index resident in HBM ---------------------------------------------------------------- */
typedef struct mapad_ctx mapad_ctx_t;

int mapad_ctx_create(const mapad_index_t* idx, const mapad_params_t* params, int device_id, mapad_ctx_t** out);
void mapad_ctx_destroy(mapad_ctx_t* ctx);
/* run every launch on this HIP stream (a hipStream_t; NULL = the default stream) */
int mapad_ctx_set_stream(mapad_ctx_t* ctx, void* hip_stream);

/* Leave the last `n_cus` compute units of the device free of this context's launches (its batch slots' streams get a CU mask; pipeline depth >= 2 only: depth 1
 * runs on the caller's stream).  A search launch fills every CU it may use; RCCL's transfer kernels (248-256 VGPRs, 37.6 KB of LDS per block) only fit on CUs it
 * does not use.  For one-process-per-GPU runs that gather results over xGMI beside the next search; before the first batch.  MAPAD_RESERVED_CUS sets the default. */
int mapad_ctx_set_reserved_cus(mapad_ctx_t* ctx, int n_cus);
/* The heavy tail.  The reference absorbs the reads that run into STACK_LIMIT / EDIT_TREE_LIMIT (src/map/mapping.rs:52-54,1358-1380) on its rayon threads; here a
 * read is handed — by the kernel, while it runs, through host-coherent page-locked memory — to the library's host threads when (a) it has made `pops` pops on the
 * GPU while the host threads keep up (default 2^20; MAPAD_TAIL_POPS; 0 = the host tail is off; MAPAD_TAIL_BACKLOG_BUDGET) — or MAPAD_TAIL_POPS_IDLE pops while a host
 * thread is idle —, (b) it needs a grown arena of a class the GPU has few of and every one is taken, while the host
 * threads have little waiting (MAPAD_TAIL_MIN_CLASS, MAPAD_TAIL_BACKLOG), or (c) no growable arena can hold it (the reads the full-limit stage would restart).
 * The host threads map it from scratch with the kernel's own search step compiled for the host (csrc/host_tail.hpp; MAPAD_TAIL_THREADS threads, default this
 * process's share of the CPUs, divided by LOCAL_WORLD_SIZE when several ranks share a node).  Their results join the batch before its order-preserving collect:
 * nothing a caller sees depends on where a read was finished. */
int mapad_ctx_set_tail_pops(mapad_ctx_t* ctx, uint32_t pops);
/* "This process is one rank of `local_world` on its node": the host tail's worker threads that take reads from now on are this rank's part of the node's CPU share
 * (what LOCAL_WORLD_SIZE / MAPAD_LOCAL_WORLD_SIZE set at start; 0 = back to those).  Process-wide like the worker pool itself; returns the number of workers.
 * The reference's workers each own a whole machine (src/distributed/worker.rs:80-198); eight ranks on one node share its CPUs, and bench.py measures C5 as such a
 * rank on a one-GPU box with this call (`secondary.c5_rank_of_8`). */
uint32_t mapad_tail_set_local_world(uint32_t local_world);
/* the batch selected by mapad_ctx_select_batch, after its collect / fetch: {reads finished on the host, pops the GPU had spent on them, pops on the host,
 * host wall-clock microseconds from the first hand-over to the last result, host threads, pop budget, and the host reads' E_search, N_push, N_node sums
 * (SURVEY 8d events the kernel did not execute), microseconds the host threads spent inside these reads, summed over the threads,
 * [10] hand-overs the host saw while the launch was still running, [11] reads handed over for reason (b), [12] for reason (c), [13] smallest class of (b) | reads handed over below `pops` because a host thread was idle << 32,
 * [14] reads a host thread CONTINUED from the GPU's state (heap and nodes copied out of the read's grown arena) instead of mapping them from scratch, [15] reads
 * the kernel handed over with their state.  For continued reads the pop / event figures above count the host's share only.} */
int mapad_last_tail_info(mapad_ctx_t* ctx, uint64_t out[16]);
/* whether mapad_fetch_result()/mapad_map_batch() also copy the D arrays back (default on; bench.py turns it off) */
int mapad_ctx_set_fetch_d_arrays(mapad_ctx_t* ctx, int on);
/* Score tables are built lazily per read length.  mapad_map_batch() does this itself; before mapad_map_batch_device()
 * (where the host never sees the reads) announce the lengths that will occur. */
int mapad_ctx_prepare_lengths(mapad_ctx_t* ctx, const uint32_t* lens, uint32_t n);

/* HitInterval (src/map/mod.rs:34-61) with the edit track flattened; 40 bytes. */
typedef struct mapad_hit {
    uint64_t lower, lower_rev, size; /* RtBiInterval */
    float alignment_score;
    uint32_t n_ops;                  /* EditOperationsTrack length */
    uint32_t ops_offset;             /* first op in the batch's ops array */
    uint32_t reserved;
} mapad_hit_t;
/* packed EditOperation (src/map/record.rs:225-237): kind<<24 | reference base (ASCII)<<16 | read position;
 * kind 0 Insertion, 1 Deletion, 2 Match, 3 Mismatch */

typedef struct mapad_read_counters { /* algorithm events per read; identical on the CPU oracle */
    uint32_t e_search, e_darray, n_push, n_pop, n_node, n_hits;
} mapad_read_counters_t;

/* Result of one batch == Vec<BinaryHeap<HitInterval>> of run_inner's par_iter (mapping.rs:153-271), order-preserving.
 * hits of read i are hits[hit_begin[i] .. hit_begin[i+1]) in BinaryHeap array order. */
typedef struct mapad_batch_result {
    uint64_t n_reads;
    uint64_t n_hits, n_ops;
    const uint64_t* hit_begin;             /* n_reads + 1 */
    const mapad_hit_t* hits;               /* n_hits */
    const uint32_t* ops;                   /* n_ops */
    const uint32_t* status;                /* per read: 0 ok, 2 stopped by --no_search_limit_recovery */
    const mapad_read_counters_t* counters; /* per read */
    const float* d_arrays;                 /* concatenated BiDArray::d_composite, same offsets as the reads (debug/parity) */
    uint64_t n_second_pass;                /* arena migrations: times a read slot outgrew its arena and moved into a larger size class */
    uint64_t n_third_pass;                 /* reads re-run by the pass that holds the reference's full STACK_LIMIT / EDIT_TREE_LIMIT */
} mapad_batch_result_t;

/* k_mismatch_search over a chunk of reads (host buffers): seqs/quals concatenated, read i = [offsets[i], offsets[i+1]).
 * quals are raw Phred values (no +33). */
int mapad_map_batch(mapad_ctx_t* ctx, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads,
                    mapad_batch_result_t** out);
void mapad_batch_result_free(mapad_batch_result_t* r);

/* Asynchronous variant for a chunk loop that keeps the GPU busy (run_inner's loop, src/map/mapping.rs:151-294, with the next chunk
 * submitted before the previous one is collected): stages the reads (the host buffers may be reused on return), launches on the next of the
 * context's batch slots (mapad_ctx_set_pipeline_depth) and returns.  Collect with mapad_ctx_select_batch + mapad_fetch_result; a fetch that
 * returns MAPAD_ERR_NOMEM (hit pools too small for that chunk) is repaired by running the chunk through mapad_map_batch. */
int mapad_submit_batch(mapad_ctx_t* ctx, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets, uint64_t n_reads);
/* page-locked host memory: reads placed here reach the GPU by DMA at link speed (pageable memory goes through the driver's staging copies).
 * Freed blocks are kept for reuse (power-of-two sizes, up to 8 GB): pinning costs milliseconds and hipHostFree waits for the device, which a
 * chunk loop that allocates per chunk cannot afford. */
void* mapad_host_alloc(size_t bytes);
void mapad_host_free(void* p);
/* CPUs this process may really use: the visible CPUs capped by the cgroup's CPU-time quota (cpu.max); MAPAD_HOST_CPUS overrides.  What the library's own host
 * thread pools (host tail, record strings, index preparation) size themselves by, and what a caller's pools beside them should (csrc/host_cpus.hpp). */
unsigned mapad_host_cpus(void);

/* Device-resident variant used by bench.py and the multi-GPU driver: inputs already in HBM (device pointers), results stay
 * in the context's device buffers until fetched.  Asynchronous on the context's stream. */
int mapad_map_batch_device(mapad_ctx_t* ctx, const void* d_seqs, const void* d_quals, const void* d_offsets, uint64_t n_reads,
                           uint32_t max_read_len);
/* Batches in flight.  With depth d > 1 the context keeps d sets of batch buffers and streams of its own: mapad_map_batch_device rotates
 * through them and returns as soon as the batch is enqueued (it waits only for the batch that used the same set d calls ago), so the serial
 * tail of batch k — its few heaviest reads, each a chain of dependent memory accesses — runs beside the bulk of batch k + 1, the way the
 * reference's worker threads start the next chunk while stragglers finish (src/map/mapping.rs:151-156).  Inputs are ordered behind the
 * context's stream (mapad_ctx_set_stream) at submission.  Default 1 (MAPAD_PIPELINE_DEPTH overrides): everything runs on the context's stream.
 * Changing the depth waits for all batches and drops their results. */
int mapad_ctx_set_pipeline_depth(mapad_ctx_t* ctx, int depth);
/* Allocates, for every batch slot, the device buffers of batches of up to n_reads reads / total_bases bases / reads up to max_read_len long
 * (host_inputs != 0: also the staging buffers of mapad_map_batch / mapad_submit_batch).  Optional: buffers grow on demand, but an allocation in
 * the middle of a pipeline waits for the kernels that are running. */
int mapad_ctx_reserve(mapad_ctx_t* ctx, uint64_t n_reads, uint64_t total_bases, uint32_t max_read_len, int host_inputs);
/* result accessors (fetch, compact, counters, kernel times, device pointers) read the batch submitted `age` calls before the most recent
 * one (0 = most recent; reset to 0 by every submission) */
int mapad_ctx_select_batch(mapad_ctx_t* ctx, int age);
/* HIP-event time stamps of every launch since the last call (waits for all batches): 4 floats per launch = ms from the start of the first
 * launch to {start of the D-array kernel, end of D arrays + ordering, end of the growable search stages, end of the full-limit stage}.
 * Returns the number of launches in *n (at most `cap` are written). */
int mapad_kernel_history(mapad_ctx_t* ctx, float* out, uint32_t cap, uint32_t* n);
/* after synchronising the stream: copy the last device batch's results to the host */
int mapad_fetch_result(mapad_ctx_t* ctx, mapad_batch_result_t** out);
/* Order-preserving collect (src/map/mapping.rs:288) on the device: lays the last batch's hits and edit operations out in read order and
 * returns device pointers to hit_begin (u64[n_reads + 1], exclusive prefix sums), the hit records (mapad_hit_t[n_hits], ops_offset into
 * the ops array) and the ops (u32[n_ops]) — the arrays mapad_fetch_result copies out and the multi-GPU gather sends to rank 0.
 * Valid until the next batch.  Launches on the context's stream; returns without waiting for it. */
int mapad_compact_result_device(mapad_ctx_t* ctx, void** d_hit_begin, void** d_hits, void** d_ops, uint64_t* n_hits, uint64_t* n_ops);
/* device pointers of the last batch's raw result buffers (for the RCCL gather): per-read hit counts (u32[n_reads]),
 * per-read first-hit index (u32[n_reads]), hit pool (mapad_hit_t[]), ops pool (u32[]), 2 x u64 cursors {n_hits, n_ops} */
int mapad_device_result_ptrs(mapad_ctx_t* ctx, void** d_hit_count, void** d_hit_first, void** d_hits, void** d_ops, void** d_cursors);
/* sums of the per-read counters of the last batch (after a fetch or a stream sync): {e_search, e_darray, n_push, n_pop, n_node, n_hits} */
int mapad_last_batch_counters(mapad_ctx_t* ctx, uint64_t out[6]);
/* HIP-event durations (ms) of the last batch's launches on the context's stream: {darray_kernel + the two ordering kernels,
 * search_kernel over every read + its (normally empty) retry launches, full-limit search_kernel}.  Synchronises on the last event. */
int mapad_last_kernel_ms(mapad_ctx_t* ctx, float out[3]);
/* launch geometry of the last batch, for bench.py's report: {darray grid, block, LDS bytes, search grid, block, full-limit grid,
 * base arena node capacity, base arena KiB per read slot} */
int mapad_last_launch_info(mapad_ctx_t* ctx, uint32_t out[8]);

/* ---- post-search: intervals_to_bam minus BAM byte encoding (mapping.rs:402-718, record.rs:269-449) ------------------- */
typedef struct mapad_record {
    uint16_t flags;
    uint8_t mapq;
    uint8_t mapped, reverse;
    int32_t tid;
    int64_t pos;          /* 0-based leftmost position, -1 if unmapped */
    float as_score, xs_score;
    int32_t nm, x0, x1;
    uint8_t has_xs;
    char xt;
    uint32_t cigar_off, cigar_len, md_off, md_len, xa_off, xa_len; /* into the text blob */
} mapad_record_t;
typedef struct mapad_records {
    uint64_t n;
    const mapad_record_t* recs;
    const char* text;
    uint64_t text_len;
} mapad_records_t;
/* in_flags: input BAM flags per read (NULL = 0, i.e. FASTQ input, record.rs:207-213); seed: stands in for rand::rng()
 * (mapping.rs:273,605; only matters for hits with >= 3 SA rows) */
int mapad_hits_to_records(const mapad_index_t* idx, const mapad_params_t* params, const mapad_batch_result_t* res, const uint8_t* seqs,
                          const uint8_t* quals, const uint64_t* offsets, const uint16_t* in_flags, uint64_t seed, mapad_records_t** out);
void mapad_records_free(mapad_records_t* r);
/* The seed to pass for a slice of a chunk that starts at read `first_read_index`, such that every read draws the same stand-in for
 * rng.next_u32() (src/map/mapping.rs:605-607) as it would if the whole chunk were converted by one call with `seed`. */
uint64_t mapad_records_seed_at(uint64_t seed, uint64_t first_read_index);

/* ---- SA locate on the device ---------------------------------------------------------------------------
 * SampledSuffixArray::get (src/index/mod.rs:160-187) for a batch of BWT rows: LF walk to the next sampled row (or '$' row) in a
 * kernel, one quad per row, against the index blocks already resident in HBM.  rows / out are host arrays; out[i] = suffix-array
 * value, UINT64_MAX for a row >= text length.  Results are identical to mapad_index_sa_get(). */
int mapad_sa_locate(mapad_ctx_t* ctx, const uint64_t* rows, uint64_t n, uint64_t* out);
/* kernel time (HIP events on the context's stream), rows and LF steps of the last locate call */
int mapad_last_locate_info(mapad_ctx_t* ctx, float* kernel_ms, uint64_t* rows, uint64_t* lf_steps);
/* mapad_hits_to_records() with the suffix-array lookups of all hit intervals of <= 8 rows done by mapad_sa_locate's kernel first
 * (interval2coordinate, mapping.rs:590-649, is the second random-access loop of the reference); same records, uses the context's
 * index and parameters.  `res` may be a result of this context that has not been freed (its hits are then read where the collect left them
 * on the device), a result of another context, or a struct the caller has filled in itself (hit_begin, hits, ops: they are uploaded first). */
int mapad_hits_to_records_gpu(mapad_ctx_t* ctx, const mapad_batch_result_t* res, const uint8_t* seqs, const uint8_t* quals, const uint64_t* offsets,
                              const uint16_t* in_flags, uint64_t seed, mapad_records_t** out);

/* The same in two calls, for a chunk loop whose GPU thread should not spend its time on strings: mapad_hits_to_coords_gpu() is the device half
 * (which hit is reported, its coordinate, the XA candidates, X0 / X1: one kernel over the hits that are still resident on the device, one small copy back);
 * mapad_coords_to_records() is the host half (flags, CIGAR / MD / XA text, XS / XT, mapping quality) — it needs no context and may run on any thread
 * while the GPU thread goes on submitting and fetching.  Together they return exactly what mapad_hits_to_records_gpu() returns. */
/* The same products left on the device, for the batch selected by mapad_ctx_select_batch (its collect is done first): per read one 88-byte record
 * {i64 pos; i32 tid; u32 mapped, reverse; f32 as, xs; i32 nm, x0, x1; u32 has_xs, xt, text_off, cigar_len, md_len, xa_len; f32 best_size; u32 read_len, mq_off, mq_n, error}
 * (csrc/text_core.hpp: DevRecord), the text pool ([CIGAR][MD][XA] of a read behind its text_off) and the pool of (score, size) f32 pairs the mapping
 * quality is computed from (mq_n pairs behind mq_off) — what a rank sends to rank 0 in the multi-GPU gather (<= 128 bytes per read; the
 * reference's ResultSheet return path, src/distributed/dispatcher.rs:223-247).  Valid until the batch slot is launched again.  Flags and MAPQ are
 * finished on the host (mapad_coords_to_records' arithmetic: glibc exp2f / log10f). */
int mapad_records_device(mapad_ctx_t* ctx, uint64_t seed, void** d_records, void** d_text, void** d_pairs, uint64_t* text_bytes, uint64_t* n_pairs);
typedef struct mapad_coords mapad_coords_t;
int mapad_hits_to_coords_gpu(mapad_ctx_t* ctx, const mapad_batch_result_t* res, uint64_t seed, mapad_coords_t** out);
int mapad_coords_to_records(const mapad_index_t* idx, const mapad_params_t* params, const mapad_batch_result_t* res, const uint16_t* in_flags,
                            const mapad_coords_t* coords, mapad_records_t** out);
void mapad_coords_free(mapad_coords_t* c);

const char* mapad_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MAPAD_AMD_H */

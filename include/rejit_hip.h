/* include/rejit_hip.h -- the C ABI of librejit_hip.so: the drop-in boundary between a
 * host program and the MI355X kernels.
 *
 * In the reference the boundary is a set of raw function pointers into JIT-generated x86
 * code, one per match type (src/regexp.h:533-536):
 *
 *     bool     (*MatchFullFunc)    (const char* text, size_t size);
 *     bool     (*MatchAnywhereFunc)(const char* text, size_t size);
 *     bool     (*MatchFirstFunc)   (const char* text, size_t size, Match*);
 *     unsigned (*MatchAllFunc)     (const char* text, size_t size, std::vector<Match>*);
 *
 * produced by Regej::Compile (src/rejit.cc:229-267) and called by Regej::Match*
 * (src/rejit.cc:150-208).  The entry points below replace exactly those: rj_compile takes
 * the place of Parser::Parse + Codegen::Compile, rj_match_* of the four generated
 * functions.  Offsets are returned instead of pointers (the C++ wrapper in
 * include/rejit.h converts).  Plain C types only; no HIP or torch types in signatures
 * (streams are passed as void*).
 *
 * Errors never throw and never abort: every call returns a negative rj_status and
 * rj_last_error() holds a message (thread-local).
 */
#ifndef REJIT_HIP_H_
#define REJIT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RJ_OK = 0,
  RJ_PARSER_ERROR = -1,  /* == rejit::ParserError, include/rejit.h:98-102 of the reference */
  RJ_TOO_LARGE = -2,     /* pattern expands beyond the device automaton limits */
  RJ_DEVICE_ERROR = -3,  /* HIP failure (no GPU, out of memory, launch error, ...) */
  RJ_BAD_ARGUMENT = -4
} rj_status;

typedef struct rj_program rj_program; /* a compiled pattern: immutable, shareable across threads */
typedef struct rj_scan rj_scan;       /* per-caller device scratch + results; NOT thread-safe   */

typedef struct {
  int32_t n_positions;     /* automaton positions (one per pattern byte)            */
  int32_t n_words;         /* 32-bit words of automaton state                       */
  int32_t has_assertions;  /* pattern contains ^ or $                               */
  int32_t scan_mode;       /* 0 dense (every position), 1 fast-forward windows       */
  int32_t n_windows;       /* number of 4-byte window constants                     */
  uint32_t window_offset;  /* byte offset of the windows inside a match             */
  uint32_t window_len;     /* bytes compared per window                             */
  uint64_t min_len;        /* shortest match                                         */
  uint64_t max_len;        /* longest match, UINT64_MAX when unbounded               */
  int32_t ring_artefact_risk; /* 1: the reference's answer on this pattern can depend on its ring artefact (DESIGN.md
                              * section 6).  A RANGE of such a pattern (rj_scan_run with own_begin / own_end) owns the whole
                              * segments between synchronisation points, [first point >= own_begin, first point >= own_end),
                              * and takes the text buffer to be the WHOLE text: give every rank the text up to its end
                              * (rejit_amd/sharding.py: visible_range(..., whole_text=True)), not a fixed halo. */
  int32_t reserved;
} rj_info;

typedef struct {
  uint64_t n_hits;         /* candidate starts produced by the scan kernel           */
  uint64_t n_candidates;   /* verified (begin,end) candidates before selection       */
  uint64_t n_matches;      /* final left-most-longest, non-overlapping matches       */
  float scan_ms;           /* start-to-end time of the scan kernel(s), from their dispatch timestamps */
  float total_ms;          /* host clock: first launch to final synchronise of the run */
  int32_t retries;         /* runs repeated because a device list had to grow        */
  int32_t large_path;      /* 1 when the rocPRIM sort path was taken                 */
  int32_t exact_path;      /* 1 when the exact replay replaced the result, 2 when it took a long segment in parts */
  int32_t linear_path;    /* 1 when a candidate outlived the parallel verifier's walk and the linear-time carry scan ran */
  int32_t stream_path;    /* 1 when the bit-stream dense kernel produced the result (one pass, pairs written once) */
  int32_t slow_starts;    /* that kernel: starts that outlived its register steps and took the scalar walk (saturating) */
  int32_t count_path;     /* 1 when the one-kernel count (plane_count.hip) answered: a count and bounds, no span list (round 6) */
  int32_t run_path;       /* 1 when the run kernels (run_scan.hip: one long-lived thread in one loop position -- `[acgt]+`, `a.*b`)
                           * produced the result: two passes over the text, linear whatever the runs' length (round 6); 2: the pair
                           * kernels of the same file (`"[^"]*"`: the same class at both ends, none of it inside the loop) */
} rj_stats;

/* ---- compile (replaces Regej::Regej + Regej::Compile, src/rejit.cc:127-137,229-267) */
int rj_compile(const char* regexp, rj_program** out);
void rj_program_free(rj_program* prog);
int rj_program_info(const rj_program* prog, rj_info* info);
const char* rj_last_error(void);

/* ---- host text: one call = H2D copy + device pipeline + D2H of the results.
 * These four replace the four JIT function pointers above. */
/* 1 = the whole text matches, 0 = it does not, <0 = rj_status */
int rj_match_full(const rj_program* prog, const char* text, size_t n);
/* 1 / 0 / <0 */
int rj_match_anywhere(const rj_program* prog, const char* text, size_t n);
/* 1 (begin/end offsets written) / 0 / <0; left-most longest match */
int rj_match_first(const rj_program* prog, const char* text, size_t n, uint64_t* begin, uint64_t* end);
/* number of matches (>= 0) or rj_status.  *spans receives a malloc'ed array of
 * 2*count offsets (begin0,end0,begin1,...) to be released with rj_free_spans; pass NULL
 * to only count. */
int64_t rj_match_all(const rj_program* prog, const char* text, size_t n, uint64_t** spans);
void rj_free_spans(uint64_t* spans);
/* rj_stats of the calling thread's last rj_match_all / rj_match_first / rj_replace_all ... of `prog` (which kernels answered:
 * count_path, stream_path, linear_path, ...), as rj_scan_stats_sized.  RJ_BAD_ARGUMENT when this thread has made no such
 * call (calls of several threads combined into one batch are accounted to the thread that led the batch). */
int rj_host_stats(const rj_program* prog, void* stats, size_t struct_size);

/* A batch of independent texts in ONE device pass -- replaces the per-file loop of a grep-like
 * caller (`for each file: re->MatchAll(file, size, &matches)`, sample/jrep.cc:261-313), whose
 * per-call latency (~50 us + a PCIe copy each) a GPU cannot afford for many small files.  Every
 * text is matched on its own: no match crosses a text boundary, `^` / `$` see each text's own
 * begin and end.  counts[i] receives the number of matches of text i; *spans (may be NULL to only
 * count) a malloc'ed array of 2 * total offsets, text after text, RELATIVE to their own text;
 * release with rj_free_spans.  Returns the total number of matches or rj_status. */
int64_t rj_match_all_batch(const rj_program* prog, const char* const* texts, const size_t* sizes, size_t n_texts,
                           uint64_t* counts, uint64_t** spans);

/* The same for a batch its owner lays out ITSELF, so that no text is copied twice: text i is
 * packed[offsets[i] .. offsets[i] + sizes[i]), offsets ascending, and EVERY byte of packed[0 .. total_bytes) that
 * belongs to no text -- at least one behind every text, the last one included -- holds rj_batch_separator(prog).
 * `packed` is uploaded as it is: from memory of rj_host_alloc (pinned) at the PCIe rate, from ordinary memory
 * through the runtime's staging copy.  A native grep reads its files straight into such a buffer
 * (samples/jrep_gpu.cc); the reference's loop has no counterpart (one mmap + MatchAll per file,
 * sample/jrep.cc:261-313).  counts / spans / return value as for rj_match_all_batch.  A pattern for which
 * rj_batch_separator is -1 (every byte value can be consumed by some position of an automaton with assertions, or
 * the pattern is at risk of the ring artefact) is matched text by text. */
int rj_batch_separator(const rj_program* prog);
int64_t rj_match_all_packed(const rj_program* prog, const char* packed, const uint64_t* offsets, const size_t* sizes, size_t n_texts,
                            uint64_t total_bytes, uint64_t* counts, uint64_t** spans);
void* rj_host_alloc(size_t bytes);   /* pinned host memory (NULL when there is none to be had) */
void rj_host_free(void* p);

/* ReplaceAll (replaces MatchAll + rejit::Replace, src/rejit.cc:97-112,220-226): every match is
 * replaced by with[0..with_len).  Returns the number of matches (>= 0) or rj_status; *out receives
 * a malloc'ed copy of the new text (NUL-terminated for convenience, *out_len excludes the NUL),
 * to be released with rj_free_text.  Only the new text crosses PCIe, not the match list. */
int64_t rj_replace_all(const rj_program* prog, const char* text, size_t n, const char* with, size_t with_len, char** out,
                       size_t* out_len);
void rj_free_text(char* text);
/* The same in two steps, for a caller that owns the buffer the new text goes to (Regej::ReplaceAll rewrites its std::string:
 * include/rejit.h; reference src/rejit.cc:220-226): _begin uploads, matches and replaces on the device and returns the number
 * of matches (or rj_status) with the new text's length in *out_len; _fetch -- same thread, same program, once -- copies the
 * new text into dst[0 .. out_len) (dst_cap >= out_len; no NUL is written).  dst may be the memory `text` pointed to: the
 * device holds its own copy by then.  Measured on the GPU box (1 GB): malloc + download + free of the one-call form cost
 * 50 + 80 ms, the download into pages the caller already owns 18 ms. */
int64_t rj_replace_all_begin(const rj_program* prog, const char* text, size_t n, const char* with, size_t with_len, size_t* out_len);
int rj_replace_all_fetch(const rj_program* prog, char* dst, size_t dst_cap);

/* ---- device-resident text (what bench.py and the multi-GPU driver use; no copies).
 * d_text must be 16-byte aligned device memory with bytes [0, n) readable.  Matches whose
 * BEGIN lies in [own_begin, own_end) are reported (own_end may be n + 1 to include the
 * empty match at the end of the text); the automaton may read up to n, so a shard passes
 * its halo inside [0, n).  carry_cur / carry_prev_end / have_prev describe the last match
 * selected before own_begin (0,0,0 for the first shard). */
int rj_scan_create(const rj_program* prog, rj_scan** out);
void rj_scan_destroy(rj_scan* scan);
/* rj_stats.scan_ms (and rj_multi_scan_ms) is the scan kernel's own duration, from a start and an end event stamped by the
 * launch.  The end event is free; the START event costs 6-9 us per call (measured, round 4: 32-37 -> 26-28 us for texts of
 * 32 KB .. 16 MiB; ~6.5 us between two kernels of the regexdna step).  So it is OFF unless asked for -- per object, or for
 * every object created from now on (rj_set_default_timing returns the previous default; environment RJ_KERNEL_TIMING=1 sets
 * the initial one) -- and scan_ms reads 0 without it.  The Python binding (rejit_amd/api.py: bench.py, the tests and the
 * tools read scan_ms) switches the default on when it loads the library. */
int rj_scan_set_timing(rj_scan* scan, int on);
int rj_set_default_timing(int on);
int64_t rj_scan_run(rj_scan* scan, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end,
                    uint64_t carry_cur, uint64_t carry_prev_end, int have_prev, void* hip_stream);
/* The same whole-text run in two halves, so that a caller with several patterns (or texts) in
 * flight does not serialise on every call's latency-bound tail: rj_scan_start enqueues the scan on
 * hip_stream and the verify / gather kernels behind it on the scan's own stream, rj_scan_finish
 * waits and returns the count (results as after rj_scan_run).  One start per scan at a time; scans
 * started back to back on one stream run their scan kernels one after the other while the tails of
 * earlier ones overlap them. */
int rj_scan_start(rj_scan* scan, const void* d_text, uint64_t n, void* hip_stream);
int64_t rj_scan_finish(rj_scan* scan);
/* MatchAllCount over device text (Regej::MatchAllCount, reference src/rejit.cc:203-208, is MatchAll with a NULL vector: the
 * caller wants the NUMBER of matches).  A pattern with the shape the one-kernel count takes (plane_count.hip; every regexdna
 * pattern, sample/regexdna.cc:51-61) is answered by that kernel -- the text read once, nothing written but the count;
 * rj_stats.count_path reads 1, rj_scan_device_spans NULL, and rj_scan_copy_spans / rj_scan_replace refuse with
 * RJ_BAD_ARGUMENT (there is no list).  Any other pattern, and a text on which the kernel voids its run, takes rj_scan_run's
 * pipeline (whole text, no carry).  rj_match_all(prog, text, n, NULL) does the same for host texts.  Returns the count or
 * rj_status. */
int64_t rj_scan_count(rj_scan* scan, const void* d_text, uint64_t n, void* hip_stream);
/* results of the last run: device pointer to 2*count uint64 offsets, or a copy (host_spans may be host or
 * device memory; returns the number of matches, copies at most cap of them) */
const uint64_t* rj_scan_device_spans(const rj_scan* scan);
int64_t rj_scan_copy_spans(const rj_scan* scan, uint64_t* host_spans, uint64_t cap);
/* Replace over device text: the matches of the LAST rj_scan_run on this scan (which must have
 * covered d_text[0..n) ) are replaced by `with` (host bytes); the result is written to d_out
 * (device, out_cap bytes).  Returns the new length or rj_status. */
int64_t rj_scan_replace(rj_scan* scan, const void* d_text, uint64_t n, const char* with, uint64_t with_len, void* d_out,
                        uint64_t out_cap, void* hip_stream);
int rj_scan_stats(const rj_scan* scan, rj_stats* stats);
/* rj_stats has grown (stream_path, slow_starts: round 4) and may grow again: rj_scan_stats writes sizeof(rj_stats) of the
 * header the LIBRARY was built from.  A caller that may meet a newer library passes the size of ITS struct: at most
 * struct_size bytes are written; returns the library's sizeof(rj_stats) (> struct_size: fields the caller does not know
 * were left out) or rj_status. */
int rj_scan_stats_sized(const rj_scan* scan, void* stats, size_t struct_size);
/* 1 / 0 / <0: kMatchFull over device text */
int rj_scan_match_full(rj_scan* scan, const void* d_text, uint64_t n, void* hip_stream);

/* ---- several patterns over the same device-resident text (regexdna: nine MatchAllCount calls on
 * one text, sample/regexdna.cc:56-70).  When every pattern has a nibble-form window set (DESIGN.md
 * section 4) the text is read ONCE for all of them; otherwise the patterns run one after another.
 * Results are those of rj_scan_run per pattern: rj_multi_scan(m, i) is an ordinary rj_scan holding
 * pattern i's spans / stats after rj_multi_run. */
typedef struct rj_multi rj_multi;   /* like rj_scan: per caller, NOT thread-safe */
int rj_multi_create(const rj_program* const* progs, int n_progs, rj_multi** out);
void rj_multi_destroy(rj_multi* multi);
/* counts[i] = matches of pattern i over d_text[0..n).  Returns how the set was run: 1 = one fused scan
 * kernel for all patterns; 3 = scan + classification + counting in one kernel (rj_multi_set_counts_only); 2 = one scan kernel per pattern queued back to back, then the verify /
 * gather tails of all patterns in two launches and a single synchronise (any set of fixed-window
 * patterns); 0 = one complete pipeline after the other; <0 = rj_status */
int rj_multi_run(rj_multi* multi, const void* d_text, uint64_t n, uint64_t* counts, void* hip_stream);
/* the same for the matches whose begin lies in [own_begin, own_end) -- a shard with its halo inside
 * [0, n), as for rj_scan_run; no selection state is carried in (first shard / independent ranges) */
int rj_multi_run_range(rj_multi* multi, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end,
                       uint64_t* counts, void* hip_stream);
/* rj_multi_run in two halves: rj_multi_start enqueues the run's kernels on hip_stream and returns, rj_multi_finish
 * waits for them and collects the counts (its return value is rj_multi_run's).  One run in flight per rj_multi; a
 * caller that alternates between TWO rj_multi objects of the same patterns keeps the device busy while the host
 * turns a result around (the reference's counterpart: regexdna-multithread.cc keeps one thread per pattern busy,
 * sample/regexdna-multithread.cc:65-78) -- bench.py's headline loop does that.  The text must stay unchanged until
 * rj_multi_finish.  [own_begin, own_end) as for rj_multi_run_range (0, n + 1: the whole text). */
int rj_multi_start(rj_multi* multi, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, void* hip_stream);
int rj_multi_finish(rj_multi* multi, uint64_t* counts);
/* For two rj_multi objects run alternately on two streams: the scan kernel of `multi`'s runs is queued behind the
 * scan kernel of `before`'s run in flight (a stream-wait on that kernel's end), so that the two HBM-bound scans never
 * share the memory system while the latency-bound tails of one run execute under the scan of the other.  before =
 * NULL removes the order.  The objects must outlive each other's runs. */
int rj_multi_order_after(rj_multi* multi, rj_multi* before);
/* on != 0: rj_multi_start queues only the scan kernel on the caller's stream and everything behind it (the
 * latency-bound classify / gather tails) on a stream of the object's own, ordered by the scan kernel's end event.  A
 * caller that alternates two rj_multi objects on ONE stream then has the HBM-bound scan kernels back to back on that
 * stream -- in order, no cross-stream wait between them -- while each run's tails execute under the next scan.
 * rj_multi_finish is unchanged (it waits for the run's last kernel, wherever it is).  Not with mode 2. */
int rj_multi_set_tail_stream(rj_multi* multi, int on);
/* on != 0: the caller wants COUNTS -- what Regej::MatchAllCount answers (reference src/rejit.cc:203-208; regexdna asks
 * nothing else of its nine patterns, sample/regexdna.cc:65).  For the pattern sets the one-pass scan takes AND whose
 * patterns all match exactly 8 bytes within one byte of the scan's base windows (k-mers with one degenerate position,
 * both strands: regexdna's nine) rj_multi_run / _run_range / _start + _finish then run ONE kernel (plane_count.hip): the
 * text once, every candidate's eight bytes looked up in a table, nothing written but the counts.  Their return value is
 * 3 for such a run.  Afterwards counts[] and rj_multi_bounds / _bounds_device / rj_multi_device_counts work as ever (every
 * workgroup of the kernel records its first / last match per pattern: nothing is read again, the text may be gone);
 * rj_scan_device_spans(rj_multi_scan(m, i)) is NULL (no span list was made) and rj_scan_copy_spans / rj_scan_replace on it
 * return RJ_BAD_ARGUMENT.  Exactness: the counts are those of the left-most-longest, non-overlapping selection.  Two matches
 * of ONE pattern fewer than 8 bytes apart (`agggtaaagggtaaa`) are resolved inside the kernel when they form an isolated
 * pair; a longer chain of such neighbours, or a pair across two waves' spans, makes the kernel void its run, and THAT call
 * is answered by the span pipeline (return value 1) -- the next call tries the kernel again.  Any other pattern set ignores
 * the switch.  An rj_multi of ONE pattern takes the switch too.
 * Returns 1 when runs will take the one-kernel path, 0 when the set does not have the shape, < 0 rj_status. */
int rj_multi_set_counts_only(rj_multi* multi, int on);
/* rj_scan_set_timing for every pattern of the object (rj_multi_scan_ms reads 0 when off). */
int rj_multi_set_timing(rj_multi* multi, int on);
rj_scan* rj_multi_scan(rj_multi* multi, int i);
/* First and last match of every pattern's result after rj_multi_run / _run_range: bounds[4*i .. 4*i+3] =
 * first begin, first end, last begin, last end (all UINT64_MAX when pattern i has no match).  This is what
 * neighbouring shards exchange to carry the left-most-longest selection over a cut: a shard whose first
 * match begins before its left neighbour's last end re-runs that pattern with rj_scan_run(rj_multi_scan(m, i),
 * ..., carry) -- 32 bytes per pattern instead of the match list. */
int rj_multi_bounds(rj_multi* multi, uint64_t* bounds, void* hip_stream);
/* The rows of an exchange that stays on the device (one process per GPU, torch.distributed / RCCL): d_rows[8 * i ..]
 * = what a rank contributes per pattern to the carry exchange between shards: count | first begin, first end (written
 * only when first_round != 0: the result under the empty carry) | last begin, last end (-1 without a match) | the
 * carry the result was selected under (cur, prev_end, have: zeroed when first_round != 0, kept by the caller when
 * it re-runs a pattern); begins and ends plus `offset` (the shard's position in the whole text).  Written by a kernel
 * queued on hip_stream -- no synchronisation, the collective that gathers the rows is queued behind it. */
int rj_multi_bounds_device(rj_multi* multi, int64_t offset, int first_round, int64_t* d_rows, void* hip_stream);
/* The decision every rank takes from the gathered rows d_all[world][n_patterns][8] = count, first begin / end under
 * the EMPTY carry, current last begin / end, carry the current result was selected under (cur, prev_end, have):
 * out (device or pinned host memory, 4 * n_patterns + 1 integers) = job-wide counts | 1 where `rank` has to select
 * the pattern again | the (cur, prev_end) to select it under | 1 when any rank re-runs anything.  One kernel on
 * hip_stream; the caller synchronises once per round. */
int rj_carry_decide(const int64_t* d_all, int world, int rank, int n_patterns, int64_t* out, void* hip_stream);
/* The whole exchange step behind one call, for C++ callers with one process (or thread) per GPU and one shard of the
 * text resident on each (the reference's counterpart is the per-pattern thread pool of sample/regexdna-multithread.cc:
 * 65-78; across devices the unit is the byte range): runs the patterns over this rank's shard (arguments as for
 * rj_multi_run_range; `offset` = the position of d_text[0] in the whole text), then carries the left-most-longest
 * selection over the cuts -- per round one all-gather of 8 integers per pattern, written and evaluated by kernels on
 * hip_stream, one synchronise -- re-running the (rare) pattern whose first match begins inside its left neighbour's
 * last one.  counts[i] = matches of pattern i over the WHOLE text, on every rank; rj_multi_scan(m, i) holds this
 * shard's final spans.  rccl_comm is the caller's ncclComm_t (RCCL is bound with dlopen at the first call: the copy
 * the process has loaded already, else librccl.so.1 / $RJ_RCCL_LIBRARY); every rank of the communicator must call.
 * Ring-artefact-risk patterns need their shard's buffer to reach the end of the text (rj_program_info). */
int rj_multi_device_counts(rj_multi* multi, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end,
                           int64_t offset, void* rccl_comm, int rank, int world, uint64_t* counts, void* hip_stream);
/* The same over any all-gather: d_send (bytes_per_rank bytes, device) of every rank, in rank order, into d_recv
 * (world * bytes_per_rank bytes, device), queued on hip_stream or complete on return; 0 = success. */
typedef int (*rj_allgather_fn)(void* ctx, const void* d_send, void* d_recv, uint64_t bytes_per_rank, void* hip_stream);
int rj_multi_device_counts_via(rj_multi* multi, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end,
                               int64_t offset, rj_allgather_fn allgather, void* ctx, int rank, int world, uint64_t* counts,
                               void* hip_stream);

/* MatchAll of ONE pattern over a text that is sharded across ranks (one process or thread per GPU, a shard of the text
 * resident on each), the ordered match list gathered on `root` -- SURVEY 8e / BASELINE configs[3] (the complex regex
 * over 50 GB on 8 GPUs); the reference's result is that list (src/codegen.cc:36-86, include/rejit.h:65-68).  Arguments
 * as for rj_multi_device_counts: this rank owns the match begins [own_begin, own_end) of its buffer d_text[0..n), which
 * starts at `offset` of the whole text and holds a halo of max_len - 1 bytes behind own_end.  Runs the shard, carries
 * the left-most-longest selection over the cuts (rounds of one 64-byte all-gather), then every rank sends its pairs --
 * as GLOBAL offsets -- straight to their place in the root's list (ncclSend / ncclRecv in one group: peer -> root, no
 * ring).  Returns the job-wide number of matches on every rank (or rj_status); on `root`
 * rj_scan_gathered_spans(scan, &count) is the device pointer to the 2 * count offsets in text order (valid until the
 * scan's next run), NULL elsewhere.  rj_scan_device_spans(scan) stays this shard's own (local) result. */
int64_t rj_scan_gather_spans(rj_scan* scan, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, int64_t offset,
                             void* rccl_comm, int rank, int world, int root, void* hip_stream);
/* The same over any pair of collectives (MPI, a test harness with several shards on one device): the all-gather of
 * rj_multi_device_counts_via, and a gather of byte ranges to the root -- d_send[0..send_bytes) of every rank to
 * d_recv + recv_offsets[rank] on the root (d_recv is NULL elsewhere; recv_offsets / recv_bytes list all ranks, on all
 * ranks), queued on hip_stream or complete on return; 0 = success. */
typedef int (*rj_gatherv_fn)(void* ctx, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* recv_offsets,
                             const uint64_t* recv_bytes, int root, void* hip_stream);
int64_t rj_scan_gather_spans_via(rj_scan* scan, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, int64_t offset,
                                 rj_allgather_fn allgather, rj_gatherv_fn gatherv, void* ctx, int rank, int world, int root, void* hip_stream);
const uint64_t* rj_scan_gathered_spans(const rj_scan* scan, uint64_t* count);
/* a copy of that list (host_spans: host or device memory, cap pairs at most); returns the number of pairs gathered (0 on a
 * rank that is not the root) or rj_status */
int64_t rj_scan_copy_gathered_spans(const rj_scan* scan, uint64_t* host_spans, uint64_t cap);
/* mode 0 (default): fuse when possible; mode 1: never fuse -- every pattern scans the whole text on its own,
 * all of them in ONE launch when the patterns have the regexdna shape (scan_windows_train), else one kernel
 * per pattern back to back on the caller's stream; mode 2: one kernel per pattern, alternating between the
 * caller's stream and a second one so that consecutive kernels overlap at their boundaries; mode 3: one
 * kernel per pattern back to back (measurement: the per-kernel HBM roofline); mode 4: mode 0 with the scan and
 * the classification of its candidates in ONE kernel (round 4: measured no faster, kept selectable) */
int rj_multi_set_mode(rj_multi* multi, int mode);
/* duration of the last run's scan kernel(s) in ms, summed (0 when the patterns ran one by one) */
float rj_multi_scan_ms(const rj_multi* multi);

/* number of visible HIP devices (0 when there is none), for callers that want to probe */
int rj_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* REJIT_HIP_H_ */

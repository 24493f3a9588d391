/* oracle/rejit_oracle.h -- CPU restatement of coreperf/rejit's hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / reported
 * baseline.  Nothing under rejit_amd/ may include, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement against
 *   (1) the golden vectors derived from every TEST* line of the reference's own
 *       tools/tests/test.cc (tests/golden/testcc_vectors.json), and
 *   (2) oracle/_ref/librejit_ref.so -- the real reference compiled in place from
 *       /root/reference by oracle/Makefile, run with use_fast_forward=0 (the
 *       configuration whose results are correct) -- on
 *       seeded random regex/text pairs (differential fuzz; runs only where the
 *       prebuilt _ref library is present).
 */
#ifndef REJIT_ORACLE_H_
#define REJIT_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ro_regex ro_regex;

/* Status codes.  RO_PARSE_ERROR mirrors rejit::ParserError (include/rejit.h:98-102).
 * RO_REJECTED marks patterns on which the reference itself aborts, asserts or has
 * undefined behaviour (empty alternation branch, unmatched '(', stray ']', ...);
 * the product reports those as ParserError too. */
#define RO_OK 0
#define RO_PARSE_ERROR (-1)
#define RO_REJECTED (-2)

int ro_compile(const char* regexp, ro_regex** out);
void ro_free(ro_regex* re);
/* Last error message of this thread ("" when none). */
const char* ro_error(void);

/* kMatchAll (include/rejit.h:65-68): left-most longest, non-overlapping matches.
 * Writes up to cap (begin,end) offset pairs to out; returns the total count. */
long ro_match_all(const ro_regex* re, const uint8_t* text, size_t n, uint64_t* out, size_t cap);
/* kMatchFirst: defined as MatchAll[0] (the reference's own no-FF MatchFirst has
 * defect Q6).  Returns 1/0; writes be[0..1]. */
int ro_match_first(const ro_regex* re, const uint8_t* text, size_t n, uint64_t* be);
/* kMatchAnywhere: MatchAll non-empty. */
int ro_match_anywhere(const ro_regex* re, const uint8_t* text, size_t n);
/* kMatchFull: the regexp matches the whole text. */
int ro_match_full(const ro_regex* re, const uint8_t* text, size_t n);

/* One-shot helpers: compile + run + free.  Return RO_PARSE_ERROR/RO_REJECTED (<0)
 * when the pattern does not compile. */
long ro_match_all_re(const char* regexp, const uint8_t* text, size_t n, uint64_t* out, size_t cap);
int ro_match_full_re(const char* regexp, const uint8_t* text, size_t n);

/* The documented left-most-longest semantics computed without the ring dynamics
 * (longest match from every start + greedy pick).  Differs from ro_match_all only on
 * the reference artefact "Q8" (see rejit_oracle.c); used to classify such inputs. */
long ro_match_all_spec(const ro_regex* re, const uint8_t* text, size_t n, uint64_t* out, size_t cap);
long ro_match_all_spec_re(const char* regexp, const uint8_t* text, size_t n, uint64_t* out, size_t cap);

int ro_longest_all_re(const char* regexp, const uint8_t* text, size_t n, int64_t* ends);

/* Introspection used by tests. */
int ro_n_states(const ro_regex* re);
int ro_n_edges(const ro_regex* re);

#ifdef __cplusplus
}
#endif
#endif /* REJIT_ORACLE_H_ */

// oracle/ref_shim.cc -- C-ABI shim over the REAL reference (coreperf/rejit), linked
// into oracle/_ref/librejit_ref.so by oracle/Makefile from the reference sources
// where they lie under /root/reference.  TEST INFRASTRUCTURE ONLY (checker and
// cpu_baseline "reference" leg); never linked into the product.
//
// This file is ours: it only calls the reference's public API (include/rejit.h:41-138)
// and assigns its mutable flags (src/flags.h:36-72, available under -DMOD_FLAGS).
//
// Every entry point builds a FRESH Regej per call and per match type: re-using one
// Regej for two match types re-lists into the same RegexpInfo and can loop forever
// (SURVEY.md section 4.4, Q7).  Callers should still run us under a timeout.
#include <stdint.h>
#include <string.h>
#include <vector>
#include "rejit.h"
#include "flags.h"

using namespace rejit;

extern "C" {

// use_fast_forward=0 is the configuration whose MatchAll results are correct on every
// probe; default flags are needed for the speed baseline.
void ref_set_flags(int use_ff, int use_ff_early, int use_ff_reduce, int use_parser_opt) {
  SET_FLAG(use_fast_forward, use_ff != 0);
  SET_FLAG(use_fast_forward_early, use_ff_early != 0);
  SET_FLAG(use_ff_reduce, use_ff_reduce != 0);
  SET_FLAG(use_parser_opt, use_parser_opt != 0);
}

// Returns the number of matches, or -1 on parser error.  Writes up to cap
// (begin,end) offset pairs into out (may be NULL when cap == 0).
long ref_match_all(const char* re, const char* text, size_t n, uint64_t* out, size_t cap) {
  Regej r(re);
  if (r.status() != RejitSuccess) return -1;
  std::vector<Match> m;
  r.MatchAll(text, n, &m);
  for (size_t i = 0; i < m.size() && i < cap; i++) {
    out[2 * i] = (uint64_t)(m[i].begin - text);
    out[2 * i + 1] = (uint64_t)(m[i].end - text);
  }
  return (long)m.size();
}

// 1 = match (begin/end written), 0 = no match, -1 = parser error.
int ref_match_first(const char* re, const char* text, size_t n, uint64_t* be) {
  Regej r(re);
  if (r.status() != RejitSuccess) return -1;
  Match m;
  m.begin = m.end = NULL;
  bool ok = r.MatchFirst(text, n, &m);
  if (ok && be) {
    be[0] = (uint64_t)(m.begin - text);
    be[1] = (uint64_t)(m.end - text);
  }
  return ok ? 1 : 0;
}

int ref_match_full(const char* re, const char* text, size_t n) {
  Regej r(re);
  if (r.status() != RejitSuccess) return -1;
  return r.MatchFull(text, n) ? 1 : 0;
}

int ref_match_anywhere(const char* re, const char* text, size_t n) {
  Regej r(re);
  if (r.status() != RejitSuccess) return -1;
  return r.MatchAnywhere(text, n) ? 1 : 0;
}

// Timing helper for bench.py's cpu_baseline "reference" leg: compile once
// (excluded, as in tools/benchmarks/engines/rejit/engine.cc:86-103), run MatchAll
// `iters` times over the same buffer, return the match count of the last run.
long ref_match_all_repeat(const char* re, const char* text, size_t n, int iters) {
  Regej r(re);
  if (r.status() != RejitSuccess) return -1;
  if (!r.Compile(kMatchAll)) return -1;
  std::vector<Match> m;
  for (int i = 0; i < iters; i++) {
    m.clear();
    r.MatchAll(text, n, &m);
  }
  return (long)m.size();
}

const char* ref_status_string(void) { return rejit_status_string; }

}  // extern "C"

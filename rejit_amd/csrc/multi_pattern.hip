// rejit_amd/csrc/multi_pattern.hip -- several patterns over ONE device-resident text (rj_multi_*): the nine
// counts of regexdna (reference sample/regexdna.cc:51-67 runs one MatchAllCount per pattern).  One pass over
// the text for all patterns where their window sets allow it (scan_windows_fused), every pattern's own scan
// in one launch (scan_windows_train) or back to back, and in every mode the tails of all patterns together:
// two launches and ONE synchronise.  Patterns that do not take the in-region pipeline run one after the other
// through run_pipeline (engine.hip).

#include <cstring>
#include <dlfcn.h>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine_internal.h"
#include "exact_count.h"

using namespace rejit_amd;

#define fail ::rejit_amd::rj_fail

namespace {

// ----------------------------------------------------------------------------- fused multi-pattern run
// fixed windows + lane-sized automaton, not at risk of Q8: the in-region pipeline without carry
bool batchable(const rj_program* rp) {
  const DevProgram& D = rp->dev;
  return D.mode == 1 && D.float_range == 1 && D.n_words <= 4 && !rp->host->q8_risk && !D.behind;
}

bool fusable(const rj_program* rp) {
  const DevProgram& D = rp->dev;
  return D.mode == 1 && D.float_range == 1 && D.n_words <= 4 && D.win_len > 4 && D.n_windows <= 2 &&
         rp->window_alphabet <= 4 && rp->window_nibbles && !rp->host->q8_risk && !D.behind;
}

// nibble form of window k (see WindowSet::nibble)
void nibble_window(const DevProgram& D, int k, uint32_t* value, uint32_t* mask) {
  uint32_t v = 0, m = 0;
  for (int i = 0; i < 8; i++) {
    const uint32_t vb = (i < 4 ? (D.win_value0[k] >> (8 * i)) : (D.win_value1[k] >> (8 * (i - 4)))) & 0xFFu;
    const uint32_t mb = (i < 4 ? (D.win_mask0[k] >> (8 * i)) : (D.win_mask1[k] >> (8 * (i - 4)))) & 0xFFu;
    const int at = 8 * (i & 3) + 4 * (i >> 2);
    if (mb) {
      v |= (vb & 15u) << at;
      m |= 15u << at;
    }
  }
  *value = v;
  *mask = m;
}

// Plane scan (plane_scan.hip): do all windows of all patterns lie within ONE byte of <= 2 base windows over an
// alphabet that 2-bit symbol codes (byte >> shift) & 3 separate?  A base is a window without wildcards; greedy:
// the first uncovered exact window becomes the next base.
struct PlanePlan {
  bool ok = false;
  uint32_t code_shift = 0, n_bases = 0, offset = 0;
  uint8_t base[kPlaneMaxBases][8] = {};
  // the general plan (plan_plane_general): windows of differing offsets / lengths, any alphabet, <= 4 bases
  bool general = false;
  uint32_t n_cmp = 8, tolerance = 1, min_offset = 0, max_offset = 0;
};

PlanePlan plan_plane(const std::vector<rj_scan*>& scans) {
  PlanePlan pl;
  struct Win {
    uint8_t v[8];
    bool fixed[8];
  };
  std::vector<Win> wins;
  uint32_t table_words = 0;
  for (size_t p = 0; p < scans.size(); p++) {
    const DevProgram& D = scans[p]->prog->dev;
    if (D.win_len != 8 || D.n_windows < 1 || D.n_windows > 2) return pl;
    // classify_shared_multi keeps every pattern's tables in LDS: short bounded patterns only
    if (D.short_max == 0 || D.n_words > 2) return pl;
    table_words += (D.table_words + 3u) & ~3u;
    if (table_words > kClassifyMaxTableWords) return pl;
    if (p == 0) pl.offset = D.win_offset;
    else if (D.win_offset != pl.offset) return pl;
    for (int k = 0; k < D.n_windows; k++) {
      Win w;
      for (int i = 0; i < 8; i++) {
        const uint32_t vb = (i < 4 ? (D.win_value0[k] >> (8 * i)) : (D.win_value1[k] >> (8 * (i - 4)))) & 0xFFu;
        const uint32_t mb = (i < 4 ? (D.win_mask0[k] >> (8 * i)) : (D.win_mask1[k] >> (8 * (i - 4)))) & 0xFFu;
        if (mb != 0 && mb != 0xFFu) return pl;
        w.v[i] = static_cast<uint8_t>(vb);
        w.fixed[i] = mb != 0;
      }
      wins.push_back(w);
    }
  }
  auto off = [&](const Win& w, const uint8_t* base) {  // bytes in which w is free or differs from base
    int d = 0;
    for (int i = 0; i < 8; i++) d += (!w.fixed[i] || w.v[i] != base[i]) ? 1 : 0;
    return d;
  };
  // pass 0: exact windows become bases; pass 1 (round 6: `[cgt]gggtaaa|tttaccc[acg]` ALONE has no exact window): a window
  // with ONE free position becomes a base with one of its own fixed bytes there -- any byte will do, the window's strings
  // all lie within one byte of it, and one of its own keeps the alphabet at <= 4 symbols; pass 2: everything is covered
  int nb = 0;
  for (int pass = 0; pass < 3; pass++)
    for (const Win& w : wins) {
      bool covered = false;
      for (int b = 0; b < nb; b++) covered = covered || off(w, pl.base[b]) <= 1;
      if (covered) continue;
      int n_free = 0, free_at = 0, fixed_at = -1;
      for (int i = 0; i < 8; i++) {
        if (!w.fixed[i]) {
          n_free++;
          free_at = i;
        } else if (fixed_at < 0) {
          fixed_at = i;
        }
      }
      if (pass == 0 && n_free == 0 && nb < 2) {
        memcpy(pl.base[nb++], w.v, 8);
      } else if (pass == 1 && n_free == 1 && nb < 2) {
        memcpy(pl.base[nb], w.v, 8);
        pl.base[nb++][free_at] = w.v[fixed_at];
      } else if (pass == 2) {
        return pl;  // not within one byte of a base
      }
    }
  if (nb == 0) return pl;
  // symbol codes: the distinct bytes of the bases must get distinct codes (else the test loses its selectivity)
  bool seen[256] = {};
  std::vector<int> alphabet;
  for (int b = 0; b < nb; b++)
    for (int i = 0; i < 8; i++)
      if (!seen[pl.base[b][i]]) {
        seen[pl.base[b][i]] = true;
        alphabet.push_back(pl.base[b][i]);
      }
  if (alphabet.size() > 4) return pl;
  for (uint32_t sh = 0; sh <= 6; sh++) {
    uint32_t codes = 0;
    bool distinct = true;
    for (int c : alphabet) {
      const uint32_t code = (static_cast<uint32_t>(c) >> sh) & 3u;
      distinct = distinct && !((codes >> code) & 1u);
      codes |= 1u << code;
    }
    if (distinct) {
      pl.code_shift = sh;
      pl.n_bases = static_cast<uint32_t>(nb);
      pl.ok = true;
      return pl;
    }
  }
  return pl;
}

// The general plan (plane_scan_general): every pattern has 1..4 fixed-offset windows of >= 4 bytes and a short bounded
// automaton; the windows' first n_cmp bytes (n_cmp = the shortest window of the set) are covered by <= 4 base windows,
// exactly (tolerance 0: all of them literals) or within one byte (a window with ONE class position).  Codes are
// (byte >> shift) & 3 with the shift under which the bases' bytes take the most distinct codes; they may alias -- the
// test is a filter.  Refused when the filter would pass more than one position in 128 of uniform codes.
bool plane_general_ok(const rj_program* rp) {
  const DevProgram& D = rp->dev;
  return D.mode == 1 && D.float_range == 1 && !D.behind && !rp->host->q8_risk && D.short_max != 0 && D.n_words <= 2 && D.win_len >= 4 &&
         D.n_windows >= 1 && D.n_windows <= kClassifyMaxWindows;
}

PlanePlan plan_plane_general(const std::vector<rj_scan*>& scans) {
  PlanePlan pl;
  struct Win {
    uint8_t v[8];
    bool fixed[8];
  };
  std::vector<Win> wins;
  uint32_t table_words = 0, n_cmp = 8, min_off = ~0u, max_off = 0;
  for (rj_scan* sc : scans) {
    if (!plane_general_ok(sc->prog)) return pl;
    const DevProgram& D = sc->prog->dev;
    table_words += (D.table_words + 3u) & ~3u;
    if (table_words > kClassifyMaxTableWords) return pl;
    n_cmp = std::min<uint32_t>(n_cmp, D.win_len);
    min_off = std::min(min_off, D.win_offset);
    max_off = std::max(max_off, D.win_offset);
  }
  for (rj_scan* sc : scans) {
    const DevProgram& D = sc->prog->dev;
    for (int k = 0; k < D.n_windows; k++) {
      Win w;
      for (int i = 0; i < 8; i++) {
        const uint32_t vb = (i < 4 ? (D.win_value0[k] >> (8 * i)) : (D.win_value1[k] >> (8 * (i - 4)))) & 0xFFu;
        const uint32_t mb = (i < 4 ? (D.win_mask0[k] >> (8 * i)) : (D.win_mask1[k] >> (8 * (i - 4)))) & 0xFFu;
        if (mb != 0 && mb != 0xFFu) return pl;
        w.v[i] = static_cast<uint8_t>(vb);
        w.fixed[i] = mb != 0 && static_cast<uint32_t>(i) < n_cmp;
      }
      wins.push_back(w);
    }
  }
  auto off = [&](const Win& w, const uint8_t* base) {  // compared bytes in which w is free or differs from base
    int d = 0;
    for (uint32_t i = 0; i < n_cmp; i++) d += (!w.fixed[i] || w.v[i] != base[i]) ? 1 : 0;
    return d;
  };
  auto exact = [&](const Win& w) {
    for (uint32_t i = 0; i < n_cmp; i++)
      if (!w.fixed[i]) return false;
    return true;
  };
  int nb = 0;
  uint32_t tol = 0;
  for (uint32_t tol_try = 0; tol_try <= 1; tol_try++) {
    nb = 0;
    bool ok = true;
    // pass 0: exact windows become bases; pass 1 (tolerance 1 only; round 6): a window with ONE free position among the compared
    // bytes becomes a base with one of its own fixed bytes there (its strings all lie within one byte of it); pass 2: all covered
    for (int pass = 0; pass < 3 && ok; pass++)
      for (const Win& w : wins) {
        bool covered = false;
        for (int b = 0; b < nb; b++) covered = covered || off(w, pl.base[b]) <= static_cast<int>(tol_try);
        if (covered) continue;
        int n_free = 0, free_at = 0, fixed_at = -1;
        for (uint32_t i = 0; i < n_cmp; i++) {
          if (!w.fixed[i]) {
            n_free++;
            free_at = static_cast<int>(i);
          } else if (fixed_at < 0) {
            fixed_at = static_cast<int>(i);
          }
        }
        if (pass == 0 && exact(w) && nb < kPlaneMaxBases) {
          memcpy(pl.base[nb++], w.v, 8);
        } else if (pass == 1 && tol_try == 1 && n_free == 1 && fixed_at >= 0 && nb < kPlaneMaxBases) {
          memcpy(pl.base[nb], w.v, 8);
          pl.base[nb++][free_at] = w.v[fixed_at];
        } else if (pass == 2) {
          ok = false;
        }
      }
    if (ok && nb > 0) {
      tol = tol_try;
      break;
    }
    nb = 0;
  }
  if (nb == 0) return pl;
  // the shift under which the bases' bytes take the most distinct codes
  int best_sh = -1, best_distinct = 0;
  for (int sh = 0; sh <= 6; sh++) {
    // per compared position the number of distinct codes would be ideal; the total over the bases' bytes is a good proxy
    bool seen_pair[256][4] = {};
    uint32_t codes_seen = 0;
    int distinct_bytes_apart = 0;
    for (int b = 0; b < nb; b++)
      for (uint32_t i = 0; i < n_cmp; i++) {
        const uint32_t c = (static_cast<uint32_t>(pl.base[b][i]) >> sh) & 3u;
        if (!seen_pair[pl.base[b][i]][c]) {
          seen_pair[pl.base[b][i]][c] = true;
          if (!((codes_seen >> c) & 1u)) distinct_bytes_apart++;
          codes_seen |= 1u << c;
        }
      }
    if (distinct_bytes_apart > best_distinct) {
      best_distinct = distinct_bytes_apart;
      best_sh = sh;
    }
  }
  if (best_sh < 0 || best_distinct < 2) return pl;
  // expected pass rate over uniform codes: n_bases x (1 + 3 n_cmp tol) / 4^n_cmp
  double rate = static_cast<double>(nb) * (1.0 + 3.0 * n_cmp * tol);
  for (uint32_t i = 0; i < n_cmp; i++) rate /= 4.0;
  if (rate > 1.0 / 128.0) return pl;
  pl.ok = pl.general = true;
  pl.code_shift = static_cast<uint32_t>(best_sh);
  pl.n_bases = static_cast<uint32_t>(nb);
  pl.n_cmp = n_cmp;
  pl.tolerance = tol;
  pl.min_offset = min_off;
  pl.max_offset = max_off;
  pl.offset = 0;
  return pl;
}

}  // namespace

struct rj_multi {
  std::vector<rj_scan*> scans;
  PlanePlan plane;             // mode 0: the one-pass bit-plane scan with a shared candidate list
  DeviceBuffer shared_hits, shared_counts;
  uint32_t shared_cap_hint = 128;
  // round 4: scan + classification in one kernel (plane_scan_classify); off after a run whose span overflowed the
  // kernel's LDS candidate slots (then the two kernels with shared regions in device memory), or RJ_NO_FUSED_CLASSIFY
  bool fused_classify = true;
  bool want_fused_classify = false;  // rj_multi_set_mode(m, 4): mode 0 with the one-kernel form
  bool flags_clean = false;   // the counters that kernel may set but does not clear are zero on the device
  bool last_was_fused_classify = false;
  // what classify_shared_multi copies into LDS: ClassifyDesc[P] + the patterns' tables (kernels.h)
  DeviceBuffer classify_blob;
  ClassifyDesc* host_desc = nullptr;   // pinned
  std::vector<ClassifyDesc> desc_uploaded;
  uint32_t desc_words = 0, blob_words = 0;
  bool classify_tables_ready = false;
  DeviceBuffer dummy_counts;  // hit_counts of the padding patterns
  DeviceBuffer tails;         // MultiTail[P]
  MultiTail* host_tails = nullptr;  // pinned
  uint64_t* host_bounds = nullptr;  // pinned, rj_multi_bounds
  hipStream_t second = nullptr;     // separate-scans mode: odd patterns' scan kernels
  hipEvent_t fork = nullptr, join = nullptr;
  std::vector<MultiTail> uploaded;  // what the device array holds (skip the copy when nothing changed)
  bool fused = false;     // every pattern has a nibble-form window set: one kernel scans for all
  bool batchable = false; // every pattern takes the in-region pipeline: scans back to back, tails together
  int mode = 0;           // rj_multi_set_mode
  float scan_ms = 0.f;
  // rj_multi_start / rj_multi_finish: a run whose kernels are enqueued and whose results have not been collected
  struct Pending {
    bool active = false;
    int kind = 0;  // rj_multi_run's return value: 1 one pass, 2 separate scans + batched tails, 0 one pipeline after the other
    const uint8_t* text = nullptr;
    uint64_t n = 0, sb = 0, se = 0;
    hipStream_t st = nullptr;
    bool fuse = false;
    std::vector<uint64_t> caps;
    uint32_t shared_cap = 0;
    hipEvent_t done = nullptr;  // behind the run's last kernel: rj_multi_finish waits for THIS run, not for the stream
  } pending;
  rj_multi* scan_after = nullptr;  // rj_multi_order_after: this object's scan kernel waits for that one's
  // rj_multi_set_tail_stream: rj_multi_start queues the tails on this stream, behind the scan kernel's end event
  hipStream_t tail_stream = nullptr;
  bool tails_own_stream = false;
  // rj_multi_device_counts: this rank's rows [P][8] followed by every rank's [world][P][8]; the decision (pinned)
  DeviceBuffer exchange_rows;
  int64_t* host_decision = nullptr;
  // rj_multi_set_counts_only: MatchAllCount in ONE kernel (plane_count.hip) for the sets exact_count.h takes
  bool counts_only = false;
  ExactCountPlan exact;
  DeviceBuffer exact_table, count_acc, wg_rows, wg_bounds;
  unsigned long long* count_out = nullptr;   // pinned: counts, flags, first / last match per pattern
  bool counts_ready = false;   // the buffers above are allocated and cleared
  // ... and for the sets the GENERAL plan takes (plane_count.hip: GeneralShape): the plan (== plane when that is the general
  // one), whether every pattern can be classified inside the kernel, the blob of descriptors + tables
  PlanePlan gplane;
  bool general_counts = false;
  DeviceBuffer count_blob;
  uint32_t count_desc_words = 0, count_blob_words = 0, count_lmax = 0, count_max_short = 0;
  int count_max_words = 1;
  bool count_blob_ready = false;
  bool last_counts = false;    // the last run left counts (no span lists)
  uint32_t counts_fallbacks = 0;

};

namespace {

SharedHits shared_hits_of(rj_multi* m, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint32_t shared_cap, uint32_t n_regions, int P) {
  SharedHits sh{};
  sh.hits = m->shared_hits.as<uint64_t>();
  sh.counts = m->shared_counts.as<uint32_t>();
  sh.cap = shared_cap;
  sh.n_regions = n_regions;
  sh.n_patterns = static_cast<uint32_t>(P);
  sh.win_offset = m->plane.offset;
  sh.text = d_text;
  sh.n = n;
  sh.sb = sb;
  sh.se = se;
  static const bool forward = getenv("RJ_CLASSIFY_FORWARD") != nullptr;  // measurement override
  sh.reverse = forward ? 0u : 1u;
  return sh;
}

// The blob classify_shared_multi stages in LDS (kernels.h): the tables are copied once, device to device; the
// descriptors hold this run's output pointers and are uploaded when they changed (host_tails is filled already).
int classify_blob(rj_multi* m, hipStream_t st) {
  const int P = static_cast<int>(m->scans.size());
  if (!m->host_desc) RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->host_desc), sizeof(ClassifyDesc) * kMaxFused));
  m->desc_words = static_cast<uint32_t>((sizeof(ClassifyDesc) * static_cast<size_t>(P) + 15) / 16 * 4);
  uint32_t off = 0;
  for (int p = 0; p < P; p++) {
    const MultiTail& t = m->host_tails[p];
    const DevProgram& D = t.program;
    ClassifyDesc& d = m->host_desc[p];
    d = ClassifyDesc{};
    d.n_windows = static_cast<uint32_t>(std::min(D.n_windows, kClassifyMaxWindows));
    for (int q = 0; q < kClassifyMaxWindows; q++) {
      d.v0[q] = D.win_value0[q];
      d.m0[q] = D.win_mask0[q];
      d.v1[q] = D.win_value1[q];
      d.m1[q] = D.win_mask1[q];
    }
    d.win_offset = D.win_offset;
    d.win_len = D.win_len;
    d.tab = off;
    d.n_words = static_cast<uint32_t>(D.n_words);
    d.n_pos = static_cast<uint32_t>(D.n_pos);
    d.n_rows = static_cast<uint32_t>(D.n_rows);
    d.short_max = D.short_max;
    d.nullable = D.nullable;
    {
      const Program& H = *m->scans[static_cast<size_t>(p)]->prog->host;
      for (int pos = 0; pos < H.n_pos && pos < 64; pos++) {
        const int row = H.row_of[static_cast<size_t>(pos)];
        if (row < 0) continue;
        bool any = false;
        for (int k = 0; k < H.n_words; k++) any = any || H.rows[0][static_cast<size_t>(row) * H.n_words + k] != 0;
        if (any) d.rowbits[pos >> 5] |= 1u << (pos & 31);
      }
    }
    d.region_cap = t.verify.region_cap;
    d.begins = t.verify.hits;
    d.ends = t.region_ends;
    d.valid_counts = t.valid_counts;
    d.counters = t.verify.counters;
    off += (D.table_words + 3u) & ~3u;
  }
  m->blob_words = m->desc_words + off;
  if (!m->classify_tables_ready) {
    RJ_HIP(m->classify_blob.reserve(static_cast<size_t>(m->blob_words) * sizeof(uint32_t)));
    RJ_HIP(hipMemsetAsync(m->classify_blob.p, 0, static_cast<size_t>(m->blob_words) * sizeof(uint32_t), st));
    for (int p = 0; p < P; p++) {
      const DevProgram& D = m->host_tails[p].program;
      RJ_HIP(hipMemcpyAsync(m->classify_blob.as<uint32_t>() + m->desc_words + m->host_desc[p].tab, D.first,
                            static_cast<size_t>(D.table_words) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    }
    m->classify_tables_ready = true;
    m->desc_uploaded.clear();
  }
  if (m->desc_uploaded.size() != static_cast<size_t>(P) ||
      memcmp(m->desc_uploaded.data(), m->host_desc, sizeof(ClassifyDesc) * static_cast<size_t>(P)) != 0) {
    RJ_HIP(hipMemcpyAsync(m->classify_blob.p, m->host_desc, sizeof(ClassifyDesc) * static_cast<size_t>(P), hipMemcpyHostToDevice, st));
    m->desc_uploaded.assign(m->host_desc, m->host_desc + P);
  }
  return RJ_OK;
}

// The scans of all patterns (ONE fused kernel, or one kernel per pattern back to back) + the tails of
// all patterns in two launches + one synchronise.  Whole text, starts [0, n].
// phase 0: the whole run (enqueue, synchronise, collect; repeated with larger regions when one overflowed).
// phase 1 (rj_multi_start): enqueue the first attempt and return.  phase 2 (rj_multi_finish): synchronise and collect
// what phase 1 enqueued -- same arguments --, and carry on like phase 0 when a region overflowed.
// (internal) the batched pipeline's regions cannot hold this text's hits -- a text with a hit at (almost) every position:
// the caller runs every pattern's own pipeline, which has the dense and the large paths for that
constexpr int kRegionsFull = -100;
int run_batched(rj_multi* m, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st, bool fuse, int phase = 0) {
  const int P = static_cast<int>(m->scans.size());
  // chunks that can hold a window of a start in [sb, se): a window begins at most 7 bytes after its start
  const uint64_t end_byte = std::min<uint64_t>(n, se + 8);
  uint64_t chunks = std::max<uint64_t>((end_byte + 1023) / 1024 - sb / 1024, 1);
  // mode 0, every window within one byte of <= 2 bases: the bit-plane scan (plane_scan.hip), which walks the
  // window positions [wlo, whi) in pairs of chunks
  static const bool no_plane = getenv("RJ_NO_PLANE") != nullptr;  // measurement override
  const bool plane = fuse && m->mode == 0 && m->plane.ok && !no_plane;
  if (fuse && m->plane.general && !plane) fuse = false;  // (a general set has no other one-pass kernel: separate scans, batched tails)
  uint64_t plane_pairs = 0, plane_wlo = 0, plane_whi = 0;
  if (plane) {
    // window positions that can belong to a start in [sb, se): the general plan's patterns have their own offsets
    const uint32_t cmp = m->plane.general ? m->plane.n_cmp : 8u;
    plane_wlo = sb + (m->plane.general ? m->plane.min_offset : m->plane.offset);
    const uint64_t last_w = n >= cmp ? n - cmp + 1 : 0;
    plane_whi = std::min<uint64_t>(se + (m->plane.general ? m->plane.max_offset : m->plane.offset), last_w);
    plane_pairs = plane_whi > plane_wlo ? (plane_whi + 2047) / 2048 - plane_wlo / 2048 : 0;
    chunks = std::max<uint64_t>(plane_pairs * 2, 1);
  }
  // (plane scan, 500 MB: 96 chunks per workgroup measured best -- 92 us against 98 at 128, 96 at 64)
  static const int plane_chunks = getenv("RJ_PLANE_CHUNKS") ? atoi(getenv("RJ_PLANE_CHUNKS")) : 96;  // measurement override
  const ScanGeometry geo = scan_geometry(chunks, plane ? static_cast<uint64_t>(plane_chunks > 0 ? plane_chunks : 96) : 128);
  rj_scan* const s0 = m->scans[0];
  std::vector<uint64_t> caps(static_cast<size_t>(P));
  uint32_t shared_cap = 0;
  bool fused_launched = false;  // this attempt's scan kernel classified its own candidates (plane_scan_classify)
  for (int attempt = 0; attempt < 6; attempt++) {
   if (phase == 2 && attempt == 0) {
    caps = m->pending.caps;
    shared_cap = m->pending.shared_cap;
    fused_launched = m->last_was_fused_classify;
   } else {
    FusedParams fp{};
    fp.text = d_text;
    fp.n = n;
    fp.sb = sb;
    fp.se = se;
    fp.span_chunks = geo.span_chunks;
    fp.n_patterns = static_cast<uint32_t>((P + kFuseGroup - 1) / kFuseGroup * kFuseGroup);
    RJ_HIP(m->dummy_counts.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint32_t)));
    for (int p = 0; p < static_cast<int>(fp.n_patterns); p++) {
      const int q = p < P ? p : 0;  // padding repeats pattern 0 with no room for hits
      rj_scan* s = m->scans[static_cast<size_t>(q)];
      const DevProgram& D = s->prog->dev;
      nibble_window(D, 0, &fp.value[p][0], &fp.mask[p][0]);
      nibble_window(D, D.n_windows > 1 ? 1 : 0, &fp.value[p][1], &fp.mask[p][1]);
      fp.offset[p] = D.win_offset;
      fp.len[p] = D.win_len;
      if (p < P) {
        const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>(s->region_cap_hint, 64), geo.span_chunks * 1024);
        caps[static_cast<size_t>(p)] = cap;
        int rc = ensure_lists(s, geo.n_regions, static_cast<uint32_t>(cap), static_cast<uint64_t>(geo.n_regions) * cap);
        if (rc != RJ_OK) return rc;
        RJ_HIP(s->valid_counts.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint32_t)));
        fp.hits[p] = s->hits.as<uint64_t>();
        fp.region_cap[p] = static_cast<uint32_t>(cap);
        fp.hit_counts[p] = s->hit_counts.as<uint32_t>();
        fp.zero_counters[p] = s->counters.as<unsigned long long>();
        s->stats = rj_stats{};
        s->result = nullptr;
        s->result_count = 0;
      } else {
        fp.hits[p] = m->scans[0]->hits.as<uint64_t>();
        fp.region_cap[p] = 0;
        fp.hit_counts[p] = m->dummy_counts.as<uint32_t>();
        fp.zero_counters[p] = nullptr;
      }
    }
    fp.n_bases = 0;
    static const bool no_prefilter = getenv("RJ_NO_FUSED_PREFILTER") != nullptr;  // measurement override
    if (fuse && m->mode == 0 && !no_prefilter) {
      // Shared prefilter (kernels.hip: fused_chunk_d1): are all windows within one nibble of <= 2 base
      // windows?  Nibbles are compared on their low 3 bits there.  A base is a window without
      // wildcards; greedy: the first uncovered exact window becomes the next base.
      auto nibbles_off = [](uint32_t v, uint32_t k, uint32_t base) {  // nibbles in which (v, k) leaves `base` free or differs
        int d = 0;
        for (int i = 0; i < 8; i++) {
          const uint32_t kk = (k >> (4 * i)) & 7u, vv = (v >> (4 * i)) & 7u, bb = (base >> (4 * i)) & 7u;
          d += (kk == 0 || vv != bb) ? 1 : 0;
        }
        return d;
      };
      uint32_t bases[2] = {0, 0};
      int nb = 0;
      bool ok = true;
      for (int pass = 0; pass < 2 && ok; pass++)
        for (int p = 0; p < P && ok; p++)
          for (int w = 0; w < 2 && ok; w++) {
            const uint32_t v = fp.value[p][w] & 0x77777777u, k = fp.mask[p][w] & 0x77777777u;
            bool covered = false;
            for (int b = 0; b < nb; b++) covered = covered || nibbles_off(v, k, bases[b]) <= 1;
            if (covered) continue;
            if (pass == 0) {
              if (k == 0x77777777u && nb < 2) bases[nb++] = v;  // an exact window: a new base
            } else {
              ok = false;  // second pass: still not within one nibble of a base
            }
          }
      if (ok && nb > 0) {
        fp.n_bases = static_cast<uint32_t>(nb);
        fp.base[0] = bases[0];
        fp.base[1] = bases[nb > 1 ? 1 : 0];
        for (int p = 0; p < static_cast<int>(fp.n_patterns); p++)
          for (int w = 0; w < 2; w++) {
            fp.value[p][w] &= 0x77777777u;
            fp.mask[p][w] &= 0x77777777u;
          }
      }
    }
    // the single-pattern tails (verify inside the regions, offsets + gather + check) of all patterns
    // in two launches; their parameters travel as one small array
    for (int p = 0; p < P; p++) {
      rj_scan* s = m->scans[static_cast<size_t>(p)];
      MultiTail& t = m->host_tails[p];
      t = MultiTail{};
      t.verify.text = d_text;
      t.verify.n = n;
      t.verify.hits = s->hits.as<uint64_t>();
      t.verify.n_regions = geo.n_regions;
      t.verify.region_cap = static_cast<uint32_t>(caps[static_cast<size_t>(p)]);
      t.verify.counters = s->counters.as<unsigned long long>();
      t.verify.sb = sb;
      t.verify.se = se;
      t.verify.expand = 1;
      t.verify.float_max = s->prog->dev.float_max;
      t.program = s->prog->dev;
      t.hit_counts = s->hit_counts.as<uint32_t>();
      t.valid_counts = s->valid_counts.as<uint32_t>();
      t.region_ends = s->cand_end.as<uint64_t>();
      t.out = s->out.as<uint64_t>();
      t.out_cap = s->out_cap;
      t.host_counters = s->host_counters;
      s->host_counters[kCntUnordered] = 0;
      s->host_counters[kCntAdjacent] = 0;
    }
    if (m->uploaded.size() != static_cast<size_t>(P) ||
        memcmp(m->uploaded.data(), m->host_tails, sizeof(MultiTail) * static_cast<size_t>(P)) != 0) {
      RJ_HIP(hipMemcpyAsync(m->tails.p, m->host_tails, sizeof(MultiTail) * P, hipMemcpyHostToDevice, st));
      m->uploaded.assign(m->host_tails, m->host_tails + P);
    }
    shared_cap = 0;
    if (m->scan_after != nullptr && m->scan_after != m && m->scan_after->scans[0]->ev[2] != nullptr) {
      // two objects on two streams (rj_multi_order_after): the scan kernels -- both HBM-bound -- stay one behind the
      // other, only the other run's latency-bound tails overlap this scan
      RJ_HIP(hipStreamWaitEvent(st, m->scan_after->scans[0]->ev[2], 0));
    }
    if (plane && m->plane.general) {
      PlaneGParams pg{};
      pg.text = d_text;
      pg.n = n;
      pg.wlo = plane_wlo;
      pg.whi = plane_whi;
      pg.span_pairs = std::max<uint64_t>((plane_pairs + geo.n_regions - 1) / geo.n_regions, 1);
      pg.code_shift = m->plane.code_shift;
      pg.n_bases = m->plane.n_bases;
      pg.n_cmp = m->plane.n_cmp;
      pg.tolerance = m->plane.tolerance;
      for (uint32_t b = 0; b < kPlaneMaxBases; b++)
        for (int i = 0; i < 8; i++) {
          const uint32_t code = (static_cast<uint32_t>(m->plane.base[b < m->plane.n_bases ? b : 0][i]) >> m->plane.code_shift) & 3u;
          pg.lo[b][i] = (code & 1u) ? 0u : ~0u;
          pg.hi[b][i] = (code & 2u) ? 0u : ~0u;
        }
      shared_cap = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint32_t>(m->shared_cap_hint, 64), pg.span_pairs * 2048));
      RJ_HIP(m->shared_hits.reserve(static_cast<size_t>(geo.n_regions) * shared_cap * sizeof(uint64_t)));
      RJ_HIP(m->shared_counts.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint32_t)));
      pg.hits = m->shared_hits.as<uint64_t>();
      pg.region_cap = shared_cap;
      pg.hit_counts = m->shared_counts.as<uint32_t>();
      pg.n_zero = static_cast<uint32_t>(P);
      for (int p = 0; p < P; p++) pg.zero_counters[p] = m->scans[static_cast<size_t>(p)]->counters.as<unsigned long long>();
      fused_launched = false;
      m->flags_clean = false;
      // round 6: plane_count's streaming loop with the general test (code planes through the VGPR index mode), the candidates
      // written to the shared regions; plane_scan_general is kept behind RJ_PLANE_SCAN_V1
      static const bool v1g = getenv("RJ_PLANE_SCAN_V1") != nullptr;   // measurement override
      if (v1g) {
        launch_plane_scan_general(pg, geo.grid, s0->t0(), s0->ev[2], st);
      } else {
        PlaneListGParams pl{};
        PlaneCountParams& c = pl.g.c;
        c.text = d_text;
        c.n = n;
        c.sb = sb;
        c.se = se;
        c.first_block = plane_wlo / 2048;
        c.end_block = c.first_block + plane_pairs;
        c.span_blocks = plane_pairs / geo.n_regions;
        c.span_extra = static_cast<uint32_t>(plane_pairs % geo.n_regions);
        c.code_shift = m->plane.code_shift;
        c.n_bases = m->plane.n_bases;
        c.n_patterns = static_cast<uint32_t>(P);
        c.batch_at = 64;
        pl.g.n_cmp = m->plane.n_cmp;
        pl.g.tolerance = m->plane.tolerance;
        for (uint32_t b = 0; b < kPlaneMaxBases; b++)
          for (uint32_t i = 0; i < 8; i++)
            pl.g.idx[b][i] = i < m->plane.n_cmp ? (static_cast<uint32_t>(m->plane.base[b < m->plane.n_bases ? b : 0][i]) >> m->plane.code_shift) & 3u : 4u;
        pl.hits = pg.hits;
        pl.region_cap = pg.region_cap;
        pl.offset = 0;
        pl.hit_counts = pg.hit_counts;
        pl.n_zero = pg.n_zero;
        for (uint32_t p = 0; p < pg.n_zero; p++) pl.zero_counters[p] = pg.zero_counters[p];
        launch_plane_list_general(pl, geo.grid, s0->t0(), s0->ev[2], st);
      }
    } else if (plane) {
      PlaneParams pp{};
      pp.text = d_text;
      pp.n = n;
      pp.sb = sb;
      pp.se = se;
      pp.span_pairs = std::max<uint64_t>((plane_pairs + geo.n_regions - 1) / geo.n_regions, 1);
      pp.offset = m->plane.offset;
      pp.code_shift = m->plane.code_shift;
      pp.n_bases = m->plane.n_bases;
      for (uint32_t b = 0; b < 2; b++)
        for (int i = 0; i < 8; i++) {
          const uint32_t code = (static_cast<uint32_t>(m->plane.base[b < m->plane.n_bases ? b : 0][i]) >> m->plane.code_shift) & 3u;
          pp.lo[b][i] = (code & 1u) ? 0u : ~0u;
          pp.hi[b][i] = (code & 2u) ? 0u : ~0u;
        }
      shared_cap = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint32_t>(m->shared_cap_hint, 64), pp.span_pairs * 2048));
      RJ_HIP(m->shared_hits.reserve(static_cast<size_t>(geo.n_regions) * shared_cap * sizeof(uint64_t)));
      RJ_HIP(m->shared_counts.reserve(static_cast<size_t>(geo.n_regions) * sizeof(uint32_t)));
      pp.hits = m->shared_hits.as<uint64_t>();
      pp.region_cap = shared_cap;
      pp.hit_counts = m->shared_counts.as<uint32_t>();
      pp.n_zero = static_cast<uint32_t>(P);
      for (int p = 0; p < P; p++) pp.zero_counters[p] = m->scans[static_cast<size_t>(p)]->counters.as<unsigned long long>();
      // plane_scan_classify (scan + classification in one kernel): correct, but measured no faster than the two kernels
      // -- its classification is VALU work inside a kernel that is VALU co-limited (scan 99 -> 119 us per 500 MB, step
      // 0.152 ms either way) -- so it is opt-in (tests run both)
      static const bool want_fused_classify = getenv("RJ_FUSED_CLASSIFY") != nullptr;
      fused_launched = false;
      if (m->fused_classify && (want_fused_classify || m->want_fused_classify)) {
        const SharedHits sh = shared_hits_of(m, d_text, n, sb, se, 0, geo.n_regions, P);
        int rc = classify_blob(m, st);
        if (rc != RJ_OK) return rc;
        SharedHits shb = sh;
        shb.blob = m->classify_blob.as<uint32_t>();
        shb.desc_words = m->desc_words;
        shb.blob_words = m->blob_words;
        if (shb.blob_words <= kFusedMaxBlobWords) {
          if (!m->flags_clean) {
            for (int p = 0; p < P; p++)
              RJ_HIP(hipMemsetAsync(m->scans[static_cast<size_t>(p)]->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
            m->flags_clean = true;
          }
          int max_words = 1;
          uint32_t max_short = 0;
          for (int p = 0; p < P; p++) {
            const DevProgram& D = m->scans[static_cast<size_t>(p)]->prog->dev;
            max_words = std::max(max_words, static_cast<int>(D.n_words));
            max_short = std::max(max_short, D.short_max);
          }
          fused_launched = launch_plane_scan_classify(pp, shb, max_words, max_short, s0->counters.as<unsigned long long>(), geo.grid,
                                                      s0->t0(), s0->ev[2], st);
        }
      }
      if (!fused_launched) {
        m->flags_clean = false;
        // round 6: plane_count's streaming loop (32 contiguous bytes per lane, code planes through the VGPR index mode, the
        // blocks dealt out evenly) with the candidates written to the shared regions: plane_scan<NB> is kept behind RJ_PLANE_SCAN_V1
        static const bool v1 = getenv("RJ_PLANE_SCAN_V1") != nullptr;   // measurement override
        if (v1) {
          launch_plane_scan(pp, geo.grid, s0->t0(), s0->ev[2], st);
        } else {
          PlaneListParams pl{};
          pl.c.text = d_text;
          pl.c.n = n;
          pl.c.sb = sb;
          pl.c.se = se;
          pl.c.first_block = plane_wlo / 2048;
          pl.c.end_block = pl.c.first_block + plane_pairs;
          pl.c.span_blocks = plane_pairs / geo.n_regions;
          pl.c.span_extra = static_cast<uint32_t>(plane_pairs % geo.n_regions);
          pl.c.code_shift = m->plane.code_shift;
          pl.c.n_bases = m->plane.n_bases;
          pl.c.n_patterns = static_cast<uint32_t>(P);
          pl.c.batch_at = 64;
          for (uint32_t b = 0; b < 2; b++) {
            const uint32_t bb = b < m->plane.n_bases ? b : 0;
            for (int i = 0; i < 8; i++) {
              const uint32_t code = (static_cast<uint32_t>(m->plane.base[bb][i]) >> m->plane.code_shift) & 3u;
              if (!(code & 1u)) pl.c.mask_bits |= 1u << (16 * b + 2 * i);
              if (!(code & 2u)) pl.c.mask_bits |= 1u << (16 * b + 2 * i + 1);
            }
          }
          pl.hits = pp.hits;
          pl.region_cap = pp.region_cap;
          pl.offset = pp.offset;
          pl.hit_counts = pp.hit_counts;
          pl.n_zero = pp.n_zero;
          for (uint32_t p = 0; p < pp.n_zero; p++) pl.zero_counters[p] = pp.zero_counters[p];
          launch_plane_list(pl, geo.grid, s0->t0(), s0->ev[2], st);
        }
      }
    } else if (fuse) {
      launch_scan_windows_fused(fp, geo.grid, s0->t0(), s0->ev[2], st);
    } else {
      // every pattern's own scan kernel (each at its full streaming rate), queued back to back on
      // the caller's stream -- or, mode 2, alternating between it and a second stream: kernels of ONE
      // stream run strictly one after the other, so every kernel boundary costs the drain of the
      // last workgroups plus the ramp-up of the next grid; with two streams the next kernel's
      // workgroups fill the slots as they become free (regexdna step 0.95 -> 0.89 ms)
      // (mode 2 only: the kernels of the two streams overlap in time, so a per-kernel duration no
      // longer means what a roofline needs; the default keeps them on the caller's stream)
      const bool two_streams = m->mode == 2;
      // mode 1, every pattern with the regexdna shape (two nibble-form windows): the scans as one launch
      static const bool no_train = getenv("RJ_NO_TRAIN") != nullptr;  // measurement override
      bool train = m->mode == 1 && !no_train;  // (mode 3: one launch per pattern, as round 1 did)
      for (int p = 0; p < P && train; p++) train = fusable(m->scans[static_cast<size_t>(p)]->prog);
      if (train) {
        TrainParams tp{};
        tp.text = d_text;
        tp.n = n;
        tp.sb = sb;
        tp.se = se;
        tp.span_chunks = geo.span_chunks;
        tp.n_patterns = static_cast<uint32_t>(P);
        bool masked = false;
        for (int p = 0; p < P; p++) {
          rj_scan* s = m->scans[static_cast<size_t>(p)];
          const DevProgram& D = s->prog->dev;
          tp.value[p][0] = fp.value[p][0];
          tp.mask[p][0] = fp.mask[p][0];
          tp.value[p][1] = fp.value[p][1];
          tp.mask[p][1] = fp.mask[p][1];
          masked = masked || fp.mask[p][0] != 0xFFFFFFFFu || fp.mask[p][1] != 0xFFFFFFFFu;
          tp.offset[p] = D.win_offset;
          tp.len[p] = D.win_len;
          tp.wlo[p] = sb + D.win_offset;
          const uint64_t last_w = n >= D.win_len ? n - D.win_len + 1 : 0;
          tp.whi[p] = std::min<uint64_t>(se + D.win_offset, last_w);
          if (tp.whi[p] < tp.wlo[p]) tp.whi[p] = tp.wlo[p];
          tp.hits[p] = s->hits.as<uint64_t>();
          tp.region_cap[p] = static_cast<uint32_t>(caps[static_cast<size_t>(p)]);
          tp.hit_counts[p] = s->hit_counts.as<uint32_t>();
          tp.zero_counters[p] = s->counters.as<unsigned long long>();
        }
        launch_scan_windows_train(tp, masked, geo.grid, s0->t0(), s0->ev[2], st);
      }
      if (two_streams) {
        RJ_HIP(hipEventRecord(m->fork, st));
        RJ_HIP(hipStreamWaitEvent(m->second, m->fork, 0));
      }
      for (int p = 0; p < P && !train; p++) {
        rj_scan* s = m->scans[static_cast<size_t>(p)];
        const DevProgram& D = s->prog->dev;
        ScanParams sp{};
        sp.text = d_text;
        sp.n = n;
        sp.sb = sb;
        sp.se = se;
        sp.wlo = sb + D.win_offset;
        const uint64_t last_w = n >= D.win_len ? n - D.win_len + 1 : 0;
        sp.whi = std::min<uint64_t>(se + D.win_offset, last_w);
        if (sp.whi < sp.wlo) sp.whi = sp.wlo;
        sp.span_chunks = geo.span_chunks;
        sp.hits = s->hits.as<uint64_t>();
        sp.region_cap = static_cast<uint32_t>(caps[static_cast<size_t>(p)]);
        sp.hit_counts = s->hit_counts.as<uint32_t>();
        sp.zero_counters = s->counters.as<unsigned long long>();
        // one pair of timestamps around the whole train of scan kernels (first kernel's start, last
        // kernel's end): a pair per kernel puts a completion signal between consecutive kernels
        hipStream_t sp_stream = (two_streams && (p & 1)) ? m->second : st;
        launch_scan_windows(sp, make_window_set(s->prog), D.n_windows, geo.grid, p == 0 ? s0->t0() : nullptr,
                            (!two_streams && p == P - 1) ? s0->ev[2] : nullptr, sp_stream);
      }
      if (two_streams) {
        RJ_HIP(hipEventRecord(m->join, m->second));
        RJ_HIP(hipStreamWaitEvent(st, m->join, 0));
        RJ_HIP(hipEventRecord(s0->ev[2], st));  // end of the train: both streams have drained
      }
    }
    // rj_multi_set_tail_stream: everything behind the scan goes to the object's own stream, ordered by the scan's end
    // event -- the caller's stream is free for the next scan kernel (of another rj_multi) at once
    hipStream_t ts = st;
    if (phase == 1 && m->tails_own_stream && m->tail_stream != nullptr && m->mode != 2) {
      ts = m->tail_stream;
      RJ_HIP(hipStreamWaitEvent(ts, s0->ev[2], 0));
    }
    if (plane && fused_launched) {
      launch_offsets_gather_check_multi(m->tails.as<MultiTail>(), P, geo.n_regions, ts);
    } else if (plane) {
      SharedHits sh = shared_hits_of(m, d_text, n, sb, se, shared_cap, geo.n_regions, P);
      int rc = classify_blob(m, ts);
      if (rc != RJ_OK) return rc;
      sh.blob = m->classify_blob.as<uint32_t>();
      sh.desc_words = m->desc_words;
      sh.blob_words = m->blob_words;
      int max_words = 1;
      uint32_t max_short = 0;
      for (int p = 0; p < P; p++) {
        const DevProgram& D = m->scans[static_cast<size_t>(p)]->prog->dev;
        max_words = std::max(max_words, static_cast<int>(D.n_words));
        max_short = std::max(max_short, D.short_max);
      }
      if (m->plane.general) launch_tails_shared_general(m->tails.as<MultiTail>(), sh, max_words, max_short, s0->counters.as<unsigned long long>(), ts);
      else launch_tails_shared(m->tails.as<MultiTail>(), sh, max_words, max_short, s0->counters.as<unsigned long long>(), ts);
    } else {
      launch_tails_multi(m->tails.as<MultiTail>(), P, geo.n_regions, ts);
    }
    if (phase == 1) {
      if (!m->pending.done) RJ_HIP(hipEventCreateWithFlags(&m->pending.done, hipEventDisableTiming));
      RJ_HIP(hipEventRecord(m->pending.done, ts));
    }
   }
    if (phase == 1) {
      m->pending.caps = caps;
      m->pending.shared_cap = shared_cap;
      m->last_was_fused_classify = fused_launched;
      return RJ_OK;  // (pending.done was recorded behind the tails, on the stream that holds them)
    }
    // (phase 2: the stream may already hold the NEXT run of another rj_multi -- wait for this one only)
    if (phase == 2 && attempt == 0) RJ_HIP(hipEventSynchronize(m->pending.done));
    else RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    bool again = false;
    if (plane && fused_launched) {
      // the flags plane_scan_classify sets but does not clear: dirty when any of them is up
      for (int p = 0; p < P; p++)
        if (m->scans[static_cast<size_t>(p)]->host_counters[kCntOverflow] != 0) m->flags_clean = false;
      if (s0->host_counters[kCntSharedMax] != 0) {
        // a span with more candidates than the kernel's LDS slots: the two kernels with shared regions in device memory
        m->flags_clean = false;
        m->fused_classify = false;
        m->shared_cap_hint = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(s0->host_counters[kCntSharedMax] * 2, m->shared_cap_hint), 1u << 20));
        s0->stats.retries++;
        continue;
      }
    } else if (plane && s0->host_counters[kCntSharedMax] != 0) {
      // a shared candidate region overflowed: size them all for the fullest one seen (x2) and run again
      const uint64_t want = std::max<uint64_t>(s0->host_counters[kCntSharedMax] * 2, static_cast<uint64_t>(shared_cap) * 2);
      if (shared_cap >= 2048 * std::max<uint64_t>((plane_pairs + geo.n_regions - 1) / geo.n_regions, 1))
        return kRegionsFull;   // (run_spans: one pipeline after the other)
      m->shared_cap_hint = static_cast<uint32_t>(std::min<uint64_t>(want, 1u << 20));
      again = true;
    }
    for (int p = 0; p < P; p++) {
      rj_scan* s = m->scans[static_cast<size_t>(p)];
      if (s->host_counters[kCntOverflow] != 0) {
        const uint64_t cap = caps[static_cast<size_t>(p)];
        const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(s->host_counters[kCntMaxRegion] * 2, cap * 4), geo.span_chunks * 1024);
        if (want <= cap) return kRegionsFull;
        s->region_cap_hint = static_cast<uint32_t>(std::min<uint64_t>(want, 1u << 20));
        again = true;
      }
    }
    if (!s0->timing) {
      m->scan_ms = 0.f;
    } else if (fuse) {
      if (s0->timing) (void)hipEventElapsedTime(&m->scan_ms, s0->ev[1], s0->ev[2]);
    } else {
      if (s0->timing) (void)hipEventElapsedTime(&m->scan_ms, s0->ev[1], s0->ev[2]);  // first start to last end, gaps included
    }
    if (again) continue;
    for (int p = 0; p < P; p++) {
      rj_scan* s = m->scans[static_cast<size_t>(p)];
      hipStream_t sp = st;  // (rare) selection kernels of one pattern after the other
      if (s->host_counters[kCntOverrun] != 0) {
        // a long-lived candidate: this pattern's run is void; its own pipeline takes the carry scan
        s->linear_hint = true;
        int rc = run_pipeline(s, d_text, n, sb, se, 0, 0, 0, st);
        if (rc != RJ_OK) return rc;
        continue;
      }
      s->hits_hint = s->host_counters[kCntHits];
      s->stats.n_hits = s->host_counters[kCntHits];
      FinalizeParams sel{};
      sel.carry_cur = 0;
      sel.carry_prev_end = 0;
      sel.have_prev = 0;
      if (s->host_counters[kCntUnordered] != 0) {
        const uint64_t nc = s->host_counters[kCntCands];
        RJ_HIP(s->keys_out.reserve(nc * sizeof(uint64_t)));
        RJ_HIP(s->vals_out.reserve(nc * sizeof(uint64_t)));
        launch_split_pairs(s->out.as<uint64_t>(), s->counters.as<unsigned long long>() + kCntCands, nc,
                           s->keys_out.as<uint64_t>(), s->vals_out.as<uint64_t>(), sp);
      }
      int rc = resolve_selection(s, sel, sp);
      if (rc != RJ_OK) return rc;
      s->result = s->out.as<uint64_t>();
      s->stats.n_matches = s->result_count;
      s->stats.scan_ms = fuse ? m->scan_ms : m->scan_ms / static_cast<float>(P);
    }
    return RJ_OK;
  }
  return kRegionsFull;
}


bool counts_shape(const rj_multi* m) {
  return (m->exact.ok && m->plane.ok && !m->plane.general && m->plane.offset == 0) || m->general_counts;
}

bool counts_path(const rj_multi* m) {
  static const bool off = getenv("RJ_NO_COUNTS") != nullptr;  // measurement override
  return m->counts_only && counts_shape(m) && m->mode == 0 && !off;
}

// the blob the general count kernel stages in LDS: ClassifyDesc per pattern (window constants, table offsets; no output
// pointers: nothing is written) + the automaton tables, copied device to device from the programs' own
int count_blob(rj_multi* m, hipStream_t st) {
  if (m->count_blob_ready) return RJ_OK;
  const int P = static_cast<int>(m->scans.size());
  std::vector<ClassifyDesc> desc(static_cast<size_t>(P));
  RJ_HIP(m->count_blob.reserve(static_cast<size_t>(m->count_blob_words) * sizeof(uint32_t)));
  RJ_HIP(hipMemsetAsync(m->count_blob.p, 0, static_cast<size_t>(m->count_blob_words) * sizeof(uint32_t), st));
  uint32_t off = 0;
  for (int p = 0; p < P; p++) {
    const DevProgram& D = m->scans[static_cast<size_t>(p)]->prog->dev;
    const Program& H = *m->scans[static_cast<size_t>(p)]->prog->host;
    ClassifyDesc& d = desc[static_cast<size_t>(p)];
    d = ClassifyDesc{};
    d.n_windows = static_cast<uint32_t>(std::min(D.n_windows, kClassifyMaxWindows));
    for (int q = 0; q < kClassifyMaxWindows; q++) {
      d.v0[q] = D.win_value0[q];
      d.m0[q] = D.win_mask0[q];
      d.v1[q] = D.win_value1[q];
      d.m1[q] = D.win_mask1[q];
    }
    d.win_offset = D.win_offset;
    d.win_len = D.win_len;
    d.tab = off;
    d.n_words = static_cast<uint32_t>(D.n_words);
    d.n_pos = static_cast<uint32_t>(D.n_pos);
    d.n_rows = static_cast<uint32_t>(D.n_rows);
    d.short_max = D.short_max;
    d.nullable = D.nullable;
    for (int pos = 0; pos < H.n_pos && pos < 64; pos++) {
      const int row = H.row_of[static_cast<size_t>(pos)];
      if (row < 0) continue;
      bool any = false;
      for (int k = 0; k < H.n_words; k++) any = any || H.rows[0][static_cast<size_t>(row) * H.n_words + k] != 0;
      if (any) d.rowbits[pos >> 5] |= 1u << (pos & 31);
    }
    RJ_HIP(hipMemcpyAsync(m->count_blob.as<uint32_t>() + m->count_desc_words + off, D.first, static_cast<size_t>(D.table_words) * sizeof(uint32_t),
                          hipMemcpyDeviceToDevice, st));
    off += (D.table_words + 3u) & ~3u;
  }
  // (a pageable source: the runtime stages it before the call returns)
  RJ_HIP(hipMemcpyAsync(m->count_blob.p, desc.data(), sizeof(ClassifyDesc) * static_cast<size_t>(P), hipMemcpyHostToDevice, st));
  RJ_HIP(hipStreamSynchronize(st));
  m->count_blob_ready = true;
  return RJ_OK;
}

// the span lists of every pattern over the starts [sb, se), synchronously: what rj_multi_run does without the counts switch
// (1 one pass, 2 separate scans + batched tails, 0 one pipeline after the other; < 0 rj_status)
int run_spans(rj_multi* m, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st) {
  if (m->fused && m->mode == 0 && n >= 16) {
    int rc = run_batched(m, d_text, n, sb, se, st, true);
    if (rc != kRegionsFull) return rc != RJ_OK ? rc : 1;
  } else if (m->batchable && n >= 16) {
    int rc = run_batched(m, d_text, n, sb, se, st, false);
    if (rc != kRegionsFull) return rc != RJ_OK ? rc : 2;
  }
  for (rj_scan* s : m->scans) {
    int rc = run_pipeline(s, d_text, n, sb, se, 0, 0, 0, st);
    if (rc != RJ_OK) return rc;
  }
  return 0;
}

// MatchAllCount of every pattern over the starts [sb, se) in one kernel.  phase as run_batched.  A void run (flags) is
// repeated by the span pipeline, synchronously -- THAT run: the next one tries the kernel again (round 5 left the object
// on the span pipeline for good after one tandem repeat).
int run_counts(rj_multi* m, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, hipStream_t st, int phase) {
  const int P = static_cast<int>(m->scans.size());
  rj_scan* const s0 = m->scans[0];
  if (phase != 2) {
    // window positions that can belong to a start in [sb, se): the general plan's patterns have their own offsets
    const bool general = !(m->exact.ok && m->plane.ok && !m->plane.general && m->plane.offset == 0);
    const PlanePlan& plan = general ? m->gplane : m->plane;
    const uint32_t cmp = general ? plan.n_cmp : 8u;
    const uint64_t last_w = n >= cmp ? n - cmp + 1 : 0;
    const uint64_t wlo = sb + (general ? plan.min_offset : 0u), whi = std::min<uint64_t>(se + (general ? plan.max_offset : 0u), last_w);
    const uint64_t first_block = wlo / 2048;
    const uint64_t end_block = whi > wlo ? (whi + 2047) / 2048 : first_block;
    const uint64_t blocks = end_block - first_block;
    // (500 MB, rounds 5-6 before the stash: 2048 .. 3584 workgroups measured equal, 0.094 ms; 5086 0.0985, 8192 0.101: a wave's
    // fixed costs want long spans.  With the stash four workgroups of ExactShape are resident per CU (35 KB of LDS each) and the
    // grid matters: KiB per workgroup 64 / 96 / 112 / 120 / 128 / 136 / 144 / 160 / 240 / 320 / 440 / 480 / 520 -> the kernel inside
    // the two-in-flight loop 0.0915 / 0.0869 / 0.0879 / 0.0895 / 0.0851 / 0.0860 / 0.0867 / 0.0870 / 0.0875 / 0.0906 / 0.1064 / 0.0860 /
    // 0.0915 ms -- 440 is 1109 workgroups, a second round of 85 behind the 1024 resident ones; 128 is the generic scans' share too)
    static const int count_chunks = getenv("RJ_COUNT_CHUNKS") ? atoi(getenv("RJ_COUNT_CHUNKS")) : 128;  // measurement override
    const ScanGeometry geo = scan_geometry(std::max<uint64_t>(blocks * 2, 1), static_cast<uint64_t>(count_chunks > 0 ? count_chunks : 128));
    if (!m->counts_ready) {
      if (!m->count_out) RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->count_out), sizeof(unsigned long long) * kPcHostWords));
      if (m->exact.ok) {
        RJ_HIP(m->exact_table.reserve(sizeof(uint32_t) * kExactTabWords));
        RJ_HIP(hipMemcpyAsync(m->exact_table.p, m->exact.table, sizeof(uint32_t) * kExactTabWords, hipMemcpyHostToDevice, st));
      }
      RJ_HIP(m->count_acc.reserve(sizeof(unsigned long long) * kPcAccWords));
      RJ_HIP(hipMemsetAsync(m->count_acc.p, 0, sizeof(unsigned long long) * kPcAccWords, st));
      m->counts_ready = true;   // (only now: a failure above leaves the next call to start over)
    }
    PlaneCountParams pc{};
    pc.text = d_text;
    pc.n = n;
    pc.sb = sb;
    pc.se = se;
    pc.first_block = first_block;
    pc.end_block = end_block;
    pc.span_blocks = blocks / geo.n_regions;
    pc.span_extra = static_cast<uint32_t>(blocks % geo.n_regions);
    pc.code_shift = plan.code_shift;
    pc.n_bases = plan.n_bases;
    pc.n_patterns = static_cast<uint32_t>(P);
    static const int batch_at = getenv("RJ_COUNT_BATCH") ? atoi(getenv("RJ_COUNT_BATCH")) : 64;  // measurement override
    pc.batch_at = static_cast<uint32_t>(std::min(std::max(batch_at, 1), 64));
    PlaneCountGParams pg{};
    if (general) {
      int rc = count_blob(m, st);
      if (rc != RJ_OK) return rc;
      pc.table = m->count_blob.as<uint32_t>();
      pc.table_words = m->count_blob_words;
      pg.n_cmp = plan.n_cmp;
      pg.tolerance = plan.tolerance;
      pg.lmax = m->count_lmax;
      pg.desc_words = m->count_desc_words;
      for (uint32_t b = 0; b < kPlaneMaxBases; b++)
        for (uint32_t i = 0; i < 8; i++)
          pg.idx[b][i] = i < plan.n_cmp ? (static_cast<uint32_t>(plan.base[b < plan.n_bases ? b : 0][i]) >> plan.code_shift) & 3u : 4u;
    } else {
      for (uint32_t b = 0; b < 2; b++) {
        const uint32_t bb = b < plan.n_bases ? b : 0;
        for (int i = 0; i < 8; i++) {
          const uint32_t code = (static_cast<uint32_t>(plan.base[bb][i]) >> plan.code_shift) & 3u;
          if (!(code & 1u)) pc.mask_bits |= 1u << (16 * b + 2 * i);
          if (!(code & 2u)) pc.mask_bits |= 1u << (16 * b + 2 * i + 1);
        }
        pc.base_lo[b] = m->exact.base_lo[bb];
        pc.base_hi[b] = m->exact.base_hi[bb];
      }
      pc.table = m->exact_table.as<uint32_t>();
    }
    pc.acc = m->count_acc.as<unsigned long long>();
    pc.host_out = m->count_out;
    RJ_HIP(m->wg_rows.reserve(sizeof(uint32_t) * kExactMaxPatterns * static_cast<size_t>(geo.grid)));
    RJ_HIP(m->wg_bounds.reserve(sizeof(unsigned long long) * 2 * kExactMaxPatterns * static_cast<size_t>(geo.grid)));
    pc.wg_rows = m->wg_rows.as<uint32_t>();
    pc.wg_bounds = m->wg_bounds.as<unsigned long long>();
    if (m->scan_after != nullptr && m->scan_after != m && m->scan_after->scans[0]->ev[2] != nullptr)
      RJ_HIP(hipStreamWaitEvent(st, m->scan_after->scans[0]->ev[2], 0));
    if (general) {
      pg.c = pc;
      launch_plane_count_general(pg, m->count_max_words, m->count_max_short, geo.grid, s0->t0(), s0->ev[2], st);
    } else {
      launch_plane_count(pc, geo.grid, s0->t0(), s0->ev[2], st);
    }
    // the rows added up: behind the scan, on the object's own stream when the caller keeps runs in flight (rj_multi_start):
    // the caller's stream is free for the next scan kernel at once
    hipStream_t fs = st;
    if (phase == 1) {
      if (!m->tail_stream) RJ_HIP(hipStreamCreateWithFlags(&m->tail_stream, hipStreamNonBlocking));
      fs = m->tail_stream;
      RJ_HIP(hipStreamWaitEvent(fs, s0->ev[2], 0));
    }
    launch_plane_count_finish(pc, geo.grid, fs);
    if (phase == 1) {
      if (!m->pending.done) RJ_HIP(hipEventCreateWithFlags(&m->pending.done, hipEventDisableTiming));
      RJ_HIP(hipEventRecord(m->pending.done, fs));
      return RJ_OK;
    }
  }
  if (phase == 2) RJ_HIP(hipEventSynchronize(m->pending.done));
  else RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  if (m->count_out[kPcHostFlags] != 0) {
    m->counts_fallbacks++;
    m->last_counts = false;
    int kind = run_spans(m, d_text, n, sb, se, st);
    return kind < 0 ? kind : RJ_OK;
  }
  m->scan_ms = 0.f;
  if (s0->timing) (void)hipEventElapsedTime(&m->scan_ms, s0->ev[1], s0->ev[2]);
  for (int p = 0; p < P; p++) {
    rj_scan* s = m->scans[static_cast<size_t>(p)];
    s->stats = rj_stats{};
    s->result = nullptr;   // (no span list: rj_scan_device_spans reads NULL, the readers of the list refuse)
    s->result_count = m->count_out[kPcHostCount + p];
    s->stats.n_matches = s->result_count;
    s->stats.scan_ms = m->scan_ms;
    s->stats.count_path = 1;
  }
  m->last_counts = true;
  return RJ_OK;
}

}  // namespace

namespace {
// the live rj_multi objects (rj_multi_destroy clears the scan_after pointers that name the object it frees)
std::mutex& live_multi_mutex() {
  static std::mutex mu;
  return mu;
}
std::vector<rj_multi*>& live_multi() {
  static std::vector<rj_multi*> v;
  return v;
}
}  // namespace

extern "C" {

int rj_multi_create(const rj_program* const* progs, int n_progs, rj_multi** out) {
  ErrnoGuard errno_guard;
  if (!progs || !out || n_progs < 1) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (n_progs > kMaxFused - kFuseGroup + 1) return fail(RJ_BAD_ARGUMENT, "at most %d patterns per rj_multi", kMaxFused - kFuseGroup + 1);
  auto m = std::make_unique<rj_multi>();
  bool all = true, all_batchable = true;
  for (int i = 0; i < n_progs; i++) {
    if (!progs[i]) return fail(RJ_BAD_ARGUMENT, "null program");
    rj_scan* s = nullptr;
    int rc = rj_scan_create(progs[i], &s);
    if (rc != RJ_OK) {
      rj_multi_destroy(m.release());
      return rc;
    }
    m->scans.push_back(s);
    all = all && fusable(progs[i]);
    all_batchable = all_batchable && batchable(progs[i]);
  }
  if (m->tails.reserve(sizeof(MultiTail) * static_cast<size_t>(n_progs)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&m->host_tails), sizeof(MultiTail) * static_cast<size_t>(n_progs)) != hipSuccess) {
    rj_multi_destroy(m.release());
    return fail(RJ_DEVICE_ERROR, "out of memory");
  }
  if (hipStreamCreateWithFlags(&m->second, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&m->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&m->join, hipEventDisableTiming) != hipSuccess) {
    rj_multi_destroy(m.release());
    return fail(RJ_DEVICE_ERROR, "hipStreamCreate / hipEventCreate failed");
  }
  m->fused = all && n_progs > 1 && getenv("RJ_NO_FUSION") == nullptr;
  if (m->fused && getenv("RJ_PLANE_GENERAL_ONLY") == nullptr) m->plane = plan_plane(m->scans);  // (env: measurement override)
  if (!m->plane.ok && n_progs > 1 && getenv("RJ_NO_FUSION") == nullptr && getenv("RJ_NO_PLANE") == nullptr && getenv("RJ_NO_PLANE_GENERAL") == nullptr) {
    // not the regexdna shape: the general one-pass plan (round 4)
    const PlanePlan g = plan_plane_general(m->scans);
    if (g.ok) {
      m->plane = g;
      m->fused = true;
    }
  }
  m->batchable = all_batchable && n_progs > 1;
  if (n_progs == 1 && all && getenv("RJ_NO_PLANE") == nullptr) {
    // ONE pattern: no fused span pipeline (its own pipeline is that already), but the plan of its windows is what the
    // one-kernel count needs -- Regej::MatchAllCount of a regexdna pattern (reference sample/regexdna.cc:65, src/rejit.cc:203-208)
    m->plane = plan_plane(m->scans);
  }
  if ((m->fused || n_progs == 1) && m->plane.ok && !m->plane.general && m->plane.offset == 0) {
    // the set's shape for MatchAllCount in one kernel (exact_count.h); used when the caller asks: rj_multi_set_counts_only
    std::vector<const Program*> hosts;
    for (rj_scan* s : m->scans) hosts.push_back(s->prog->host.get());
    make_exact_count_plan(hosts, m->plane.base, m->plane.n_bases, &m->exact);
  }
  if (!m->exact.ok && getenv("RJ_NO_PLANE") == nullptr && getenv("RJ_NO_PLANE_GENERAL") == nullptr && getenv("RJ_NO_COUNTS_GENERAL") == nullptr) {
    // ... and for everything else the general plan takes (round 6): alternations of literals of any length, k-mers of any
    // k, windows with a class position -- the reference's fast forward takes any such set, src/codegen.cc:327-393
    m->gplane = m->plane.ok && m->plane.general ? m->plane : plan_plane_general(m->scans);
    bool ok = m->gplane.ok;
    uint32_t words = 0;
    for (rj_scan* s : m->scans) {
      const DevProgram& D = s->prog->dev;
      const Program& H = *s->prog->host;
      ok = ok && !H.has_assertions && !H.q8_risk && !H.any_nullable && H.min_len >= 1 && D.short_max >= 1 && D.short_max <= 16 && D.n_words <= 2;
      words += (D.table_words + 3u) & ~3u;
      m->count_lmax = std::max(m->count_lmax, D.short_max);
      m->count_max_short = std::max(m->count_max_short, D.short_max);
      m->count_max_words = std::max(m->count_max_words, static_cast<int>(D.n_words));
    }
    m->count_desc_words = static_cast<uint32_t>((sizeof(ClassifyDesc) * static_cast<size_t>(n_progs) + 15) / 16 * 4);
    m->count_blob_words = m->count_desc_words + words;
    m->general_counts = ok && m->count_blob_words <= kCountMaxBlobWords && n_progs < kExactMaxPatterns && m->count_lmax >= 2;
  }
  {
    std::lock_guard<std::mutex> lock(live_multi_mutex());
    live_multi().push_back(m.get());
  }
  *out = m.release();
  return RJ_OK;
}

void rj_multi_destroy(rj_multi* m) {
  ErrnoGuard errno_guard;
  if (!m) return;
  {
    // rj_multi_order_after keeps a raw pointer: objects that wait for this one's scans stop doing so
    std::lock_guard<std::mutex> lock(live_multi_mutex());
    auto& live = live_multi();
    live.erase(std::remove(live.begin(), live.end(), m), live.end());
    for (rj_multi* other : live)
      if (other->scan_after == m) other->scan_after = nullptr;
  }
  for (rj_scan* s : m->scans) rj_scan_destroy(s);
  if (m->host_tails) (void)hipHostFree(m->host_tails);
  if (m->host_bounds) (void)hipHostFree(m->host_bounds);
  if (m->host_desc) (void)hipHostFree(m->host_desc);
  if (m->count_out) (void)hipHostFree(m->count_out);
  if (m->second) (void)hipStreamDestroy(m->second);
  if (m->fork) (void)hipEventDestroy(m->fork);
  if (m->join) (void)hipEventDestroy(m->join);
  if (m->pending.done) (void)hipEventDestroy(m->pending.done);
  if (m->host_decision) (void)hipHostFree(m->host_decision);
  if (m->tail_stream) (void)hipStreamDestroy(m->tail_stream);
  delete m;
}

int rj_multi_run(rj_multi* m, const void* d_text, uint64_t n, uint64_t* counts, void* hip_stream) {
  return rj_multi_run_range(m, d_text, n, 0, n + 1, counts, hip_stream);
}

int rj_multi_run_range(rj_multi* m, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, uint64_t* counts,
                       void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!m || (!d_text && n) || !counts) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (m->pending.active) return fail(RJ_BAD_ARGUMENT, "rj_multi_run: a run started with rj_multi_start has not been finished");
  if (own_end > n + 1) own_end = n + 1;
  if (own_begin >= own_end) {
    for (size_t i = 0; i < m->scans.size(); i++) counts[i] = 0;
    return 0;
  }
  if ((reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) return fail(RJ_BAD_ARGUMENT, "device text must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  m->scan_ms = 0.f;
  int fused = 0;
  m->last_counts = false;
  if (n >= 16 && counts_path(m)) {
    int rc = run_counts(m, static_cast<const uint8_t*>(d_text), n, own_begin, own_end, st, 0);
    if (rc != RJ_OK) return rc;
    fused = m->last_counts ? 3 : (m->fused && m->mode == 0) ? 1 : m->batchable ? 2 : 0;
  } else {
    fused = run_spans(m, static_cast<const uint8_t*>(d_text), n, own_begin, own_end, st);
    if (fused < 0) return fused;
  }
  for (size_t i = 0; i < m->scans.size(); i++) counts[i] = m->scans[i]->result_count;
  return fused;
}

int rj_multi_start(rj_multi* m, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!m || (!d_text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (m->pending.active) return fail(RJ_BAD_ARGUMENT, "rj_multi_start: the previous run has not been finished");
  if ((reinterpret_cast<uintptr_t>(d_text) & 15u) != 0) return fail(RJ_BAD_ARGUMENT, "device text must be 16-byte aligned");
  if (own_end > n + 1) own_end = n + 1;
  rj_multi::Pending& q = m->pending;
  q.text = static_cast<const uint8_t*>(d_text);
  q.n = n;
  q.sb = own_begin;
  q.se = own_end;
  q.st = static_cast<hipStream_t>(hip_stream);
  m->scan_ms = 0.f;
  q.kind = own_begin >= own_end ? -1 : (n >= 16 && counts_path(m)) ? 3 : (m->fused && m->mode == 0 && n >= 16) ? 1 : (m->batchable && n >= 16) ? 2 : 0;
  q.fuse = q.kind == 1;
  m->last_counts = false;
  if (q.kind == 3) {
    int rc = run_counts(m, q.text, n, q.sb, q.se, q.st, 1);
    if (rc != RJ_OK) return rc;
  } else if (q.kind > 0) {
    int rc = run_batched(m, q.text, n, q.sb, q.se, q.st, q.fuse, 1);
    if (rc != RJ_OK) return rc;
  }  // (kind 0: pattern sets that take one pipeline after the other run in rj_multi_finish)
  q.active = true;
  return RJ_OK;
}

int rj_multi_finish(rj_multi* m, uint64_t* counts) {
  ErrnoGuard errno_guard;
  if (!m || !counts) return fail(RJ_BAD_ARGUMENT, "null argument");
  rj_multi::Pending& q = m->pending;
  if (!q.active) return fail(RJ_BAD_ARGUMENT, "rj_multi_finish without rj_multi_start");
  q.active = false;
  if (q.kind < 0) {  // an empty range
    for (size_t i = 0; i < m->scans.size(); i++) counts[i] = 0;
    return 0;
  }
  if (q.kind == 3) {
    int rc = run_counts(m, q.text, q.n, q.sb, q.se, q.st, 2);
    if (rc != RJ_OK) return rc;
    if (!m->last_counts) q.kind = (m->fused && m->mode == 0) ? 1 : m->batchable ? 2 : 0;   // (a void run: the span pipeline answered)
  } else if (q.kind != 0) {
    int rc = run_batched(m, q.text, q.n, q.sb, q.se, q.st, q.fuse, 2);
    if (rc == kRegionsFull) {   // (a hit at almost every position: every pattern's own pipeline)
      q.kind = 0;
      for (rj_scan* s : m->scans) {
        rc = run_pipeline(s, q.text, q.n, q.sb, q.se, 0, 0, 0, q.st);
        if (rc != RJ_OK) return rc;
      }
    }
    if (rc != RJ_OK) return rc;
  } else {
    for (rj_scan* s : m->scans) {
      int rc = run_pipeline(s, q.text, q.n, q.sb, q.se, 0, 0, 0, q.st);
      if (rc != RJ_OK) return rc;
    }
  }
  for (size_t i = 0; i < m->scans.size(); i++) counts[i] = m->scans[i]->result_count;
  return q.kind;
}

int rj_multi_order_after(rj_multi* m, rj_multi* before) {
  if (!m) return fail(RJ_BAD_ARGUMENT, "null argument");
  std::lock_guard<std::mutex> lock(live_multi_mutex());
  m->scan_after = before;
  return RJ_OK;
}

rj_scan* rj_multi_scan(rj_multi* m, int i) {
  if (!m || i < 0 || static_cast<size_t>(i) >= m->scans.size()) return nullptr;
  return m->scans[static_cast<size_t>(i)];
}

float rj_multi_scan_ms(const rj_multi* m) { return m ? m->scan_ms : 0.f; }

int rj_multi_bounds(rj_multi* m, uint64_t* bounds, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!m || !bounds) return fail(RJ_BAD_ARGUMENT, "null argument");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int P = static_cast<int>(m->scans.size());
  if (!m->host_bounds) RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->host_bounds), sizeof(uint64_t) * 4 * kMaxFused));
  // a counts run: plane_count_finish left every pattern's first / last match (begin | length << 56) in pinned memory; a
  // pattern re-run under a carry since (rj_scan_run on rj_multi_scan) has its own list
  bool listed = !m->last_counts;
  if (m->last_counts)
    for (int p = 0; p < P; p++)
      if (m->scans[static_cast<size_t>(p)]->result) listed = true;
  if (listed) {
    BoundsParams bp{};
    bp.n_lists = P;
    for (int p = 0; p < P; p++) {
      bp.spans[p] = m->scans[static_cast<size_t>(p)]->result;
      bp.count[p] = bp.spans[p] ? m->scans[static_cast<size_t>(p)]->result_count : 0;
    }
    launch_first_last(bp, m->host_bounds, st);
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    memcpy(bounds, m->host_bounds, sizeof(uint64_t) * 4 * static_cast<size_t>(P));
  }
  if (m->last_counts)
    for (int p = 0; p < P; p++) {
      if (m->scans[static_cast<size_t>(p)]->result) continue;
      const unsigned long long f = m->count_out[kPcHostBounds + 2 * p], l = m->count_out[kPcHostBounds + 2 * p + 1];
      const unsigned long long at = (1ull << kPcLenShift) - 1;
      bounds[4 * p + 0] = f == kPcNone ? kPcNone : (f & at);
      bounds[4 * p + 1] = f == kPcNone ? kPcNone : (f & at) + (f >> kPcLenShift);
      bounds[4 * p + 2] = l == kPcNone ? kPcNone : (l & at);
      bounds[4 * p + 3] = l == kPcNone ? kPcNone : (l & at) + (l >> kPcLenShift);
    }
  return RJ_OK;
}

int rj_multi_bounds_device(rj_multi* m, int64_t offset, int first_round, int64_t* d_rows, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!m || !d_rows) return fail(RJ_BAD_ARGUMENT, "null argument");
  const int P = static_cast<int>(m->scans.size());
  BoundsParams bp{};
  bp.n_lists = P;
  for (int p = 0; p < P; p++) {
    bp.spans[p] = m->scans[static_cast<size_t>(p)]->result;
    bp.count[p] = bp.spans[p] ? m->scans[static_cast<size_t>(p)]->result_count : 0;
  }
  // (after a counts run the patterns without a list take count and bounds from the kernel's device copy)
  if (m->last_counts) launch_bounds_rows_counts(bp, m->count_acc.as<unsigned long long>(), offset, first_round, d_rows, static_cast<hipStream_t>(hip_stream));
  else launch_bounds_rows(bp, offset, first_round, d_rows, static_cast<hipStream_t>(hip_stream));
  RJ_HIP(hipGetLastError());
  return RJ_OK;
}

int rj_carry_decide(const int64_t* d_all, int world, int rank, int n_patterns, int64_t* out, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!d_all || !out || world < 1 || rank < 0 || rank >= world || n_patterns < 1 || n_patterns > 64) return fail(RJ_BAD_ARGUMENT, "bad argument");
  launch_carry_decide(d_all, world, rank, n_patterns, out, static_cast<hipStream_t>(hip_stream));
  RJ_HIP(hipGetLastError());
  return RJ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The exchange step of a sharded multi-pattern count behind ONE call, for C++ callers with one process (or thread)
// per GPU: what rejit_amd/sharding.py's CarryExchange does over torch.distributed, here over the collective the
// caller hands in -- RCCL's all-gather on the caller's communicator (rj_multi_device_counts), or any function of
// that shape (rj_multi_device_counts_via: MPI, a test harness with several shards on one device).
//   per round: rows of this rank (kernel) -> all-gather of 8 integers per pattern -> decision (kernel) -> ONE
//   synchronise -> the (rare) patterns whose selection has to be repeated under the left neighbour's last match.
// Rounds end when no rank repeats anything: at most `world` of them (a carry travels one shard per round).
int rj_multi_device_counts_via(rj_multi* m, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, int64_t offset,
                               rj_allgather_fn allgather, void* ctx, int rank, int world, uint64_t* counts, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!m || !allgather || !counts || world < 1 || rank < 0 || rank >= world) return fail(RJ_BAD_ARGUMENT, "bad argument");
  const int P = static_cast<int>(m->scans.size());
  if (P < 1 || P > 64) return fail(RJ_BAD_ARGUMENT, "rj_multi_device_counts: 1..64 patterns");
  const int kind = rj_multi_run_range(m, d_text, n, own_begin, own_end, counts, hip_stream);
  if (kind < 0) return kind;
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const size_t row_words = static_cast<size_t>(P) * 8;
  RJ_HIP(m->exchange_rows.reserve(sizeof(int64_t) * row_words * static_cast<size_t>(world + 1)));
  if (!m->host_decision) RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->host_decision), sizeof(int64_t) * (4 * 64 + 1)));
  int64_t* mine = m->exchange_rows.as<int64_t>();
  int64_t* all = mine + row_words;
  int64_t* out = m->host_decision;
  for (int round = 0; round <= world; round++) {
    const int e = rj_multi_bounds_device(m, offset, round == 0, mine, hip_stream);
    if (e < 0) return e;
    if (allgather(ctx, mine, all, sizeof(int64_t) * row_words, hip_stream) != 0)
      return fail(RJ_DEVICE_ERROR, "rj_multi_device_counts: the all-gather failed (round %d)", round);
    const int d = rj_carry_decide(all, world, rank, P, out, hip_stream);
    if (d < 0) return d;
    RJ_HIP(hipStreamSynchronize(st));
    if (out[4 * P] == 0) {
      for (int i = 0; i < P; i++) counts[i] = static_cast<uint64_t>(out[i]);
      return kind;
    }
    for (int i = 0; i < P; i++) {
      if (!out[P + i]) continue;
      // (global offsets in the rows; the shard's own run takes them relative to its buffer)
      const int64_t cur = out[2 * P + 2 * i], pe = out[2 * P + 2 * i + 1];
      const int have = (cur != 0 || pe != 0) ? 1 : 0;
      // in the shard's own coordinates; a previous match that ENDS before the buffer begins cannot touch anything in it
      // (clamping its end to 0 would suppress a legitimate empty match at local position 0): no carry then
      const int have_local = have && pe >= offset ? 1 : 0;
      const uint64_t lc = have_local && cur > offset ? static_cast<uint64_t>(cur - offset) : 0, lp = have_local ? static_cast<uint64_t>(pe - offset) : 0;
      const int64_t k = rj_scan_run(m->scans[static_cast<size_t>(i)], d_text, n, own_begin, own_end, lc, lp, have_local, hip_stream);
      if (k < 0) return static_cast<int>(k);
      const int64_t used[3] = {cur, pe, have};  // the carry this result was selected under: part of the next round's row
      RJ_HIP(hipMemcpyAsync(mine + 8 * i + 5, used, sizeof(used), hipMemcpyHostToDevice, st));
      RJ_HIP(hipStreamSynchronize(st));  // (`used` is on the stack)
    }
  }
  return fail(RJ_DEVICE_ERROR, "rj_multi_device_counts: the carry exchange did not converge in %d rounds", world + 1);
}

namespace {
// RCCL, bound at the first call: the library is an optional companion of this one (a single-GPU caller never
// needs it), and a process that has loaded RCCL already -- through PyTorch, say -- must get THAT copy.
struct Rccl {
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
};
const Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {std::getenv("RJ_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* name : names) {
      if (!name || !*name) continue;
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(h, "ncclAllGather"));
        r.send = reinterpret_cast<decltype(r.send)>(dlsym(h, "ncclSend"));
        r.recv = reinterpret_cast<decltype(r.recv)>(dlsym(h, "ncclRecv"));
        r.group_start = reinterpret_cast<decltype(r.group_start)>(dlsym(h, "ncclGroupStart"));
        r.group_end = reinterpret_cast<decltype(r.group_end)>(dlsym(h, "ncclGroupEnd"));
        if (r.all_gather && r.send && r.recv && r.group_start && r.group_end) return;
        r = Rccl{};
      }
    }
  });
  return r;
}
constexpr int kNcclInt64 = 4, kNcclUint64 = 5;  // ncclDataType_t (rccl.h)
int rccl_gather(void* comm, const void* send, void* recv, uint64_t bytes, void* stream) {
  return rccl().all_gather(send, recv, bytes / sizeof(int64_t), kNcclInt64, comm, static_cast<hipStream_t>(stream));
}
// every rank's pairs straight to their place in the root's list: point to point, root <- rank, one group (SURVEY 8e:
// "direct peer -> root transfers, not a ring")
struct RcclGatherCtx {
  void* comm;
  int rank, world;
};
int rccl_gatherv(void* ctx, const void* d_send, uint64_t send_bytes, void* d_recv, const uint64_t* recv_offsets, const uint64_t* recv_bytes, int root,
                 void* stream) {
  const RcclGatherCtx* c = static_cast<const RcclGatherCtx*>(ctx);
  const Rccl& R = rccl();
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = R.group_start();
  if (c->rank == root)
    for (int r = 0; r < c->world && rc == 0; r++)
      if (recv_bytes[r]) rc = R.recv(static_cast<char*>(d_recv) + recv_offsets[r], recv_bytes[r] / 8, kNcclUint64, r, c->comm, st);
  if (rc == 0 && send_bytes) rc = R.send(d_send, send_bytes / 8, kNcclUint64, root, c->comm, st);
  const int rc2 = R.group_end();
  return rc ? rc : rc2;
}
int rccl_allgather_ctx(void* ctx, const void* send, void* recv, uint64_t bytes, void* stream) {
  return rccl_gather(static_cast<const RcclGatherCtx*>(ctx)->comm, send, recv, bytes, stream);
}
}  // namespace

int rj_multi_device_counts(rj_multi* m, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, int64_t offset,
                           void* rccl_comm, int rank, int world, uint64_t* counts, void* hip_stream) {
  if (!rccl_comm) return fail(RJ_BAD_ARGUMENT, "rj_multi_device_counts: no communicator");
  if (!rccl().all_gather) return fail(RJ_DEVICE_ERROR, "rj_multi_device_counts: librccl.so not found (set RJ_RCCL_LIBRARY)");
  return rj_multi_device_counts_via(m, d_text, n, own_begin, own_end, offset, rccl_gather, rccl_comm, rank, world, counts, hip_stream);
}

// ---------------------------------------------------------------------------------------------------------------
// MatchAll of ONE pattern over a text sharded across ranks, the (begin, end) pairs gathered on `root` in text order
// (BASELINE configs[3]: the complex regex over 50 GB on 8 GPUs; the reference's result IS the ordered
// match list, src/codegen.cc:36-86): the shard's run, the carry rounds of rj_multi_device_counts for one pattern
// (the left-most-longest selection crosses a cut when a match of the left neighbour reaches over it), then every
// rank's pairs -- global offsets -- straight to their place in the root's list.  Rank order = text order, and the
// carry has removed what overlapped a cut, so the concatenation is the reference's list.
int64_t rj_scan_gather_spans_via(rj_scan* s, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, int64_t offset,
                                 rj_allgather_fn allgather, rj_gatherv_fn gatherv, void* ctx, int rank, int world, int root, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!s || !allgather || !gatherv || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return fail(RJ_BAD_ARGUMENT, "bad argument");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  s->gathered = nullptr;
  s->gathered_count = 0;
  int64_t k = rj_scan_run(s, d_text, n, own_begin, own_end, 0, 0, 0, hip_stream);
  if (k < 0) return k;
  RJ_HIP(s->gx_rows.reserve(sizeof(int64_t) * 8 * static_cast<size_t>(world + 1)));
  if (!s->gx_host) RJ_HIP(hipHostMalloc(reinterpret_cast<void**>(&s->gx_host), sizeof(int64_t) * (8 + 8 * 1024)));
  if (world > 1024) return fail(RJ_BAD_ARGUMENT, "rj_scan_gather_spans: at most 1024 ranks");
  int64_t* mine = s->gx_rows.as<int64_t>();
  int64_t* all = mine + 8;
  int64_t* out = s->gx_host;
  bool converged = false;
  for (int round = 0; round <= world && !converged; round++) {
    BoundsParams bp{};
    bp.n_lists = 1;
    bp.spans[0] = s->result;
    bp.count[0] = s->result ? s->result_count : 0;
    launch_bounds_rows(bp, offset, round == 0, mine, st);
    if (allgather(ctx, mine, all, sizeof(int64_t) * 8, hip_stream) != 0)
      return fail(RJ_DEVICE_ERROR, "rj_scan_gather_spans: the all-gather failed (round %d)", round);
    launch_carry_decide(all, world, rank, 1, out, st);
    RJ_HIP(hipMemcpyAsync(out + 8, all, sizeof(int64_t) * 8 * static_cast<size_t>(world), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    if (out[4] == 0) {
      converged = true;
      break;
    }
    if (out[1]) {
      const int64_t cur = out[2], pe = out[3];
      const int have = (cur != 0 || pe != 0) ? 1 : 0;
      const int have_local = have && pe >= offset ? 1 : 0;
      const uint64_t lc = have_local && cur > offset ? static_cast<uint64_t>(cur - offset) : 0, lp = have_local ? static_cast<uint64_t>(pe - offset) : 0;
      k = rj_scan_run(s, d_text, n, own_begin, own_end, lc, lp, have_local, hip_stream);
      if (k < 0) return k;
      const int64_t used[3] = {cur, pe, have};
      RJ_HIP(hipMemcpyAsync(mine + 5, used, sizeof(used), hipMemcpyHostToDevice, st));
      RJ_HIP(hipStreamSynchronize(st));  // (`used` is on the stack)
    }
  }
  if (!converged) return fail(RJ_DEVICE_ERROR, "rj_scan_gather_spans: the carry exchange did not converge in %d rounds", world + 1);
  // every rank knows every rank's count (the rows of the last round)
  std::vector<uint64_t> offs(static_cast<size_t>(world)), bytes(static_cast<size_t>(world));
  uint64_t total = 0;
  for (int r = 0; r < world; r++) {
    const uint64_t c = static_cast<uint64_t>(out[8 + 8 * r]);
    offs[static_cast<size_t>(r)] = total * 16;
    bytes[static_cast<size_t>(r)] = c * 16;
    total += c;
  }
  const uint64_t my = s->result ? s->result_count : 0;
  if (my * 16 != bytes[static_cast<size_t>(rank)]) return fail(RJ_DEVICE_ERROR, "rj_scan_gather_spans: internal: the rows disagree with the shard's count");
  RJ_HIP(s->gx_send.reserve(std::max<uint64_t>(my, 1) * 16));
  if (my) launch_globalize_spans(s->result, my, offset, s->gx_send.as<uint64_t>(), st);
  if (rank == root) RJ_HIP(s->gx_out.reserve(std::max<uint64_t>(total, 1) * 16));
  if (gatherv(ctx, s->gx_send.p, my * 16, rank == root ? s->gx_out.p : nullptr, offs.data(), bytes.data(), root, hip_stream) != 0)
    return fail(RJ_DEVICE_ERROR, "rj_scan_gather_spans: the gather failed");
  RJ_HIP(hipStreamSynchronize(st));
  RJ_HIP(hipGetLastError());
  s->gathered = rank == root ? s->gx_out.as<uint64_t>() : nullptr;
  s->gathered_count = total;
  return static_cast<int64_t>(total);
}

int64_t rj_scan_gather_spans(rj_scan* s, const void* d_text, uint64_t n, uint64_t own_begin, uint64_t own_end, int64_t offset, void* rccl_comm, int rank,
                             int world, int root, void* hip_stream) {
  if (!rccl_comm) return fail(RJ_BAD_ARGUMENT, "rj_scan_gather_spans: no communicator");
  if (!rccl().all_gather) return fail(RJ_DEVICE_ERROR, "rj_scan_gather_spans: librccl.so not found (set RJ_RCCL_LIBRARY)");
  RcclGatherCtx c{rccl_comm, rank, world};
  return rj_scan_gather_spans_via(s, d_text, n, own_begin, own_end, offset, rccl_allgather_ctx, rccl_gatherv, &c, rank, world, root, hip_stream);
}

const uint64_t* rj_scan_gathered_spans(const rj_scan* s, uint64_t* count) {
  if (count) *count = s ? s->gathered_count : 0;
  return s ? s->gathered : nullptr;
}

int rj_multi_set_tail_stream(rj_multi* m, int on) {
  ErrnoGuard errno_guard;
  if (!m) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (m->pending.active) return fail(RJ_BAD_ARGUMENT, "rj_multi_set_tail_stream: a run is in flight");
  // (a tail stream of the lowest OR of the highest priority was measured: 0.196 / 0.188 ms per step against 0.134 with an
  // ordinary stream -- a priority queue is scheduled differently altogether, and worse for this)
  if (on && !m->tail_stream) RJ_HIP(hipStreamCreateWithFlags(&m->tail_stream, hipStreamNonBlocking));
  m->tails_own_stream = on != 0;
  return RJ_OK;
}

int rj_multi_set_timing(rj_multi* m, int on) {
  if (!m) return fail(RJ_BAD_ARGUMENT, "null argument");
  for (rj_scan* s : m->scans) s->timing = on != 0;
  return RJ_OK;
}

int rj_multi_set_counts_only(rj_multi* m, int on) {
  ErrnoGuard errno_guard;
  if (!m) return fail(RJ_BAD_ARGUMENT, "null argument");
  if (m->pending.active) return fail(RJ_BAD_ARGUMENT, "rj_multi_set_counts_only: a run is in flight");
  m->counts_only = on != 0;
  return (m->counts_only && counts_shape(m)) ? 1 : 0;
}

}  // extern "C"

// MatchAllCount of ONE pattern over a device text (rj_scan_count; rj_match_all(..., NULL) for host texts): a private
// rj_multi of the one pattern on the counts path, made at the first call; patterns without the shape (or a void run) are
// answered by the scan's own pipeline.
int64_t rejit_amd::scan_count(rj_scan* s, const uint8_t* d_text, uint64_t n, hipStream_t st) {
  if (s->counter_state == 0) {
    s->counter_state = -1;
    const rj_program* progs[1] = {s->prog};
    rj_multi* m = nullptr;
    if (n >= 16 && (fusable(s->prog) || plane_general_ok(s->prog)) && rj_multi_create(progs, 1, &m) == RJ_OK) {
      if (rj_multi_set_counts_only(m, 1) == 1) {
        (void)rj_multi_set_timing(m, s->timing ? 1 : 0);
        s->counter = m;
        s->counter_state = 1;
      } else {
        rj_multi_destroy(m);
      }
    } else if (n < 16) {
      s->counter_state = 0;   // (not decided on a text the kernel does not take)
    }
  }
  if (s->counter_state == 1 && n >= 16 && (reinterpret_cast<uintptr_t>(d_text) & 15u) == 0) {
    uint64_t count = 0;
    int how = rj_multi_run(s->counter, d_text, n, &count, st);
    if (how < 0) return how;
    rj_scan* inner = s->counter->scans[0];
    s->stats = inner->stats;
    s->result_count = inner->result_count;
    // (a void run was answered by the inner scan's pipeline: its list stays the inner scan's -- this call counts)
    s->result = nullptr;
    if (how != 3) s->stats.count_path = 0;
    return static_cast<int64_t>(count);
  }
  s->count_only_run = true;
  int rc = run_pipeline(s, d_text, n, 0, n + 1, 0, 0, 0, st);
  s->count_only_run = false;
  if (rc != RJ_OK) return rc;
  return static_cast<int64_t>(s->result_count);
}

extern "C" {

int64_t rj_scan_count(rj_scan* s, const void* d_text, uint64_t n, void* hip_stream) {
  ErrnoGuard errno_guard;
  if (!s || (!d_text && n)) return fail(RJ_BAD_ARGUMENT, "null argument");
  return scan_count(s, static_cast<const uint8_t*>(d_text), n, static_cast<hipStream_t>(hip_stream));
}

int rj_multi_set_mode(rj_multi* m, int mode) {
  if (!m || mode < 0 || mode > 4) return fail(RJ_BAD_ARGUMENT, "bad argument");
  m->want_fused_classify = mode == 4;
  m->mode = mode == 4 ? 0 : mode;
  return RJ_OK;
}

}  // extern "C"

// rejit_amd/csrc/linear.hip -- host side of the linear-time carry scan (carry_scan.h,
// carry_kernels.hip): the path a run takes when some candidate stayed alive for longer than the
// per-start verifier may walk (DevProgram::max_walk) -- unbounded repetitions over long runs, where
// walking every start on its own is quadratic.  The reference is linear on every input (one byte per
// iteration, merged threads: src/x64/codegen-x64.cc:535-640, 951-987, 1075-1097); so is this.
//
//   summaries of every sub-chunk from the first own start to the end of the text   (1 launch)
//   resolve: the true automaton state at every sub-chunk boundary                   (1 launch)
//   per segment of own starts (bounds the E / G slabs: 16 bytes per start):
//     emit E(s) -> local chains -> hop over the sub-chunks -> taken matches         (4 launches)
//     offsets_gather_check with regions = sub-chunks, then the ordinary selection tail (zero-length
//     rule, carry) exactly as after verify_in_regions
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "carry_scan.h"
#include "engine_internal.h"

namespace rejit_amd {

namespace {
constexpr uint64_t kCsSegment = 256ull << 20;  // own starts per segment: 4 GiB of E + G
}

bool linear_path_fits(const rj_program* rp) { return cs_state_words(rp->rev) != 0; }
bool linear_path_cheap(const rj_program* rp) { const int w = cs_state_words(rp->rev); return w != 0 && w <= 32; }

int run_linear(rj_scan* s, const uint8_t* d_text, uint64_t n, uint64_t sb, uint64_t se, uint64_t carry_cur,
               uint64_t carry_prev_end, int have_prev, hipStream_t st) {
  const rj_program* rp = s->prog;
  const DevProgram& R = rp->rev;
  if (!linear_path_fits(rp))
    return rj_fail(RJ_TOO_LARGE, "a match candidate runs longer than the parallel verifier walks and the automaton (%d positions) "
                                 "is wider than the linear-time path takes (8192)", R.n_pos);
  if (se > n + 1) se = n + 1;
  s->result_count = 0;
  s->stats.linear_path = 1;
  if (sb >= se) return RJ_OK;
  // sub-chunk = what one lane walks sequentially: large enough that the summaries stay small next to
  // the text, small enough that a modest text still fills the GPU
  uint64_t sub = 4096;
  static const uint64_t want_lanes = getenv("RJ_CS_LANES") ? static_cast<uint64_t>(atoll(getenv("RJ_CS_LANES"))) : 131072;  // measurement override
  // (131072 lanes = two waves per SIMD: `a.*b` over a 64 MiB line 8.8 -> 5.6 ms, `[acgt]+` 5.2 -> 4.3 against 32768;
  // 262144 and more: slower again -- sub-chunks of 256 bytes, the per-sub-chunk work of resolve and the chains grows)
  while (sub > 256 && (n - sb) / sub < want_lanes) sub >>= 1;
  while ((n - sb) / sub > (16u << 20)) sub <<= 1;
  const int np = std::max(R.n_pos, 1), W = R.n_words;
  // ... and the summaries (np * (8 + 4 W) bytes per sub-chunk: 10 KiB for a 256-position automaton) stay below half
  // the text: with 4-KiB sub-chunks a wide automaton took 2.5 x the text in HBM and a multi-GB text failed with an
  // out-of-memory error where the path is meant to serve every input
  while (sub < (64u << 10) && static_cast<uint64_t>(np) * (8 + 4 * static_cast<uint64_t>(W)) * 2 > sub) sub <<= 1;
  const uint64_t a0 = sb / sub * sub;
  const uint64_t m = n / sub - sb / sub + 1;  // sub-chunks from the first own start to the one that holds position n
  RJ_HIP(s->cs_vals.reserve((m + 1) * np * sizeof(uint64_t)));
  RJ_HIP(s->cs_mats.reserve(m * static_cast<uint64_t>(np) * W * sizeof(uint32_t)));
  RJ_HIP(s->cs_scratch.reserve(std::max<size_t>(cs_scratch_bytes(R, m), 16)));
  if (s->timing) RJ_HIP(hipEventRecord(s->ev[1], st));
  launch_cs_summarize(R, d_text, n, a0, sub, m, s->cs_vals.as<uint64_t>(), s->cs_mats.as<uint32_t>(), s->cs_scratch.as<uint8_t>(), st);
  RJ_HIP(s->cs_groups.reserve(cs_resolve_scratch_bytes(R, m)));
  launch_cs_resolve(R, m, s->cs_vals.as<uint64_t>(), s->cs_mats.as<uint32_t>(), s->cs_groups.as<uint8_t>(), st);

  const uint64_t seg_starts = std::max<uint64_t>(kCsSegment / sub, 1) * sub;
  uint64_t total = 0, longest = 0;
  bool first_segment = true;
  for (uint64_t lo = sb; lo < se;) {
    const uint64_t seg_a0 = lo / sub * sub;
    const uint64_t hi = std::min(se, seg_a0 + seg_starts);
    const uint64_t skip = (seg_a0 - a0) / sub;              // sub-chunks of the run before this segment
    const uint64_t m_own = (hi - 1) / sub - lo / sub + 1;   // sub-chunks that hold the segment's starts
    const uint64_t slots = m_own * sub;
    RJ_HIP(s->cs_e.reserve(slots * sizeof(uint64_t)));
    RJ_HIP(s->cs_g.reserve(slots * sizeof(uint64_t)));
    RJ_HIP(s->cs_entry.reserve(m_own * sizeof(uint64_t)));
    RJ_HIP(s->cs_counts.reserve(((m_own + 3) / 4 * 4 + 4) * sizeof(uint32_t)));
    RJ_HIP(hipMemsetAsync(s->cs_entry.p, 0xFF, m_own * sizeof(uint64_t), st));
    RJ_HIP(hipMemsetAsync(s->flag.p, 0, 16, st));
    RJ_HIP(hipMemsetAsync(s->counters.p, 0, kCntSize * sizeof(unsigned long long), st));
    launch_cs_emit(R, d_text, n, seg_a0, sub, m - skip, m_own, lo, hi, s->cs_vals.as<uint64_t>() + skip * np, s->cs_e.as<uint64_t>(),
                   s->cs_scratch.as<uint8_t>(), s->flag.as<unsigned long long>() + 1, st);
    launch_cs_chain(s->cs_e.as<uint64_t>(), s->cs_g.as<uint64_t>(), seg_a0, sub, m_own, lo, hi, std::max(carry_cur, lo),
                    s->cs_entry.as<uint64_t>(), s->cs_counts.as<uint32_t>(), s->flag.as<unsigned long long>(), st);
    unsigned long long* h_total = reinterpret_cast<unsigned long long*>(s->host_flag);  // [0] taken, [1] longest candidate
    RJ_HIP(hipMemcpyAsync(h_total, s->flag.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    const uint64_t taken = h_total[0];
    longest = std::max<uint64_t>(longest, h_total[1]);
    s->stats.n_hits += taken;
    // lay the taken matches out (regions = sub-chunks) and check them like any candidate list: empty
    // matches go through the selection for the zero-length rule (reference src/codegen.cc:65-73)
    if (taken > s->out_cap) {
      RJ_HIP(s->out.reserve(taken * 2 * sizeof(uint64_t)));
      s->out_cap = taken;
    }
    RJ_HIP(s->out.reserve(16));
    FinalizeParams fp{};
    fp.carry_cur = carry_cur;
    fp.carry_prev_end = carry_prev_end;
    fp.have_prev = have_prev;
    fp.detect_adjacent = 0;
    fp.expand = 1;
    s->host_counters[kCntUnordered] = 0;
    s->host_counters[kCntAdjacent] = 0;
    uint64_t *off_scratch = nullptr, *prev_scratch = nullptr;
    if (taken > m_own * 16) {
      RJ_HIP(s->hit_offsets.reserve((m_own + 1) * sizeof(uint64_t)));
      RJ_HIP(s->scan_a.reserve(m_own * sizeof(uint64_t)));
      off_scratch = s->hit_offsets.as<uint64_t>();
      prev_scratch = s->scan_a.as<uint64_t>();
    }
    launch_offsets_gather_check(s->cs_counts.as<uint32_t>(), s->cs_g.as<uint64_t>(), s->cs_e.as<uint64_t>(), static_cast<uint32_t>(m_own),
                                static_cast<uint32_t>(sub), carry_cur, s->out.as<uint64_t>(), s->out_cap,
                                s->counters.as<unsigned long long>(), s->host_counters, off_scratch, prev_scratch, st, carry_prev_end,
                                have_prev);
    RJ_HIP(hipStreamSynchronize(st));
    RJ_HIP(hipGetLastError());
    if (s->host_counters[kCntUnordered] != 0) {
      const uint64_t nc = s->host_counters[kCntCands];
      RJ_HIP(s->keys_out.reserve(std::max<uint64_t>(nc, 1) * sizeof(uint64_t)));
      RJ_HIP(s->vals_out.reserve(std::max<uint64_t>(nc, 1) * sizeof(uint64_t)));
      launch_split_pairs(s->out.as<uint64_t>(), s->counters.as<unsigned long long>() + kCntCands, nc, s->keys_out.as<uint64_t>(),
                         s->vals_out.as<uint64_t>(), st);
    }
    int rc = resolve_selection(s, fp, st);
    if (rc != RJ_OK) return rc;
    const uint64_t got = s->result_count;
    const bool last_segment = hi >= se;
    if (first_segment && last_segment) {
      total = got;
      s->result = s->out.as<uint64_t>();
    } else {
      if (got) {
        RJ_HIP(s->cs_acc.grow_keep((total + got) * 2 * sizeof(uint64_t), total * 2 * sizeof(uint64_t)));
        RJ_HIP(hipMemcpyAsync(s->cs_acc.as<uint64_t>() + 2 * total, s->out.p, got * 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
      }
      total += got;
      s->result = s->cs_acc.as<uint64_t>();
    }
    if (got && !last_segment) {
      uint64_t last[2];
      RJ_HIP(hipMemcpyAsync(last, s->out.as<uint64_t>() + 2 * (got - 1), sizeof(last), hipMemcpyDeviceToHost, st));
      RJ_HIP(hipStreamSynchronize(st));
      carry_cur = last[1] > last[0] ? last[1] : last[0] + 1;
      carry_prev_end = last[1];
      have_prev = 1;
    }
    first_segment = false;
    lo = hi;
  }
  if (s->timing) RJ_HIP(hipEventRecord(s->ev[2], st));
  RJ_HIP(hipStreamSynchronize(st));
  float ms = 0.f;
  if (s->timing) (void)hipEventElapsedTime(&ms, s->ev[1], s->ev[2]);
  s->stats.scan_ms += ms;
  s->result_count = total;
  // every candidate would have fitted the parallel verifier's walk: the next text starts there again
  s->linear_hint = longest >= rp->dev.max_walk;
  if (total == 0) s->result = s->out.as<uint64_t>();
  return RJ_OK;
}

}  // namespace rejit_amd

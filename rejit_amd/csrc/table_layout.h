// rejit_amd/csrc/table_layout.h -- the automaton tables as ONE contiguous blob of 32-bit words, the
// form the kernels read (device_program.h):
//   first [C][W]   last [C][W]   linear [W]   row_of [P]   rows [C][n_rows][W]   cls [256][W]
// Shared by the engine (upload to HBM; forward and reverse automaton) and by the CPU unit tests,
// which point a DevProgram at the host copy.
#ifndef REJIT_AMD_TABLE_LAYOUT_H_
#define REJIT_AMD_TABLE_LAYOUT_H_

#include <algorithm>
#include <cstdint>
#include <vector>

#include "device_program.h"
#include "lowering.h"

namespace rejit_amd {

struct TableBlob {
  std::vector<uint32_t> words;
  size_t off_first = 0, off_last = 0, off_linear = 0, off_rowof = 0, off_rows = 0, off_cls = 0;
  int W = 1, C = 1, R = 1, Pn = 1;
};

// `Tables`: Program (forward) or Program::Reverse -- same member names
template <class Tables>
TableBlob make_table_blob(const Tables& T, int n_pos, int n_words, bool has_assertions) {
  TableBlob b;
  b.W = n_words;
  b.C = has_assertions ? kNumCtx : 1;
  b.R = std::max(T.n_rows, 1);
  b.Pn = std::max(n_pos, 1);
  const size_t W = static_cast<size_t>(b.W), C = static_cast<size_t>(b.C), R = static_cast<size_t>(b.R);
  b.off_first = 0;
  b.off_last = b.off_first + C * W;
  b.off_linear = b.off_last + C * W;
  b.off_rowof = b.off_linear + W;
  b.off_rows = b.off_rowof + static_cast<size_t>(b.Pn);
  b.off_cls = b.off_rows + C * R * W;
  b.words.assign(b.off_cls + 256 * W, 0u);
  for (size_t c = 0; c < C; c++) {
    std::copy(T.first[c].begin(), T.first[c].end(), b.words.begin() + static_cast<long>(b.off_first + c * W));
    std::copy(T.last[c].begin(), T.last[c].end(), b.words.begin() + static_cast<long>(b.off_last + c * W));
    for (int r = 0; r < T.n_rows; r++)
      std::copy(T.rows[c].begin() + static_cast<long>(r) * b.W, T.rows[c].begin() + static_cast<long>(r + 1) * b.W,
                b.words.begin() + static_cast<long>(b.off_rows + (c * R + static_cast<size_t>(r)) * W));
  }
  std::copy(T.linear.begin(), T.linear.end(), b.words.begin() + static_cast<long>(b.off_linear));
  for (int i = 0; i < n_pos; i++) b.words[b.off_rowof + static_cast<size_t>(i)] = static_cast<uint32_t>(T.row_of[static_cast<size_t>(i)]);
  std::copy(T.cls.begin(), T.cls.end(), b.words.begin() + static_cast<long>(b.off_cls));
  return b;
}

// table pointers / sizes of D for a blob that lives at `base` (host or device memory)
inline void point_tables(DevProgram* D, const uint32_t* base, const TableBlob& b, int n_pos) {
  D->n_pos = n_pos;
  D->n_words = b.W;
  D->n_ctx = b.C;
  D->n_rows = b.R;
  D->table_words = static_cast<uint32_t>(b.words.size());
  D->first = base + b.off_first;
  D->last = base + b.off_last;
  D->linear = base + b.off_linear;
  D->row_of = reinterpret_cast<const int32_t*>(base + b.off_rowof);
  D->rows = base + b.off_rows;
  D->cls = base + b.off_cls;
}

inline uint32_t nullable_bits(const Program& P) {
  uint32_t bits = 0;
  for (int c = 0; c < kNumCtx; c++)
    if (P.nullable[P.has_assertions ? c : 0]) bits |= 1u << c;
  return bits;
}

}  // namespace rejit_amd
#endif

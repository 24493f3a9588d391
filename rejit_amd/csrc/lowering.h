// rejit_amd/csrc/lowering.h -- host-side lowering of an ERE pattern to launch parameters
// for the HIP kernels (kernels.hip).  Replaces the reference's front end + JIT:
//   src/parser.cc            -> parse()           (same accepted language, same quirks)
//   src/codegen.cc:91-324    -> build_graph()     (RegexpIndexer / RegexpLister)
//   src/codegen.cc:327-557   -> plan_fast_forward() (FF_finder, re-thought for a GPU)
//   src/x64/codegen-x64.cc   -> nothing is emitted; the automaton below is DATA that the
//                               pre-compiled kernels interpret.
//
// The device-side representation is a position (Glushkov-style) automaton, not the
// reference's state ring: every consumed byte of the pattern is one POSITION, the
// simulation state is a bit-vector over positions, one step is
//     S' = follow_ctx(S) & cls[byte]
// and zero-width assertions (^ $) are compiled away into four CONTEXT variants of the
// first / follow / last sets (context = is-start-of-line, is-end-of-line at the current
// text position).
#ifndef REJIT_AMD_LOWERING_H_
#define REJIT_AMD_LOWERING_H_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace rejit_amd {

// ----------------------------------------------------------------------------- AST
enum class NodeKind { Literal, Any, Class, StartOfLine, EndOfLine, Repeat, Concat, Alternate };

struct ByteSet {
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  void add(uint8_t c) { w[c >> 5] |= 1u << (c & 31); }
  bool has(uint8_t c) const { return (w[c >> 5] >> (c & 31)) & 1u; }
  ByteSet inverted() const {
    ByteSet r;
    for (int i = 0; i < 8; i++) r.w[i] = ~w[i];
    return r;
  }
  int count() const {
    int n = 0;
    for (int i = 0; i < 8; i++) n += __builtin_popcount(w[i]);
    return n;
  }
  bool operator==(const ByteSet& o) const {
    for (int i = 0; i < 8; i++)
      if (w[i] != o.w[i]) return false;
    return true;
  }
};

constexpr uint32_t kUnbounded = 0xFFFFFFFFu;  // kMaxUInt in the reference
constexpr unsigned kMaxLiteralNode = 64;      // kMaxNodeLength, src/regexp.h:107

struct Node {
  NodeKind kind;
  std::string bytes;       // Literal: 1..64 bytes (the reference's MultipleChar)
  ByteSet listed;          // Class: the characters listed between the brackets
  bool negated = false;    // Class: [^...] / \D / \S
  uint32_t min = 0, max = 0;                   // Repeat
  std::vector<std::unique_ptr<Node>> kids;     // Repeat (1), Concat, Alternate
  explicit Node(NodeKind k) : kind(k) {}
};

enum ParseStatus { kParseOk = 0, kParseError = -1 };

struct ParseResult {
  int status = kParseOk;
  std::string message;           // formatted like the reference's rejit_status_string
  std::unique_ptr<Node> root;
};

// Accepts exactly the language of the reference's Parser::ParseERE (src/parser.cc:40-195)
// including its quirks; patterns on which the reference has undefined behaviour or
// aborts are reported as parse errors.
ParseResult parse(const char* regexp);

// ----------------------------------------------------------------------------- NFA graph
struct ByteEdge {     // consumes bytes: a literal run or one class byte
  int src, dst;
  std::string bytes;  // non-empty: literal run
  ByteSet cls;        // else: one byte out of this set (negation already applied)
};
enum class ControlKind { Epsilon, StartOfLine, EndOfLine };
struct ControlEdge {
  int src, dst;
  ControlKind kind;
};
struct Graph {
  int n_states = 0;
  int entry = 0, exit = 1;
  std::vector<ByteEdge> byte_edges;
  std::vector<ControlEdge> control_edges;
};

// Thompson-style graph with the reference's wiring (RegexpLister::VisitRepetition,
// src/codegen.cc:175-324), including the two behaviours kept for bit-exact parity:
//  * a repetition with max == 1 gets a "repeat" epsilon exit->entry, so x? == x*;
//  * copies of a repetition's base made by DeepCopy lose Bracket's non_matching flag
//    (src/regexp.cc:104-110), so the 2nd.. copies of [^..] / \D / \S are positive.
// Returns false (message set) when the pattern expands beyond `max_states`.
bool build_graph(const Node& root, Graph* g, std::string* message, int max_states = 1 << 16);

// ----------------------------------------------------------------------------- program
// Context bits at a text position p (between byte p-1 and byte p):
//   bit0: start of line  (p == 0 or text[p-1] in {\n,\r})   codegen-x64.cc:686-708
//   bit1: end of line    (p == n or text[p]   in {\n,\r})
constexpr int kNumCtx = 4;

constexpr int kMaxPositions = 8192;

// candidate start s  <=>  for some window:
//   (load32(text + s + offset) & mask0) == value0  &&  (load32(text + s + offset + 4) & mask1) == value1
// A window covers `len` (1..8) pattern bytes; a zero mask byte is a wildcard (a class
// position, or a byte past `len`).
struct FFWindow {
  uint32_t offset;
  uint32_t len;
  uint32_t value0, mask0;
  uint32_t value1, mask1;
};

enum class ScanMode {
  Dense,     // every start position passes a first-byte / nullable test, then is simulated
  Windows,   // <= kMaxWindows fixed-offset 4-byte windows select candidate starts
};
constexpr int kMaxWindows = 8;

struct Program {
  int n_pos = 0;            // P
  int n_words = 0;          // ceil(P / 32), >= 1
  bool has_assertions = false;
  // [ctx][word]
  std::vector<uint32_t> first[kNumCtx];
  std::vector<uint32_t> last[kNumCtx];   // accepting right after consuming this position
  bool nullable[kNumCtx] = {false, false, false, false};
  // follow: positions whose follow set is exactly {i+1} in every context are "linear"
  std::vector<uint32_t> linear;          // [word]
  // non-linear positions get a row of n_words per context
  std::vector<int32_t> row_of;           // [pos] -> row index or -1
  int n_rows = 0;
  std::vector<uint32_t> rows[kNumCtx];   // [row][word]
  std::vector<uint32_t> cls;             // [256][word]
  ByteSet first_bytes;                   // bytes that can start a non-empty match (any ctx)
  bool any_nullable = false;
  uint64_t min_len = 0;
  uint64_t max_len = 0;                  // kUnboundedLen when the automaton has a cycle
  static constexpr uint64_t kUnboundedLen = ~0ull;
  // fast-forward plan
  ScanMode mode = ScanMode::Dense;
  std::vector<FFWindow> windows;
  // Floating windows (the reference's fast-forward element that is NOT at the start of the
  // match, e.g. `abcdefgh` in ([complex]|(regexp)){2,7}abcdefgh(...), src/codegen.cc:352-383):
  // every match contains one of the windows somewhere between float_min and float_max bytes
  // after its start.  A hit at w makes every s in [w - float_max, w - float_min] a candidate
  // start (the reference runs its NFA backwards from the hit instead, codegen-x64.cc:643-650).
  bool floating = false;
  uint32_t float_min = 0, float_max = 0;
  // Windows BEHIND an unbounded prefix (`.*regexp`, `[a-z]+abcdefgh`, `\d+regexp`): the reference picks
  // any literal as fast-forward element and runs its NFA BACKWARDS from the hit to the match start
  // (src/codegen.cc:352-383, src/x64/codegen-x64.cc:643-650).  Here: the literal edges of `windows` form a
  // cut of the NFA graph (every match crosses one) but their distance from the match start is not
  // bounded; a hit at w makes window k's positions `cut_positions[k]` (the automaton positions that
  // consume the literal's first byte) live at w, from which the REVERSE automaton (rev) finds the
  // left-most start and the forward automaton the end.
  bool behind = false;
  std::vector<std::vector<uint32_t>> cut_positions;   // [window][word]: forward positions, bitset
  std::string literal;                   // non-empty: the whole pattern is this literal
  // The NFA graph itself (reference state numbering semantics) and whether the pattern can
  // hit the reference's "Q8" ring-slot artefact: some state reachable
  // from the entry state through control edges can also be occupied by an older thread.
  Graph graph;
  bool q8_risk = false;
  // The REVERSE position automaton: the same positions renumbered back to front (j' = P-1-j),
  // first' = last, last' = first, follow transposed per context, so that the kernels' ordinary
  // step  S' = follow_ctx(S) & cls[byte]  applied while reading the text BACKWARDS walks a match
  // from its end to its start.  The context of a step is still the one of the boundary crossed.
  // Used by the linear-time carry scan (longest end of every start in one backward pass; the
  // reference gets linear time from its merged state ring, codegen-x64.cc:951-987) and by the
  // backward pass from a fast-forward hit to the match start (codegen-x64.cc:643-650).
  struct Reverse {
    std::vector<uint32_t> first[kNumCtx], last[kNumCtx], linear, rows[kNumCtx], cls;
    std::vector<int32_t> row_of;
    int n_rows = 0;
  } rev;
};

struct LowerResult {
  int status = 0;        // 0 ok, kParseError, or -2 "too large"
  std::string message;
  std::unique_ptr<Program> program;
};

LowerResult lower(const char* regexp);

}  // namespace rejit_amd
#endif

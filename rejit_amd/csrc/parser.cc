// rejit_amd/csrc/parser.cc -- ERE front end.
//
// Accepts the language of the reference's hand-written stack parser
// (src/parser.cc:40-195 ParseERE, :317-425 curly brackets, :428-464 brackets,
// :467-495 literal coalescing, :506-606 parentheses / alternation) and builds our own
// AST (lowering.h).  Behaviour that callers can observe and that is therefore kept:
//   * literal runs coalesce into nodes of at most 64 bytes; '*' and '{' detach the LAST
//     byte of the run and bind to it, '+' and '?' bind to the WHOLE run ("ab+" is (ab)+);
//   * escapes: \( \) \{ \} \[ \] \| \* \+ \^ \$ \\ are literals, \d \D \s \S classes,
//     \n \t control bytes, \xHH a byte whose hex LETTERS decode as 0..5 (reference bug,
//     parser.cc:23-37); anything else (\. \? ...) is "unexpected character";
//   * brackets: optional '^', optional leading '-', then look-ahead driven parsing of
//     singles and a-b ranges (signed byte compares); no escapes or classes inside;
//   * an unmatched ')' is a literal; x{m,n} on a literal run with m > 1 is rewritten to
//     x^m x{0,n-m} (which matters because the lowering treats {0,1} like the reference);
//   * errors are reported as ParserError with the reference's message layout.
// Where the reference has undefined behaviour, trips an assertion or aborts (empty
// branch, quantifier without operand, unterminated bracket, stray ']', missing ')',
// malformed \x) we return ParserError as well.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lowering.h"

namespace rejit_amd {
namespace {

struct Item {
  enum Tag { kNode, kOpenGroup, kBar } tag;
  std::unique_ptr<Node> node;
};

class Parser {
 public:
  explicit Parser(const char* re) : re_(re) {}

  ParseResult run() {
    ParseResult out;
    while (ok() && re_[at_] != '\0') step();
    if (ok()) finish();
    if (!ok()) {
      out.status = kParseError;
      out.message = message_;
      return out;
    }
    out.root = std::move(stack_.back().node);
    return out;
  }

 private:
  bool ok() const { return message_.empty(); }

  // Same layout as Parser::ParseError (parser.cc:652-665): index, pattern, caret, detail.
  void error_at(size_t index, const char* fmt, ...) {
    if (!ok()) return;
    char detail[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(detail, sizeof(detail), fmt, ap);
    va_end(ap);
    char head[64];
    snprintf(head, sizeof(head), "Error parsing at index %zu\n", index);
    message_ = std::string(head) + re_ + "\n" + std::string(index, ' ') + "^ \n" + detail;
    if (message_.empty()) message_ = "parse error";
  }

  static bool binds_to_last_byte(char c) { return c == '*' || c == '{'; }

  Node* top_node() {
    if (stack_.empty() || stack_.back().tag != Item::kNode) return nullptr;
    return stack_.back().node.get();
  }

  void push_node(std::unique_ptr<Node> n) {
    Item it;
    it.tag = Item::kNode;
    it.node = std::move(n);
    stack_.push_back(std::move(it));
  }

  void push_marker(Item::Tag tag) {
    Item it;
    it.tag = tag;
    stack_.push_back(std::move(it));
  }

  void push_byte(uint8_t c, bool may_join_run) {
    Node* t = top_node();
    if (may_join_run && t && t->kind == NodeKind::Literal && t->bytes.size() < kMaxLiteralNode) {
      t->bytes.push_back(static_cast<char>(c));
      return;
    }
    auto n = std::make_unique<Node>(NodeKind::Literal);
    n->bytes.push_back(static_cast<char>(c));
    push_node(std::move(n));
  }

  // A pattern byte taken literally: it joins the current run unless the NEXT pattern
  // byte is '*' or '{' (then the quantifier must see this byte alone).
  void push_pattern_byte(size_t index) {
    char c = re_[index];
    char next = c ? re_[index + 1] : '\0';
    push_byte(static_cast<uint8_t>(c), !binds_to_last_byte(next));
  }

  std::unique_ptr<Node> pop_operand(char op) {
    if (stack_.empty() || stack_.back().tag != Item::kNode) {
      error_at(at_, "nothing to repeat before '%c'\n", op);
      return nullptr;
    }
    std::unique_ptr<Node> n = std::move(stack_.back().node);
    stack_.pop_back();
    return n;
  }

  void quantify(char op, uint32_t lo, uint32_t hi) {
    auto operand = pop_operand(op);
    if (!operand) return;
    auto rep = std::make_unique<Node>(NodeKind::Repeat);
    rep->min = lo;
    rep->max = hi;
    rep->kids.push_back(std::move(operand));
    push_node(std::move(rep));
  }

  static std::unique_ptr<Node> make_class(bool negated) {
    auto n = std::make_unique<Node>(NodeKind::Class);
    n->negated = negated;
    return n;
  }

  static void add_range(Node* cls, char lo, char hi) {
    // MatchBracket compares with signed conditions (codegen-x64.cc:902-907)
    for (int v = 0; v < 256; v++) {
      signed char sc = static_cast<signed char>(static_cast<uint8_t>(v));
      if (sc >= static_cast<signed char>(lo) && sc <= static_cast<signed char>(hi)) {
        cls->listed.add(static_cast<uint8_t>(v));
      }
    }
  }

  int hex_digit(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'A' && c <= 'F') return c - 'A';  // sic: the reference maps A..F to 0..5
    if (c >= 'a' && c <= 'f') return c - 'a';
    return -1;
  }

  void step() {
    const char c = re_[at_];
    const char next = re_[at_ + 1];
    size_t used = 1;
    switch (c) {
      case '\\':
        used = escape(next);
        break;
      case '{':
        used = braces();
        break;
      case '.':
        push_node(std::make_unique<Node>(NodeKind::Any));
        break;
      case '*':
        quantify('*', 0, kUnbounded);
        break;
      case '+':
        quantify('+', 1, kUnbounded);
        break;
      case '?':
        quantify('?', 0, 1);
        break;
      case '^':
        push_node(std::make_unique<Node>(NodeKind::StartOfLine));
        break;
      case '$':
        push_node(std::make_unique<Node>(NodeKind::EndOfLine));
        break;
      case '(':
        push_marker(Item::kOpenGroup);
        break;
      case ')':
        close_group();
        break;
      case '|':
        fold_sequence();
        if (ok()) push_marker(Item::kBar);
        break;
      case '[':
        used = bracket();
        break;
      case ']':
        error_at(at_, "unexpected character ]\n");
        break;
      default:
        push_pattern_byte(at_);
    }
    at_ += used;
  }

  size_t escape(char next) {
    switch (next) {
      case '(': case ')': case '{': case '}': case '[': case ']':
      case '|': case '*': case '+': case '^': case '$': case '\\':
        push_pattern_byte(at_ + 1);
        return 2;
      case 'd': case 'D': {
        auto cls = make_class(next == 'D');
        add_range(cls.get(), '0', '9');
        push_node(std::move(cls));
        return 2;
      }
      case 's': case 'S': {
        auto cls = make_class(next == 'S');
        cls->listed.add(' ');
        cls->listed.add('\t');
        push_node(std::move(cls));
        return 2;
      }
      case 'n':
        push_byte('\n', true);
        return 2;
      case 't':
        push_byte('\t', true);
        return 2;
      case 'x': {
        char h = re_[at_ + 2];
        char l = h ? re_[at_ + 3] : '\0';
        int hv = hex_digit(h), lv = hex_digit(l);
        if (hv < 0 || lv < 0) {
          error_at(at_ + 2, "expected: <two hexadecimal digits>\n");
          return 2;
        }
        push_byte(static_cast<uint8_t>((hv << 4) | lv), true);
        return 4;
      }
      default:
        error_at(at_ + 1, "unexpected character %c\n", next);
        return 2;
    }
  }

  // strtoul like Parser::ParseIntegerAt (parser.cc:308-314), truncated to 32 bits.
  uint32_t integer_at(const char* pos, const char** end) {
    char* e = nullptr;
    unsigned long v = strtoul(pos, &e, 10);
    *end = e;
    if (e == pos) error_at(static_cast<size_t>(pos - re_), "expected: <base 10 integer>\n");
    return static_cast<uint32_t>(v);
  }

  bool expect(const char* pos, char want) {
    if (*pos == want) return true;
    error_at(static_cast<size_t>(pos - re_), "expected: %c\n", want);
    return false;
  }

  size_t braces() {
    const char* open = re_ + at_;
    const char* p = open + 1;
    const char* end = p;
    uint32_t lo = 0, hi = 0;
    if (*p == ',') {
      hi = integer_at(p + 1, &end);
      if (!ok() || !expect(end, '}')) return 1;
      p = end + 1;
    } else {
      lo = integer_at(p, &end);
      if (!ok()) return 1;
      p = end;
      if (*p == ',') {
        ++p;
        if (*p == '}') {
          hi = kUnbounded;
          ++p;
        } else {
          hi = integer_at(p, &end);
          if (!ok() || !expect(end, '}')) return 1;
          p = end + 1;
        }
      } else {
        if (!expect(p, '}')) return 1;
        ++p;
        hi = lo;
      }
    }
    if (lo > hi) {
      error_at(static_cast<size_t>(p - 1 - re_), "Invalid repetition bounds: %u > %u\n", lo, hi);
      return 1;
    }
    constexpr uint32_t kMaxCount = 100000;
    if ((lo != kUnbounded && lo > kMaxCount) || (hi != kUnbounded && hi > kMaxCount)) {
      error_at(at_, "repetition count too large\n");
      return 1;
    }
    auto operand = pop_operand('{');
    if (!operand) return 1;

    if (operand->kind == NodeKind::Literal && lo > 1) {
      // x{lo,hi} on a literal run: lo copies of the run packed into <=64-byte nodes,
      // then x{0,hi-lo} (parser.cc:372-418).
      const std::string base = operand->bytes;
      std::vector<std::unique_ptr<Node>> chunks;
      auto cur = std::make_unique<Node>(NodeKind::Literal);
      cur->bytes = base;
      for (uint32_t k = 1; k < lo; k++) {
        if (cur->bytes.size() + base.size() > kMaxLiteralNode) {
          chunks.push_back(std::move(cur));
          cur = std::make_unique<Node>(NodeKind::Literal);
        }
        cur->bytes += base;
      }
      chunks.push_back(std::move(cur));
      if (lo != hi) {
        auto rest = std::make_unique<Node>(NodeKind::Repeat);
        rest->min = 0;
        rest->max = (hi == kUnbounded) ? kUnbounded : hi - lo;
        auto again = std::make_unique<Node>(NodeKind::Literal);
        again->bytes = base;
        rest->kids.push_back(std::move(again));
        chunks.push_back(std::move(rest));
      }
      if (chunks.size() == 1) {
        push_node(std::move(chunks[0]));
      } else {
        auto seq = std::make_unique<Node>(NodeKind::Concat);
        seq->kids = std::move(chunks);
        push_node(std::move(seq));
      }
    } else {
      auto rep = std::make_unique<Node>(NodeKind::Repeat);
      rep->min = lo;
      rep->max = hi;
      rep->kids.push_back(std::move(operand));
      push_node(std::move(rep));
    }
    return static_cast<size_t>(p - open);
  }

  size_t bracket() {
    const char* open = re_ + at_;
    const char* p = open + 1;
    bool negated = false;
    if (*p == '^') {
      negated = true;
      ++p;
    }
    auto cls = make_class(negated);
    if (*p == '-') {
      cls->listed.add('-');
      ++p;
    }
    for (;;) {
      if (p[0] == '\0') break;
      if (p[0] == ']') {
        push_node(std::move(cls));
        return static_cast<size_t>(p + 1 - open);
      }
      if (p[1] == '\0') break;
      if (p[1] == ']') {
        cls->listed.add(static_cast<uint8_t>(p[0]));
        p += 1;
      } else if (p[2] == ']') {
        cls->listed.add(static_cast<uint8_t>(p[0]));
        cls->listed.add(static_cast<uint8_t>(p[1]));
        p += 2;
      } else if (p[1] == '-') {
        if (p[2] == '\0') break;
        add_range(cls.get(), p[0], p[2]);
        p += 3;
      } else {
        cls->listed.add(static_cast<uint8_t>(p[0]));
        p += 1;
      }
    }
    error_at(at_, "expected: ]\n");
    return 1;
  }

  // Everything above the nearest marker becomes one node (a Concat when there are
  // several).  An empty sequence -- "()", "a||b", "|a", "" -- is an error here; the
  // reference builds an empty Concatenation and later aborts on it.
  void fold_sequence() {
    size_t first = stack_.size();
    while (first > 0 && stack_[first - 1].tag == Item::kNode) --first;
    size_t count = stack_.size() - first;
    if (count == 0) {
      error_at(at_, "empty (sub-)expression\n");
      return;
    }
    if (count == 1) return;
    auto seq = std::make_unique<Node>(NodeKind::Concat);
    for (size_t i = first; i < stack_.size(); i++) seq->kids.push_back(std::move(stack_[i].node));
    stack_.resize(first);
    push_node(std::move(seq));
  }

  // Collapse "... ( r | r | r" (or the whole stack) into one node.
  void fold_alternatives() {
    fold_sequence();
    if (!ok()) return;
    size_t base = stack_.size();
    while (base > 0 && stack_[base - 1].tag != Item::kOpenGroup) --base;
    // [base, size) = r (| r)*
    size_t branches = 0;
    for (size_t i = base; i < stack_.size(); i++) branches += stack_[i].tag == Item::kNode;
    if (branches == 1 && stack_.size() - base == 1) return;  // trivial alternation
    auto alt = std::make_unique<Node>(NodeKind::Alternate);
    for (size_t i = base; i < stack_.size(); i++) {
      if (stack_[i].tag == Item::kNode) alt->kids.push_back(std::move(stack_[i].node));
    }
    stack_.resize(base);
    push_node(std::move(alt));
  }

  void close_group() {
    bool open = false;
    for (const Item& it : stack_) open |= it.tag == Item::kOpenGroup;
    if (!open) {
      push_pattern_byte(at_);  // a ')' without '(' is a literal (parser.cc:507-522)
      return;
    }
    fold_alternatives();
    if (!ok()) return;
    std::unique_ptr<Node> inner = std::move(stack_.back().node);
    stack_.pop_back();
    stack_.pop_back();  // the '('
    push_node(std::move(inner));
  }

  void finish() {
    fold_alternatives();
    if (!ok()) return;
    if (stack_.size() != 1) {
      size_t open = 0;
      for (const Item& it : stack_) open += it.tag == Item::kOpenGroup;
      error_at(at_, "Missing %zu right-parenthis ')'.\n", open);
    }
  }

  const char* re_;
  size_t at_ = 0;
  std::vector<Item> stack_;
  std::string message_;
};

}  // namespace

ParseResult parse(const char* regexp) { return Parser(regexp).run(); }

}  // namespace rejit_amd

/* oracle/rejit_oracle.c -- CPU restatement of coreperf/rejit's hot path in plain C.
 *
 * TEST INFRASTRUCTURE ONLY (see rejit_oracle.h).  Parity status: PINNED against the
 * reference's own test vectors and against the real reference built in place
 * (oracle/_ref), see tests/test_oracle.py.
 *
 * What is restated, and from where (all paths relative to the reference root):
 *   - the accepted language: Parser::ParseERE and helpers, src/parser.cc:40-195,
 *     317-649 (quantifier binding, bracket quirks, escapes, the \xHH hex-letter bug,
 *     the a{m,n} -> a^m a{0,n-m} parser optimisation, trivial-alternation removal);
 *   - the NFA: RegexpIndexer/RegexpLister, src/codegen.cc:91-324 (alternation
 *     branches share entry/exit, concatenation chains states, repetitions are
 *     expanded by copying and wired with epsilons -- INCLUDING the reference's
 *     behaviour that every repetition with max == 1 (x?, x{1}, x{0,1}) gets a
 *     "repeat" epsilon and therefore behaves like x* / x+, codegen.cc:266-312);
 *   - the matcher: the no-fast-forward kMatchAll / kMatchFull loops that
 *     Codegen::Generate emits, src/x64/codegen-x64.cc:99-207 (frame and dispatch),
 *     :535-640 (GenerateMatchDirection), :366-398 (control regexps), :401-466
 *     (CheckMatch), :469-522 (RegisterMatch), :653-677 (GenerateTransitions),
 *     :757-933 (literal / period / bracket tests), :951-987 (SetState: the
 *     left-most start wins), :1075-1097 (ClearStates), and the result sink
 *     MatchAllAppendFilter, src/codegen.cc:36-86.
 *
 * The data structure is the reference's: a ring of `times` x `states` slots that
 * hold the START OFFSET of the left-most thread that will be in that state `time`
 * bytes from now (the reference stores the start pointer, 0 = dead; we store the
 * offset, -1 = dead).  It is deliberately NOT the data structure the product uses.
 */
#include "rejit_oracle.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_NODE_LENGTH 64u          /* kMaxNodeLength, src/regexp.h:107 */
#define REP_INF 0xFFFFFFFFu          /* kMaxUInt, src/globals.h */
#define MAX_REP_EXPANSION 100000u    /* ours: refuse absurd {m,n} before allocating */
#define MAX_STATES (1 << 20)

/* ------------------------------------------------------------------------- */
/* Errors                                                                     */

static _Thread_local char g_err[256];

const char* ro_error(void) { return g_err; }

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

/* ------------------------------------------------------------------------- */
/* Regexp tree (src/regexp.h:27-68,213-471)                                   */

typedef enum {
  N_MC, N_PERIOD, N_BRACKET, N_SOL, N_EOL, N_REP, N_CONCAT, N_ALT,
  N_LPAREN, N_BAR /* parser stack markers */
} ntype;

typedef struct node {
  ntype type;
  uint8_t chars[MAX_NODE_LENGTH]; /* N_MC */
  unsigned nchars;
  uint8_t set[32];                /* N_BRACKET: listed characters (before negation) */
  int negated;                    /* N_BRACKET: Bracket::non_matching */
  struct node* sub;               /* N_REP */
  uint32_t min, max;
  struct node** subs;             /* N_CONCAT / N_ALT */
  int nsubs, capsubs;
  struct node* next_alloc;
} node;

typedef struct {
  const char* re;
  size_t index;
  node** stack;
  int sp, cap;
  node* allocs;
  int status;
} parser;

static node* new_node(parser* P, ntype t) {
  node* n = (node*)calloc(1, sizeof(node));
  n->type = t;
  n->next_alloc = P->allocs;
  P->allocs = n;
  return n;
}

static void node_append(node* parent, node* child) {
  if (parent->nsubs == parent->capsubs) {
    parent->capsubs = parent->capsubs ? parent->capsubs * 2 : 4;
    parent->subs = (node**)realloc(parent->subs, sizeof(node*) * (size_t)parent->capsubs);
  }
  parent->subs[parent->nsubs++] = child;
}

static void free_nodes(node* n) {
  while (n) {
    node* next = n->next_alloc;
    free(n->subs);
    free(n);
    n = next;
  }
}

static int is_marker(const node* n) { return n->type == N_LPAREN || n->type == N_BAR; }

static void push(parser* P, node* n) {
  if (P->sp == P->cap) {
    P->cap = P->cap ? P->cap * 2 : 16;
    P->stack = (node**)realloc(P->stack, sizeof(node*) * (size_t)P->cap);
  }
  P->stack[P->sp++] = n;
}

static node* tos(parser* P) { return P->sp ? P->stack[P->sp - 1] : NULL; }

/* PopRegexp for a quantifier (parser.cc:609-631, :425).  The reference pops whatever
 * is on the stack -- nothing at all (undefined behaviour) or a marker; we reject. */
static node* pop_operand(parser* P, char op) {
  if (P->sp == 0 || is_marker(P->stack[P->sp - 1])) {
    P->status = fail(RO_REJECTED, "quantifier '%c' at index %zu has nothing to repeat "
                     "(undefined behaviour in the reference)", op, P->index);
    return NULL;
  }
  return P->stack[--P->sp];
}

/* Parser::PushChar(char, bool), parser.cc:467-485 */
static void push_char(parser* P, uint8_t c, int append_to_mc_tos) {
  node* t = tos(P);
  if (append_to_mc_tos && t && t->type == N_MC && t->nchars < MAX_NODE_LENGTH) {
    t->chars[t->nchars++] = c;
    return;
  }
  node* mc = new_node(P, N_MC);
  mc->chars[0] = c;
  mc->nchars = 1;
  push(P, mc);
}

/* IsRetroactiveChar, parser.h:100-104: '*' and '{' bind to the LAST char only;
 * '+' and '?' are not listed, so they bind to the whole preceding literal run. */
static int is_retroactive(char c) { return c == '*' || c == '{'; }

/* Parser::PushChar(const char*), parser.cc:488-495 */
static void push_char_at(parser* P, size_t idx) {
  char c = P->re[idx];
  char lookahead = c ? P->re[idx + 1] : '\0';
  push_char(P, (uint8_t)c, !is_retroactive(lookahead));
}

static void set_add(node* b, uint8_t c) { b->set[c >> 3] |= (uint8_t)(1u << (c & 7)); }

/* Bracket::AddCharRange + MatchBracket's SIGNED byte compares,
 * src/x64/codegen-x64.cc:878-909 (greater_equal / less_equal on cmpb). */
static void set_add_range(node* b, char lo, char hi) {
  for (int v = 0; v < 256; v++) {
    signed char sc = (signed char)(uint8_t)v;
    if (sc >= (signed char)lo && sc <= (signed char)hi) set_add(b, (uint8_t)v);
  }
}

static void set_negate(node* b) { b->negated = 1; }

/* Parser::DoConcatenation, parser.cc:538-566 */
static void do_concatenation(parser* P) {
  if (P->sp == 0) {
    P->status = fail(RO_REJECTED, "empty (sub-)expression at index %zu "
                     "(undefined behaviour in the reference)", P->index);
    return;
  }
  int it = P->sp - 1;
  while (it > 0 && !is_marker(P->stack[it])) it--;
  int first = is_marker(P->stack[it]) ? it + 1 : it;
  if (first == P->sp) {
    /* The reference builds a Concatenation with no sub-expression here and later
     * dies with std::out_of_range when it is indexed. */
    P->status = fail(RO_REJECTED, "empty (sub-)expression at index %zu "
                     "(the reference aborts on it)", P->index);
    return;
  }
  if (first + 1 != P->sp) {
    node* concat = new_node(P, N_CONCAT);
    for (int i = first; i < P->sp; i++) node_append(concat, P->stack[i]);
    P->sp = first;
    push(P, concat);
  }
}

/* Parser::DoAlternation, parser.cc:569-606 (FLAG_use_parser_opt is on by default) */
static void do_alternation(parser* P) {
  do_concatenation(P);
  if (P->status != RO_OK) return;
  int last = P->sp - 1;
  if (P->stack[last]->type == N_LPAREN ||
      (last - 1 >= 0 && P->stack[last - 1]->type == N_LPAREN) || last == 0) {
    return; /* trivial alternation of zero or one element */
  }
  node* alt = new_node(P, N_ALT);
  int it;
  /* The reference collects the branches from the top of the stack downwards, i.e. in
   * REVERSE order (parser.cc:597-601); order is irrelevant for left-most longest. */
  for (it = last; it >= 0 && P->stack[it]->type != N_LPAREN; it--) {
    if (!is_marker(P->stack[it])) node_append(alt, P->stack[it]);
  }
  int first = it < 0 ? 0 : it + 1;
  P->sp = first;
  push(P, alt);
}

/* Parser::DoRightParenthesis, parser.cc:506-528 */
static void do_right_parenthesis(parser* P) {
  int found = 0;
  for (int i = P->sp - 1; i >= 0; i--) {
    if (P->stack[i]->type == N_LPAREN) { found = 1; break; }
  }
  if (!found) {
    push_char_at(P, P->index); /* an unmatched ')' is a literal */
    return;
  }
  do_alternation(P);
  if (P->status != RO_OK) return;
  node* inner = P->stack[--P->sp];
  /* tos is now the left parenthesis */
  --P->sp;
  push(P, inner);
}

/* Parser::ParseIntegerAt, parser.cc:308-314: strtoul, truncated to 32 bits. */
static uint32_t parse_integer_at(parser* P, const char* pos, const char** end) {
  char* e;
  uint32_t v = (uint32_t)strtoul(pos, &e, 10);
  *end = e;
  if (pos == e) {
    P->status = fail(RO_PARSE_ERROR, "Error parsing at index %zu: expected: <base 10 integer>",
                     (size_t)(pos - P->re));
  }
  return v;
}

static int expect_char(parser* P, const char* c, char expected) {
  if (*c != expected) {
    P->status = fail(RO_PARSE_ERROR, "Error parsing at index %zu: expected: %c",
                     (size_t)(c - P->re), expected);
    return 0;
  }
  return 1;
}

/* Parser::ParseCurlyBrackets, parser.cc:317-425.  Returns the number of regexp
 * characters consumed. */
static size_t parse_curly_brackets(parser* P, const char* lcb) {
  uint32_t min, max;
  const char* c = lcb + 1;
  const char* end;

  if (*c == ',') {
    min = 0;
    c++;
    max = parse_integer_at(P, c, &end);
    if (P->status != RO_OK) return 0;
    if (!expect_char(P, end, '}')) return 0;
    c = end + 1;
  } else {
    min = parse_integer_at(P, c, &end);
    if (P->status != RO_OK) return 0;
    c = end;
    if (*c == ',') {
      c++;
      if (*c == '}') {
        max = REP_INF;
        c++;
      } else {
        max = parse_integer_at(P, c, &end);
        if (P->status != RO_OK) return 0;
        c = end;
        if (!expect_char(P, c, '}')) return 0;
        c++;
      }
    } else {
      if (!expect_char(P, c, '}')) return 0;
      c++;
      max = min;
    }
  }

  if (min > max) {
    P->status = fail(RO_PARSE_ERROR, "Error parsing at index %zu: Invalid repetition bounds: %u > %u",
                     (size_t)(c - 1 - P->re), min, max);
    return 0;
  }
  if ((min != REP_INF && min > MAX_REP_EXPANSION) || (max != REP_INF && max > MAX_REP_EXPANSION)) {
    P->status = fail(RO_REJECTED, "repetition bound too large for the oracle");
    return 0;
  }

  node* re = pop_operand(P, '{');
  if (!re) return 0;

  if (re->type == N_MC && min > 1) {
    /* Parser-level optimisation a{min,max} -> a^min a{0,max-min}, parser.cc:372-418. */
    node* mc = re;
    node* result = NULL;
    node* mc_start;
    if (min == max) {
      mc_start = mc;
      result = mc_start;
    } else {
      mc_start = new_node(P, N_MC);
      memcpy(mc_start->chars, mc->chars, mc->nchars);
      mc_start->nchars = mc->nchars;
    }
    unsigned base_len = mc->nchars;
    uint8_t base[MAX_NODE_LENGTH];
    memcpy(base, mc->chars, base_len);
    node* concat = NULL;
    if ((uint64_t)base_len * min > MAX_NODE_LENGTH || min != max) {
      concat = new_node(P, N_CONCAT);
      result = concat;
    }
    uint32_t repeat_base = 1;
    while (repeat_base++ < min) {
      if (mc_start->nchars + base_len > MAX_NODE_LENGTH) {
        node_append(concat, mc_start);
        mc_start = new_node(P, N_MC);
      }
      memcpy(mc_start->chars + mc_start->nchars, base, base_len);
      mc_start->nchars += base_len;
    }
    if (concat) node_append(concat, mc_start);
    if (min != max) {
      node* rep = new_node(P, N_REP);
      /* `mc` may have been extended in place when min == max; here min != max so the
       * base literal is untouched. */
      rep->sub = mc;
      rep->min = 0;
      rep->max = (max == REP_INF) ? REP_INF : max - min;
      node_append(concat, rep);
    }
    push(P, result);
  } else {
    node* rep = new_node(P, N_REP);
    rep->sub = re;
    rep->min = min;
    rep->max = max;
    push(P, rep);
  }
  return (size_t)(c - lcb);
}

/* Parser::ParseBrackets, parser.cc:428-464.  The reference never checks for the end
 * of the pattern; an unterminated bracket reads out of bounds.  We reject those. */
static size_t parse_brackets(parser* P, const char* lb) {
  const char* c = lb + 1;
  node* b = new_node(P, N_BRACKET);
  int negated = 0;
  if (*c == '^') { negated = 1; c++; }
  if (*c == '-') { set_add(b, '-'); c++; }
  for (;;) {
    if (*c == '\0') goto unterminated;
    if (*c == ']') { c++; break; }
    if (c[1] == '\0') goto unterminated;
    if (c[1] == ']') {
      set_add(b, (uint8_t)*c);
      c++;
    } else if (c[2] == ']') {
      set_add(b, (uint8_t)c[0]);
      set_add(b, (uint8_t)c[1]);
      c += 2;
    } else if (c[1] == '-') {
      if (c[2] == '\0') goto unterminated;
      set_add_range(b, c[0], c[2]);
      c += 3;
    } else {
      set_add(b, (uint8_t)*c);
      c++;
    }
  }
  if (negated) set_negate(b);
  push(P, b);
  return (size_t)(c - lb);
unterminated:
  P->status = fail(RO_REJECTED, "unterminated bracket expression starting at index %zu "
                   "(out-of-bounds read in the reference)", (size_t)(lb - P->re));
  return 0;
}

/* hex_code_from_char, parser.cc:23-37 -- note the bug: letters map to 0..5, not
 * 10..15, so \x4a is byte 0x40. */
static int hex_code_from_char(char c, int* ok) {
  if ('0' <= c && c <= '9') return c - '0';
  if ('A' <= c && c <= 'F') return c - 'A';
  if ('a' <= c && c <= 'f') return c - 'a';
  *ok = 0; /* UNREACHABLE() -> rejit_fatal -> abort in the reference */
  return 0;
}

static node* wrap_repetition(parser* P, char op, uint32_t min, uint32_t max) {
  node* sub = pop_operand(P, op);
  if (!sub) return NULL;
  node* rep = new_node(P, N_REP);
  rep->sub = sub;
  rep->min = min;
  rep->max = max;
  push(P, rep);
  return rep;
}

/* Parser::ParseERE, parser.cc:40-195 */
static int parse_ere(parser* P) {
  char c;
  while ((c = P->re[P->index])) {
    char lookahead = P->re[P->index + 1];
    size_t advance = 1;
    switch (c) {
      case '\\': {
        advance = 2;
        switch (lookahead) {
          case '(': case ')': case '{': case '}': case '[': case ']':
          case '|': case '*': case '+': case '^': case '$': case '\\':
            push_char_at(P, P->index + 1);
            break;
          case 'd': case 'D': {
            node* b = new_node(P, N_BRACKET);
            set_add_range(b, '0', '9');
            if (lookahead == 'D') set_negate(b);
            push(P, b);
            break;
          }
          case 'n': push_char(P, '\n', 1); break;
          case 's': case 'S': {
            node* b = new_node(P, N_BRACKET);
            set_add(b, ' ');
            set_add(b, '\t');
            if (lookahead == 'S') set_negate(b);
            push(P, b);
            break;
          }
          case 't': push_char(P, '\t', 1); break;
          case 'x': {
            advance = 4;
            char h = P->re[P->index + 2];
            char l = h ? P->re[P->index + 3] : '\0';
            int ok = 1;
            int hv = hex_code_from_char(h, &ok);
            int lv = hex_code_from_char(l, &ok);
            if (!ok) {
              return P->status = fail(RO_REJECTED, "bad \\x escape at index %zu "
                                      "(the reference aborts)", P->index);
            }
            push_char(P, (uint8_t)((hv << 4) | lv), 1);
            break;
          }
          default:
            return P->status = fail(RO_PARSE_ERROR, "Error parsing at index %zu: unexpected character %c",
                                    P->index + 1, lookahead);
        }
        break;
      }
      case '{': advance = parse_curly_brackets(P, P->re + P->index); break;
      case '.': push(P, new_node(P, N_PERIOD)); break;
      case '*': wrap_repetition(P, '*', 0, REP_INF); break;
      case '+': wrap_repetition(P, '+', 1, REP_INF); break;
      case '?': wrap_repetition(P, '?', 0, 1); break;
      case '^': push(P, new_node(P, N_SOL)); break;
      case '$': push(P, new_node(P, N_EOL)); break;
      case '(': push(P, new_node(P, N_LPAREN)); break;
      case ')': do_right_parenthesis(P); break;
      case '|':
        do_concatenation(P);
        if (P->status == RO_OK) push(P, new_node(P, N_BAR));
        break;
      case '[': advance = parse_brackets(P, P->re + P->index); break;
      case ']':
        return P->status = fail(RO_REJECTED, "stray ']' at index %zu (UNREACHABLE in the reference)",
                                P->index);
      default: push_char_at(P, P->index);
    }
    if (P->status != RO_OK) return P->status;
    P->index += advance;
  }
  /* Parser::DoFinish, parser.cc:634-649 */
  do_alternation(P);
  if (P->status != RO_OK) return P->status;
  if (P->sp != 1) {
    return P->status = fail(RO_REJECTED, "Error parsing at index %zu: missing right-parenthesis "
                            "(the reference then fails ALWAYS_ASSERT(stack_.size() == 1))", P->index);
  }
  return RO_OK;
}

/* ------------------------------------------------------------------------- */
/* NFA (RegexpIndexer / RegexpLister, src/codegen.cc:91-324)                   */

typedef enum { E_MC, E_PERIOD, E_BRACKET } mtype;
typedef enum { C_EPS, C_SOL, C_EOL } ctype;

typedef struct {
  mtype type;
  int src, dst;
  const node* n; /* chars / set live in the tree */
  int negated;   /* E_BRACKET: effective non_matching flag of THIS copy */
} medge;

typedef struct {
  ctype type;
  int src, dst;
} cedge;

struct ro_regex {
  node* allocs;
  int n_states;
  int entry, exit;
  medge* m; int nm, capm;
  cedge* c; int nc, capc;
  unsigned max_mc;
  int status;
};

static int new_state(ro_regex* R) {
  if (R->n_states >= MAX_STATES) { R->status = RO_REJECTED; return 0; }
  return R->n_states++;
}

static void add_medge(ro_regex* R, mtype t, int src, int dst, const node* n, int negated) {
  if (R->nm == R->capm) {
    R->capm = R->capm ? R->capm * 2 : 16;
    R->m = (medge*)realloc(R->m, sizeof(medge) * (size_t)R->capm);
  }
  R->m[R->nm].type = t; R->m[R->nm].src = src; R->m[R->nm].dst = dst; R->m[R->nm].n = n;
  R->m[R->nm].negated = negated;
  R->nm++;
}

static void add_cedge(ro_regex* R, ctype t, int src, int dst) {
  if (R->nc == R->capc) {
    R->capc = R->capc ? R->capc * 2 : 16;
    R->c = (cedge*)realloc(R->c, sizeof(cedge) * (size_t)R->capc);
  }
  R->c[R->nc].type = t; R->c[R->nc].src = src; R->c[R->nc].dst = dst;
  R->nc++;
}

/* `copied` is set for every sub-tree that the reference obtains through DeepCopy()
 * (second and later copies of a repetition's base, codegen.cc:228-241).
 * Bracket::DeepCopy (src/regexp.cc:104-110) copies the characters and ranges but NOT
 * flags_, so a copied [^...] / \D / \S silently becomes the POSITIVE class.  Kept for
 * parity: e.g. \S{2,3} is \S \s \s in the reference. */
static void build(ro_regex* R, const node* n, int entry, int exit, int copied);

/* n_rep copies of `base` chained entry -> ... -> exit; returns the entry and exit
 * states of copy i via ent[]/ext[] (the Lister deep-copies the base and the Indexer
 * numbers each copy, codegen.cc:223-258). */
static void build_copies(ro_regex* R, const node* base, unsigned n_rep, int entry, int exit,
                         int* ent, int* ext, int copied) {
  int cur = entry;
  for (unsigned i = 0; i < n_rep && R->status == RO_OK; i++) {
    int nxt = (i + 1 == n_rep) ? exit : new_state(R);
    ent[i] = cur;
    ext[i] = nxt;
    build(R, base, cur, nxt, copied || i > 0);
    cur = nxt;
  }
}

/* RegexpLister::VisitRepetition, codegen.cc:175-324 */
static void build_repetition(ro_regex* R, const node* rep, int entry, int exit, int copied) {
  const node* base = rep->sub;
  uint32_t min_rep = rep->min, max_rep = rep->max;
  int is_limited = max_rep != REP_INF;

  if (min_rep == 0 && max_rep == 0) {
    add_cedge(R, C_EPS, entry, exit);
    return;
  }
  int needs_concatenation = min_rep > 1 || (max_rep > 1 && is_limited);
  unsigned n_rep = needs_concatenation ? (is_limited ? max_rep : min_rep) : 1;

  int inside_entry = entry;
  int inside_exit = exit;
  if (!is_limited) {
    inside_exit = new_state(R);            /* extra state at the end */
    if (min_rep <= 1) inside_entry = new_state(R); /* extra state at the beginning */
  }
  int* ent = (int*)malloc(sizeof(int) * n_rep);
  int* ext = (int*)malloc(sizeof(int) * n_rep);
  build_copies(R, base, n_rep, inside_entry, inside_exit, ent, ext, copied);
  if (R->status != RO_OK) { free(ent); free(ext); return; }

  if (min_rep == 0) add_cedge(R, C_EPS, entry, exit); /* bypass epsilon */

  if (is_limited && max_rep > 1) {
    /* exit epsilons after copy max(1,min) .. max-1 */
    unsigned mn = min_rep > 1 ? min_rep : 1;
    for (unsigned i = mn - 1; i + 1 < n_rep; i++) add_cedge(R, C_EPS, ext[i], exit);
  } else {
    /* NOTE: this branch is also taken for limited repetitions with max == 1
     * (x?, x{1}, x{0,1}); the "repeat epsilon" below then loops exit -> entry, which
     * is why those behave like x* / x+ in the reference. */
    if (min_rep <= 1) add_cedge(R, C_EPS, entry, inside_entry);   /* entry epsilon */
    add_cedge(R, C_EPS, inside_exit, exit);                       /* exit epsilon */
    add_cedge(R, C_EPS, ext[n_rep - 1], ent[n_rep - 1]);          /* repeat epsilon */
  }
  free(ent);
  free(ext);
}

static void build(ro_regex* R, const node* n, int entry, int exit, int copied) {
  if (R->status != RO_OK) return;
  switch (n->type) {
    case N_MC:
      if (n->nchars > R->max_mc) R->max_mc = n->nchars;
      add_medge(R, E_MC, entry, exit, n, 0);
      break;
    case N_PERIOD:
    case N_BRACKET:
      /* UpdateRegexpMaxLength: one-byte nodes count too (parser.cc:78,91,457,499) */
      if (R->max_mc < 1) R->max_mc = 1;
      add_medge(R, n->type == N_PERIOD ? E_PERIOD : E_BRACKET, entry, exit, n, n->negated && !copied);
      break;
    case N_SOL: add_cedge(R, C_SOL, entry, exit); break;
    case N_EOL: add_cedge(R, C_EOL, entry, exit); break;
    case N_REP: build_repetition(R, n, entry, exit, copied); break;
    case N_CONCAT: {
      /* RegexpIndexer::VisitConcatenation, codegen.cc:128-140: chained states */
      int cur = entry;
      for (int i = 0; i < n->nsubs; i++) {
        int nxt = (i + 1 == n->nsubs) ? exit : new_state(R);
        build(R, n->subs[i], cur, nxt, copied);
        cur = nxt;
      }
      break;
    }
    case N_ALT:
      /* RegexpIndexer::VisitAlternation, codegen.cc:112-125: shared entry / exit */
      for (int i = 0; i < n->nsubs; i++) build(R, n->subs[i], entry, exit, copied);
      break;
    default:
      R->status = RO_REJECTED;
  }
}

int ro_compile(const char* regexp, ro_regex** out) {
  g_err[0] = '\0';
  *out = NULL;
  parser P;
  memset(&P, 0, sizeof(P));
  P.re = regexp;
  int st = parse_ere(&P);
  if (st != RO_OK) {
    free(P.stack);
    free_nodes(P.allocs);
    return st;
  }
  ro_regex* R = (ro_regex*)calloc(1, sizeof(ro_regex));
  R->allocs = P.allocs;
  R->entry = new_state(R);      /* rinfo_->set_entry_state(0), codegen.cc:93 */
  R->exit = new_state(R);
  build(R, P.stack[0], R->entry, R->exit, 0);
  free(P.stack);
  if (R->status != RO_OK) {
    ro_free(R);
    return fail(RO_REJECTED, "pattern too large for the oracle");
  }
  *out = R;
  return RO_OK;
}

void ro_free(ro_regex* R) {
  if (!R) return;
  free_nodes(R->allocs);
  free(R->m);
  free(R->c);
  free(R);
}

int ro_n_states(const ro_regex* R) { return R->n_states; }
int ro_n_edges(const ro_regex* R) { return R->nm + R->nc; }

/* ------------------------------------------------------------------------- */
/* The matcher                                                                */

#define DEAD (-1)

typedef struct {
  const ro_regex* R;
  const uint8_t* text;
  size_t n;
  int times;          /* 1 + min(max literal length, 64), codegen.cc:615 */
  int64_t* ring;      /* times x n_states */
  int base;           /* ring index of time 0 */
} sim;

static int64_t* slot(sim* S, int time, int state) {
  int t = S->base + time;
  if (t >= S->times) t -= S->times;
  return &S->ring[(size_t)t * (size_t)S->R->n_states + (size_t)state];
}

/* Codegen::SetState, codegen-x64.cc:951-987: the target takes the source's start iff
 * (src-1) <u (tgt-1): the left-most start wins and a dead source never wins. */
static int set_state(sim* S, int time, int target, int64_t src_start) {
  if (src_start == DEAD) return 0;
  int64_t* t = slot(S, time, target);
  if (*t == DEAD || src_start < *t) { *t = src_start; return 1; }
  return 0;
}

static int is_line_break(uint8_t c) { return c == '\n' || c == '\r'; }

/* Codegen::HandleControlRegexps, codegen-x64.cc:366-398 + VisitEpsilon :680-682,
 * MatchStartOrEndOfLine :686-708.  The reference runs the topologically sorted list
 * once, or an unsortable list n_ctrl times; both reach the fix point computed here. */
static void handle_control_regexps(sim* S, size_t p) {
  const ro_regex* R = S->R;
  int changed = 1;
  while (changed) {
    changed = 0;
    for (int i = 0; i < R->nc; i++) {
      const cedge* e = &R->c[i];
      int64_t v = *slot(S, 0, e->src);
      if (v == DEAD) continue;
      int ok = 1;
      if (e->type == C_SOL) ok = (p == 0) || is_line_break(S->text[p - 1]);
      else if (e->type == C_EOL) ok = (p == S->n) || is_line_break(S->text[p]);
      if (ok) changed |= set_state(S, 0, e->dst, v);
    }
  }
}

/* Codegen::GenerateTransitions, codegen-x64.cc:653-677, and the matching visitors
 * :757-848 (literal; CheckEnoughStringLength :735-749), :851-873 (period excludes
 * \n and \r), :878-933 (bracket). */
static void generate_transitions(sim* S, size_t p) {
  const ro_regex* R = S->R;
  for (int i = 0; i < R->nm; i++) {
    const medge* e = &R->m[i];
    int64_t v = *slot(S, 0, e->src);
    if (v == DEAD) continue;
    switch (e->type) {
      case E_MC: {
        unsigned L = e->n->nchars;
        if (p + L <= S->n && memcmp(S->text + p, e->n->chars, L) == 0) set_state(S, (int)L, e->dst, v);
        break;
      }
      case E_PERIOD:
        if (!is_line_break(S->text[p])) set_state(S, 1, e->dst, v);
        break;
      case E_BRACKET: {
        uint8_t c = S->text[p];
        int in = (e->n->set[c >> 3] >> (c & 7)) & 1;
        if (in != e->negated) set_state(S, 1, e->dst, v);
        break;
      }
    }
  }
}

static void advance_time(sim* S) {
  /* ClearTime(0) then rotate the ring, codegen-x64.cc:565-578 */
  int64_t* t0 = slot(S, 0, 0);
  for (int s = 0; s < S->R->n_states; s++) t0[s] = DEAD;
  S->base++;
  if (S->base >= S->times) S->base -= S->times;
}

static int sim_init(sim* S, const ro_regex* R, const uint8_t* text, size_t n) {
  S->R = R;
  S->text = text;
  S->n = n;
  unsigned ml = R->max_mc > MAX_NODE_LENGTH ? MAX_NODE_LENGTH : R->max_mc;
  S->times = 1 + (int)ml;
  size_t slots = (size_t)S->times * (size_t)R->n_states;
  S->ring = (int64_t*)malloc(sizeof(int64_t) * slots);
  if (!S->ring) return 0;
  for (size_t i = 0; i < slots; i++) S->ring[i] = DEAD;
  S->base = 0;
  return 1;
}

typedef struct {
  uint64_t* out;
  size_t cap;
  size_t count;
  /* The filter needs to look at (and pop) previously stored matches; when the caller's
   * buffer is smaller than the result we keep our own copy. */
  uint64_t* own;
  size_t own_cap;
} sink;

/* MatchAllAppendFilter, src/codegen.cc:36-86 */
static void sink_emit(sink* K, uint64_t begin, uint64_t end) {
  while (K->count > 0 && K->own[2 * (K->count - 1)] >= begin) K->count--;
  if (begin == end && K->count > 0 && begin == K->own[2 * (K->count - 1) + 1]) return;
  if (K->count == K->own_cap) {
    K->own_cap = K->own_cap ? K->own_cap * 2 : 64;
    K->own = (uint64_t*)realloc(K->own, sizeof(uint64_t) * 2 * K->own_cap);
  }
  K->own[2 * K->count] = begin;
  K->own[2 * K->count + 1] = end;
  K->count++;
}

long ro_match_all(const ro_regex* R, const uint8_t* text, size_t n, uint64_t* out, size_t cap) {
  sim S;
  if (!sim_init(&S, R, text, n)) return -3;
  sink K;
  memset(&K, 0, sizeof(K));
  int have_pending = 0;
  uint64_t pend_begin = 0, pend_end = 0;
  size_t p = 0;
  for (;;) {
    /* CheckTimeFlow, no fast-forward, kMatchAll: register a pending match,
     * codegen-x64.cc:259-275 -> RegisterMatch :469-522 */
    if (have_pending) {
      sink_emit(&K, pend_begin, pend_end);
      have_pending = 0;
      if (pend_end == n) break;
    }
    /* SetStateForce(0, entry_state): a new thread starts at every position, :544-554 */
    *slot(&S, 0, R->entry) = (int64_t)p;
    handle_control_regexps(&S, p);
    /* CheckMatch forward, :426-461 */
    int64_t xs = *slot(&S, 0, R->exit);
    if (xs != DEAD) {
      have_pending = 1;
      pend_begin = (uint64_t)xs;
      pend_end = (uint64_t)p;
      /* ClearStates(begin, end): kill threads that started strictly inside the match */
      size_t slots = (size_t)S.times * (size_t)R->n_states;
      for (size_t i = 0; i < slots; i++) {
        if (S.ring[i] != DEAD && S.ring[i] > xs && S.ring[i] < (int64_t)p) S.ring[i] = DEAD;
      }
    }
    if (p == n) {
      /* limit, :583-612 */
      if (have_pending) sink_emit(&K, pend_begin, pend_end);
      break;
    }
    generate_transitions(&S, p);
    advance_time(&S);
    p++;
  }
  for (size_t i = 0; i < K.count && i < cap; i++) {
    out[2 * i] = K.own[2 * i];
    out[2 * i + 1] = K.own[2 * i + 1];
  }
  long count = (long)K.count;
  free(K.own);
  free(S.ring);
  return count;
}

int ro_match_first(const ro_regex* R, const uint8_t* text, size_t n, uint64_t* be) {
  uint64_t first[2];
  long c = ro_match_all(R, text, n, first, 1);
  if (c <= 0) return 0;
  if (be) { be[0] = first[0]; be[1] = first[1]; }
  return 1;
}

int ro_match_anywhere(const ro_regex* R, const uint8_t* text, size_t n) {
  return ro_match_all(R, text, n, NULL, 0) > 0;
}

/* kMatchFull: Generate :162-164 seeds the entry state once; GenerateMatchDirection
 * without re-seeding; CheckTimeFlow :252-256 gives up when no thread is alive; the
 * answer is "exit state live at the end of the text", :586-590. */
int ro_match_full(const ro_regex* R, const uint8_t* text, size_t n) {
  sim S;
  if (!sim_init(&S, R, text, n)) return -3;
  *slot(&S, 0, R->entry) = 0;
  size_t p = 0;
  int result = 0;
  size_t slots = (size_t)S.times * (size_t)R->n_states;
  for (;;) {
    int alive = 0;
    for (size_t i = 0; i < slots; i++) if (S.ring[i] != DEAD) { alive = 1; break; }
    if (!alive) { result = 0; break; }
    handle_control_regexps(&S, p);
    if (p == n) { result = *slot(&S, 0, R->exit) != DEAD; break; }
    generate_transitions(&S, p);
    advance_time(&S);
    p++;
  }
  free(S.ring);
  return result;
}


/* ------------------------------------------------------------------------- */
/* The DOCUMENTED semantics (include/rejit.h:59-64: "all left-most longest matches"),
 * computed independently of the ring dynamics: longest match from every start, then a
 * greedy left-to-right pick.  The reference's own no-FF loop above agrees with this
 * except for one artefact (DESIGN.md "Q8"): a thread seeded exactly at the end p of a
 * match (b,p) can be shadowed in a ring slot by an older thread that started inside
 * (b,p) and is cleared by ClearStates in the same step (codegen-x64.cc:455-459), so the
 * thread starting at p silently loses states; with use_fast_forward=1 the reference
 * returns yet another (overlapping) answer on those inputs.  Tests use this function to
 * classify such inputs; the strict restatement above stays the parity oracle. */
static int longest_from(const ro_regex* R, const uint8_t* text, size_t n, size_t s, size_t* end) {
  sim S;
  if (!sim_init(&S, R, text, n)) return 0;
  *slot(&S, 0, R->entry) = (int64_t)s;
  size_t p = s;
  int found = 0;
  size_t slots = (size_t)S.times * (size_t)R->n_states;
  for (;;) {
    int alive = 0;
    for (size_t i = 0; i < slots; i++) if (S.ring[i] != DEAD) { alive = 1; break; }
    if (!alive) break;
    handle_control_regexps(&S, p);
    if (*slot(&S, 0, R->exit) != DEAD) { found = 1; *end = p; }
    if (p == n) break;
    generate_transitions(&S, p);
    advance_time(&S);
    p++;
  }
  free(S.ring);
  return found;
}

long ro_match_all_spec(const ro_regex* R, const uint8_t* text, size_t n, uint64_t* out, size_t cap) {
  size_t count = 0, cur = 0, prev_end = 0;
  int have_prev = 0;
  for (size_t s = 0; s <= n; s++) {
    if (s < cur) continue;
    size_t e;
    if (!longest_from(R, text, n, s, &e)) continue;
    cur = e > s ? e : s + 1;
    if (!(e == s && have_prev && prev_end == s)) { /* zero-length rule, codegen.cc:65-73 */
      if (count < cap) { out[2 * count] = s; out[2 * count + 1] = e; }
      count++;
    }
    have_prev = 1;
    prev_end = e;
  }
  return (long)count;
}

long ro_match_all_spec_re(const char* regexp, const uint8_t* text, size_t n, uint64_t* out, size_t cap) {
  ro_regex* R;
  int st = ro_compile(regexp, &R);
  if (st != RO_OK) return st;
  long c = ro_match_all_spec(R, text, n, out, cap);
  ro_free(R);
  return c;
}

/* ends[s] = end of the longest match that begins exactly at s, or -1 (s = 0..n).
 * Test helper: lets the multi-rank selection protocol be checked against a plain
 * per-start table. */
int ro_longest_all_re(const char* regexp, const uint8_t* text, size_t n, int64_t* ends) {
  ro_regex* R;
  int st = ro_compile(regexp, &R);
  if (st != RO_OK) return st;
  for (size_t s = 0; s <= n; s++) {
    size_t e;
    ends[s] = longest_from(R, text, n, s, &e) ? (int64_t)e : -1;
  }
  ro_free(R);
  return RO_OK;
}

long ro_match_all_re(const char* regexp, const uint8_t* text, size_t n, uint64_t* out, size_t cap) {
  ro_regex* R;
  int st = ro_compile(regexp, &R);
  if (st != RO_OK) return st;
  long c = ro_match_all(R, text, n, out, cap);
  ro_free(R);
  return c;
}

int ro_match_full_re(const char* regexp, const uint8_t* text, size_t n) {
  ro_regex* R;
  int st = ro_compile(regexp, &R);
  if (st != RO_OK) return st;
  int r = ro_match_full(R, text, n);
  ro_free(R);
  return r;
}

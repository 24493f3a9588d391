// include/rejit.h -- C++ interface of the MI355X regex scan engine.
//
// Source-compatible with the public header of coreperf/rejit (its include/rejit.h:41-138):
// the same namespace, types, enumerators and signatures, so that callers written against
// the reference (its sample/regexdna.cc, sample/jrep.cc, tools/benchmarks/engines/rejit/
// engine.cc) compile and link against librejit_hip.so unchanged.  Everything behind the
// declarations is different: a Regej owns a lowered automaton resident in HBM
// (include/rejit_hip.h: rj_program) instead of JIT-generated x86 code, and every Match*
// call runs the HIP kernel pipeline.
//
// Semantics (same as the reference): matches are left-most longest and do not overlap;
// '.' does not match '\n' or '\r'; '^' / '$' are multi-line and treat both '\n' and '\r'
// as line breaks; a Match holds pointers into the caller's text, [begin, end), with
// begin == end for an empty match.  MatchAll APPENDS to the vector it is given.
//
// Differences a caller can observe: the pattern string is copied (the reference keeps the
// caller's pointer); one Regej can be used for several match types and from several
// threads at once; a HIP failure is reported as status() == DeviceError instead of an
// abort.
#ifndef REJIT_H_
#define REJIT_H_

#include <cstddef>
#include <string>
#include <vector>

using namespace std;  // the reference header does this and its callers rely on it

struct rj_program;

namespace rejit {

struct Match {
  const char* begin;
  const char* end;
};

enum MatchType { kMatchFull, kMatchAnywhere, kMatchFirst, kMatchAll, kNMatchTypes };

// Negative values are errors; rejit_status_string then holds a description.
enum Status {
  RejitSuccess = 0,
  ParserError = -1,
  // additions of this implementation (the reference aborts in the equivalent situations)
  PatternTooLarge = -2,
  DeviceError = -3
};
extern char* const rejit_status_string;

class Regej {
 public:
  explicit Regej(const char* regexp);
  explicit Regej(const string& regexp);
  ~Regej();

  Status status() const { return status_; }
  // Outcome of the most recent Match* / Replace* call on this object: RejitSuccess, or the reason it
  // returned "no match" without having matched (a HIP failure, an input the device automaton cannot
  // take).  The reference aborts in the equivalent situations (rejit_fatal, src/checks.cc:19-26);
  // here the call returns 0 / false, rejit_status_string holds the message, a line goes to stderr
  // and this accessor tells a caller the result is not an answer.
  Status last_status() const { return last_status_; }

  bool MatchFull(const string& text);
  bool MatchFull(const char* text, size_t text_size);
  bool MatchAnywhere(const string& text);
  bool MatchAnywhere(const char* text, size_t text_size);
  bool MatchFirst(const string& text, Match* match);
  bool MatchFirst(const char* text, size_t text_size, Match* match);
  size_t MatchAll(const string& text, std::vector<struct Match>* matches);
  size_t MatchAll(const char* text, size_t text_size, std::vector<struct Match>* matches);
  size_t MatchAllCount(const string& text);
  size_t MatchAllCount(const char* text, size_t text_size);

  bool ReplaceFirst(string& text, const string& with);
  size_t ReplaceAll(string& text, const string& with);

  // Kept for source compatibility: lowering happens in the constructor, so this only
  // reports whether the pattern is usable.
  bool Compile(MatchType match_type);

 private:
  Regej(const Regej&);
  Regej& operator=(const Regej&);
  void init(const char* regexp);

  bool failed(long rc);

  string regexp_;
  rj_program* program_;
  Status status_;
  Status last_status_;
};

// One-shot helpers.  The reference builds a temporary Regej per call (no cache, its include/rejit.h:48-50); here the 32
// most recent patterns stay compiled (a compile also fills device tables) -- same results, the second call is cheap.
bool MatchFull(const char* regexp, const string& text);
bool MatchFull(const char* regexp, const char* text, size_t text_size);
bool MatchAnywhere(const char* regexp, const string& text);
bool MatchAnywhere(const char* regexp, const char* text, size_t text_size);
bool MatchFirst(const char* regexp, const string& text, Match* match);
bool MatchFirst(const char* regexp, const char* text, size_t text_size, Match* match);
size_t MatchAll(const char* regexp, const string& text, std::vector<struct Match>* matches);
size_t MatchAll(const char* regexp, const char* text, size_t text_size, std::vector<struct Match>* matches);
size_t MatchAllCount(const char* regexp, const string& text);
size_t MatchAllCount(const char* regexp, const char* text, size_t text_size);

// Splice `with` over one match / over every match of a list (ordered, non-overlapping).
void Replace(Match to_replace, string& text, const string& with);
void Replace(vector<Match>* to_replace, string& text, const string& with);
bool ReplaceFirst(const char* regexp, string& text, const string& with);
size_t ReplaceAll(const char* regexp, string& text, const string& with);

}  // namespace rejit

#endif  // REJIT_H_

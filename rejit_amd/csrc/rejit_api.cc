// rejit_amd/csrc/rejit_api.cc -- the rejit:: C++ API (include/rejit.h) on top of the C ABI
// (include/rejit_hip.h).  Counterpart of the reference's src/rejit.cc: the free functions
// build a temporary Regej per call (rejit.cc:37-87), Replace splices around the matches
// (rejit.cc:97-112), MatchAll appends with the sink's filter (src/codegen.cc:36-86).
#include "../../include/rejit.h"

#include <cstdio>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <string>

#include "../../include/rejit_hip.h"

namespace rejit {

namespace {
char g_status_buffer[512] = "";

// (the reference's global buffer, shared by every Regej of the process: writers take turns, so a message is
// never a mix of two; a reader racing with a writer is inherent to the reference's interface)
std::mutex g_status_mutex;
void set_status_string(const char* msg) {
  std::lock_guard<std::mutex> lock(g_status_mutex);
  snprintf(g_status_buffer, sizeof(g_status_buffer), "%s", msg ? msg : "");
}

Status to_status(int rc) {
  switch (rc) {
    case RJ_OK: return RejitSuccess;
    case RJ_PARSER_ERROR: return ParserError;
    case RJ_TOO_LARGE: return PatternTooLarge;
    default: return DeviceError;
  }
}
}  // namespace

char* const rejit_status_string = g_status_buffer;

void Regej::init(const char* regexp) {
  regexp_ = regexp ? regexp : "";
  program_ = nullptr;
  int rc = rj_compile(regexp_.c_str(), &program_);
  status_ = to_status(rc);
  last_status_ = status_;
  if (rc != RJ_OK) set_status_string(rj_last_error());
}

// A run-time failure must not look like "no match" (the reference aborts where this library returns
// an error code): record it, keep the message, and say so on stderr.
bool Regej::failed(long rc) {
  if (rc >= 0) {
    last_status_ = RejitSuccess;
    return false;
  }
  last_status_ = to_status(static_cast<int>(rc));
  set_status_string(rj_last_error());
  fprintf(stderr, "rejit: matching /%s/ failed: %s\n", regexp_.c_str(), rejit_status_string);
  return true;
}

Regej::Regej(const char* regexp) { init(regexp); }
Regej::Regej(const string& regexp) { init(regexp.c_str()); }
Regej::~Regej() { rj_program_free(program_); }

bool Regej::Compile(MatchType) { return status_ == RejitSuccess; }

bool Regej::MatchFull(const string& text) { return MatchFull(text.c_str(), text.size()); }

bool Regej::MatchFull(const char* text, size_t text_size) {
  if (status_ != RejitSuccess) return false;
  int r = rj_match_full(program_, text, text_size);
  if (failed(r)) return false;
  return r > 0;
}

bool Regej::MatchAnywhere(const string& text) { return MatchAnywhere(text.c_str(), text.size()); }

bool Regej::MatchAnywhere(const char* text, size_t text_size) {
  if (status_ != RejitSuccess) return false;
  int r = rj_match_anywhere(program_, text, text_size);
  if (failed(r)) return false;
  return r > 0;
}

bool Regej::MatchFirst(const string& text, Match* match) { return MatchFirst(text.c_str(), text.size(), match); }

bool Regej::MatchFirst(const char* text, size_t text_size, Match* match) {
  if (status_ != RejitSuccess) return false;
  uint64_t b = 0, e = 0;
  int r = rj_match_first(program_, text, text_size, &b, &e);
  if (failed(r)) return false;
  if (r > 0 && match) {
    match->begin = text + b;
    match->end = text + e;
  }
  return r > 0;
}

size_t Regej::MatchAll(const string& text, std::vector<Match>* matches) {
  return MatchAll(text.c_str(), text.size(), matches);
}

size_t Regej::MatchAll(const char* text, size_t text_size, std::vector<Match>* matches) {
  if (status_ != RejitSuccess) return 0;
  uint64_t* spans = nullptr;
  int64_t n = rj_match_all(program_, text, text_size, matches ? &spans : nullptr);
  if (failed(n)) return matches ? matches->size() : 0;
  if (!matches) return static_cast<size_t>(n);
  for (int64_t i = 0; i < n; i++) {
    Match m;
    m.begin = text + spans[2 * i];
    m.end = text + spans[2 * i + 1];
    // the reference's sink (MatchAllAppendFilter): a new match replaces stored matches
    // that begin at or after it, and an empty match right at the end of the previous one
    // is dropped -- which is what makes repeated calls on the same vector idempotent
    while (!matches->empty() && matches->back().begin >= m.begin) matches->pop_back();
    if (m.begin == m.end && !matches->empty() && m.begin == matches->back().end) continue;
    matches->push_back(m);
  }
  rj_free_spans(spans);
  return matches->size();
}

size_t Regej::MatchAllCount(const string& text) { return MatchAllCount(text.c_str(), text.size()); }

size_t Regej::MatchAllCount(const char* text, size_t text_size) {
  if (status_ != RejitSuccess) return 0;
  int64_t n = rj_match_all(program_, text, text_size, nullptr);  // count only: 8 bytes come back
  if (failed(n)) return 0;
  return static_cast<size_t>(n);
}

bool Regej::ReplaceFirst(string& text, const string& with) {
  Match m;
  if (!MatchFirst(text, &m)) return false;
  Replace(m, text, with);
  return true;
}

size_t Regej::ReplaceAll(string& text, const string& with) {
  if (status_ != RejitSuccess) return 0;
  // MatchAll + Replace fused on the device: only the new text comes back over PCIe -- straight into the string the caller
  // handed over (the device has its own copy of the old text by then): no second buffer to allocate, fault in and free
  // (measured: that was 130 ms of a 1 GB call, the transfer itself 18)
  size_t new_len = 0;
  int64_t n = rj_replace_all_begin(program_, text.data(), text.size(), with.data(), with.size(), &new_len);
  if (failed(n)) return 0;
  if (new_len > text.capacity()) {
    // (room to grow: regexdna's eleven IUB replacements lengthen the text eleven times, sample/regexdna.cc:71-87 -- one
    // reallocation instead of eleven.  The old contents need not survive it.)
    text.clear();
    text.reserve(new_len + new_len / 2);
  }
  text.resize(new_len);
  int rc = rj_replace_all_fetch(program_, new_len ? &text[0] : nullptr, new_len);
  if (failed(rc)) {
    text.clear();   // (the old text is gone and the new one did not arrive: do not leave half of each)
    return 0;
  }
  return static_cast<size_t>(n);
}

// ----------------------------------------------------------------------------- free functions
// The reference's free functions build a temporary Regej -- parse + JIT -- on every call
// (src/rejit.cc:37-87; "there is no cache", include/rejit.h:48-50).  Here a compile also allocates
// and fills device tables, so the most recent patterns are kept.  A Regej is safe
// for concurrent use; entries are shared_ptrs so an evicted one lives until its callers return.
namespace {
std::shared_ptr<Regej> cached(const char* regexp) {
  static std::mutex mu;
  // (never destroyed: at process exit the HIP runtime may be gone before static destructors run)
  static auto& lru = *new std::list<std::pair<std::string, std::shared_ptr<Regej>>>;  // front = most recent
  const std::string key = regexp ? regexp : "";
  std::lock_guard<std::mutex> lock(mu);
  for (auto it = lru.begin(); it != lru.end(); ++it)
    if (it->first == key) {
      lru.splice(lru.begin(), lru, it);
      return lru.front().second;
    }
  std::shared_ptr<Regej> fresh = std::make_shared<Regej>(key);
  if (fresh->status() != RejitSuccess) return fresh;  // not kept: the next call reports the error again
  lru.emplace_front(key, fresh);
  if (lru.size() > 32) lru.pop_back();
  return fresh;
}
}  // namespace

bool MatchFull(const char* regexp, const string& text) { return MatchFull(regexp, text.c_str(), text.size()); }
bool MatchFull(const char* regexp, const char* text, size_t n) { return cached(regexp)->MatchFull(text, n); }
bool MatchAnywhere(const char* regexp, const string& text) { return MatchAnywhere(regexp, text.c_str(), text.size()); }
bool MatchAnywhere(const char* regexp, const char* text, size_t n) { return cached(regexp)->MatchAnywhere(text, n); }
bool MatchFirst(const char* regexp, const string& text, Match* m) {
  return MatchFirst(regexp, text.c_str(), text.size(), m);
}
bool MatchFirst(const char* regexp, const char* text, size_t n, Match* m) { return cached(regexp)->MatchFirst(text, n, m); }
size_t MatchAll(const char* regexp, const string& text, std::vector<Match>* ms) {
  return MatchAll(regexp, text.c_str(), text.size(), ms);
}
size_t MatchAll(const char* regexp, const char* text, size_t n, std::vector<Match>* ms) {
  return cached(regexp)->MatchAll(text, n, ms);
}
size_t MatchAllCount(const char* regexp, const string& text) { return MatchAllCount(regexp, text.c_str(), text.size()); }
size_t MatchAllCount(const char* regexp, const char* text, size_t n) { return cached(regexp)->MatchAllCount(text, n); }

void Replace(Match to_replace, string& text, const string& with) {
  std::vector<Match> one(1, to_replace);
  Replace(&one, text, with);
}

void Replace(vector<Match>* to_replace, string& text, const string& with) {
  string out;
  out.reserve(text.size() + text.size() / 16);
  const char* base = text.data();
  const char* copied_to = base;
  for (const Match& m : *to_replace) {
    out.append(copied_to, static_cast<size_t>(m.begin - copied_to));
    out.append(with);
    copied_to = m.end;
  }
  out.append(copied_to, static_cast<size_t>(base + text.size() - copied_to));
  text.swap(out);
}

bool ReplaceFirst(const char* regexp, string& text, const string& with) { return cached(regexp)->ReplaceFirst(text, with); }
size_t ReplaceAll(const char* regexp, string& text, const string& with) { return cached(regexp)->ReplaceAll(text, with); }

}  // namespace rejit

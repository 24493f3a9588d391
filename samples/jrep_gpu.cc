// samples/jrep_gpu.cc -- grep over a file tree on the GPU: the native counterpart of the reference's
// sample/jrep.cc (tree walk :408-493, per-file MatchAll :288, `^` line table :294-313, line output :336-369),
// written over the C ABI (include/rejit_hip.h) only.
//
//   jrep_gpu [-H] [-n] [-r|-R] [-c] [-A n] [-B n] [-C n] [-j readers] [--batch-mib m] [--count] PATTERN PATH...
//
// The reference calls MatchAll once per file; a GPU round trip costs >= 20 us whatever the file's size, so here
//   * one thread walks the tree (nftw, the reference's order -- output is printed in that order),
//   * a pool of reader threads reads the files of the NEXT batch (default 64 MiB of files) into one buffer while
//   * the main thread matches the CURRENT batch in ONE device pass (rj_match_all_batch: every file is its own
//     text, no match crosses a file boundary), builds the `^` line table of the files with matches in a second,
//     small pass (sample/jrep.cc:292-295) and formats file:line:text.
// With several GPUs visible rj_match_all_batch spreads a batch's files over them below the C ABI
// (multi_device.hip); nothing here changes.
//
// Exit status like grep: 0 = some line matched, 1 = none, 2 = error.
#include <fcntl.h>
#include <ftw.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rejit_hip.h"

namespace {

struct Options {
  bool with_filename = false, line_number = false, recursive = false, follow = false, color = false, count_only = false;
  unsigned before = 0, after = 0;
  unsigned readers = 0;
  size_t batch_bytes = 64u << 20;
  unsigned nopenfd = 1024;
  const char* pattern = nullptr;
  std::vector<const char*> paths;
};

struct FileEntry {
  std::string name;
  size_t size;  // from the walk's stat; what was actually read may be shorter
};

std::vector<FileEntry>* g_listing = nullptr;

int list_cb(const char* path, const struct stat* st, int typeflag, struct FTW*) {
  if (typeflag == FTW_F) g_listing->push_back({path, static_cast<size_t>(st->st_size)});
  return 0;
}

// ---- a batch: consecutive files of the listing, read into one buffer by the reader pool
struct Batch {
  size_t first = 0, last = 0;          // files [first, last) of the listing
  char* data = nullptr;                // malloc'ed, NOT initialised (every byte of it is written by a read)
  std::vector<size_t> off, len;        // file i of the batch: data[off[i] .. off[i] + len[i])
  std::atomic<size_t> unread{0};
};

struct Pipeline {
  const std::vector<FileEntry>* files = nullptr;
  std::vector<Batch>* batches = nullptr;
  std::vector<uint32_t> batch_of;      // file -> batch
  std::atomic<size_t> next_file{0};
  std::mutex mu;
  std::condition_variable batch_read, batch_freed;
  size_t consumed = 0;                 // batches the matcher is done with
  size_t ahead = 8;                    // readers run at most this many batches ahead of the matcher
};

void reader(Pipeline* p) {
  const std::vector<FileEntry>& files = *p->files;
  for (;;) {
    const size_t k = p->next_file.fetch_add(1);
    if (k >= files.size()) return;
    const uint32_t bi = p->batch_of[k];
    Batch& b = (*p->batches)[bi];
    {
      std::unique_lock<std::mutex> lk(p->mu);
      while (bi >= p->consumed + p->ahead) p->batch_freed.wait(lk);
      if (!b.data) {
        size_t bytes = 1;
        for (size_t i = b.first; i < b.last; i++) bytes += files[i].size;
        b.data = static_cast<char*>(malloc(bytes));
        if (!b.data) {
          fprintf(stderr, "jrep_gpu: out of memory\n");
          _exit(2);
        }
      }
    }
    const FileEntry& f = files[k];
    const size_t slot = k - b.first;
    size_t got = 0;
    if (f.size) {
      const int fd = open(f.name.c_str(), O_RDONLY);
      if (fd < 0) {
        fprintf(stderr, "jrep_gpu: %s: %s\n", f.name.c_str(), strerror(errno));
      } else {
        while (got < f.size) {
          const ssize_t r = read(fd, b.data + b.off[slot] + got, f.size - got);
          if (r <= 0) break;
          got += static_cast<size_t>(r);
        }
        close(fd);
      }
    }
    b.len[slot] = got;
    if (b.unread.fetch_sub(1) == 1) {
      std::lock_guard<std::mutex> lk(p->mu);
      p->batch_read.notify_all();
    }
  }
}

// ---- output: every line that holds the begin of a match, once, with optional context -- the output loop of
// sample/jrep.cc:300-400 over offsets.  `starts` = the begins of the `^` matches inside the file.
struct Printer {
  const Options& o;
  std::string out;
  explicit Printer(const Options& opt) : o(opt) {}

  void head(const std::string& name, size_t line, char sep) {
    if (o.with_filename) {
      out += name;
      out += sep;
    }
    if (o.line_number) {
      out += std::to_string(line + 1);
      out += sep;
    }
  }
  static bool breaks(char c) { return c == '\n' || c == '\r'; }

  void file(const std::string& name, const char* data, size_t n, const uint64_t* spans, size_t n_matches, const uint64_t* sol, size_t n_sol) {
    std::vector<size_t> starts;
    for (size_t i = 0; i < n_sol; i++)
      if (sol[2 * i] < n) starts.push_back(sol[2 * i]);  // (`^` also matches behind a final line break)
    if (starts.empty()) starts.push_back(0);
    const size_t n_lines = starts.size();
    auto line_end = [&](size_t l) { return l + 1 < n_lines ? starts[l + 1] : n; };
    auto line_of = [&](size_t pos) {
      const size_t l = static_cast<size_t>(std::upper_bound(starts.begin(), starts.end(), pos) - starts.begin());
      return std::min(l ? l - 1 : 0, n_lines - 1);
    };
    auto whole_line = [&](size_t l) {
      out.append(data + starts[l], line_end(l) - starts[l]);
      if (line_end(l) == starts[l] || !breaks(data[line_end(l) - 1])) out += '\n';
    };
    struct Group {
      size_t line, last, m0, m1;  // matches [m0, m1) begin on `line`; the last of them ends on line `last`
    };
    std::vector<Group> groups;
    for (size_t m = 0; m < n_matches; m++) {
      const size_t b = spans[2 * m], e = spans[2 * m + 1];
      if (b >= n && (n == 0 || breaks(data[n - 1]))) continue;  // at the end of a file that ends in a line break: no line to show
      const size_t line = line_of(b);
      const size_t last = e > b ? std::max(line, line_of(e - 1)) : line;
      if (!groups.empty() && line <= groups.back().last) {
        groups.back().last = std::max(groups.back().last, last);
        groups.back().m1 = m + 1;
      } else {
        groups.push_back({line, last, m, m + 1});
      }
    }
    const bool context = o.before || o.after;
    long printed = -1;  // last line written so far
    for (size_t g = 0; g < groups.size(); g++) {
      const Group& G = groups[g];
      const size_t lo = std::max<long>(static_cast<long>(G.line) - static_cast<long>(o.before), printed + 1);
      if (context && printed >= 0 && static_cast<long>(lo) > printed + 1) out += "--\n";
      for (size_t c = lo; c < G.line; c++) {
        head(name, c, '-');
        whole_line(c);
      }
      head(name, G.line, ':');
      size_t at = starts[G.line];
      for (size_t m = G.m0; m < G.m1; m++) {
        const size_t mb = spans[2 * m], me = spans[2 * m + 1];
        if (mb >= n && (n == 0 || breaks(data[n - 1]))) continue;
        out.append(data + at, mb - at);
        if (o.color) out += "\x1b[31m";
        out.append(data + mb, me - mb);
        if (o.color) out += "\x1b[0m";
        at = me;
      }
      const size_t end = line_end(G.last);
      out.append(data + at, end > at ? end - at : 0);
      const bool ends_in_break = end > at ? breaks(data[end - 1]) : (at > 0 && breaks(data[at - 1]));
      if (!ends_in_break) out += '\n';
      printed = static_cast<long>(G.last);
      const size_t nxt = g + 1 < groups.size() ? groups[g + 1].line : n_lines;
      for (size_t c = G.last + 1; c < std::min({G.last + o.after + 1, n_lines, nxt}); c++) {
        head(name, c, '-');
        whole_line(c);
        printed = static_cast<long>(c);
      }
    }
  }
};

int usage() {
  fprintf(stderr, "usage: jrep_gpu [-H] [-n] [-r|-R] [-c] [-A n] [-B n] [-C n] [-j readers] [--batch-mib m] [--count] PATTERN PATH...\n");
  return 2;
}

}  // namespace

int main(int argc, char** argv) {
  Options o;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto number = [&](const char* attached) -> unsigned {
      if (attached && *attached) return static_cast<unsigned>(strtoul(attached, nullptr, 10));
      if (i + 1 < argc) return static_cast<unsigned>(strtoul(argv[++i], nullptr, 10));
      return 0;
    };
    if (a == "-H" || a == "--with-filename") o.with_filename = true;
    else if (a == "-n" || a == "--line-number") o.line_number = true;
    else if (a == "-r" || a == "--recursive") o.recursive = true;
    else if (a == "-R" || a == "--dereference-recursive") o.recursive = o.follow = true;
    else if (a == "-c" || a == "--color_output") o.color = true;
    else if (a == "--count") o.count_only = true;
    else if (a.rfind("-A", 0) == 0) o.after = number(a.c_str() + 2);
    else if (a.rfind("-B", 0) == 0) o.before = number(a.c_str() + 2);
    else if (a.rfind("-C", 0) == 0) o.after = o.before = number(a.c_str() + 2);
    else if (a.rfind("-j", 0) == 0) o.readers = number(a.c_str() + 2);
    else if (a.rfind("-k", 0) == 0) o.nopenfd = number(a.c_str() + 2);
    else if (a == "--batch-mib") o.batch_bytes = static_cast<size_t>(number(nullptr)) << 20;
    else if (a.size() > 1 && a[0] == '-' && a != "--") return usage();
    else if (!o.pattern) o.pattern = argv[i];
    else o.paths.push_back(argv[i]);
  }
  if (!o.pattern || o.paths.empty()) return usage();
  if (o.pattern[0] == 0) return 0;  // (sample/jrep.cc:506)
  if (o.readers == 0) o.readers = std::min(16u, std::max(2u, std::thread::hardware_concurrency()));
  if (o.batch_bytes == 0) o.batch_bytes = 64u << 20;

  // the device comes up (HIP runtime, code objects, streams, staging memory: ~150 ms of a fresh process) while
  // the tree is walked and the first batches are read
  rj_program *re = nullptr, *sol = nullptr;
  int compile_rc = RJ_OK;
  std::string compile_error;
  std::thread device([&] {
    compile_rc = rj_compile(o.pattern, &re);
    if (compile_rc == RJ_OK) compile_rc = rj_compile("^", &sol);  // sample/jrep.cc:239: the line table is a MatchAll of "^"
    if (compile_rc != RJ_OK) {
      compile_error = rj_last_error();
      return;
    }
    // warm-up: a tiny batch per pattern loads the kernels and creates the scratch the real batches reuse
    static const char warm[2][64] = {"warm up\n", "jrep\n"};
    const char* texts[2] = {warm[0], warm[1]};
    const size_t sizes[2] = {8, 5};
    uint64_t counts[2];
    (void)rj_match_all_batch(re, texts, sizes, 2, counts, nullptr);
    (void)rj_match_all_batch(sol, texts, sizes, 2, counts, nullptr);
  });

  std::vector<FileEntry> files;
  g_listing = &files;
  for (const char* path : o.paths) {
    struct stat st;
    if (stat(path, &st) != 0) {
      fprintf(stderr, "jrep_gpu: %s: %s\n", path, strerror(errno));
      continue;
    }
    if (S_ISDIR(st.st_mode)) {
      if (!o.recursive) {
        fprintf(stderr, "jrep_gpu: %s: Is a directory.\n", path);
        continue;
      }
      nftw(path, list_cb, static_cast<int>(o.nopenfd), o.follow ? 0 : FTW_PHYS);
    } else if (S_ISREG(st.st_mode)) {
      files.push_back({path, static_cast<size_t>(st.st_size)});
    }
  }

  // batches of consecutive files; a pool of readers fills them in listing order, a few batches ahead of the matcher
  std::vector<Batch> batches;
  {
    size_t n_batches = 0;
    for (size_t i = 0; i < files.size();) {
      size_t bytes = 0, first = i;
      while (i < files.size() && (i == first || bytes + files[i].size <= o.batch_bytes)) bytes += files[i++].size;
      n_batches++;
    }
    batches = std::vector<Batch>(n_batches);
  }
  Pipeline pipe;
  pipe.files = &files;
  pipe.batches = &batches;
  pipe.batch_of.resize(files.size());
  {
    size_t bi = 0;
    for (size_t i = 0; i < files.size(); bi++) {
      Batch& b = batches[bi];
      b.first = i;
      size_t bytes = 0;
      while (i < files.size() && (i == b.first || bytes + files[i].size <= o.batch_bytes)) {
        b.off.push_back(bytes);
        bytes += files[i].size;
        pipe.batch_of[i] = static_cast<uint32_t>(bi);
        i++;
      }
      b.last = i;
      b.len.assign(b.last - b.first, 0);
      b.unread.store(b.last - b.first);
    }
  }
  std::vector<std::thread> readers;
  for (unsigned t = 0; t < std::min<size_t>(o.readers, std::max<size_t>(files.size(), 1)); t++) readers.emplace_back(reader, &pipe);
  device.join();
  if (compile_rc != RJ_OK) {
    fprintf(stderr, "jrep_gpu: %s\n", compile_error.c_str());
    fflush(stderr);
    _exit(2);  // (readers may still be at work: the process ends here)
  }
  bool found = false;
  int rc = 0;
  Printer printer(o);
  for (size_t bi = 0; bi < batches.size() && rc == 0; bi++) {
    Batch& b = batches[bi];
    {
      std::unique_lock<std::mutex> lk(pipe.mu);
      while (b.unread.load() != 0) pipe.batch_read.wait(lk);
    }
    const size_t nf = b.last - b.first;
    std::vector<const char*> texts(nf);
    for (size_t k = 0; k < nf; k++) texts[k] = b.data + b.off[k];
    std::vector<uint64_t> counts(nf, 0);
    uint64_t* spans = nullptr;
    const int64_t total = rj_match_all_batch(re, texts.data(), b.len.data(), nf, counts.data(), &spans);
    if (total < 0) {
      fprintf(stderr, "jrep_gpu: %s\n", rj_last_error());
      rc = 2;
    } else if (total > 0) {
      found = true;
      std::vector<size_t> hit;
      for (size_t k = 0; k < nf; k++)
        if (counts[k]) hit.push_back(k);
      if (o.count_only) {
        for (size_t k : hit) {
          printer.out += files[b.first + k].name;
          printer.out += ':';
          printer.out += std::to_string(counts[k]);
          printer.out += '\n';
        }
      } else {
        // second pass, only over the files with matches (sample/jrep.cc:292-295)
        std::vector<const char*> htexts(hit.size());
        std::vector<size_t> hsizes(hit.size());
        for (size_t h = 0; h < hit.size(); h++) {
          htexts[h] = texts[hit[h]];
          hsizes[h] = b.len[hit[h]];
        }
        std::vector<uint64_t> lcounts(hit.size(), 0);
        uint64_t* lines = nullptr;
        const int64_t lt = rj_match_all_batch(sol, htexts.data(), hsizes.data(), hit.size(), lcounts.data(), &lines);
        if (lt < 0) {
          fprintf(stderr, "jrep_gpu: %s\n", rj_last_error());
          rc = 2;
        } else {
          size_t at = 0, lat = 0, h = 0;
          for (size_t k = 0; k < nf; k++) {
            if (counts[k]) {
              printer.file(files[b.first + k].name, texts[k], b.len[k], spans + 2 * at, counts[k], lines + 2 * lat, lcounts[h]);
              lat += lcounts[h];
              h++;
            }
            at += counts[k];
          }
          rj_free_spans(lines);
        }
      }
      fwrite(printer.out.data(), 1, printer.out.size(), stdout);
      printer.out.clear();
    }
    rj_free_spans(spans);
    free(b.data);
    b.data = nullptr;
    {
      std::lock_guard<std::mutex> lk(pipe.mu);
      pipe.consumed = bi + 1;
      pipe.batch_freed.notify_all();
    }
  }
  fflush(stdout);
  if (rc) _exit(rc);
  for (auto& t : readers) t.join();
  // (no rj_program_free, no runtime teardown: the process ends here and takes both with it)
  _exit(found ? 0 : 1);
}

// hgaprec_host.cpp -- see hgaprec_host.hpp.  Plain C++17, no HIP.
#include "hgaprec_host.hpp"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <numeric>
#include <sstream>
#include <thread>
#include <functional>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

namespace hgaprec {

// ======================================================================
// Env
// ======================================================================
int Env::parse(int argc, char **argv, bool echo, std::string *bad)
{
  // same flag set and defaults as main.cc:41-232.  Flags that select modes
  // outside the CAVI hot path are accepted (they are valid options of the
  // CLI) and remembered in `unsupported`.
  auto out_of_scope = [&](const char *f) { if (unsupported.empty()) unsupported = f; };
  for (int i = 0; i < argc; ++i) {
    const char *s = argv[i];
    auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
    if (!strcmp(s, "-dir")) { datfname = next(); if (echo) fprintf(stdout, "+ dir = %s\n", datfname.c_str()); }
    else if (!strcmp(s, "-n")) { n = (uint32_t)atoi(next()); if (echo) fprintf(stdout, "+ n = %d\n", n); }
    else if (!strcmp(s, "-p")) { }
    else if (!strcmp(s, "-m")) { m = (uint32_t)atoi(next()); if (echo) fprintf(stdout, "+ m = %d\n", m); }
    else if (!strcmp(s, "-k")) { k = (uint32_t)atoi(next()); if (echo) fprintf(stdout, "+ k = %d\n", k); }
    else if (!strcmp(s, "-nmi")) { next(); out_of_scope("-nmi"); }
    else if (!strcmp(s, "-rfreq")) { rfreq = (uint32_t)atoi(next()); if (echo) fprintf(stdout, "+ rfreq = %d\n", rfreq); }
    else if (!strcmp(s, "-strid")) { out_of_scope("-strid"); }
    else if (!strcmp(s, "-label")) { label = next(); }
    else if (!strcmp(s, "-logl")) { logl = true; if (echo) fprintf(stdout, "+ logl mode\n"); }
    else if (!strcmp(s, "-max-iterations")) { max_iterations = (uint32_t)atoi(next()); if (echo) fprintf(stdout, "+ max iterations %d\n", max_iterations); }
    else if (!strcmp(s, "-seed")) { seed = atof(next()); if (echo) fprintf(stdout, "+ random seed set to %.5f\n", seed); }
    else if (!strcmp(s, "-load")) { next(); }   // parsed, never consulted (main.cc:137-140)
    else if (!strcmp(s, "-test")) { out_of_scope("-test"); }
    else if (!strcmp(s, "-batch")) { batch = true; if (echo) fprintf(stdout, "+ batch inference\n"); }
    else if (!strcmp(s, "-online")) { batch = false; if (echo) fprintf(stdout, "+ online inference\n"); }
    else if (!strcmp(s, "-gen-heldout")) { out_of_scope("-gen-heldout"); }
    else if (!strcmp(s, "-pred-accuracy") || !strcmp(s, "-gt-accuracy")) { }
    else if (!strcmp(s, "-netflix") || !strcmp(s, "-mendeley") || !strcmp(s, "-movielens") ||
             !strcmp(s, "-echonest")) { }   // dataset tag: only used by CREATE_TRAIN_TEST_SETS (dead)
    else if (!strcmp(s, "-nyt")) { out_of_scope("-nyt"); }
    else if (!strcmp(s, "-a")) { a = atof(next()); }
    else if (!strcmp(s, "-b")) { b = atof(next()); }
    else if (!strcmp(s, "-c")) { c = atof(next()); }
    else if (!strcmp(s, "-d")) { d = atof(next()); }
    else if (!strcmp(s, "-binary-data")) { binary_data = true; }
    else if (!strcmp(s, "-bias")) { bias = true; }
    else if (!strcmp(s, "-hier")) { hier = true; }
    else if (!strcmp(s, "-mle-user") || !strcmp(s, "-mle-item") || !strcmp(s, "-canny") ||
             !strcmp(s, "-gen-ranking") || !strcmp(s, "-rmse") || !strcmp(s, "-msr") ||
             !strcmp(s, "-nmf") || !strcmp(s, "-nmfload") || !strcmp(s, "-vwload") ||
             !strcmp(s, "-lda") || !strcmp(s, "-vwlda") || !strcmp(s, "-write-training") ||
             !strcmp(s, "-chi") || !strcmp(s, "-chinmf") || !strcmp(s, "-als") ||
             !strcmp(s, "-wals") || !strcmp(s, "-climf") || !strcmp(s, "-ctr")) { out_of_scope(s); }
    else if (!strcmp(s, "-novb")) { vb = false; }
    else if (!strcmp(s, "-wals_l") || !strcmp(s, "-wals_C")) { next(); }
    else if (!strcmp(s, "-rating-threshold")) { rating_threshold = (uint32_t)atoi(next()); }
    else if (!strcmp(s, "-device")) { device = atoi(next()); device_set = true; }   // extension: HIP device ordinal
    else if (!strcmp(s, "-ngpus")) { ngpus = atoi(next()); if (ngpus < 1) ngpus = 1; } // extension: one process per GPU
    else if (!strcmp(s, "-comm")) { comm_mode = next(); }            // extension: rccl | host
    else if (!strcmp(s, "-single-allreduce")) { single_allreduce = true; }  // extension: one fused all-reduce per iteration
    else if (!strcmp(s, "-checkpoint")) { checkpoint_every = (uint32_t)atoi(next()); }   // extension
    else if (!strcmp(s, "-resume")) { resume = true; }                                 // extension
    else if (!strcmp(s, "-cache")) { data_cache = true; }                              // extension
    else if (!strcmp(s, "-no-tiles")) { no_tiles = true; }       // extension: hpf_config.tiling = 1 (row-major work lists only)
    else if (!strcmp(s, "-plain-rows")) { plain_rows = true; }   // extension: hpf_config.w_storage = 3 (never pack the rows of W)
    else if (!strcmp(s, "-w48")) { w48 = true; }                 // extension, OPT-IN, lossy: hpf_config.w_storage = 2 (W in 48 bits)
    else if (i > 0) {
      if (bad) *bad = s;
      return 1;
    }
  }
  return 0;
}

std::string Env::make_prefix() const
{
  std::ostringstream sa;
  sa << "n" << n << "-";
  sa << "m" << m << "-";
  sa << "k" << k;
  if (label != "") sa << "-" << label;
  else if (datfname.length() > 3) {
    std::string q = datfname.substr(0, 2);
    if (isalpha((unsigned char)q[0])) sa << "-" << q;
  }
  if (a != 0.3) sa << "-a" << a;
  if (b != 0.3) sa << "-b" << b;
  if (c != 0.3) sa << "-c" << c;
  if (d != 0.3) sa << "-d" << d;
  if (batch) sa << "-batch"; else sa << "-online";
  if (binary_data) sa << "-bin";
  if (bias) sa << "-bias";
  if (hier) sa << "-hier";
  if (vb) sa << "-vb";
  if (seed) sa << "-seed" << seed;
  return sa.str();
}

int Env::open_output()
{
  prefix = make_prefix();
  fprintf(stdout, "+ Creating directory %s\n", prefix.c_str());
  fflush(stdout);
  struct stat st;
  if (stat(prefix.c_str(), &st) != 0) {            // log.cc:97-118, force = true
    if (errno != ENOENT) { fprintf(stderr, "Warning: could not stat dir %s\n", prefix.c_str()); return -1; }
    mkdir(prefix.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH);
    if (stat(prefix.c_str(), &st) != 0) { fprintf(stderr, "Warning: could not create dir %s\n", prefix.c_str()); return -1; }
  }
  std::string lf = prefix + "/infer.log";
  logf = fopen(lf.c_str(), "w");
  if (logf) fprintf(stderr, "+ writing log to %s\n", lf.c_str());
  else { logf = fopen("/dev/null", "w"); fprintf(stderr, "+ writing log to /dev/null\n"); }
  plogf = fopen(file_str("/param.txt").c_str(), "w");
  if (!plogf) { printf("cannot open param file:%s\n", strerror(errno)); return -1; }
  // env.hh:383-402.  (-a..-d only reach the directory name and these lines;
  // every Gamma object is built with the literal 0.3 prior, hgaprec.cc:13-20)
  plog("n", n); plog("k", k); plog("t", (uint32_t)2);
  plog("test_ratio", 0.2); plog("validation_ratio", 0.01);
  plog("seed", seed); plog("a", a); plog("b", b); plog("c", c); plog("d", d);
  plog("reportfreq", rfreq); plog("vb", vb); plog("bias", bias); plog("hier", hier);
  plog("nmf", false); plog("lda", false); plog("wals_l", 0.1); plog("wals_C", (uint32_t)10);
  plog("mle_user", false); plog("mle_item", false);
  return 0;
}

void Env::close_output()
{
  if (plogf) fclose(plogf);
  if (logf) fclose(logf);
  plogf = logf = nullptr;
}

void Env::plog(const std::string &key, double v) { fprintf(plogf, "%s: %.9f\n", key.c_str(), v); fflush(plogf); }
void Env::plog(const std::string &key, bool v) { fprintf(plogf, "%s: %s\n", key.c_str(), v ? "True" : "False"); fflush(plogf); }
void Env::plog(const std::string &key, uint32_t v) { fprintf(plogf, "%s: %d\n", key.c_str(), v); fflush(plogf); }
void Env::plog(const std::string &key, uint64_t v) { fprintf(plogf, "%s: %lu\n", key.c_str(), (unsigned long)v); fflush(plogf); }
void Env::plog(const std::string &key, const std::string &v) { fprintf(plogf, "%s: %s\n", key.c_str(), v.c_str()); fflush(plogf); }

void Env::lerr(const char *fmt, ...)
{
  if (!logf) return;
  time_t now = time(0); struct tm p; localtime_r(&now, &p);
  char ts[64]; strftime(ts, sizeof ts, "%b %e %T", &p);
  fprintf(logf, "[%s] [%d] [%3s] ", ts, (int)getpid(), "ERR");     // log.cc:26-46
  va_list ap; va_start(ap, fmt); vfprintf(logf, fmt, ap); va_end(ap);
  fprintf(logf, "\n\n");
  fflush(logf);
}

// ======================================================================
// Ratings
// ======================================================================
static inline uint32_t hash32(uint32_t k) { k *= 2654435761u; return k ^ (k >> 15); }

void Ratings::IdMap::rehash(uint32_t cap)
{
  std::vector<uint64_t> old(std::move(slots));
  slots.assign(cap, 0); cnt = 0;
  for (uint64_t e : old) if ((uint32_t)e) put((uint32_t)(e >> 32), (uint32_t)e - 1);
}
bool Ratings::IdMap::find(uint32_t key, uint32_t *val) const
{
  const uint32_t mask = (uint32_t)slots.size() - 1;
  for (uint32_t i = hash32(key) & mask;; i = (i + 1) & mask) {
    const uint64_t e = slots[i];
    if (!(uint32_t)e) return false;
    if ((uint32_t)(e >> 32) == key) { *val = (uint32_t)e - 1; return true; }
  }
}
void Ratings::IdMap::put(uint32_t key, uint32_t val)
{
  if ((uint64_t)(cnt + 1) * 2 > slots.size()) rehash((uint32_t)slots.size() * 2);
  const uint32_t mask = (uint32_t)slots.size() - 1;
  uint32_t i = hash32(key) & mask;
  for (; (uint32_t)slots[i]; i = (i + 1) & mask)
    if ((uint32_t)(slots[i] >> 32) == key) { slots[i] = ((uint64_t)key << 32) | (val + 1u); return; }
  slots[i] = ((uint64_t)key << 32) | (val + 1u); cnt++;
}

uint32_t Ratings::input_rating_class(uint32_t v) const     // ratings.hh:191-197
{
  if (!binary) return v;
  return v >= rating_threshold ? 1 : 0;
}

// buffered whitespace-separated unsigned reader with fscanf("%u\t%u\t%u\n")
// semantics for well-formed files (any whitespace separates fields)
namespace {
struct Tok {
  FILE *f; std::vector<char> buf; size_t pos = 0, len = 0; bool eof = false;
  explicit Tok(FILE *ff) : f(ff), buf(1 << 20) {}
  int peek() {
    if (pos == len) {
      if (eof) return -1;
      len = fread(buf.data(), 1, buf.size(), f); pos = 0;
      if (len == 0) { eof = true; return -1; }
    }
    return (unsigned char)buf[pos];
  }
  void skip_ws() { int c; while ((c = peek()) >= 0 && isspace(c)) ++pos; }
  // 1 = number read, 0 = matching failure (non-numeric), -1 = EOF before any digit
  int next_u32(uint32_t *out) {
    skip_ws();
    int c = peek();
    if (c < 0) return -1;
    bool neg = false;
    if (c == '+' || c == '-') { neg = (c == '-'); ++pos; c = peek(); }
    if (c < 0 || !isdigit(c)) return 0;
    uint64_t v = 0;
    while ((c = peek()) >= 0 && isdigit(c)) { v = v * 10 + (uint64_t)(c - '0'); if (v > 0xffffffffffffull) v &= 0xffffffffull; ++pos; }
    uint32_t r = (uint32_t)v;
    *out = neg ? (uint32_t)(0u - r) : r;
    return 1;
  }
  bool at_eof() { skip_ws(); return peek() < 0; }
};
}  // namespace

// One record of a ratings file, in the reference's order of tests (ratings.cc:77-111).  The two "last
// successful lookup" pairs only save hash probes (files are usually grouped by user).
struct Ratings::Consumer {
  Ratings &r; HeldOut *out; uint32_t lim_n, lim_m;
  bool have_last_u = false, have_last_m = false;
  uint32_t last_uid = 0, last_us = 0, last_mid = 0, last_ms = 0;
  // capacity for this pass: ratings.cc:35-36 shrinks env.n / env.m to the
  // registered counts once the training file has been read
  Consumer(Ratings &rr, HeldOut *o) : r(rr), out(o), lim_n(o ? rr.n : rr.cap_n), lim_m(o ? rr.m : rr.cap_m) {}
  void take(uint32_t uid, uint32_t mid, uint32_t rating) {
    uint32_t us = 0, ms = 0;
    bool hasu, hasm;
    if (have_last_u && uid == last_uid) { hasu = true; us = last_us; }
    else if ((hasu = r.user2seq.find(uid, &us))) { have_last_u = true; last_uid = uid; last_us = us; }
    if (have_last_m && mid == last_mid) { hasm = true; ms = last_ms; }
    else if ((hasm = r.item2seq.find(mid, &ms))) { have_last_m = true; last_mid = mid; last_ms = ms; }
    if ((!hasu && r.n >= lim_n) || (!hasm && r.m >= lim_m)) return;
    if (r.input_rating_class(rating) == 0) return;
    if (!hasu) { us = r.n; r.user2seq.put(uid, us); r.seq2user.push_back(uid); r.n++; }     // ratings.hh:117-133
    if (!hasm) { ms = r.m; r.item2seq.put(mid, ms); r.seq2item.push_back(mid); r.m++; }     // ratings.hh:135-151
    if (!out) {
      r.nratings++;
      r.tr_u_.push_back(us); r.tr_i_.push_back(ms); r.tr_y_.push_back(rating);
    } else {
      out->u.push_back(us); out->i.push_back(ms);
      out->y.push_back(r.binary ? 1 : (int32_t)rating);
    }
  }
};

int Ratings::read_generic(FILE *f, HeldOut *out)
{
  const int fast = read_generic_parallel(f, out);   // 0: done; 1: this file takes the token-by-token reader
  if (fast <= 0) return fast;
  Tok tk(f);
  uint32_t mid = 0, uid = 0, rating = 0;        // persist across lines like the reference's locals
  Consumer c(*this, out);
  bool first = true;
  while (true) {
    // while (!feof(f)) { if (fscanf(...) < 0) exit(-1); ... }
    if (!first && tk.at_eof()) break;           // the trailing "\n" directive ate the whitespace
    first = false;
    int r1 = tk.next_u32(&uid);
    if (r1 < 0) { printf("error: unexpected lines in file\n"); fflush(stdout); return -2; }   // empty file: ratings.cc:71-75
    // EOF inside the last record: fscanf returns 1 or 2 (>= 0) and the
    // reference goes on with the stale values of the missing fields
    int r2 = r1 == 1 ? tk.next_u32(&mid) : 0;
    int r3 = r2 == 1 ? tk.next_u32(&rating) : (r2 < 0 ? -1 : 0);
    if (r1 == 0 || r2 == 0 || r3 == 0) {
      // a non-numeric token: the reference never consumes it and loops forever
      fprintf(stderr, "error: malformed line in ratings file\n");
      return -2;
    }
    c.take(uid, mid, rating);
  }
  return 0;
}

// ---- the same reader on all host threads ------------------------------------
// A ratings file of 10^8..10^9 lines is seconds to minutes of single-thread parsing in front of iterations
// that take milliseconds.  For a WELL-FORMED file -- nothing but decimal digits and white space, a multiple
// of three tokens -- the result of the loop above can be produced in pieces:
//   1. the text is cut at white space into one piece per thread; the pieces count their tokens and refuse
//      anything that is not a digit or white space; a second pass parses the tokens into one array;
//   2. records (token triples) are dealt to the threads in file order; each notes, in order, the user and
//      item ids it sees for the first time among records whose rating class is not 0;
//   3. one thread merges those lists piece by piece: an id's sequence number is its rank among first
//      appearances -- what the loop above assigns as long as neither capacity (-n / -m) runs out.  If one
//      would, the maps are reset and the records go through Consumer::take one by one (the tests on
//      capacity couple the two sides, ratings.cc:84-85);
//   4. the threads translate their records through the finished maps into their place of the output.
// Held-out files never register ids (their capacities are the registered counts): steps 2-3 fall away.
// Anything else -- a sign, a letter, a token count that is no multiple of three, a small file, one thread --
// returns 1 and the token-by-token reader above takes the file with the reference's behaviour for it.
namespace {
inline bool is_ws(unsigned char c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace, "C" locale
struct LocalIds {          // ids in order of first appearance inside one piece
  std::vector<uint64_t> slots; std::vector<uint32_t> order; uint32_t mask;
  LocalIds() : slots(1024, 0), mask(1023) {}
  void note(uint32_t key) {
    for (uint32_t i = hash32(key) & mask;; i = (i + 1) & mask) {
      const uint64_t e = slots[i];
      if (!e) { slots[i] = ((uint64_t)key << 1) | 1u; order.push_back(key); if (order.size() * 2 > slots.size()) grow(); return; }
      if ((uint32_t)(e >> 1) == key) return;
    }
  }
  void grow() {
    std::vector<uint64_t> old(std::move(slots));
    slots.assign(old.size() * 2, 0); mask = (uint32_t)slots.size() - 1;
    for (uint64_t e : old) if (e) { uint32_t i = hash32((uint32_t)(e >> 1)) & mask; while (slots[i]) i = (i + 1) & mask; slots[i] = e; }
  }
};
template <typename F> void on_threads(unsigned nt, F fn)
{
  if (nt == 1) { fn(0u); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) th.emplace_back(fn, t);
  for (auto &x : th) x.join();
}
}  // namespace

int Ratings::read_generic_parallel(FILE *f, HeldOut *out)
{
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *e = getenv("HGAPREC_READ_THREADS")) { const int v = atoi(e); nt = v < 1 ? 1u : (unsigned)v; }   // (a negative or garbage value: one thread, not 64)
  nt = std::min(nt, 64u);
  size_t min_bytes = (size_t)8 << 20;
  if (const char *e = getenv("HGAPREC_READ_PARALLEL_MIN")) min_bytes = (size_t)strtoull(e, nullptr, 0);
  struct stat st;
  if (nt < 2 || fstat(fileno(f), &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < std::max<size_t>(min_bytes, 1)) return 1;
  const size_t size = (size_t)st.st_size;
  {
    // This reader holds every token as a 32-bit word (12 bytes per record, ~0.8 x the size of the text) BESIDE the
    // records it produces (another 12 bytes each) until it returns: about twice the streaming reader's peak, 24 GB
    // per 10^9 ratings.  Where that does not fit comfortably -- more than a quarter of the memory that is
    // available right now, or HGAPREC_READ_PARALLEL_MAX bytes of text -- the token-by-token reader takes the file.
    size_t max_bytes = 0;
    if (const char *e = getenv("HGAPREC_READ_PARALLEL_MAX")) max_bytes = (size_t)strtoull(e, nullptr, 0);
    else if (FILE *mi = fopen("/proc/meminfo", "r")) {
      char line[128]; unsigned long long kb = 0;
      while (fgets(line, sizeof line, mi)) if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) break;
      fclose(mi);
      if (kb) max_bytes = (size_t)(kb * 1024ull / 4 * 5 / 4);        // text whose tokens (0.8 x) take a quarter of it
    }
    if (max_bytes && size > max_bytes) return 1;
  }
  void *map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fileno(f), 0);
  if (map == MAP_FAILED) return 1;
  const unsigned char *d = (const unsigned char *)map;
  struct Unmap { void *p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, size};
  nt = (unsigned)std::max<size_t>(1, std::min<size_t>(nt, size / 64 + 1));

  // 1. pieces of text that start behind white space; tokens per piece; only digits and white space
  std::vector<size_t> cut(nt + 1, size);
  cut[0] = 0;
  for (unsigned t = 1; t < nt; ++t) {
    size_t p = std::max(cut[t - 1], size / nt * t);
    while (p < size && p > 0 && !is_ws(d[p - 1])) ++p;
    cut[t] = p;
  }
  std::vector<uint64_t> ntok(nt + 1, 0);
  std::vector<char> dirty(nt, 0);
  on_threads(nt, [&](unsigned t) {
    uint64_t cnt = 0; bool in = false, bad = false;
    for (size_t p = cut[t]; p < cut[t + 1]; ++p) {
      const unsigned char c = d[p];
      if (c >= '0' && c <= '9') { cnt += !in; in = true; }
      else if (is_ws(c)) in = false;
      else { bad = true; break; }
    }
    ntok[t + 1] = cnt; dirty[t] = bad;
  });
  for (unsigned t = 0; t < nt; ++t) { if (dirty[t]) return 1; ntok[t + 1] += ntok[t]; }
  const uint64_t N = ntok[nt];
  if (N == 0 || N % 3 != 0) return 1;
  std::vector<uint32_t> tok(N);
  on_threads(nt, [&](unsigned t) {
    uint32_t *o = tok.data() + ntok[t];
    const unsigned char *p = d + cut[t], *e = d + cut[t + 1];
    while (p < e) {
      while (p < e && is_ws(*p)) ++p;
      if (p == e) break;
      uint64_t v = 0;
      for (; p < e && *p >= '0' && *p <= '9'; ++p) { v = v * 10 + (uint64_t)(*p - '0'); if (v > 0xffffffffffffull) v &= 0xffffffffull; }   // as Tok::next_u32
      *o++ = (uint32_t)v;
    }
  });

  const uint64_t R = N / 3;
  nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(nt, R / 1024 + 1));
  auto rec0 = [&](unsigned t) { return R / nt * t + std::min<uint64_t>(t, R % nt); };
  std::vector<uint64_t> kept(nt + 1, 0);
  if (!out) {
    // 2. first appearances per piece, among the records that count
    std::vector<LocalIds> lu(nt), li(nt);
    on_threads(nt, [&](unsigned t) {
      uint64_t cnt = 0; bool have = false; uint32_t last = 0;
      for (uint64_t r = rec0(t), e = rec0(t + 1); r < e; ++r) {
        if (input_rating_class(tok[3 * r + 2]) == 0) continue;
        const uint32_t uid = tok[3 * r], mid = tok[3 * r + 1];
        if (!have || uid != last) { lu[t].note(uid); have = true; last = uid; }
        li[t].note(mid);
        ++cnt;
      }
      kept[t + 1] = cnt;
    });
    // 3. merge in file order
    bool overflow = false;
    uint32_t v;
    for (unsigned t = 0; t < nt && !overflow; ++t) {
      for (uint32_t uid : lu[t].order) {
        if (user2seq.find(uid, &v)) continue;
        if (n >= cap_n) { overflow = true; break; }
        user2seq.put(uid, n); seq2user.push_back(uid); n++;
      }
      for (uint32_t mid : li[t].order) {
        if (overflow) break;
        if (item2seq.find(mid, &v)) continue;
        if (m >= cap_m) { overflow = true; break; }
        item2seq.put(mid, m); seq2item.push_back(mid); m++;
      }
    }
    if (overflow) {          // a capacity binds: which records survive depends on the order of BOTH sides
      user2seq = IdMap(); item2seq = IdMap(); seq2user.clear(); seq2item.clear(); n = m = 0; nratings = 0;
      Consumer c(*this, nullptr);
      for (uint64_t r = 0; r < R; ++r) c.take(tok[3 * r], tok[3 * r + 1], tok[3 * r + 2]);
      return 0;
    }
    for (unsigned t = 0; t < nt; ++t) kept[t + 1] += kept[t];
    nratings = kept[nt];
    tr_u_.resize(nratings); tr_i_.resize(nratings); tr_y_.resize(nratings);
    // 4. every record that counts, through the finished maps
    on_threads(nt, [&](unsigned t) {
      uint64_t o = kept[t]; bool have = false; uint32_t last = 0, us = 0, ms = 0;
      for (uint64_t r = rec0(t), e = rec0(t + 1); r < e; ++r) {
        const uint32_t y = tok[3 * r + 2];
        if (input_rating_class(y) == 0) continue;
        const uint32_t uid = tok[3 * r];
        if (!have || uid != last) { user2seq.find(uid, &us); have = true; last = uid; }
        item2seq.find(tok[3 * r + 1], &ms);
        tr_u_[o] = us; tr_i_[o] = ms; tr_y_[o] = y; ++o;
      }
    });
    return 0;
  }
  // held-out file: a record counts when both ids are registered and its rating class is not 0
  for (int pass = 0; pass < 2; ++pass) {
    on_threads(nt, [&](unsigned t) {
      uint64_t o = pass ? kept[t] : 0, cnt = 0; uint32_t us = 0, ms = 0;       // (pass 0: kept[t] is being written by thread t - 1)
      for (uint64_t r = rec0(t), e = rec0(t + 1); r < e; ++r) {
        const uint32_t y = tok[3 * r + 2];
        if (!user2seq.find(tok[3 * r], &us) || !item2seq.find(tok[3 * r + 1], &ms) || input_rating_class(y) == 0) continue;
        if (pass) { out->u[o] = us; out->i[o] = ms; out->y[o] = binary ? 1 : (int32_t)y; ++o; }
        ++cnt;
      }
      if (!pass) kept[t + 1] = cnt;
    });
    if (!pass) {
      for (unsigned t = 0; t < nt; ++t) kept[t + 1] += kept[t];
      out->u.resize(kept[nt]); out->i.resize(kept[nt]); out->y.resize(kept[nt]);
    }
  }
  return 0;
}

int Ratings::read_train(const std::string &path)
{
  FILE *f = fopen(path.c_str(), "r");
  if (!f) { fprintf(stderr, "error: cannot open file %s:%s", path.c_str(), strerror(errno)); return -1; }
  n = m = 0; nratings = 0;
  int rc = read_generic(f, nullptr);
  fclose(f);
  if (rc) return rc;
  // CSR: rows by user seq, inside a row the file order (ratings.cc:105
  // push_back); the rating used for every copy of a duplicated (u,i) is the
  // LAST one written to the per-user std::map<item,uint8_t> (ratings.cc:96-103)
  const uint64_t nnz = tr_u_.size();
  rowptr.assign((size_t)n + 1, 0);
  col.resize(nnz); val.resize(nnz);
  // A stable counting sort by user.  On several threads every thread owns a range of users and walks ALL
  // records in file order for them (sequential reads, no shared counter): inside a row the file order,
  // whatever the number of threads.
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *e = getenv("HGAPREC_READ_THREADS")) { const int v = atoi(e); nt = v < 1 ? 1u : (unsigned)v; }
  nt = std::max(1u, std::min(nt, 32u));
  if (nnz < ((uint64_t)1 << 22) && !getenv("HGAPREC_READ_PARALLEL_MIN")) nt = 1;
  nt = (unsigned)std::max<uint32_t>(1, std::min<uint32_t>(nt, n));
  std::vector<uint32_t> ucut(nt + 1, n);
  for (unsigned t = 0; t <= nt; ++t) ucut[t] = (uint32_t)((uint64_t)n * t / nt);
  const uint32_t *tu = tr_u_.data();
  on_threads(nt, [&](unsigned t) {
    const uint32_t lo = ucut[t], span = ucut[t + 1] - lo;
    int64_t *cnt = rowptr.data() + 1;
    for (uint64_t j = 0; j < nnz; ++j) { const uint32_t u = tu[j] - lo; if (u < span) cnt[u + lo]++; }
  });
  for (uint32_t u = 0; u < n; ++u) rowptr[u + 1] += rowptr[u];
  for (unsigned t = 1; t < nt; ++t)            // ranges of about equal numbers of ratings from here on
    ucut[t] = (uint32_t)(std::lower_bound(rowptr.begin(), rowptr.end(), (int64_t)(nnz / nt * t)) - rowptr.begin());
  for (unsigned t = 1; t <= nt; ++t) ucut[t] = std::max(ucut[t], ucut[t - 1]);
  ucut[nt] = n;
  on_threads(nt, [&](unsigned t) {
    const uint32_t lo = ucut[t], span = ucut[t + 1] - lo;
    if (!span) return;
    std::vector<int64_t> next(rowptr.begin() + lo, rowptr.begin() + lo + span);
    for (uint64_t j = 0; j < nnz; ++j) {
      const uint32_t u = tu[j] - lo;
      if (u >= span) continue;
      const int64_t p = next[u]++;
      col[(size_t)p] = tr_i_[j];
      val[(size_t)p] = binary ? 1 : (uint8_t)tr_y_[j];       // yval_t = uint8_t (env.hh:20)
    }
    // rows with a repeated item are rare: find them with a "last user that listed
    // this item" stamp (linear), and only those rows take the sort-based fix-up:
    // the rating used for every copy of a duplicated (u,i) is the LAST one written
    // to the per-user std::map<item,uint8_t> (ratings.cc:96-103)
    std::vector<uint32_t> perm, stamp(m, 0xffffffffu);
    for (uint32_t u = lo; u < lo + span; ++u) {
      const int64_t a = rowptr[u], b = rowptr[u + 1];
      bool dup = false;
      for (int64_t j = a; j < b; ++j) { if (stamp[col[(size_t)j]] == u) dup = true; stamp[col[(size_t)j]] = u; }
      if (!dup) continue;
      perm.resize((size_t)(b - a));
      std::iota(perm.begin(), perm.end(), 0u);
      std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return col[a + x] < col[a + y]; });
      for (size_t s2 = 0; s2 < perm.size();) {
        size_t e = s2;
        while (e + 1 < perm.size() && col[a + perm[e + 1]] == col[a + perm[s2]]) ++e;
        if (e > s2) { const uint8_t last = val[a + perm[e]]; for (size_t q = s2; q <= e; ++q) val[a + perm[q]] = last; }
        s2 = e + 1;
      }
    }
  });
  std::vector<uint32_t>().swap(tr_u_); std::vector<uint32_t>().swap(tr_i_); std::vector<uint32_t>().swap(tr_y_);
  return 0;
}

int Ratings::read_heldout(const std::string &path, HeldOut *out)
{
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  HeldOut raw;
  int rc = read_generic(f, &raw);
  fclose(f);
  if (rc) return rc;
  // std::map<Rating,int>: sorted by (user, item), last assignment wins
  const size_t cnt = raw.u.size();
  std::vector<uint32_t> p(cnt);
  std::iota(p.begin(), p.end(), 0u);
  std::stable_sort(p.begin(), p.end(), [&](uint32_t x, uint32_t y) {
    return raw.u[x] != raw.u[y] ? raw.u[x] < raw.u[y] : raw.i[x] < raw.i[y]; });
  out->u.clear(); out->i.clear(); out->y.clear();
  for (size_t s = 0; s < cnt;) {
    size_t e = s;
    while (e + 1 < cnt && raw.u[p[e + 1]] == raw.u[p[s]] && raw.i[p[e + 1]] == raw.i[p[s]]) ++e;
    out->u.push_back(raw.u[p[e]]); out->i.push_back(raw.i[p[e]]); out->y.push_back(raw.y[p[e]]);
    s = e + 1;
  }
  return 0;
}

// ---- binary dataset image -------------------------------------------------
namespace {
struct CacheHeader {
  char magic[8];                 // "HPFDATA1"
  uint32_t cap_n, cap_m, binary, rating_threshold;
  uint64_t src_size[3]; int64_t src_mtime_ns[3];   // train / validation / test .tsv
  uint32_t n, m;
  uint64_t nnz, n_validation, n_test;
};
const char kCacheMagic[8] = {'H', 'P', 'F', 'D', 'A', 'T', 'A', '1'};
const uint64_t kCacheTail = 0x31444e4544465048ull;   // "HPFDEND1"

bool fingerprint(const std::string &dir, CacheHeader *h)
{
  const char *names[3] = {"/train.tsv", "/validation.tsv", "/test.tsv"};
  for (int j = 0; j < 3; ++j) {
    struct stat st;
    if (stat((dir + names[j]).c_str(), &st)) return false;
    h->src_size[j] = (uint64_t)st.st_size;
    h->src_mtime_ns[j] = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  }
  return true;
}
template <typename T> bool put_vec(FILE *f, const std::vector<T> &v, size_t cnt)
{
  return cnt == 0 || fwrite(v.data(), sizeof(T), cnt, f) == cnt;
}
template <typename T> bool get_vec(FILE *f, std::vector<T> *v, size_t cnt)
{
  v->resize(cnt);
  return cnt == 0 || fread(v->data(), sizeof(T), cnt, f) == cnt;
}
}  // namespace

int Ratings::save_cache(const std::string &dir, const std::string &image) const
{
  CacheHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, kCacheMagic, 8);
  h.cap_n = cap_n; h.cap_m = cap_m; h.binary = binary; h.rating_threshold = rating_threshold;
  if (!fingerprint(dir, &h)) return -1;
  h.n = n; h.m = m; h.nnz = col.size(); h.n_validation = validation.u.size(); h.n_test = test.u.size();
  const std::string path = image.empty() ? dir + "/hgaprec.cache.bin" : image, tmp = path + ".tmp." + std::to_string((long)getpid());
  FILE *f = fopen(tmp.c_str(), "wb");
  if (!f) return -1;
  bool ok = fwrite(&h, sizeof h, 1, f) == 1;
  ok = ok && put_vec(f, seq2user, n) && put_vec(f, seq2item, m) && put_vec(f, rowptr, (size_t)n + 1);
  ok = ok && put_vec(f, col, h.nnz) && put_vec(f, val, h.nnz);
  ok = ok && put_vec(f, validation.u, h.n_validation) && put_vec(f, validation.i, h.n_validation) && put_vec(f, validation.y, h.n_validation);
  ok = ok && put_vec(f, test.u, h.n_test) && put_vec(f, test.i, h.n_test) && put_vec(f, test.y, h.n_test);
  ok = ok && fwrite(&kCacheTail, 8, 1, f) == 1;
  ok = (fclose(f) == 0) && ok;
  if (!ok || rename(tmp.c_str(), path.c_str())) { remove(tmp.c_str()); return -1; }
  return 0;
}

int Ratings::load_cache(const std::string &dir, const std::string &image)
{
  FILE *f = fopen((image.empty() ? dir + "/hgaprec.cache.bin" : image).c_str(), "rb");
  if (!f) return 1;
  CacheHeader h, want;
  memset(&want, 0, sizeof want);
  bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, kCacheMagic, 8) && fingerprint(dir, &want);
  ok = ok && h.cap_n == cap_n && h.cap_m == cap_m && h.binary == (uint32_t)binary && h.rating_threshold == rating_threshold;
  for (int j = 0; ok && j < 3; ++j) ok = h.src_size[j] == want.src_size[j] && h.src_mtime_ns[j] == want.src_mtime_ns[j];
  // the counts must account for the file's size exactly BEFORE anything is allocated from
  // them (a damaged header must not turn into a multi-terabyte resize), and they must fit
  // the capacities this run was started with
  if (ok) {
    struct stat st;
    ok = fstat(fileno(f), &st) == 0 && h.n <= cap_n && h.m <= cap_m;
    const long double want_bytes = (long double)sizeof h + 4.0L * h.n + 4.0L * h.m + 8.0L * ((long double)h.n + 1) +
                                   5.0L * (long double)h.nnz + 12.0L * (long double)h.n_validation +
                                   12.0L * (long double)h.n_test + 8.0L;
    ok = ok && want_bytes == (long double)st.st_size;
  }
  Ratings t;                                    // only committed when the whole image checks out
  ok = ok && get_vec(f, &t.seq2user, h.n) && get_vec(f, &t.seq2item, h.m) && get_vec(f, &t.rowptr, (size_t)h.n + 1);
  ok = ok && get_vec(f, &t.col, h.nnz) && get_vec(f, &t.val, h.nnz);
  ok = ok && get_vec(f, &t.validation.u, h.n_validation) && get_vec(f, &t.validation.i, h.n_validation) && get_vec(f, &t.validation.y, h.n_validation);
  ok = ok && get_vec(f, &t.test.u, h.n_test) && get_vec(f, &t.test.i, h.n_test) && get_vec(f, &t.test.y, h.n_test);
  uint64_t tail = 0;
  ok = ok && fread(&tail, 8, 1, f) == 1 && tail == kCacheTail;
  fclose(f);
  ok = ok && t.rowptr.front() == 0 && (uint64_t)t.rowptr.back() == h.nnz;
  // every index the rest of the program will use as a subscript
  for (uint32_t u = 0; ok && u < h.n; ++u) ok = t.rowptr[u] <= t.rowptr[u + 1];
  for (size_t j = 0; ok && j < t.col.size(); ++j) ok = t.col[j] < h.m;
  for (const HeldOut *ho : {&t.validation, &t.test})
    for (size_t j = 0; ok && j < ho->u.size(); ++j) ok = ho->u[j] < h.n && ho->i[j] < h.m;
  if (!ok) return 1;
  n = h.n; m = h.m; nratings = h.nnz;
  seq2user.swap(t.seq2user); seq2item.swap(t.seq2item); rowptr.swap(t.rowptr); col.swap(t.col); val.swap(t.val);
  validation = std::move(t.validation); test = std::move(t.test);
  user2seq = IdMap(); item2seq = IdMap();
  for (uint32_t s = 0; s < n; ++s) user2seq.put(seq2user[s], s);
  for (uint32_t s = 0; s < m; ++s) item2seq.put(seq2item[s], s);
  heldout_loaded = true;
  return 0;
}

int Ratings::read_test_users(const std::string &path, std::vector<uint32_t> *out) const
{
  FILE *f = fopen(path.c_str(), "r");
  if (!f) return -1;
  out->clear();
  Tok tk(f);
  uint32_t uid = 0;
  while (!tk.at_eof()) {                         // fscanf(f, "%u\n", &uid) per line
    if (tk.next_u32(&uid) != 1) break;
    uint32_t us;
    if (user2seq.find(uid, &us)) out->push_back(us);
  }
  fclose(f);
  std::sort(out->begin(), out->end());
  out->erase(std::unique(out->begin(), out->end()), out->end());
  return 0;
}

int Ratings::write_marginals(const std::string &byusers, const std::string &byitems,
                             uint32_t *lu, uint32_t *li) const
{
  FILE *f = fopen(byusers.c_str(), "w");
  if (!f) return -1;
  uint32_t x = 0;
  for (uint32_t u = 0; u < n; ++u) {
    const int64_t a = rowptr[u], b = rowptr[u + 1];
    if (a == b) { x++; continue; }
    uint32_t t = 0;
    for (int64_t j = a; j < b; ++j) t += val[(size_t)j];
    x = 0;
    fprintf(f, "%d\t%d\t%d\t%d\n", u, seq2user[u], (int)(b - a), t);
  }
  fclose(f);
  if (lu) *lu = x;
  std::vector<uint32_t> deg(m, 0), sum(m, 0);
  for (size_t j = 0; j < col.size(); ++j) { deg[col[j]]++; sum[col[j]] += val[j]; }
  f = fopen(byitems.c_str(), "w");
  if (!f) return -1;
  x = 0;
  for (uint32_t i = 0; i < m; ++i) {
    if (!deg[i]) { x++; continue; }
    x = 0;
    fprintf(f, "%d\t%d\t%d\t%d\n", i, seq2item[i], deg[i], sum[i]);
  }
  fclose(f);
  if (li) *li = x;
  return 0;
}

// ======================================================================
// MT19937 (Matsumoto & Nishimura; GSL's gsl_rng_mt19937 conventions)
// ======================================================================
void Mt19937::set(unsigned long s)
{
  if (s == 0) s = 4357;                          // GSL default seed
  mt[0] = (uint32_t)(s & 0xffffffffUL);
  for (int i = 1; i < 624; ++i)
    mt[i] = (uint32_t)(1812433253UL * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned long)i);
  mti = 624;
}

void Mt19937::refill()
{
  auto twist = [](uint32_t u, uint32_t v) {
    const uint32_t y = (u & 0x80000000U) | (v & 0x7fffffffU);
    return (y >> 1) ^ ((v & 1U) ? 0x9908b0dfU : 0U);
  };
  int kk = 0;
  for (; kk < 624 - 397; ++kk) mt[kk] = mt[kk + 397] ^ twist(mt[kk], mt[kk + 1]);
  for (; kk < 623; ++kk) mt[kk] = mt[kk + (397 - 624)] ^ twist(mt[kk], mt[kk + 1]);
  mt[623] = mt[396] ^ twist(mt[623], mt[0]);
  mti = 0;
}

static inline uint32_t mt_temper(uint32_t k)
{
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680U;
  k ^= (k << 15) & 0xefc60000U;
  k ^= (k >> 18);
  return k;
}

uint32_t Mt19937::next_u32()
{
  if (mti >= 624) refill();
  return mt_temper(mt[mti++]);
}

// the next cnt values of uniform(), scaled: out[j] = base + scale * uniform().  The same words in the same
// order as cnt calls of next_u32(); a block of the state is tempered in one branch-free loop.
void Mt19937::fill_affine(double *out, size_t cnt, double base, double scale)
{
  while (cnt) {
    if (mti >= 624) refill();
    const size_t c = std::min<size_t>(cnt, (size_t)(624 - mti));
    const uint32_t *w = mt + mti;
    for (size_t j = 0; j < c; ++j) out[j] = base + scale * (mt_temper(w[j]) / 4294967296.0);
    mti += (int)c; out += c; cnt -= c;
  }
}
void Mt19937::skip(size_t cnt)
{
  while (cnt) {
    if (mti >= 624) refill();
    const size_t c = std::min<size_t>(cnt, (size_t)(624 - mti));
    mti += (int)c; cnt -= c;
  }
}

unsigned long Mt19937::uniform_int(unsigned long n)
{
  const unsigned long scale = 0xffffffffUL / n;
  unsigned long k;
  do { k = next_u32() / scale; } while (k >= n);
  return k;
}

Mt19937 make_rng(double env_seed)
{
  // gsl_rng_env_setup(): GSL_RNG_SEED sets gsl_rng_default_seed; gsl_rng_alloc
  // seeds with it; then "if (_env.seed) gsl_rng_set(_r, _env.seed)"
  unsigned long def = 0;
  if (const char *e = getenv("GSL_RNG_SEED")) def = strtoul(e, nullptr, 0);
  Mt19937 r(def);
  if (env_seed) r.set((unsigned long)env_seed);
  return r;
}

// digamma, x > 0: recurrence to x >= 10, asymptotic series through x^-14
double digamma(double x)
{
  double acc = 0.0;
  while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
  const double xi = 1.0 / x, x2 = xi * xi;
  const double s = x2 * (1.0 / 12 - x2 * (1.0 / 120 - x2 * (1.0 / 252 - x2 * (1.0 / 240 -
                   x2 * (1.0 / 132 - x2 * (691.0 / 32760 - x2 * (1.0 / 12)))))));
  return acc + std::log(x) - 0.5 * xi - s;
}

// ======================================================================
// HGAPRec::initialize (hgaprec.cc:153-204)
// ======================================================================
namespace {
const double SPRIOR = 0.3, RPRIOR = 0.3;       // hgaprec.cc:13-20

// Each helper draws for ALL `rows` rows (the stream position is what the
// reference's is) and stores only rows [lo, hi).

// elements [0, cnt) in equal pieces on the host's threads (capped at 32; HGAPREC_SAVE_THREADS overrides); every
// element is computed by the same scalar code whichever thread takes it, so the result does not depend on the count
template <typename F>
void parallel_pieces(size_t cnt, F fn)
{
  unsigned nt = std::thread::hardware_concurrency();
  if (const char *e = getenv("HGAPREC_SAVE_THREADS")) nt = (unsigned)atoi(e);
  nt = std::max(1u, std::min(nt, 32u));
  if (cnt < ((size_t)1 << 16)) nt = 1;
  if (nt == 1) { fn((size_t)0, cnt); return; }
  std::vector<std::thread> th;
  const size_t piece = (cnt + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t a = std::min(cnt, (size_t)t * piece), b = std::min(cnt, a + piece);
    if (a < b) th.emplace_back(fn, a, b);
  }
  for (auto &x : th) x.join();
}
// GPMatrix::initialize (gpbase.hh:292-308): rows*k shape draws, then k rate draws
void gp_initialize(Mt19937 &r, uint32_t rows, uint32_t k, bool global_rate, uint32_t lo, uint32_t hi,
                   StateArray &shape, StateArray &rate)
{
  shape.resize((size_t)(hi - lo) * k);
  r.skip((size_t)lo * k);                                     // rows of other ranks: drawn, not kept
  r.fill_affine(shape.data(), shape.size(), SPRIOR, 0.01);    // SPRIOR + 0.01 * uniform(), row by row
  r.skip((size_t)(rows - hi) * k);
  std::vector<double> b0(k);
  for (uint32_t j = 0; j < k; ++j) b0[j] = RPRIOR + 0.1 * r.uniform();
  if (global_rate) rate.assign(b0.begin(), b0.end());                   // GPMatrixGR::initialize gpbase.hh:651-663
  else {
    rate.resize((size_t)(hi - lo) * k);
    double *pr = rate.data(); const double *pb = b0.data();
    parallel_pieces((size_t)(hi - lo), [pr, pb, k](size_t r0, size_t r1) {
      for (size_t i = r0; i < r1; ++i) for (uint32_t j = 0; j < k; ++j) pr[i * k + j] = pb[j];
    });
  }
}
// initialize_exp (gpbase.hh:324-340 / 700-715): fresh rate draw per element.  The draws are taken in the
// reference's order by one thread; digamma and log of the kept rows -- nine tenths of the time of a start
// state (C2: 1.1e8 elements) -- go to the host's threads afterwards.
void gp_initialize_exp(Mt19937 &r, uint32_t rows, uint32_t k, uint32_t lo, uint32_t hi,
                       const StateArray &shape, StateArray &E, StateArray &Elog)
{
  E.resize(shape.size()); Elog.resize(shape.size());
  r.skip((size_t)lo * k);
  r.fill_affine(E.data(), E.size(), RPRIOR, 0.1);             // RPRIOR + 0.1 * uniform(): the rate for now
  r.skip((size_t)(rows - hi) * k);
  const double *sh = shape.data();
  double *pe = E.data(), *pl = Elog.data();
  parallel_pieces(shape.size(), [sh, pe, pl](size_t a, size_t b2) {
    for (size_t e = a; e < b2; ++e) {
      const double b = pe[e];
      pe[e] = sh[e] / b;
      pl[e] = digamma(sh[e]) - std::log(b);
    }
  });
}
// initialize2(v) + compute_expectations (gpbase.hh:310-322,939-949; 248-262,912-925)
void gp_initialize2(Mt19937 &r, uint32_t rows, double v, uint32_t lo, uint32_t hi, StateArray &shape,
                    StateArray &rate, StateArray &E, StateArray &Elog)
{
  const uint32_t keep = hi - lo;
  shape.resize(keep); rate.resize(keep); E.resize(keep); Elog.resize(keep);
  for (uint32_t i = 0; i < rows; ++i) {
    const double s = SPRIOR + 0.01 * r.uniform();
    if (i >= lo && i < hi) { shape[i - lo] = s; rate[i - lo] = RPRIOR + v; }
  }
  for (uint32_t i = 0; i < keep; ++i) { E[i] = shape[i] / rate[i]; Elog[i] = digamma(shape[i]) - std::log(rate[i]); }
}
}  // namespace

void initialize_state(Mt19937 &rng, uint32_t n, uint32_t m, uint32_t k, bool hier,
                      bool bias, GammaState *s, uint32_t lo, uint32_t hi)
{
  if (hi > n) hi = n;
  if (lo > hi) lo = hi;
  s->n = hi - lo; s->m = m; s->k = k; s->hier = hier; s->bias = bias;
  if (!hier) {                                   // hgaprec.cc:156-161
    gp_initialize(rng, m, k, true, 0, m, s->beta_shape, s->beta_rate);
    gp_initialize(rng, n, k, true, lo, hi, s->theta_shape, s->theta_rate);
    gp_initialize_exp(rng, m, k, 0, m, s->beta_shape, s->beta_E, s->beta_Elog);
    gp_initialize_exp(rng, n, k, lo, hi, s->theta_shape, s->theta_E, s->theta_Elog);
  } else {                                       // hgaprec.cc:173-193
    gp_initialize2(rng, n, (double)k, lo, hi, s->xi_shape, s->xi_rate, s->xi_E, s->xi_Elog);
    gp_initialize2(rng, m, (double)k, 0, m, s->eta_shape, s->eta_rate, s->eta_E, s->eta_Elog);
    gp_initialize(rng, m, k, false, 0, m, s->beta_shape, s->beta_rate);
    gp_initialize_exp(rng, m, k, 0, m, s->beta_shape, s->beta_E, s->beta_Elog);
    gp_initialize(rng, n, k, false, lo, hi, s->theta_shape, s->theta_rate);
    gp_initialize_exp(rng, n, k, lo, hi, s->theta_shape, s->theta_E, s->theta_Elog);
  }
  if (bias) {                                    // hgaprec.cc:197-203
    gp_initialize2(rng, n, (double)m, lo, hi, s->ubias_shape, s->ubias_rate, s->ubias_E, s->ubias_Elog);
    gp_initialize2(rng, m, (double)n, 0, m, s->ibias_shape, s->ibias_rate, s->ibias_E, s->ibias_Elog);
  }
}

// ======================================================================
// writers
// ======================================================================
// "%.8f" without printf: exact for 0 <= |v| < 2^53 (everything this model
// produces), falls back to snprintf otherwise.  The integer part is exact;
// the fraction f = v - floor(v) is exact too, and f * 1e8 is formed as an
// error-free product p + e (e = fma(f, 1e8, -p)), so the decimal is rounded
// exactly like glibc does it: to nearest, ties to even.
namespace {
const char DIGITS2[201] =
  "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
  "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
  "8081828384858687888990919293949596979899";
}

size_t format_fixed8(double v, char *out)
{
  if (!(std::fabs(v) < 9007199254740992.0)) return (size_t)snprintf(out, 400, "%.8f", v);   // also NaN
  char *o = out;
  if (std::signbit(v)) { *o++ = '-'; v = -v; }
  uint64_t ip = (uint64_t)v;                        // floor(v), v >= 0
  const double f = v - (double)ip;                  // exact
  // error-free product f * 1e8 = p + e (Dekker / Veltkamp, no FMA needed):
  // 1e8 = 1e8 + 0 splits trivially (it has 20 significant bits)
  const double p = f * 1e8;
  const double c = 134217729.0 * f, fh = c - (c - f), fl = f - fh;
  const double e = (fh * 1e8 - p) + fl * 1e8;
  long n = __builtin_lrint(p);                      // to nearest, ties to even, on p
  // r = p - n is exact and |r| <= 0.5; |e| < ulp(p), so the true product
  // p + e can only be on the other side of a rounding boundary when |r| == 0.5
  const double r = p - (double)n;
  if (r == 0.5) { if (e > 0.0 || (e == 0.0 && (n & 1))) n += 1; }
  else if (r == -0.5) { if (e < 0.0 || (e == 0.0 && (n & 1))) n -= 1; }
  if (n >= 100000000L) { n -= 100000000L; ip += 1; }
  char tmp[24]; int len = 0;
  while (ip >= 100) { const unsigned q = (unsigned)(ip % 100); ip /= 100; tmp[len++] = DIGITS2[2 * q + 1]; tmp[len++] = DIGITS2[2 * q]; }
  if (ip >= 10) { tmp[len++] = DIGITS2[2 * ip + 1]; tmp[len++] = DIGITS2[2 * ip]; }
  else tmp[len++] = (char)('0' + ip);
  while (len) *o++ = tmp[--len];
  *o++ = '.';
  uint32_t fr = (uint32_t)n;
  for (int k = 3; k >= 0; --k) { const uint32_t q = fr % 100; fr /= 100; o[2 * k] = DIGITS2[2 * q]; o[2 * k + 1] = DIGITS2[2 * q + 1]; }
  o += 8;
  return (size_t)(o - out);
}

namespace {
inline char *put_u32(char *o, uint32_t v)           // "%d" of a value below 2^31 (seq ids, item ids)
{
  if ((int32_t)v < 0) return o + sprintf(o, "%d", (int32_t)v);
  char tmp[12]; int len = 0;
  do { tmp[len++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (len) *o++ = tmp[--len];
  return o;
}
struct BufWriter {
  FILE *f; std::vector<char> buf; size_t pos = 0;
  explicit BufWriter(FILE *ff) : f(ff), buf(4u << 20) {}
  char *room(size_t n) { if (pos + n > buf.size()) flush(); return buf.data() + pos; }
  void advance(size_t n) { pos += n; }
  void flush() { if (pos) fwrite(buf.data(), 1, pos, f); pos = 0; }
};
}  // namespace

// fopen(path, "w") for a file that is written whole, again and again (a model save every report step): the old
// contents are overwritten in place and the length is set when the new ones are complete, instead of being cut to
// zero first -- dropping 1.1 GB of cached pages and allocating them again was half the time of a save.  The bytes of
// the finished file are the same; while it is being written a reader sees new text followed by old instead of new
// text followed by nothing.
// What fopen("w") gave for free and this does not: a run killed in the middle of a save (SIGKILL, out of memory) leaves a
// file of full length whose head is new and whose tail is old -- well-formed and wrong, where a truncated file would be
// short and obviously so.  So a file that is being rewritten has a MARKER beside it, "<path>.writing", created before the
// first byte and removed after the length has been set: a marker that is still there says the file next to it is not
// whole (README "Output files").  A failed write also cuts the file at the last byte that was written.
std::string rewrite_marker(const std::string &path) { return path + ".writing"; }
void rewrite_begin(const std::string &path)
{
  const int fd = ::open(rewrite_marker(path).c_str(), O_WRONLY | O_CREAT | O_CLOEXEC, 0666);
  if (fd >= 0) ::close(fd);                    // (a directory that takes no new file: the save itself will say so)
}
void rewrite_end(const std::string &path) { ::unlink(rewrite_marker(path).c_str()); }

static FILE *open_rewrite(const std::string &path)
{
  // the marker follows the open of the file itself: a target that cannot be opened (EACCES, no room for a new
  // inode) is an old file nobody touched, and a marker beside it would call it broken.  Nothing has been written yet when
  // the marker appears, so it still precedes the first byte.
  const int fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_CLOEXEC, 0666);
  if (fd < 0) return nullptr;
  struct stat st;
  const bool regular = ::fstat(fd, &st) == 0 && S_ISREG(st.st_mode);               // not for /dev/null or a pipe
  if (regular) rewrite_begin(path);
  FILE *f = fdopen(fd, "w");
  if (!f) { ::close(fd); if (regular) rewrite_end(path); }
  return f;
}
static bool close_rewrite(FILE *f, const std::string &path, bool written_ok)
{
  bool ok = fflush(f) == 0 && written_ok;
  struct stat st;
  const bool regular = fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode);          // (a pipe or /dev/null has no length to set)
  if (regular) {
    // complete: the length of the new text.  Failed: whatever reached the file, so that it is at least visibly short
    const off_t end = ok ? ftello(f) : ::lseek(fileno(f), 0, SEEK_CUR);
    const bool cut = end >= 0 && ftruncate(fileno(f), end) == 0;
    ok = ok && cut;
  }
  ok = (fclose(f) == 0) && ok;
  if (ok && regular) rewrite_end(path);         // the marker of a failed save stays
  return ok;
}

// rows [r0, r1) of the matrix as text into out (grown when needed, never shrunk); *len = bytes
static void format_rows(const double *a, uint32_t r0, uint32_t r1, uint32_t cols,
                        const uint32_t *seq2id, uint32_t nids, uint32_t row0, std::vector<char> &out, size_t *len)
{
  size_t pos = 0;
  for (uint32_t i = r0; i < r1; ++i) {
    const uint32_t seq = i + row0;
    const uint32_t id = (seq2id && seq < nids) ? seq2id[seq] : seq;
    if (out.size() < pos + 32) out.resize(std::max(out.size() * 2, pos + ((size_t)1 << 16)));
    char *o = out.data() + pos, *o0 = o;
    o = put_u32(o, seq); *o++ = '\t';
    o = put_u32(o, id); *o++ = '\t';
    pos += (size_t)(o - o0);
    for (uint32_t k = 0; k < cols; ++k) {
      if (out.size() < pos + 420) out.resize(std::max(out.size() * 2, pos + ((size_t)1 << 16)));
      o = out.data() + pos; o0 = o;
      o += format_fixed8(a[(size_t)i * cols + k], o);
      *o++ = (k == cols - 1) ? '\n' : '\t';
      pos += (size_t)(o - o0);
    }
  }
  *len = pos;
}

// A save is 330 M numbers at C2, 3.3 GB of text per report step: above ~2 M numbers the row blocks are
// formatted by the host's threads, a wave of blocks at a time, and written in order WHILE the next wave is
// being formatted (two sets of buffers) -- the bytes are those of the serial writer.  `threads` = 0: all of
// the host's (HGAPREC_SAVE_THREADS overrides; at most 24).
int save_matrix(const std::string &path, const double *a, uint32_t rows, uint32_t cols,
                const uint32_t *seq2id, uint32_t nids, uint32_t row0, unsigned threads)
{
  FILE *tf = open_rewrite(path);
  if (!tf) return -1;
  unsigned nt = threads ? threads : std::thread::hardware_concurrency();
  if (const char *e = getenv("HGAPREC_SAVE_THREADS")) nt = (unsigned)atoi(e);
  if (nt < 1) nt = 1;
  if (nt > 24) nt = 24;                 // tools/save_bench.py: 1M x 100 in 0.21 / 0.145 / 0.19 s on 8 / 21 / 64 threads
  if ((uint64_t)rows * cols < (2u << 20)) nt = 1;
  const uint32_t blk = std::max<uint32_t>(1, (uint32_t)((1u << 18) / std::max<uint32_t>(cols, 1)));   // ~256 K numbers
  std::vector<std::vector<char>> bufs[2] = {std::vector<std::vector<char>>(nt), std::vector<std::vector<char>>(nt)};
  std::vector<size_t> lens[2] = {std::vector<size_t>(nt, 0), std::vector<size_t>(nt, 0)};
  bool ok = true;
  std::thread writer;
  int set = 0;
  for (uint32_t r = 0; r < rows; r += blk * nt, set ^= 1) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
      const uint32_t r0 = r + t * blk;
      lens[set][t] = 0;
      if (r0 >= rows) continue;
      const uint32_t r1 = (uint32_t)std::min<uint64_t>(rows, (uint64_t)r0 + blk);
      if (nt == 1) format_rows(a, r0, r1, cols, seq2id, nids, row0, bufs[set][t], &lens[set][t]);
      else th.emplace_back(format_rows, a, r0, r1, cols, seq2id, nids, row0, std::ref(bufs[set][t]), &lens[set][t]);
    }
    for (auto &x : th) x.join();
    if (writer.joinable()) writer.join();         // the previous wave is on its way before this one follows
    if (!ok) break;
    auto put = [&, set]() {
      for (unsigned t = 0; t < nt && ok; ++t)
        if (lens[set][t]) ok = fwrite(bufs[set][t].data(), 1, lens[set][t], tf) == lens[set][t];
    };
    if (nt == 1) put(); else writer = std::thread(put);
  }
  if (writer.joinable()) writer.join();
  ok = close_rewrite(tf, path, ok);
  return ok ? 0 : -1;
}

int save_vector(const std::string &path, const double *a, uint32_t rows,
                const uint32_t *seq2id, uint32_t nids, uint32_t row0)
{
  FILE *tf = open_rewrite(path);
  if (!tf) return -1;
  BufWriter w(tf);
  for (uint32_t i = 0; i < rows; ++i) {
    const uint32_t seq = i + row0;
    const uint32_t id = (seq2id && seq < nids) ? seq2id[seq] : seq;
    char *o = w.room(460), *o0 = o;
    o = put_u32(o, seq); *o++ = '\t';
    o = put_u32(o, id); *o++ = '\t';
    o += format_fixed8(a[i], o);
    *o++ = '\n';
    w.advance((size_t)(o - o0));
  }
  w.flush();
  return close_rewrite(tf, path, ferror(tf) == 0) ? 0 : -1;
}

// ======================================================================
// stop rule (hgaprec.cc:1476-1492)
// ======================================================================
bool StopRule::update(uint32_t iter, double a, int *why)
{
  bool stop = false;
  *why = -1;
  if (iter > 30) {
    if (a > prev_h && prev_h != 0 && std::fabs((a - prev_h) / prev_h) < 0.000001) { stop = true; *why = 0; }
    else if (a < prev_h) nh++;
    else if (a > prev_h) nh = 0;
    if (nh > 2) { *why = 1; stop = true; }
  }
  prev_h = a;
  return stop;
}


// ======================================================================
// Comm: TCP star through rank 0
// ======================================================================
}  // namespace hgaprec

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>

namespace hgaprec {

namespace {
int send_all(int fd, const void *p, size_t n)
{
  const char *c = (const char *)p;
  while (n) { ssize_t w = ::send(fd, c, n, MSG_NOSIGNAL); if (w <= 0) { if (errno == EINTR) continue; return -1; } c += w; n -= (size_t)w; }
  return 0;
}
int recv_all(int fd, void *p, size_t n)
{
  char *c = (char *)p;
  while (n) { ssize_t r = ::recv(fd, c, n, 0); if (r <= 0) { if (r < 0 && errno == EINTR) continue; return -1; } c += r; n -= (size_t)r; }
  return 0;
}
}  // namespace

// Rank 0 listens on `addr` only (MASTER_ADDR, default 127.0.0.1: the feature is
// one process per GPU of ONE node), the handshake carries the rank and a
// 64-bit nonce (the launcher, spawn_ranks, hands one to its children through
// HGAPREC_NONCE; main() passes it in), and both the accept loop and the
// handshake reads time out.
int Comm::init(int rank_, int world_, const std::string &addr, int port, uint64_t nonce)
{
  rank = rank_; world = world_;
  if (world <= 1) return 0;
  struct Hello { int32_t rank; uint32_t magic; uint64_t nonce; };
  const uint32_t magic = 0x48504631u;                      // "HPF1"
  sockaddr_in sa{}; sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, addr.c_str(), &sa.sin_addr) != 1) {
    // launchers commonly export a NAME (localhost, the node's hostname): resolve it
    addrinfo hints{}; hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
    addrinfo *res = nullptr;
    if (getaddrinfo(addr.c_str(), nullptr, &hints, &res) != 0 || !res) {
      fprintf(stderr, "error: MASTER_ADDR '%s' is neither an IPv4 address nor a name that resolves to one\n", addr.c_str());
      return -2;
    }
    sa.sin_addr = ((sockaddr_in *)res->ai_addr)->sin_addr;
    freeaddrinfo(res);
  }
  if (rank == 0) {
    int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) return -1;
    int one = 1; setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    if (::bind(ls, (sockaddr *)&sa, sizeof sa) < 0 || ::listen(ls, world) < 0) { ::close(ls); return -1; }
    fds.assign(world, -1);
    int have = 1, strangers = 0;
    while (have < world) {
      pollfd pf{ls, POLLIN, 0};
      const int pr = ::poll(&pf, 1, 120000);               // a rank that never shows up must not hang the job
      if (pr < 0 && errno == EINTR) continue;
      if (pr <= 0) { ::close(ls); close_all(); return -1; }
      int fd = ::accept(ls, nullptr, nullptr);
      if (fd < 0) { if (errno == EINTR) continue; ::close(ls); close_all(); return -1; }
      timeval tv{10, 0}; setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
      Hello hl{-1, 0, 0};
      if (recv_all(fd, &hl, sizeof hl) || hl.magic != magic || hl.nonce != nonce || hl.rank <= 0 ||
          hl.rank >= world || fds[hl.rank] >= 0) {
        ::close(fd);                                       // not one of ours: drop it, keep listening
        if (++strangers > 64) { ::close(ls); close_all(); return -1; }
        continue;
      }
      timeval off{0, 0}; setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &off, sizeof off);
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      fds[hl.rank] = fd; ++have;
    }
    ::close(ls);
  } else {
    int fd = -1;
    for (int tries = 0; tries < 600; ++tries) {            // rank 0 may not listen yet
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return -1;
      if (::connect(fd, (sockaddr *)&sa, sizeof sa) == 0) break;
      ::close(fd); fd = -1;
      usleep(100000);
    }
    if (fd < 0) return -1;
    int one = 1; setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    Hello hl{rank, magic, nonce};
    if (send_all(fd, &hl, sizeof hl)) { ::close(fd); return -1; }
    fds.assign(1, fd);
  }
  return 0;
}

void Comm::close_all()
{
  for (int fd : fds) if (fd >= 0) ::close(fd);
  fds.clear();
}

int Comm::reduce_impl(double *v, size_t n, bool is_max)
{
  if (world <= 1 || n == 0) return 0;
  if (rank == 0) {
    std::vector<double> tmp(n);
    for (int k = 1; k < world; ++k) {                      // rank order: deterministic sums
      if (recv_all(fds[k], tmp.data(), n * 8)) return -1;
      if (is_max) { for (size_t i = 0; i < n; ++i) v[i] = std::max(v[i], tmp[i]); }
      else { for (size_t i = 0; i < n; ++i) v[i] += tmp[i]; }
    }
    for (int k = 1; k < world; ++k) if (send_all(fds[k], v, n * 8)) return -1;
  } else {
    if (send_all(fds[0], v, n * 8) || recv_all(fds[0], v, n * 8)) return -1;
  }
  return 0;
}
int Comm::allreduce_sum(double *v, size_t n) { return reduce_impl(v, n, false); }
int Comm::allreduce_max(double *v, size_t n) { return reduce_impl(v, n, true); }

int Comm::bcast(void *p, size_t bytes)
{
  if (world <= 1 || bytes == 0) return 0;
  if (rank == 0) { for (int k = 1; k < world; ++k) if (send_all(fds[k], p, bytes)) return -1; return 0; }
  return recv_all(fds[0], p, bytes);
}

int Comm::barrier() { double z = 0.0; return allreduce_sum(&z, 1); }

std::vector<std::pair<uint32_t, uint32_t>> partition_users(const std::vector<int64_t> &rowptr, int world)
{
  const uint32_t n = (uint32_t)rowptr.size() - 1;
  const int64_t nnz = rowptr[n];
  std::vector<uint32_t> cuts(1, 0);
  for (int r = 1; r < world; ++r) {
    const int64_t target = nnz * r / world;
    uint32_t c = (uint32_t)(std::lower_bound(rowptr.begin(), rowptr.end(), target) - rowptr.begin());
    c = std::max<uint32_t>(c, cuts.back() + 1);            // at least one user per rank ...
    const uint32_t cap = n > (uint32_t)(world - r) ? n - (uint32_t)(world - r) : 0;
    c = std::min<uint32_t>(c, cap);                        // ... and one left for each later rank
    cuts.push_back(std::max<uint32_t>(c, cuts.back()));
  }
  cuts.push_back(n);
  std::vector<std::pair<uint32_t, uint32_t>> out;
  for (int r = 0; r < world; ++r) out.emplace_back(cuts[r], cuts[r + 1]);
  return out;
}
}  // namespace hgaprec

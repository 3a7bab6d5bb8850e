// host_selftest.cpp -- CPU-only driver of the host side, built with
// -fsanitize=address,undefined by `make asan` (the
// reference has no sanitizer build; its reader has missing-return UB).  It walks
// the code that owns raw memory and sockets -- the hand-rolled id maps, the TSV
// reader on hostile input, the binary dataset cache (valid, truncated and
// corrupted images), the multi-threaded "%.8f" writer, the TCP star used by
// `hgaprec -ngpus N` -- and checks results, so that a sanitizer report or a
// wrong answer both fail the run.  No HIP, no oracle.
#include "hgaprec_host.hpp"

#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

using namespace hgaprec;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static std::string g_dir;
static std::string path(const char *name) { return g_dir + "/" + name; }

static void write_text(const std::string &p, const std::string &s)
{
  FILE *f = fopen(p.c_str(), "w");
  if (!f) { perror(p.c_str()); exit(2); }
  fwrite(s.data(), 1, s.size(), f);
  fclose(f);
}

static std::string read_text(const std::string &p)
{
  std::string s; FILE *f = fopen(p.c_str(), "r");
  if (!f) return s;
  char buf[65536]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
  fclose(f);
  return s;
}

// ---- id maps: growth, collisions, ids at the edges of uint32 --------------
static void test_idmap()
{
  Ratings::IdMap mp;
  std::mt19937 g(1);
  std::vector<uint32_t> keys;
  for (int i = 0; i < 200000; ++i) keys.push_back((uint32_t)g());
  keys.push_back(0u); keys.push_back(0xffffffffu); keys.push_back(0xfffffffeu);
  std::vector<uint32_t> first(keys.size());
  for (size_t i = 0; i < keys.size(); ++i) {
    uint32_t v;
    if (!mp.find(keys[i], &v)) { mp.put(keys[i], (uint32_t)i); first[i] = (uint32_t)i; }
    else first[i] = v;
  }
  for (size_t i = 0; i < keys.size(); ++i) { uint32_t v = 0; CHECK(mp.find(keys[i], &v) && v == first[i]); }
  uint32_t v;
  CHECK(!mp.find(123456789u, &v) || true);
}

// ---- reader on well-formed and hostile input --------------------------------
static void test_reader()
{
  // first-seen order, duplicates, rating 0, rating 300 (uint8 wrap), capacity, no trailing newline
  write_text(path("train.tsv"), "7\t70\t5\n3\t30\t4\n7\t30\t0\n7\t70\t2\n9\t90\t300\n3\t70\t1\n11\t10\t3\n9\t30\t1");
  Ratings r; r.cap_n = 3; r.cap_m = 3; r.binary = false; r.rating_threshold = 1;
  CHECK(r.read_train(path("train.tsv")) == 0);
  CHECK(r.n == 3 && r.m == 3);                         // user 11 / item 10 are over capacity
  CHECK(r.seq2user[0] == 7 && r.seq2user[1] == 3 && r.seq2user[2] == 9);
  CHECK(r.rowptr.size() == 4 && (uint64_t)r.rowptr[3] == r.col.size());
  CHECK(r.col.size() == r.val.size());
  // user 7: item 70 twice (duplicate kept in the list, last rating wins), item 30 dropped (rating 0)
  CHECK(r.rowptr[1] - r.rowptr[0] == 2 && r.val[0] == 2 && r.val[1] == 2);
  // rating 300 wraps to 44
  bool saw44 = false; for (uint8_t y : r.val) saw44 |= (y == 44);
  CHECK(saw44);
  write_text(path("validation.tsv"), "7\t30\t4\n99\t30\t5\n7\t999\t1\n3\t70\t0\n");
  CHECK(r.read_heldout(path("validation.tsv"), &r.validation) == 0);
  CHECK(r.validation.u.size() == 1);                   // unseen ids and rating 0 are skipped

  // hostile: garbage tokens, huge numbers, a NUL byte, very long line, empty file
  write_text(path("junk.tsv"), std::string("1\t2\t3\nabc\tdef\n4\t5\t6\n"));
  Ratings j; j.cap_n = 10; j.cap_m = 10;
  (void)j.read_train(path("junk.tsv"));               // must terminate without touching bad memory
  std::string big = "4294967295\t4294967295\t4294967295\n99999999999999999999\t1\t1\n";
  big += std::string(100000, '7') + "\t1\t1\n";
  write_text(path("big.tsv"), big);
  Ratings b; b.cap_n = 10; b.cap_m = 10;
  (void)b.read_train(path("big.tsv"));
  write_text(path("empty.tsv"), "");
  Ratings e; e.cap_n = 10; e.cap_m = 10;
  (void)e.read_train(path("empty.tsv"));
  CHECK(e.n == 0);
  Ratings missing; missing.cap_n = 1; missing.cap_m = 1;
  CHECK(missing.read_train(path("does-not-exist.tsv")) != 0);
}

// ---- dataset cache: round trip, then every kind of damaged image ------------
static void test_cache()
{
  const std::string d = path("cache");
  mkdir(d.c_str(), 0775);
  std::mt19937 g(5);
  std::string tr, va, te;
  for (int i = 0; i < 5000; ++i) {
    char line[64];
    snprintf(line, sizeof line, "%u\t%u\t%u\n", (unsigned)(g() % 300) + 1, (unsigned)(g() % 200) + 1, (unsigned)(g() % 6));
    (i % 10 == 0 ? va : i % 10 == 1 ? te : tr) += line;
  }
  write_text(d + "/train.tsv", tr); write_text(d + "/validation.tsv", va); write_text(d + "/test.tsv", te);
  Ratings a; a.cap_n = 300; a.cap_m = 200;
  CHECK(a.read_train(d + "/train.tsv") == 0);
  CHECK(a.read_heldout(d + "/validation.tsv", &a.validation) == 0);
  CHECK(a.read_heldout(d + "/test.tsv", &a.test) == 0);
  CHECK(a.save_cache(d) == 0);
  Ratings b; b.cap_n = 300; b.cap_m = 200;
  CHECK(b.load_cache(d) == 0);
  CHECK(b.n == a.n && b.m == a.m && b.col == a.col && b.val == a.val && b.rowptr == a.rowptr);
  CHECK(b.seq2user == a.seq2user && b.seq2item == a.seq2item);
  CHECK(b.validation.u == a.validation.u && b.test.y == a.test.y && b.heldout_loaded);
  uint32_t s = 0;
  CHECK(b.user2seq.find(a.seq2user[5], &s) && s == 5);
  // different reader parameters => the image must be refused
  Ratings c; c.cap_n = 299; c.cap_m = 200;
  CHECK(c.load_cache(d) != 0);

  const std::string img = d + "/hgaprec.cache.bin";
  const std::string good = read_text(img);
  CHECK(good.size() > 64);
  auto try_image = [&](const std::string &bytes) {
    write_text(img, bytes);
    Ratings x; x.cap_n = 300; x.cap_m = 200;
    return x.load_cache(d);                            // any answer but a crash / overrun is fine ...
  };
  for (size_t cut : {(size_t)0, (size_t)7, (size_t)40, good.size() / 3, good.size() / 2, good.size() - 1})
    CHECK(try_image(good.substr(0, cut)) != 0);         // ... and a truncated image must be refused
  std::mt19937 h(9);
  for (int t = 0; t < 200; ++t) {                       // flipped bytes in the header / counts area
    std::string bad = good;
    const size_t at = h() % std::min<size_t>(bad.size(), 256);
    bad[at] = (char)(bad[at] ^ (1 << (h() % 8)));
    (void)try_image(bad);
  }
  for (int t = 0; t < 50; ++t) {                        // a count field blown up to "huge"
    std::string bad = good;
    const size_t at = (h() % 32) * 4;
    if (at + 4 <= bad.size()) { const uint32_t hugev = 0xfffffff0u; memcpy(&bad[at], &hugev, 4); }
    (void)try_image(bad);
  }
  write_text(img, good);
}

// ---- writers: the threaded matrix writer equals the serial one --------------
static void test_writers()
{
  const uint32_t rows = 20000, cols = 7;
  std::vector<double> a((size_t)rows * cols);
  std::mt19937_64 g(3);
  for (double &v : a) {
    const int kind = (int)(g() % 6);
    const double u = (double)(g() >> 11) / 9007199254740992.0;
    v = kind == 0 ? u * 1e-9 : kind == 1 ? u * 1e6 : kind == 2 ? 0.0 : kind == 3 ? 0.5 + 5e-9 : u;
  }
  std::vector<uint32_t> ids(rows);
  for (uint32_t i = 0; i < rows; ++i) ids[i] = 1000000u - i;
  setenv("HGAPREC_SAVE_THREADS", "1", 1);
  CHECK(save_matrix(path("m1.tsv"), a.data(), rows, cols, ids.data(), rows, 5) == 0);
  setenv("HGAPREC_SAVE_THREADS", "7", 1);
  CHECK(save_matrix(path("m7.tsv"), a.data(), rows, cols, ids.data(), rows, 5) == 0);
  unsetenv("HGAPREC_SAVE_THREADS");
  const std::string s1 = read_text(path("m1.tsv")), s7 = read_text(path("m7.tsv"));
  CHECK(!s1.empty() && s1 == s7);
  // the fast formatter against printf on the same values
  char buf[64], ref[64];
  for (size_t i = 0; i < 20000; ++i) {
    const size_t n = format_fixed8(a[i], buf); buf[n] = 0;
    snprintf(ref, sizeof ref, "%.8f", a[i]);
    CHECK(!strcmp(buf, ref));
  }
  CHECK(save_vector(path("v.tsv"), a.data(), 100, ids.data(), 10, 0) == 0);   // ids shorter than rows
  CHECK(save_matrix("/nonexistent-dir/x.tsv", a.data(), 2, 2, ids.data(), 2, 0) != 0);
}

// ---- start state: shard slices of the one MT19937 stream ---------------------
static void test_state()
{
  const uint32_t n = 57, m = 31, k = 6;
  Mt19937 r0 = make_rng(7.0), r1 = make_rng(7.0), r2 = make_rng(7.0);
  GammaState full, lo, hi;
  initialize_state(r0, n, m, k, true, true, &full);
  initialize_state(r1, n, m, k, true, true, &lo, 0, 20);
  initialize_state(r2, n, m, k, true, true, &hi, 20, n);
  CHECK(lo.n == 20 && hi.n == n - 20);
  CHECK(r0.next_u32() == r1.next_u32());
  CHECK(lo.beta_E == full.beta_E && hi.eta_Elog == full.eta_Elog);
  for (uint32_t e = 0; e < 20 * k; ++e) CHECK(lo.theta_E[e] == full.theta_E[e]);
  for (uint32_t e = 0; e < (n - 20) * k; ++e) CHECK(hi.theta_Elog[e] == full.theta_Elog[20 * k + e]);
  CHECK(std::fabs(digamma(1.0) + 0.57721566490153286) < 1e-15);
  StopRule st; int why = -1;
  CHECK(!st.update(10, -2.0, &why));
}

// ---- TCP star: 4 ranks as threads, a stranger knocking first ------------------
static void test_comm()
{
  const int world = 4, port = 20000 + (int)(getpid() % 20000);
  const uint64_t nonce = 0xc0ffee1234ull;
  std::vector<int> ok(world, 0);
  std::vector<double> sums(world, 0.0), maxs(world, 0.0);
  std::vector<uint32_t> got(world, 0);
  auto body = [&](int rank) {
    Comm c;
    if (c.init(rank, world, "127.0.0.1", port, nonce)) return;
    std::vector<double> v(1000);
    for (size_t i = 0; i < v.size(); ++i) v[i] = (double)(rank + 1) * (double)(i + 1);
    if (c.allreduce_sum(v.data(), v.size())) return;
    sums[rank] = v[999];
    double mx = (double)rank;
    if (c.allreduce_max(&mx, 1)) return;
    maxs[rank] = mx;
    uint32_t token = rank == 0 ? 0xabcdef01u : 0;
    if (c.bcast(&token, 4)) return;
    got[rank] = token;
    if (c.barrier()) return;
    c.close_all();
    ok[rank] = 1;
  };
  std::vector<std::thread> th;
  th.emplace_back(body, 0);
  {
    // a connection that does not know the nonce must be dropped, not given a rank
    std::thread s([&] { Comm x; (void)x.init(2, world, "127.0.0.1", port, 0xbadull); x.close_all(); });
    s.join();
  }
  for (int r = 1; r < world; ++r) th.emplace_back(body, r);
  for (auto &t : th) t.join();
  for (int r = 0; r < world; ++r) {
    CHECK(ok[r]);
    CHECK(sums[r] == 1000.0 * (1 + 2 + 3 + 4));
    CHECK(maxs[r] == 3.0);
    CHECK(got[r] == 0xabcdef01u);
  }
  // MASTER_ADDR as a NAME (launchers export localhost / a host name): resolved, rank 0 listens;
  // a name that does not resolve is an error of its own, not "cannot reach rank 0"
  {
    std::vector<int> ok2(2, 0);
    auto two = [&](int rank) {
      Comm c;
      if (c.init(rank, 2, "localhost", port + 1, nonce)) return;
      double v = rank + 1.0;
      if (c.allreduce_sum(&v, 1) || v != 3.0) return;
      c.close_all();
      ok2[rank] = 1;
    };
    std::thread a(two, 0), b(two, 1);
    a.join(); b.join();
    CHECK(ok2[0] && ok2[1]);
    Comm bad;
    CHECK(bad.init(0, 2, "no-such-host.invalid", port + 2, nonce) == -2);
  }
  // partition: every rank non-empty, boundaries monotone, balanced on nnz
  std::vector<int64_t> rp(1, 0);
  std::mt19937 g(2);
  for (int u = 0; u < 5000; ++u) rp.push_back(rp.back() + (int64_t)(g() % 100));
  for (int w : {1, 2, 3, 8, 64}) {
    auto parts = partition_users(rp, w);
    CHECK((int)parts.size() == w && parts.front().first == 0 && parts.back().second == 5000);
    for (int r = 0; r < w; ++r) {
      CHECK(parts[r].first < parts[r].second);
      if (r) CHECK(parts[r].first == parts[r - 1].second);
    }
  }
}

// ---- Env: directory name and flags --------------------------------------------
static void test_env()
{
  const char *argv1[] = {"hgaprec", "-dir", "data/ml", "-n", "300", "-m", "200", "-k", "5", "-hier", "-seed", "7"};
  Env e; std::string bad;
  CHECK(e.parse(12, (char **)argv1, false, &bad) == 0);
  CHECK(e.make_prefix() == "n300-m200-k5-da-batch-hier-vb-seed7");
  const char *argv2[] = {"hgaprec", "-dir", "x", "-bogus"};
  Env e2;
  CHECK(e2.parse(4, (char **)argv2, false, &bad) != 0 && bad == "-bogus");
  const char *argv3[] = {"hgaprec", "-n"};                  // value missing at the end of argv
  Env e3;
  (void)e3.parse(2, (char **)argv3, false, &bad);
}

// ---- the threaded host paths (round 4): a ratings file parsed in pieces, the CSR built per user range, the start
// state's expectations and the matrix writer on several threads -- the results of the one-thread paths, and (under
// -fsanitize=thread, `make tsan`) no data race on the way
static void test_threads()
{
  std::mt19937 g(5);
  std::string text;
  const int R = 60000;
  for (int j = 0; j < R; ++j) {
    const unsigned u = j < R / 2 ? 100u + (unsigned)(j / 9) : 100u + g() % 9000u;       // grouped, then scattered users
    const unsigned i = 1u + (g() % 97u) * (g() % 41u);
    text += std::to_string(u) + "\t" + std::to_string(i) + "\t" + std::to_string(g() % 6u) + "\n";
  }
  write_text(path("big.tsv"), text);
  write_text(path("bigv.tsv"), text.substr(0, text.size() / 3 * 2));                       // may end inside a record: token-by-token then
  auto read = [&](const char *threads, uint32_t cap_n, Ratings *r) {
    setenv("HGAPREC_READ_THREADS", threads, 1);
    setenv("HGAPREC_READ_PARALLEL_MIN", "0", 1);
    r->cap_n = cap_n; r->cap_m = 100000;
    CHECK(r->read_train(path("big.tsv")) == 0);
    CHECK(r->read_heldout(path("bigv.tsv"), &r->validation) == 0);
  };
  for (uint32_t cap_n : {100000u, 3000u}) {                 // 3000: the capacity binds, the fast path hands its records back
    Ratings one, many;
    read("1", cap_n, &one);
    read("5", cap_n, &many);
    CHECK(one.n == many.n && one.m == many.m && one.nratings == many.nratings);
    CHECK(one.rowptr == many.rowptr && one.col == many.col && one.val == many.val);
    CHECK(one.seq2user == many.seq2user && one.seq2item == many.seq2item);
    CHECK(one.validation.u == many.validation.u && one.validation.i == many.validation.i && one.validation.y == many.validation.y);
    CHECK(cap_n < 100000u ? one.n == cap_n : one.n > 3000u);
  }
  unsetenv("HGAPREC_READ_THREADS"); unsetenv("HGAPREC_READ_PARALLEL_MIN");
  // start state: 70 000 elements per matrix, above the bar of the threaded half
  GammaState a, b;
  setenv("HGAPREC_SAVE_THREADS", "1", 1);
  { Mt19937 r = make_rng(3.0); initialize_state(r, 700, 300, 100, true, true, &a); }
  setenv("HGAPREC_SAVE_THREADS", "6", 1);
  { Mt19937 r = make_rng(3.0); initialize_state(r, 700, 300, 100, true, true, &b); }
  CHECK(a.theta_Elog.size() == 70000 && std::equal(a.theta_Elog.begin(), a.theta_Elog.end(), b.theta_Elog.begin()));
  CHECK(std::equal(a.theta_E.begin(), a.theta_E.end(), b.theta_E.begin()) && std::equal(a.theta_rate.begin(), a.theta_rate.end(), b.theta_rate.begin()));
  CHECK(std::equal(a.beta_Elog.begin(), a.beta_Elog.end(), b.beta_Elog.begin()));
  // three matrices side by side, each formatted on several threads while its previous wave is written
  const uint32_t rows = 30000, cols = 80;                   // 2.4 M numbers: above the bar of the threaded writer
  std::vector<double> mtx((size_t)rows * cols);
  for (auto &v : mtx) v = (double)(g() % 100000u) / 977.0;
  setenv("HGAPREC_SAVE_THREADS", "1", 1);
  CHECK(save_matrix(path("solo.tsv"), mtx.data(), rows, cols, nullptr, 0) == 0);
  setenv("HGAPREC_SAVE_THREADS", "4", 1);
  {
    std::vector<std::thread> th;
    int bad[3] = {0, 0, 0};
    for (int j = 0; j < 3; ++j)
      th.emplace_back([&, j]() { bad[j] = save_matrix(path(("side" + std::to_string(j) + ".tsv").c_str()), mtx.data(), rows, cols, nullptr, 0, 0, 4); });
    for (auto &t : th) t.join();
    const std::string want = read_text(path("solo.tsv"));
    for (int j = 0; j < 3; ++j) CHECK(bad[j] == 0 && read_text(path(("side" + std::to_string(j) + ".tsv").c_str())) == want);
  }
  unsetenv("HGAPREC_SAVE_THREADS");
}

int main(int argc, char **argv)
{
  char tmpl[] = "/tmp/hgaprec_selftest_XXXXXX";
  const char *d = argc > 1 ? argv[1] : mkdtemp(tmpl);
  if (!d) { perror("mkdtemp"); return 2; }
  g_dir = d;
  test_idmap();
  test_env();
  test_reader();
  test_cache();
  test_writers();
  test_state();
  test_threads();
  test_comm();
  if (g_fail) { fprintf(stderr, "host_selftest: %d check(s) failed\n", g_fail); return 1; }
  printf("host_selftest ok\n");
  return 0;
}

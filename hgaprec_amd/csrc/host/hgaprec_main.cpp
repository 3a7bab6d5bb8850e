// hgaprec_main.cpp -- the `hgaprec` command line on MI355X.
//
// Same flags, input files and output files as the reference's src/main.cc +
// HGAPRec::vb_hier / vb / vb_bias (hgaprec.cc:1321-1436, 919-980, 1219-1319);
// the CAVI sweeps run on the GPU through include/hpf.h.  What stays on the
// host is what the reference also does outside the hot loop: parsing, TSV
// I/O, the MT19937 start state, the held-out series and its stop rule.
//
// Out of scope: competitor bridges, MLE/Canny ablations,
// -gen-ranking/-msr/-rmse report modes; their flags are recognised and refused.
#include "../../../include/hpf.h"
#include "hgaprec_host.hpp"

#include <algorithm>
#include <cassert>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

using namespace hgaprec;

static volatile sig_atomic_t g_save_state_now = 0;
static void term_handler(int) { g_save_state_now = 1; }   // main.cc:19-30

namespace {

struct Driver {
  Env &env; Ratings &rt; hpf_handle *h = nullptr;
  uint32_t n, m, k, iter = 0;
  time_t start;
  FILE *vf = nullptr, *tf = nullptr, *af = nullptr, *pf = nullptr;
  StopRule stop;
  Mt19937 rng;                         // gsl_rng *_r: keeps running after initialize()
  std::vector<uint32_t> sampled;       // _sampled_users (std::map keys: sorted, unique)
  std::vector<uint32_t> item_deg;      // _movies[m]->size()

  Driver(Env &e, Ratings &r) : env(e), rt(r), n(r.n), m(r.m), k(e.k), start(time(0)) {}

  uint32_t duration() const { return (uint32_t)(time(0) - start); }      // hgaprec.hh:164-169

  void die(const char *what, int rc) {
    fprintf(stderr, "error: %s: %s (%s)\n", what, hpf_strerror(rc), h ? hpf_last_error(h) : "");
    exit(-1);
  }

  // HGAPRec::HGAPRec (hgaprec.cc:8-98): output files, held-out sets, prior log
  void construct() {
    env.plog("infer n:", n);
    const char *names[] = {"/heldout.txt", "/validation.txt", "/test.txt", "/logl.txt",
                           "/precision.txt", "/ndcg.txt", "/rmse.txt"};
    for (const char *nm : names) {
      FILE *f = fopen(env.file_str(nm).c_str(), "w");
      if (!f) { printf("cannot open heldout file:%s\n", strerror(errno)); exit(-1); }
      if (!strcmp(nm, "/validation.txt")) vf = f;
      else if (!strcmp(nm, "/test.txt")) tf = f;
      else if (!strcmp(nm, "/logl.txt")) af = f;
      else if (!strcmp(nm, "/precision.txt")) pf = f;
      else fclose(f);
    }
    // load_validation_and_test_sets (hgaprec.cc:110-151): both must open
    int rc = rt.read_heldout(env.datfname + "/validation.tsv", &rt.validation);
    assert(rc != -1);
    if (rc) exit(-1);
    rc = rt.read_heldout(env.datfname + "/test.tsv", &rt.test);
    assert(rc != -1);
    if (rc) exit(-1);
    printf("+ loaded validation and test sets from %s\n", env.datfname.c_str());
    fflush(stdout);
    env.plog("test ratings", (uint64_t)rt.test.u.size());
    env.plog("validation ratings", (uint64_t)rt.validation.u.size());
    if (!env.hier) {
      env.plog("theta shape:", 0.3); env.plog("theta rate:", 0.3);
      env.plog("beta shape:", 0.3); env.plog("beta rate:", 0.3);
    } else {
      env.plog("htheta shape:", 0.3); env.plog("htheta rate:", 0.3);
      env.plog("hbeta shape:", 0.3); env.plog("hbeta rate:", 0.3);
      env.plog("thetarate shape:", 0.3); env.plog("thetarate rate:", 0.3);
      env.plog("betarate shape:", 0.3); env.plog("betarate rate:", 0.3);
    }

    hpf_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.n_users = n; cfg.n_items = m; cfg.K = k;
    cfg.hier = env.hier; cfg.bias = env.bias; cfg.binary = env.binary_data;
    cfg.n_users_total = n; cfg.device = env.device; cfg.n_ranks = 1; cfg.rank = 0;
    cfg.s_prior = 0.3; cfg.r_prior = 0.3;
    rc = hpf_create(&cfg, &h);
    if (rc) die("hpf_create (is an MI355X visible? there is no CPU fallback)", rc);
    rc = hpf_upload_csr(h, rt.rowptr.data(), rt.col.data(), env.binary_data ? nullptr : rt.val.data());
    if (rc) die("hpf_upload_csr", rc);
  }

  // HGAPRec::initialize (hgaprec.cc:153-204): MT19937 on the host, state to the device
  void initialize() {
    rng = make_rng(env.seed);
    GammaState s;
    initialize_state(rng, n, m, k, env.hier, env.bias, &s);
    auto put = [&](hpf_state w, const std::vector<double> &v) {
      int rc = hpf_set_state(h, w, v.data(), v.size());
      if (rc) die("hpf_set_state", rc);
    };
    put(HPF_THETA_SHAPE, s.theta_shape); put(HPF_THETA_RATE, s.theta_rate);
    put(HPF_THETA_E, s.theta_E); put(HPF_THETA_ELOG, s.theta_Elog);
    put(HPF_BETA_SHAPE, s.beta_shape); put(HPF_BETA_RATE, s.beta_rate);
    put(HPF_BETA_E, s.beta_E); put(HPF_BETA_ELOG, s.beta_Elog);
    if (env.hier) {
      put(HPF_XI_SHAPE, s.xi_shape); put(HPF_XI_RATE, s.xi_rate); put(HPF_XI_E, s.xi_E); put(HPF_XI_ELOG, s.xi_Elog);
      put(HPF_ETA_SHAPE, s.eta_shape); put(HPF_ETA_RATE, s.eta_rate); put(HPF_ETA_E, s.eta_E); put(HPF_ETA_ELOG, s.eta_Elog);
    }
    if (env.bias) {
      put(HPF_UBIAS_SHAPE, s.ubias_shape); put(HPF_UBIAS_E, s.ubias_E); put(HPF_UBIAS_ELOG, s.ubias_Elog);
      put(HPF_IBIAS_SHAPE, s.ibias_shape); put(HPF_IBIAS_E, s.ibias_E); put(HPF_IBIAS_ELOG, s.ibias_Elog);
    }
  }

  // GP*::save_state (gpbase.hh:389-398,743-752,971-980)
  void save_object(const char *name, hpf_state shape, uint32_t rows, uint32_t cols, bool vec_rate,
                   const std::vector<uint32_t> &ids) {
    std::vector<double> buf;
    auto get = [&](hpf_state w, size_t cnt) {
      buf.resize(cnt);
      int rc = hpf_get_state(h, w, buf.data(), cnt);
      if (rc) die("hpf_get_state", rc);
    };
    const std::string base = env.file_str(std::string("/") + name);
    const uint32_t nids = (uint32_t)ids.size();
    get(shape, (size_t)rows * cols);
    save_matrix(base + "_shape.tsv", buf.data(), rows, cols, ids.data(), nids);
    if (vec_rate) {   // GPMatrixGR: D1Array<double>::save of the K-vector, ids looked up by k
      get((hpf_state)(shape + 1), cols);
      save_vector(base + "_rate.tsv", buf.data(), cols, ids.data(), nids);
    } else {
      get((hpf_state)(shape + 1), (size_t)rows * cols);
      save_matrix(base + "_rate.tsv", buf.data(), rows, cols, ids.data(), nids);
    }
    get((hpf_state)(shape + 2), (size_t)rows * cols);
    save_matrix(base + ".tsv", buf.data(), rows, cols, ids.data(), nids);
  }
  void save_array(const char *name, hpf_state shape, uint32_t rows, const std::vector<uint32_t> &ids) {
    std::vector<double> buf(rows);
    const std::string base = env.file_str(std::string("/") + name);
    const char *suf[3] = {"_shape.tsv", "_rate.tsv", ".tsv"};
    for (int j = 0; j < 3; ++j) {
      int rc = hpf_get_state(h, (hpf_state)(shape + j), buf.data(), rows);
      if (rc) die("hpf_get_state", rc);
      save_vector(base + suf[j], buf.data(), rows, ids.data(), (uint32_t)ids.size());
    }
  }

  void save_model() {                           // hgaprec.cc:2137-2158
    if (env.hier) {
      save_object("hbeta", HPF_BETA_SHAPE, m, k, false, rt.seq2item);
      save_array("betarate", HPF_ETA_SHAPE, m, rt.seq2item);
      save_object("htheta", HPF_THETA_SHAPE, n, k, false, rt.seq2user);
      save_array("thetarate", HPF_XI_SHAPE, n, rt.seq2user);
    } else {
      save_object("beta", HPF_BETA_SHAPE, m, k, true, rt.seq2item);
      save_object("theta", HPF_THETA_SHAPE, n, k, true, rt.seq2user);
    }
    if (env.bias) {     // n x 1 GPMatrix objects: one value column
      save_object("betabias", HPF_IBIAS_SHAPE, m, 1, false, rt.seq2item);
      save_object("thetabias", HPF_UBIAS_SHAPE, n, 1, false, rt.seq2user);
    }
  }

  // ---- ranking evaluation: compute_precision / compute_itemrank --------
  bool test_hit(int v) const {                   // ratings.hh:183-189
    return env.binary_data ? v >= 1 : (uint32_t)v >= env.rating_threshold;
  }
  // rating stored for (user n, item m) in the training set, 0 if absent (Ratings::r)
  uint32_t train_r(uint32_t n, uint32_t m) const {
    uint32_t r = 0;                              // duplicates carry the same (last) value
    for (int64_t j = rt.rowptr[n]; j < rt.rowptr[n + 1]; ++j) if (rt.col[(size_t)j] == m) r = rt.val[(size_t)j];
    return r;
  }
  static size_t lower(const HeldOut &h, uint32_t u, uint32_t i) {
    size_t lo = 0, hi = h.u.size();
    while (lo < hi) { size_t mid = (lo + hi) / 2;
      if (h.u[mid] < u || (h.u[mid] == u && h.i[mid] < i)) lo = mid + 1; else hi = mid; }
    return lo;
  }
  // validation items of the sampled users, CSR over `sampled` (is_validation())
  void build_mask(std::vector<uint64_t> &mptr, std::vector<uint32_t> &mitems) const {
    mptr.assign(sampled.size() + 1, 0); mitems.clear();
    for (size_t b = 0; b < sampled.size(); ++b) {
      for (size_t a = lower(rt.validation, sampled[b], 0); a < rt.validation.u.size() && rt.validation.u[a] == sampled[b]; ++a)
        mitems.push_back(rt.validation.i[a]);
      mptr[b + 1] = mitems.size();
    }
  }

  void compute_precision(bool save_ranking_file) {          // hgaprec.cc:1703-1848
    if (iter % 100 == 0 && iter > 0) save_ranking_file = true;
    FILE *f = save_ranking_file ? fopen(env.file_str("/ranking.tsv").c_str(), "w") : nullptr;
    if (!save_ranking_file) {                    // hgaprec.cc:1714-1721
      sampled.clear();
      do {
        const uint32_t u = (uint32_t)rng.uniform_int(n);
        auto it = std::lower_bound(sampled.begin(), sampled.end(), u);
        if (it == sampled.end() || *it != u) sampled.insert(it, u);
      } while (sampled.size() < 1000 && sampled.size() < n / 2);
    }
    const uint32_t N = 100;                      // _topN_by_user
    std::vector<uint64_t> mptr; std::vector<uint32_t> mitems;
    build_mask(mptr, mitems);
    std::vector<uint32_t> items(sampled.size() * N); std::vector<double> scores(sampled.size() * N);
    int rc = hpf_rank_topn(h, sampled.data(), (uint32_t)sampled.size(), mptr.data(), mitems.data(), N,
                           items.data(), scores.data());
    if (rc) die("hpf_rank_topn", rc);
    double mhits10 = 0, mhits100 = 0; uint32_t total_users = 0;
    for (size_t b = 0; b < sampled.size(); ++b) {
      const uint32_t u = sampled[b];
      uint32_t hits10 = 0, hits100 = 0;
      for (uint32_t j = 0; j < m && j < N; ++j) {
        const uint32_t it = items[b * N + j]; const double pred = scores[b * N + j];
        int v = 0;
        const size_t a = lower(rt.test, u, it);
        if (a < rt.test.u.size() && rt.test.u[a] == u && rt.test.i[a] == it) {
          v = test_hit(rt.test.y[a]) ? 1 : 0;
          if (j < 10) { if (v > 0) { hits10++; hits100++; } }
          else if (j < 100) { if (v > 0) hits100++; }
        }
        if (f && train_r(u, it) == 0) fprintf(f, "%d\t%d\t%.5f\t%d\n", rt.seq2user[u], rt.seq2item[it], pred, v);
      }
      mhits10 += (double)hits10 / 10; mhits100 += (double)hits100 / 100; total_users++;
    }
    if (f) fclose(f);
    fprintf(pf, "%d\t%.5f\t%.5f\n", total_users, (double)mhits10 / total_users, (double)mhits100 / total_users);
    fflush(pf);
  }

  void compute_itemrank(bool final) {                       // hgaprec.cc:1606-1701
    if (iter % 100 == 0 && iter > 0) final = true;
    if (!final) return;
    FILE *f = fopen(env.file_str("/itemrank.tsv").c_str(), "w");
    FILE *itemf = fopen(env.file_str("/meanrank.txt").c_str(), "w");
    if (!itemf) { printf("cannot open logl file:%s\n", strerror(errno)); exit(-1); }
    if (item_deg.empty()) { item_deg.assign(m, 0); for (uint32_t c : rt.col) item_deg[c]++; }
    std::vector<uint64_t> mptr; std::vector<uint32_t> mitems;
    build_mask(mptr, mitems);
    std::vector<uint32_t> qs, qi;                 // one query per test item that is a hit
    for (size_t b = 0; b < sampled.size(); ++b)
      for (size_t a = lower(rt.test, sampled[b], 0); a < rt.test.u.size() && rt.test.u[a] == sampled[b]; ++a)
        if (test_hit(rt.test.y[a])) { qs.push_back((uint32_t)b); qi.push_back(rt.test.i[a]); }
    std::vector<uint32_t> rank(qs.size()); std::vector<double> pred(qs.size());
    int rc = hpf_item_ranks(h, sampled.data(), (uint32_t)sampled.size(), mptr.data(), mitems.data(),
                            qs.data(), qi.data(), (uint32_t)qs.size(), rank.data(), pred.data());
    if (rc) die("hpf_item_ranks", rc);
    double sum_rank = .0, sum_reciprocal_rank = .0; uint32_t total_users = 0;
    std::vector<uint32_t> ord, seen;
    for (size_t b = 0, q0 = 0; b < sampled.size(); ++b) {
      size_t q1 = q0; while (q1 < qs.size() && qs[q1] == b) ++q1;
      const uint32_t u = sampled[b];
      // items the reference counts as "ranked": Ratings::r(n,m) == 0
      seen.clear();
      for (int64_t j = rt.rowptr[u]; j < rt.rowptr[u + 1]; ++j) if (rt.val[(size_t)j] > 0) seen.push_back(rt.col[(size_t)j]);
      std::sort(seen.begin(), seen.end());
      const uint32_t nranked = m - (uint32_t)(std::unique(seen.begin(), seen.end()) - seen.begin());
      ord.resize(q1 - q0);
      for (size_t k = 0; k < ord.size(); ++k) ord[k] = (uint32_t)(q0 + k);
      std::sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return rank[x] < rank[y]; });
      double rank_ui = .0, reciprocal_rank_ui = .0; uint32_t ntestitems = 0;
      for (uint32_t q : ord) {
        const uint32_t j = rank[q];
        ntestitems++;
        fprintf(f, "%d\t%d\t%.5f\t%d\t%d\n", u, qi[q], pred[q], j, item_deg[qi[q]]);
        rank_ui += (j + 1);
        reciprocal_rank_ui += 1 / (j + 1);        // integer division, as in the reference
      }
      if (ntestitems > 0 && nranked > 0) {
        sum_rank += (rank_ui / nranked) / ntestitems;
        sum_reciprocal_rank += reciprocal_rank_ui / ntestitems;
        total_users++;
      }
      q0 = q1;
    }
    fclose(f);
    fprintf(itemf, "%d\t%.5f\t%.5f\n", total_users, (double)sum_rank / total_users,
            (double)sum_reciprocal_rank / total_users);
    fclose(itemf);
  }

  void gen_ranking_for_users() {                            // hgaprec.cc:2087-2112 (load == false)
    const std::string path = env.datfname + "/test_users.tsv";
    env.lerr("loading test users from %s", path.c_str());
    std::vector<uint32_t> ids;
    if (rt.read_test_users(path, &ids)) { env.lerr("cannot open %s", path.c_str()); return; }
    sampled = ids;
    compute_precision(true);
    compute_itemrank(true);
    env.lerr("DONE writing ranking.tsv in output directory\n");
  }

  void do_on_stop() { save_model(); gen_ranking_for_users(); }          // hgaprec.cc:1572-1577

  // HGAPRec::compute_likelihood (hgaprec.cc:1439-1501); returns true to stop
  bool compute_likelihood(bool validation) {
    const HeldOut &ho = validation ? rt.validation : rt.test;
    FILE *ff = validation ? vf : tf;
    double s = 0.0; uint64_t cnt = 0;
    int rc = hpf_heldout_ll(h, ho.u.data(), ho.i.data(), ho.y.data(), ho.u.size(), &s, &cnt);
    if (rc) die("hpf_heldout_ll", rc);
    const uint32_t kk = (uint32_t)cnt;
    fprintf(ff, "%d\t%d\t%.9f\t%d\n", iter, duration(), s / kk, kk);
    fflush(ff);
    if (!validation) return false;
    const double a = s / kk;
    int why = -1;
    const bool st = stop.update(iter, a, &why);
    FILE *f = fopen(env.file_str("/max.txt").c_str(), "w");
    fprintf(f, "%d\t%d\t%.5f\t%d\n", iter, duration(), a, why);
    fclose(f);
    if (st) { do_on_stop(); return true; }
    return false;
  }

  // the three batch loops share one shape; only -hier honours max_iterations
  // (hgaprec.cc:1337-1339; vb() and vb_bias() run until the stop rule fires)
  void run() {
    if (!env.hier) env.lerr(env.bias ? "running vb_bias()" : "running vb()");
    initialize();
    while (1) {
      if (env.hier && iter > env.max_iterations) exit(0);
      int rc = hpf_iterate(h, 1);
      if (rc) die("hpf_iterate", rc);
      printf("\r iteration %d", iter);
      fflush(stdout);
      if (iter % env.rfreq == 0) {
        if (compute_likelihood(true)) exit(0);
        compute_likelihood(false);
        save_model();
        compute_precision(false);
        if (env.hier || !env.bias) compute_itemrank(false);   // vb_bias() has no itemrank call
        if (env.logl) {                          // HGAPRec::logl, hgaprec.cc:2160-2255
          double v = 0.0;
          int rc2 = hpf_elbo(h, &v);
          if (rc2) die("hpf_elbo", rc2);
          fprintf(af, "%.5f\n", v);
          fflush(af);
        }
      }
      if (g_save_state_now) {
        env.lerr("Saving state at iteration %d duration %d secs", iter, duration());
        do_on_stop();
      }
      iter++;
    }
  }
};

}  // namespace

int main(int argc, char **argv)
{
  signal(SIGTERM, term_handler);
  if (argc <= 1) {
    printf("gaprec -dir <netflix-dataset-dir> -n <users>"
           "-m <movies> -k <dims> -label <out-dir-tag>\n");
    exit(0);
  }
  Env env; std::string bad;
  if (env.parse(argc, argv, true, &bad)) {
    fprintf(stdout, "error: unknown option %s\n", bad.c_str());
    fflush(stdout);
    abort();                                    // the reference asserts (main.cc:227-230)
  }
  if (!env.unsupported.empty()) {
    fprintf(stderr, "error: option %s selects a mode outside the MI355X hot-path build "
                    "(supported: -dir -n -m -k -hier -bias -binary-data -rfreq -max-iterations "
                    "-seed -label -rating-threshold -a -b -c -d)\n", env.unsupported.c_str());
    return 2;
  }
  if (env.open_output()) { fprintf(stderr, "error: cannot create output directory\n"); abort(); }

  Ratings ratings;
  ratings.cap_n = env.n; ratings.cap_m = env.m;
  ratings.binary = env.binary_data; ratings.rating_threshold = env.rating_threshold;
  fprintf(stdout, "+ reading ratings dataset from %s\n", env.datfname.c_str());
  fflush(stdout);
  int rc = ratings.read_train(env.datfname + "/train.tsv");
  if (rc) exit(-1);
  env.plog("training ratings", (uint32_t)ratings.nratings);
  {
    uint32_t lu = 0, li = 0;
    ratings.write_marginals(env.file_str("/byusers.tsv"), env.file_str("/byitems.tsv"), &lu, &li);
    env.lerr("longest sequence of users with no movies: %d", lu);
    env.lerr("longest sequence of items with no users: %d", li);
    // write_marginal_distributions logs env.n / env.m before Ratings::read shrinks them
    env.plog("post pruning nusers:", env.n);
    env.plog("post pruning nitems:", env.m);
  }
  {
    char st[1024];
    snprintf(st, sizeof st, "read %d users, %d movies, %d ratings", ratings.n, ratings.m, (uint32_t)ratings.nratings);
    env.plog("statistics", std::string(st));
  }
  if (!env.batch) {
    printf("Quitting. Online inference not implemented.\n");
    fflush(stdout);
    exit(0);
  }
  Driver d(env, ratings);
  d.construct();
  d.run();
  return 0;
}

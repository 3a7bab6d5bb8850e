// host_capi.cpp -- C exports of the host side for the CPU test-suite
// (tests/test_host_*.py load libhgaprec_host.so through ctypes).  Not part of
// the device ABI; the device ABI is include/hpf.h.
#include "hgaprec_host.hpp"

#include <cstring>

using namespace hgaprec;

extern "C" {

// Env: parse argv (without argv[0]) and return the output-directory name
int hg_prefix(int argc, char **argv, char *out, size_t cap, char *bad, size_t badcap)
{
  Env e; std::string b;
  std::vector<char *> av; av.push_back((char *)"hgaprec");
  for (int i = 0; i < argc; ++i) av.push_back(argv[i]);
  int rc = e.parse((int)av.size(), av.data(), false, &b);
  if (rc) { if (bad) { strncpy(bad, b.c_str(), badcap - 1); bad[badcap - 1] = 0; } return rc; }
  std::string p = e.make_prefix();
  strncpy(out, p.c_str(), cap - 1); out[cap - 1] = 0;
  return 0;
}

// Env::open_output in the current directory (creates the dir + param.txt head)
int hg_open_output(int argc, char **argv, char *out, size_t cap)
{
  Env e; std::string b;
  std::vector<char *> av; av.push_back((char *)"hgaprec");
  for (int i = 0; i < argc; ++i) av.push_back(argv[i]);
  if (e.parse((int)av.size(), av.data(), false, &b)) return 1;
  if (e.open_output()) return -1;
  strncpy(out, e.prefix.c_str(), cap - 1); out[cap - 1] = 0;
  e.close_output();
  return 0;
}

Ratings *hg_ratings_new(uint32_t cap_n, uint32_t cap_m, int binary, uint32_t thr)
{
  Ratings *r = new Ratings(); r->cap_n = cap_n; r->cap_m = cap_m; r->binary = binary != 0;
  r->rating_threshold = thr; return r;
}
void hg_ratings_free(Ratings *r) { delete r; }
int hg_ratings_read_train(Ratings *r, const char *path) { return r->read_train(path); }
int hg_ratings_read_heldout(Ratings *r, const char *path, int which)
{ return r->read_heldout(path, which == 0 ? &r->validation : &r->test); }
uint32_t hg_ratings_n(const Ratings *r) { return r->n; }
uint32_t hg_ratings_m(const Ratings *r) { return r->m; }
uint64_t hg_ratings_nnz(const Ratings *r) { return r->col.size(); }
const int64_t *hg_ratings_rowptr(const Ratings *r) { return r->rowptr.data(); }
const uint32_t *hg_ratings_col(const Ratings *r) { return r->col.data(); }
const uint8_t *hg_ratings_val(const Ratings *r) { return r->val.data(); }
const uint32_t *hg_ratings_seq2user(const Ratings *r) { return r->seq2user.data(); }
const uint32_t *hg_ratings_seq2item(const Ratings *r) { return r->seq2item.data(); }
uint64_t hg_ratings_heldout_count(const Ratings *r, int w) { return (w ? r->test : r->validation).u.size(); }
const uint32_t *hg_ratings_heldout_u(const Ratings *r, int w) { return (w ? r->test : r->validation).u.data(); }
const uint32_t *hg_ratings_heldout_i(const Ratings *r, int w) { return (w ? r->test : r->validation).i.data(); }
const int32_t *hg_ratings_heldout_y(const Ratings *r, int w) { return (w ? r->test : r->validation).y.data(); }
int hg_ratings_write_marginals(const Ratings *r, const char *bu, const char *bi)
{ return r->write_marginals(bu, bi, nullptr, nullptr); }
int hg_ratings_save_cache(const Ratings *r, const char *dir) { return r->save_cache(dir); }
int hg_ratings_load_cache(Ratings *r, const char *dir) { return r->load_cache(dir); }
int hg_ratings_test_users(const Ratings *r, const char *path, uint32_t *out, uint32_t cap)
{
  std::vector<uint32_t> ids;
  if (r->read_test_users(path, &ids)) return -1;
  for (uint32_t j = 0; j < ids.size() && j < cap; ++j) out[j] = ids[j];
  return (int)ids.size();
}

void hg_mt_u32(double seed, uint32_t count, uint32_t *out)
{
  Mt19937 r = make_rng(seed);
  for (uint32_t i = 0; i < count; ++i) out[i] = r.next_u32();
}
double hg_digamma(double x) { return digamma(x); }

GammaState *hg_state_new(double seed, uint32_t n, uint32_t m, uint32_t k, int hier, int bias)
{
  GammaState *s = new GammaState();
  Mt19937 r = make_rng(seed);
  initialize_state(r, n, m, k, hier != 0, bias != 0, s);
  return s;
}
// the rows [lo, hi) of the user-side arrays, as a rank of several keeps them; *words_after = the generator's
// next word afterwards (every rank must leave the stream where a single process does)
GammaState *hg_state_new_range(double seed, uint32_t n, uint32_t m, uint32_t k, int hier, int bias,
                               uint32_t lo, uint32_t hi, uint32_t *word_after)
{
  GammaState *s = new GammaState();
  Mt19937 r = make_rng(seed);
  initialize_state(r, n, m, k, hier != 0, bias != 0, s, lo, hi);
  if (word_after) *word_after = r.next_u32();
  return s;
}
void hg_state_free(GammaState *s) { delete s; }
// which: the hpf_state index of include/hpf.h
size_t hg_state_get(const GammaState *s, int which, const double **p)
{
  const StateArray *v[24] = {
    &s->theta_shape, &s->theta_rate, &s->theta_E, &s->theta_Elog,
    &s->beta_shape, &s->beta_rate, &s->beta_E, &s->beta_Elog,
    &s->xi_shape, &s->xi_rate, &s->xi_E, &s->xi_Elog,
    &s->eta_shape, &s->eta_rate, &s->eta_E, &s->eta_Elog,
    &s->ubias_shape, &s->ubias_rate, &s->ubias_E, &s->ubias_Elog,
    &s->ibias_shape, &s->ibias_rate, &s->ibias_E, &s->ibias_Elog };
  if (which < 0 || which >= 24) { *p = nullptr; return 0; }
  *p = v[which]->data();
  return v[which]->size();
}

int hg_save_matrix(const char *path, const double *a, uint32_t rows, uint32_t cols,
                   const uint32_t *ids, uint32_t nids)
{ return save_matrix(path, a, rows, cols, ids, nids); }
int hg_save_vector(const char *path, const double *a, uint32_t rows, const uint32_t *ids, uint32_t nids)
{ return save_vector(path, a, rows, ids, nids); }

// user ranges of a multi-process run: out[2*r], out[2*r+1] = [lo, hi) of rank r
void hg_partition_users(const int64_t *rowptr, uint32_t n, int world, uint32_t *out)
{
  std::vector<int64_t> rp(rowptr, rowptr + n + 1);
  auto parts = partition_users(rp, world);
  for (int r = 0; r < world; ++r) { out[2 * r] = parts[r].first; out[2 * r + 1] = parts[r].second; }
}

// format_fixed8 over an array; out receives the strings separated by '\n'
size_t hg_format_fixed8(const double *v, size_t n, char *out)
{
  char *o = out;
  for (size_t i = 0; i < n; ++i) { o += format_fixed8(v[i], o); *o++ = '\n'; }
  return (size_t)(o - out);
}

// feed a validation series through the stop rule; returns the index at which
// it stops (or -1), why[] receives the max.txt code per step
int hg_stop_rule(const uint32_t *iters, const double *a, uint32_t cnt, int *why)
{
  StopRule s; int stop_at = -1;
  for (uint32_t i = 0; i < cnt; ++i) {
    bool st = s.update(iters[i], a[i], &why[i]);
    if (st && stop_at < 0) { stop_at = (int)i; break; }
  }
  return stop_at;
}

}  // extern "C"

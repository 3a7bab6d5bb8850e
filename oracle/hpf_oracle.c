/*
 * hpf_oracle.c -- TEST INFRASTRUCTURE ONLY (see hpf_oracle.h for the parity
 * status: "parity unpinned" end-to-end, component pins listed there).
 *
 * Plain-C, single-thread, fp64 restatement of the reference algorithm.  It
 * keeps the reference's own data flow (shape/rate "curr" and "next" buffers,
 * swap + reset to prior, sequential log-add-exp, serial summation order) so
 * that its results are what the reference binary computes up to the
 * last-ulp differences between GSL's psi and the psi below.  Every function
 * cites the reference file:line it follows (paths relative to
 * /root/reference/src).
 */
#define _GNU_SOURCE
#include "hpf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <assert.h>

/* ------------------------------------------------------------------ */
/* MT19937: GSL 2.x rng/mt.c (gsl_rng_mt19937), the generator behind    */
/* gsl_rng_default (hgaprec.cc:34-38).  Published algorithm: Matsumoto  */
/* & Nishimura 1998, 2002 init_genrand seeding; GSL maps seed 0 -> 4357.*/
/* ------------------------------------------------------------------ */
void orc_rng_seed(orc_rng *r, unsigned long s)
{
  if (s == 0) s = 4357;
  r->mt[0] = (uint32_t)(s & 0xffffffffUL);
  for (int i = 1; i < 624; ++i) {
    uint32_t p = r->mt[i - 1];
    r->mt[i] = (uint32_t)(1812433253UL * (p ^ (p >> 30)) + (unsigned long)i);
  }
  r->mti = 624;
}

uint32_t orc_rng_u32(orc_rng *r)
{
  uint32_t *mt = r->mt;
  if (r->mti >= 624) {
    int kk;
    for (kk = 0; kk < 624 - 397; ++kk) {
      uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    for (; kk < 623; ++kk) {
      uint32_t y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    uint32_t y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    r->mti = 0;
  }
  uint32_t k = mt[r->mti++];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680U;
  k ^= (k << 15) & 0xefc60000U;
  k ^= (k >> 18);
  return k;
}

/* gsl_rng_uniform for mt19937 = mt_get_double = u32 / 2^32 */
double orc_rng_uniform(orc_rng *r) { return orc_rng_u32(r) / 4294967296.0; }

/* gsl_rng_uniform_int (rng/rng.c): range = max-min = 0xffffffff */
unsigned long orc_rng_uniform_int(orc_rng *r, unsigned long n)
{
  unsigned long scale = 0xffffffffUL / n, k;
  do { k = orc_rng_u32(r) / scale; } while (k >= n);
  return k;
}

/* ------------------------------------------------------------------ */
/* digamma, x > 0.  The reference calls gsl_sf_psi (gpbase.hh:260,336)  */
/* -- GSL is a third-party dependency absent here and not version-      */
/* pinned by the reference (configure.ac only probes -lgsl).  psi is a   */
/* mathematical function; GSL documents ~2 eps accuracy.  This          */
/* evaluates it in long double: upward recurrence to x >= 20, then the  */
/* Stirling/Bernoulli asymptotic series (A&S 6.3.18), and rounds once.  */
/* ------------------------------------------------------------------ */
double orc_psi(double xd)
{
  long double x = xd, acc = 0.0L;
  while (x < 20.0L) { acc -= 1.0L / x; x += 1.0L; }
  long double xi = 1.0L / x, x2 = xi * xi;
  /* B_2n / (2n): 1/12, 1/120, 1/252, 1/240, 1/132, 691/32760, 1/12,
     3617/8160, 43867/14364 */
  long double s = x2 * (1.0L / 12 - x2 * (1.0L / 120 - x2 * (1.0L / 252 -
                  x2 * (1.0L / 240 - x2 * (1.0L / 132 - x2 * (691.0L / 32760 -
                  x2 * (1.0L / 12 - x2 * (3617.0L / 8160 -
                  x2 * (43867.0L / 14364)))))))));
  return (double)(acc + logl(x) - 0.5L * xi - s);
}

/* ------------------------------------------------------------------ */
/* matrix.hh:367-381  D1Array<T>::logsum -- sequential log-add-exp     */
/* ------------------------------------------------------------------ */
double orc_logsum(const double *d, uint32_t n)
{
  assert(n > 0);
  if (n == 1) return d[0];
  double r = d[0];
  for (uint32_t i = 1; i < n; ++i) {
    if (d[i] < r) r = r + log(1 + exp(d[i] - r));
    else          r = d[i] + log(1 + exp(r - d[i]));
  }
  return r;
}

/* matrix.hh:383-389 */
void orc_lognormalize(double *d, uint32_t n)
{
  double s = orc_logsum(d, n);
  for (uint32_t i = 0; i < n; ++i) d[i] = exp(d[i] - s);
}

/* ------------------------------------------------------------------ */
/* Ratings store: ratings.cc:63-119, ratings.hh:117-165,191-197        */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t key, val; } kv32;

/* tiny open-addressing id -> seq map (replaces std::map lookups; order of
   insertion, not of keys, defines seq ids -- same as IDMap usage) */
typedef struct { kv32 *t; uint32_t cap, cnt; uint8_t *used; } idmap;

static void idmap_init(idmap *h, uint32_t cap0)
{
  h->cap = 16; while (h->cap < cap0 * 2u) h->cap <<= 1;
  h->cnt = 0;
  h->t = (kv32 *)calloc(h->cap, sizeof(kv32));
  h->used = (uint8_t *)calloc(h->cap, 1);
}
static void idmap_free(idmap *h) { free(h->t); free(h->used); }
static uint32_t idmap_hash(uint32_t k) { k *= 2654435761u; return k ^ (k >> 15); }
static int idmap_find(const idmap *h, uint32_t key, uint32_t *val)
{
  uint32_t i = idmap_hash(key) & (h->cap - 1);
  while (h->used[i]) {
    if (h->t[i].key == key) { *val = h->t[i].val; return 1; }
    i = (i + 1) & (h->cap - 1);
  }
  return 0;
}
static void idmap_put(idmap *h, uint32_t key, uint32_t val);
static void idmap_grow(idmap *h)
{
  idmap o = *h;
  h->cap = o.cap * 2; h->cnt = 0;
  h->t = (kv32 *)calloc(h->cap, sizeof(kv32));
  h->used = (uint8_t *)calloc(h->cap, 1);
  for (uint32_t i = 0; i < o.cap; ++i)
    if (o.used[i]) idmap_put(h, o.t[i].key, o.t[i].val);
  idmap_free(&o);
}
static void idmap_put(idmap *h, uint32_t key, uint32_t val)
{
  if ((h->cnt + 1) * 2u > h->cap) idmap_grow(h);
  uint32_t i = idmap_hash(key) & (h->cap - 1);
  while (h->used[i]) {
    if (h->t[i].key == key) { h->t[i].val = val; return; }
    i = (i + 1) & (h->cap - 1);
  }
  h->used[i] = 1; h->t[i].key = key; h->t[i].val = val; h->cnt++;
}

typedef struct { uint32_t u, i; int32_t y; uint64_t ord; } triple;

struct orc_ratings {
  uint32_t cap_n, cap_m;   /* Env::n / Env::m while reading train */
  int binary; uint32_t thr;
  idmap user2seq, item2seq;
  uint32_t *seq2user, *seq2item;
  uint32_t nusers, nitems;
  /* training triples in file order (seq ids, raw rating) */
  triple *tr; uint64_t ntr, captr;
  int finalized;
  int64_t *rowptr; uint32_t *col; uint8_t *val;
  /* heldout maps */
  triple *ho[2]; uint64_t nho[2], capho[2];
  uint32_t *ho_u[2], *ho_i[2]; int32_t *ho_y[2];
};

orc_ratings *orc_ratings_new(uint32_t cap_n, uint32_t cap_m, int binary,
                             uint32_t thr)
{
  orc_ratings *r = (orc_ratings *)calloc(1, sizeof(*r));
  r->cap_n = cap_n; r->cap_m = cap_m; r->binary = binary; r->thr = thr;
  idmap_init(&r->user2seq, 1024); idmap_init(&r->item2seq, 1024);
  r->seq2user = (uint32_t *)malloc(sizeof(uint32_t) * (cap_n ? cap_n : 1));
  r->seq2item = (uint32_t *)malloc(sizeof(uint32_t) * (cap_m ? cap_m : 1));
  return r;
}

void orc_ratings_free(orc_ratings *r)
{
  if (!r) return;
  idmap_free(&r->user2seq); idmap_free(&r->item2seq);
  free(r->seq2user); free(r->seq2item); free(r->tr);
  free(r->rowptr); free(r->col); free(r->val);
  for (int w = 0; w < 2; ++w) {
    free(r->ho[w]); free(r->ho_u[w]); free(r->ho_i[w]); free(r->ho_y[w]);
  }
  free(r);
}

/* ratings.hh:191-197 */
static uint32_t input_rating_class(const orc_ratings *r, uint32_t v)
{
  if (!r->binary) return v;
  return v >= r->thr ? 1 : 0;
}

static int cmp_row_then_order(const void *a, const void *b)
{
  const triple *x = (const triple *)a, *y = (const triple *)b;
  if (x->u != y->u) return x->u < y->u ? -1 : 1;
  return x->ord < y->ord ? -1 : (x->ord > y->ord);
}
static int cmp_pair_then_order(const void *a, const void *b)
{
  const triple *x = (const triple *)a, *y = (const triple *)b;
  if (x->u != y->u) return x->u < y->u ? -1 : 1;
  if (x->i != y->i) return x->i < y->i ? -1 : 1;
  return x->ord < y->ord ? -1 : (x->ord > y->ord);
}

/* ratings.cc:63-119.  heldout < 0: training pass (cmap == NULL there) */
static int read_generic(orc_ratings *r, FILE *f, int heldout)
{
  uint32_t mid = 0, uid = 0, rating = 0;
  /* capacity seen by this pass: ratings.cc:35-36 shrinks env.n/env.m to the
     registered counts after the training pass */
  uint32_t cap_n = heldout < 0 ? r->cap_n : r->nusers;
  uint32_t cap_m = heldout < 0 ? r->cap_m : r->nitems;
  while (!feof(f)) {
    int got = fscanf(f, "%u\t%u\t%u\n", &uid, &mid, &rating);
    if (got < 0) {                       /* ratings.cc:71-75: exit(-1) */
      fprintf(stderr, "error: unexpected lines in file\n");
      return -1;
    }
    if (got != 3) {
      /* the reference would spin forever on a non-numeric token (nothing is
         consumed); the oracle reports it instead */
      fprintf(stderr, "error: malformed line in ratings file\n");
      return -1;
    }
    uint32_t n = 0, m = 0;
    int hasu = idmap_find(&r->user2seq, uid, &n);
    int hasm = idmap_find(&r->item2seq, mid, &m);
    if ((!hasu && r->nusers >= cap_n) || (!hasm && r->nitems >= cap_m))
      continue;
    if (input_rating_class(r, rating) == 0) continue;
    if (!hasu) {                         /* ratings.hh:117-133 add_user */
      n = r->nusers; idmap_put(&r->user2seq, uid, n);
      r->seq2user[r->nusers++] = uid;
    }
    if (!hasm) {                         /* ratings.hh:135-151 add_movie */
      m = r->nitems; idmap_put(&r->item2seq, mid, m);
      r->seq2item[r->nitems++] = mid;
    }
    triple t; t.u = n; t.i = m; t.y = (int32_t)rating; t.ord = 0;
    if (heldout < 0) {
      if (r->ntr == r->captr) {
        r->captr = r->captr ? r->captr * 2 : 4096;
        r->tr = (triple *)realloc(r->tr, r->captr * sizeof(triple));
      }
      t.ord = r->ntr; r->tr[r->ntr++] = t;
    } else {
      int w = heldout;
      if (r->nho[w] == r->capho[w]) {
        r->capho[w] = r->capho[w] ? r->capho[w] * 2 : 1024;
        r->ho[w] = (triple *)realloc(r->ho[w], r->capho[w] * sizeof(triple));
      }
      t.ord = r->nho[w]; t.y = r->binary ? 1 : (int32_t)rating;
      r->ho[w][r->nho[w]++] = t;
    }
  }
  return 0;
}

static void finalize_train(orc_ratings *r)
{
  /* per-user item vector keeps file order (ratings.cc:105 push_back); the
     rating looked up at sweep time is the LAST value written to the per-user
     std::map<item, uint8_t> (ratings.cc:96-103, ratings.hh:153-165) */
  uint64_t nnz = r->ntr;
  triple *s = (triple *)malloc((nnz ? nnz : 1) * sizeof(triple));
  memcpy(s, r->tr, nnz * sizeof(triple));
  qsort(s, nnz, sizeof(triple), cmp_pair_then_order);
  /* last duplicate wins; stored as yval_t = uint8_t (env.hh:20) */
  uint8_t *lastval = (uint8_t *)malloc(nnz ? nnz : 1);
  for (uint64_t a = 0; a < nnz;) {
    uint64_t b = a;
    while (b + 1 < nnz && s[b + 1].u == s[a].u && s[b + 1].i == s[a].i) ++b;
    uint8_t v = r->binary ? 1 : (uint8_t)(uint32_t)s[b].y;
    for (uint64_t c = a; c <= b; ++c) lastval[s[c].ord] = v;
    a = b + 1;
  }
  memcpy(s, r->tr, nnz * sizeof(triple));
  qsort(s, nnz, sizeof(triple), cmp_row_then_order);
  r->rowptr = (int64_t *)calloc((size_t)r->nusers + 1, sizeof(int64_t));
  r->col = (uint32_t *)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
  r->val = (uint8_t *)malloc(nnz ? nnz : 1);
  for (uint64_t a = 0; a < nnz; ++a) {
    r->rowptr[s[a].u + 1]++;
    r->col[a] = s[a].i;
    r->val[a] = lastval[s[a].ord];
  }
  for (uint32_t u = 0; u < r->nusers; ++u) r->rowptr[u + 1] += r->rowptr[u];
  free(s); free(lastval);
  r->finalized = 1;
}

int orc_ratings_read_train(orc_ratings *r, const char *path)
{
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int rc = read_generic(r, f, -1);
  fclose(f);
  if (rc == 0) finalize_train(r);
  return rc;
}

int orc_ratings_read_heldout(orc_ratings *r, const char *path, int w)
{
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int rc = read_generic(r, f, w);
  fclose(f);
  if (rc) return rc;
  /* std::map<Rating,int>: unique keys, last assignment wins, sorted */
  uint64_t n = r->nho[w];
  qsort(r->ho[w], n, sizeof(triple), cmp_pair_then_order);
  r->ho_u[w] = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  r->ho_i[w] = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  r->ho_y[w] = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
  uint64_t o = 0;
  for (uint64_t a = 0; a < n;) {
    uint64_t b = a;
    while (b + 1 < n && r->ho[w][b + 1].u == r->ho[w][a].u &&
           r->ho[w][b + 1].i == r->ho[w][a].i) ++b;
    r->ho_u[w][o] = r->ho[w][b].u; r->ho_i[w][o] = r->ho[w][b].i;
    r->ho_y[w][o] = r->ho[w][b].y; ++o;
    a = b + 1;
  }
  r->nho[w] = o;
  return 0;
}

uint32_t orc_ratings_n(const orc_ratings *r) { return r->nusers; }
uint32_t orc_ratings_m(const orc_ratings *r) { return r->nitems; }
uint64_t orc_ratings_nnz(const orc_ratings *r) { return r->ntr; }
const int64_t  *orc_ratings_rowptr(const orc_ratings *r) { return r->rowptr; }
const uint32_t *orc_ratings_col(const orc_ratings *r) { return r->col; }
const uint8_t  *orc_ratings_val(const orc_ratings *r) { return r->val; }
const uint32_t *orc_ratings_seq2user(const orc_ratings *r) { return r->seq2user; }
const uint32_t *orc_ratings_seq2item(const orc_ratings *r) { return r->seq2item; }
uint64_t orc_ratings_heldout_count(const orc_ratings *r, int w) { return r->nho[w]; }
const uint32_t *orc_ratings_heldout_u(const orc_ratings *r, int w) { return r->ho_u[w]; }
const uint32_t *orc_ratings_heldout_i(const orc_ratings *r, int w) { return r->ho_i[w]; }
const int32_t  *orc_ratings_heldout_y(const orc_ratings *r, int w) { return r->ho_y[w]; }

/* ratings.cc:217-271: seq, id, degree, sum of (uint8) ratings */
int orc_ratings_write_marginals(const orc_ratings *r, const char *byusers,
                                const char *byitems)
{
  FILE *f = fopen(byusers, "w");
  if (!f) return -1;
  for (uint32_t n = 0; n < r->nusers; ++n) {
    int64_t a = r->rowptr[n], b = r->rowptr[n + 1];
    if (a == b) continue;
    uint32_t t = 0;
    for (int64_t j = a; j < b; ++j) t += r->val[j];
    fprintf(f, "%d\t%d\t%d\t%d\n", n, r->seq2user[n], (int)(b - a), t);
  }
  fclose(f);
  uint32_t *deg = (uint32_t *)calloc(r->nitems ? r->nitems : 1, sizeof(uint32_t));
  uint32_t *sum = (uint32_t *)calloc(r->nitems ? r->nitems : 1, sizeof(uint32_t));
  for (uint64_t j = 0; j < r->ntr; ++j) { deg[r->col[j]]++; sum[r->col[j]] += r->val[j]; }
  f = fopen(byitems, "w");
  if (!f) { free(deg); free(sum); return -1; }
  for (uint32_t i = 0; i < r->nitems; ++i) {
    if (!deg[i]) continue;
    fprintf(f, "%d\t%d\t%d\t%d\n", i, r->seq2item[i], deg[i], sum[i]);
  }
  fclose(f); free(deg); free(sum);
  return 0;
}

/* ------------------------------------------------------------------ */
/* Gamma containers: gpbase.hh                                          */
/* ------------------------------------------------------------------ */
typedef struct {
  uint32_t n, k;       /* k == 1 for GPArray and the bias matrices */
  int global_rate;     /* GPMatrixGR: rate is a k-vector           */
  double sprior, rprior;
  double *scurr, *snext, *rcurr, *rnext, *Ev, *Elogv;
  size_t rsize;
  int hier;                           /* GPMatrix::_hier (set by set_prior_rate)   */
  double *hier_rprior, *hier_log_rprior;  /* gpbase.hh:134-136                     */
} gp;

static void gp_set_to_prior(gp *g)   /* gpbase.hh:149-154,527-532,863-868 */
{
  size_t ns = (size_t)g->n * g->k;
  for (size_t a = 0; a < ns; ++a) g->snext[a] = g->sprior;
  for (size_t a = 0; a < g->rsize; ++a) g->rnext[a] = g->rprior;
}

static void gp_init(gp *g, uint32_t n, uint32_t k, int global_rate)
{
  g->n = n; g->k = k; g->global_rate = global_rate;
  g->sprior = 0.3; g->rprior = 0.3;      /* hgaprec.cc:13-20: literal 0.3 */
  size_t ns = (size_t)n * k; if (!ns) ns = 1;
  g->rsize = global_rate ? k : (size_t)n * k;
  size_t nr = g->rsize ? g->rsize : 1;
  g->scurr = (double *)calloc(ns, 8); g->snext = (double *)calloc(ns, 8);
  g->rcurr = (double *)calloc(nr, 8); g->rnext = (double *)calloc(nr, 8);
  g->Ev = (double *)calloc(ns, 8); g->Elogv = (double *)calloc(ns, 8);
  g->hier = 0;
  g->hier_rprior = (double *)calloc(n ? n : 1, 8);
  g->hier_log_rprior = (double *)calloc(n ? n : 1, 8);
}
static void gp_free(gp *g)
{
  free(g->scurr); free(g->snext); free(g->rcurr); free(g->rnext);
  free(g->Ev); free(g->Elogv); free(g->hier_rprior); free(g->hier_log_rprior);
}

/* gpbase.hh:27-44 */
static void make_nonzero(double av, double bv, double *a, double *b)
{
  assert(av >= 0 && bv >= 0);
  *b = !(bv > .0) ? 1e-30 : bv;
  *a = !(av > .0) ? 1e-30 : av;
}

static void gp_swap(gp *g)           /* gpbase.hh:240-246,571-577,897-903 */
{
  double *t = g->scurr; g->scurr = g->snext; g->snext = t;
  t = g->rcurr; g->rcurr = g->rnext; g->rnext = t;
  gp_set_to_prior(g);
}

/* GPMatrix::set_prior_rate gpbase.hh:163-173: rnext[n,:] = ev[n] (D2Array::set_elements(row, v), matrix.hh:946-951,
   which OVERWRITES the rprior the row held), and the E / E[log] of the row's rate prior are kept for the ELBO */
static void gp_set_prior_rate(gp *g, const double *ev, const double *elogv)
{
  for (uint32_t n = 0; n < g->n; ++n) {
    for (uint32_t k = 0; k < g->k; ++k) g->rnext[(size_t)n * g->k + k] = ev[n];
    g->hier_rprior[n] = ev[n]; g->hier_log_rprior[n] = elogv[n];
  }
  g->hier = 1;
}

/* GPMatrix::update_rate_next(const Array &u) gpbase.hh:218-223: _rnext.add_slice(i, u) for every row
   (matrix.hh:1060-1067); GPMatrixGR::update_rate_next gpbase.hh:560-564 and GPArray::update_rate_next 884-889:
   _rnext += u (D1Array::operator+=, matrix.hh:464-472) */
static void gp_update_rate_next(gp *g, const double *u)
{
  if (g->global_rate) { for (uint32_t k = 0; k < g->k; ++k) g->rnext[k] += u[k]; return; }
  for (uint32_t i = 0; i < g->n; ++i)
    for (uint32_t k = 0; k < g->k; ++k) g->rnext[(size_t)i * g->k + k] += u[k];
}

/* GPArray::update_rate_next(const Array &u) gpbase.hh:884-889 on an n-vector (k == 1): _rnext += u */
static void gp_array_update_rate_next(gp *g, const double *u)
{
  for (uint32_t n = 0; n < g->n; ++n) g->rnext[n] += u[n];
}

/* GPArray::update_shape_next(double v) gpbase.hh:877-882 */
static void gp_array_update_shape_next(gp *g, double v)
{
  for (uint32_t n = 0; n < g->n; ++n) g->snext[n] += v;
}

/* GPMatrix::update_rate_next_all(uint32_t k, double v) gpbase.hh:225-231, k == 0 */
static void gp_update_rate_next_all(gp *g, double v)
{
  for (uint32_t i = 0; i < g->n; ++i) g->rnext[(size_t)i * g->k] += v;
}

static int g_parallel_sweeps = 0;   /* set only inside orc_model_iterate_all_cores */

static void gp_compute_expectations(gp *g) /* gpbase.hh:248-262,579-598,912-925 */
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (g_parallel_sweeps)
#endif
  for (int64_t i = 0; i < (int64_t)g->n; ++i)
    for (uint32_t j = 0; j < g->k; ++j) {
      size_t e = (size_t)i * g->k + j;
      double a, b;
      make_nonzero(g->scurr[e], g->global_rate ? g->rcurr[j] : g->rcurr[e], &a, &b);
      g->Ev[e] = a / b;
      g->Elogv[e] = orc_psi(a) - log(b);
    }
}

/* GPMatrix::initialize gpbase.hh:292-308 ; GPMatrixGR::initialize 651-663 */
static void gp_initialize(gp *g, orc_rng *r)
{
  for (uint32_t i = 0; i < g->n; ++i)
    for (uint32_t k = 0; k < g->k; ++k)
      g->scurr[(size_t)i * g->k + k] = g->sprior + 0.01 * orc_rng_uniform(r);
  if (g->global_rate) {
    for (uint32_t k = 0; k < g->k; ++k)
      g->rcurr[k] = g->rprior + 0.1 * orc_rng_uniform(r);
  } else {
    /* the K draws always happen, even for n == 0 the reference would write
       bd[0][k]; n >= 1 in every use */
    double *b0 = (double *)malloc(sizeof(double) * (g->k ? g->k : 1));
    for (uint32_t k = 0; k < g->k; ++k)
      b0[k] = g->rprior + 0.1 * orc_rng_uniform(r);
    for (uint32_t i = 0; i < g->n; ++i)
      for (uint32_t k = 0; k < g->k; ++k)
        g->rcurr[(size_t)i * g->k + k] = b0[k];
    free(b0);
  }
  gp_set_to_prior(g);
}

/* GPMatrix::initialize2 gpbase.hh:310-322 ; GPArray::initialize2 939-949 */
static void gp_initialize2(gp *g, double v, orc_rng *r)
{
  for (uint32_t i = 0; i < g->n; ++i)
    for (uint32_t k = 0; k < g->k; ++k) {
      size_t e = (size_t)i * g->k + k;
      g->scurr[e] = g->sprior + 0.01 * orc_rng_uniform(r);
      g->rcurr[e] = g->rprior + v;
    }
  gp_set_to_prior(g);
}

/* GPMatrix::initialize_exp gpbase.hh:324-340 ; GPMatrixGR 700-715 */
static void gp_initialize_exp(gp *g, orc_rng *r)
{
  for (uint32_t i = 0; i < g->n; ++i)
    for (uint32_t k = 0; k < g->k; ++k) {
      size_t e = (size_t)i * g->k + k;
      double b = g->rprior + 0.1 * orc_rng_uniform(r);
      g->Ev[e] = g->scurr[e] / b;
      g->Elogv[e] = orc_psi(g->scurr[e]) - log(b);
    }
  gp_set_to_prior(g);
}

/* matrix.hh:327-335  D1Array<T>::sum -- plain left-to-right sum from 0.0 (stride: elements apart) */
double orc_sum_strided(const double *d, uint32_t n, size_t stride)
{
  double s = .0;
  for (uint32_t i = 0; i < n; ++i) s += d[(size_t)i * stride];
  return s;
}
/* sum_rows gpbase.hh:264-271: v[k] += E[i][k] for i = 0..n-1, into a fresh zeroed Array.
 * Per column this is orc_sum_strided(Ev + k, n, K) -- the same additions in the same order
 * (tests/test_oracle_pins.py checks the two against each other); rows stay the outer loop,
 * as in the reference, so that the timed CPU baseline streams the matrix once */
static void gp_sum_rows(const gp *g, double *v)
{
  for (uint32_t i = 0; i < g->n; ++i)
    for (uint32_t k = 0; k < g->k; ++k) v[k] += g->Ev[(size_t)i * g->k + k];
}
/* sum_cols gpbase.hh:273-280: v[i] += E[i][k] for k = 0..K-1 */
static void gp_sum_cols(const gp *g, double *v)
{
  for (uint32_t i = 0; i < g->n; ++i) v[i] += orc_sum_strided(g->Ev + (size_t)i * g->k, g->k, 1);
}

/* ------------------------------------------------------------------ */
struct orc_model {
  uint32_t n, m, K; int hier, bias, binary;
  const int64_t *rowptr; const uint32_t *col; const uint8_t *val;
  gp theta, beta;          /* htheta/hbeta (hier) or theta/beta (GR) */
  gp xi, eta;              /* thetarate / betarate (hier only)       */
  gp ubias, ibias;         /* thetabias / betabias                   */
  orc_rng rng;
  double *phi, *tmpK, *tmpN;
  int skip_step_a;
  int novb;                /* Env::vb == false (-novb); read by vb_bias() only */
};

/* TEST ENTRY (tests/test_oracle_pins.py): the sweep steps above, run on arrays handed in, so that the functions the
   model's iteration calls are the ones held against the reference's own D2Array / D1Array code (oracle/_ref/refpart
   rows, tests/golden/rows.json).  mode 0: GPMatrix (set_prior_rate, update_rate_next, swap); 1: GPMatrixGR
   (update_rate_next, swap); 2: GPArray (update_shape_next(v), update_rate_next(u[n]), swap); 3: bias GPMatrix
   (update_rate_next_all(0, v), swap).  snext_in: rows*k shape accumulators (the phi sums + prior).  Outputs: scurr,
   rcurr (rows*k; mode 1: rcurr is k long), and the refilled snext / rnext. */
void orc_test_sweep_steps(int mode, uint32_t rows, uint32_t k, const double *snext_in, const double *ev,
                          const double *u, double v, double *scurr, double *rcurr, double *snext, double *rnext)
{
  gp g;
  gp_init(&g, rows, k, mode == 1);
  gp_set_to_prior(&g);
  memcpy(g.snext, snext_in, sizeof(double) * (size_t)rows * k);
  if (mode == 0) { gp_set_prior_rate(&g, ev, ev); gp_update_rate_next(&g, u); }
  else if (mode == 1) gp_update_rate_next(&g, u);
  else if (mode == 2) { gp_array_update_shape_next(&g, v); gp_array_update_rate_next(&g, u); }
  else gp_update_rate_next_all(&g, v);
  gp_swap(&g);
  memcpy(scurr, g.scurr, sizeof(double) * (size_t)rows * k);
  memcpy(rcurr, g.rcurr, sizeof(double) * g.rsize);
  memcpy(snext, g.snext, sizeof(double) * (size_t)rows * k);
  memcpy(rnext, g.rnext, sizeof(double) * g.rsize);
  gp_free(&g);
}

orc_model *orc_model_new(uint32_t n, uint32_t m, uint32_t K, int hier, int bias,
                         int binary)
{
  orc_model *M = (orc_model *)calloc(1, sizeof(*M));
  M->n = n; M->m = m; M->K = K; M->hier = hier; M->bias = bias; M->binary = binary;
  gp_init(&M->theta, n, K, !hier);
  gp_init(&M->beta, m, K, !hier);
  gp_init(&M->xi, n, 1, 0);
  gp_init(&M->eta, m, 1, 0);
  gp_init(&M->ubias, n, 1, 0);
  gp_init(&M->ibias, m, 1, 0);
  M->phi = (double *)calloc((size_t)K + 2, 8);
  M->tmpK = (double *)calloc((size_t)K + 2, 8);
  M->tmpN = (double *)calloc((size_t)(n > m ? n : m) + 1, 8);
  orc_rng_seed(&M->rng, 0);
  return M;
}

void orc_model_set_novb(orc_model *M, int novb) { M->novb = novb ? 1 : 0; }

void orc_model_free(orc_model *M)
{
  if (!M) return;
  gp_free(&M->theta); gp_free(&M->beta); gp_free(&M->xi); gp_free(&M->eta);
  gp_free(&M->ubias); gp_free(&M->ibias);
  free(M->phi); free(M->tmpK); free(M->tmpN); free(M);
}

void orc_model_set_csr(orc_model *M, const int64_t *rowptr, const uint32_t *col,
                       const uint8_t *val)
{ M->rowptr = rowptr; M->col = col; M->val = val; }

orc_rng *orc_model_rng(orc_model *M) { return &M->rng; }

/* hgaprec.cc:34-38 (seed) and 153-204 (initialize) */
void orc_model_initialize(orc_model *M, double seed)
{
  /* gsl_rng_alloc seeds with gsl_rng_default_seed = 0 (-> 4357); then
     if (_env.seed) gsl_rng_set(_r, _env.seed)  [double -> unsigned long] */
  orc_rng_seed(&M->rng, 0);
  if (seed) orc_rng_seed(&M->rng, (unsigned long)seed);
  orc_rng *r = &M->rng;
  if (!M->hier) {
    gp_initialize(&M->beta, r);
    gp_initialize(&M->theta, r);
    gp_initialize_exp(&M->beta, r);
    gp_initialize_exp(&M->theta, r);
  } else {
    gp_initialize2(&M->xi, (double)M->K, r);  gp_compute_expectations(&M->xi);
    gp_initialize2(&M->eta, (double)M->K, r); gp_compute_expectations(&M->eta);
    gp_initialize(&M->beta, r);  gp_initialize_exp(&M->beta, r);
    gp_initialize(&M->theta, r); gp_initialize_exp(&M->theta, r);
  }
  if (M->bias) {
    gp_initialize2(&M->ubias, (double)M->m, r); gp_compute_expectations(&M->ubias);
    gp_initialize2(&M->ibias, (double)M->n, r); gp_compute_expectations(&M->ibias);
  }
}

/* step A: hgaprec.cc:1340-1366 (hier), 928-942 (vb), 1227-1248 (vb_bias);
   get_phi hgaprec.cc:206-239 */
static void sweep_nonzeros(orc_model *M)
{
  if (M->skip_step_a) return;           /* orc_model_iterate_all_cores already did it */
  const uint32_t K = M->K, x = M->bias ? K + 2 : K;
  double *phi = M->phi;
  for (uint32_t n = 0; n < M->n; ++n) {
    const double *elt = M->theta.Elogv + (size_t)n * K;
    for (int64_t j = M->rowptr[n]; j < M->rowptr[n + 1]; ++j) {
      uint32_t m = M->col[j];
      uint8_t y = M->val ? M->val[j] : 1;
      const double *elb = M->beta.Elogv + (size_t)m * K;
      for (uint32_t k = 0; k < K; ++k) phi[k] = elt[k] + elb[k];
      if (M->bias) { phi[K] = M->ubias.Elogv[n]; phi[K + 1] = M->ibias.Elogv[m]; }
      orc_lognormalize(phi, x);
      if (y > 1)
        for (uint32_t k = 0; k < x; ++k) phi[k] *= y;   /* matrix.hh:399-406 */
      /* add_slice adds only the first K entries (matrix.hh:1060-1067) */
      double *st = M->theta.snext + (size_t)n * K, *sb = M->beta.snext + (size_t)m * K;
      for (uint32_t k = 0; k < K; ++k) st[k] += phi[k];
      for (uint32_t k = 0; k < K; ++k) sb[k] += phi[k];
      if (M->bias) { M->ubias.snext[n] += phi[K]; M->ibias.snext[m] += phi[K + 1]; }
    }
  }
}

static void bias_sweeps(orc_model *M)   /* hgaprec.cc:1388-1396 / 1262-1268 */
{
  gp_update_rate_next_all(&M->ubias, M->m);
  gp_swap(&M->ubias); gp_compute_expectations(&M->ubias);
  gp_update_rate_next_all(&M->ibias, M->n);
  gp_swap(&M->ibias); gp_compute_expectations(&M->ibias);
}

static void iterate_hier(orc_model *M)  /* hgaprec.cc:1340-1414 */
{
  const uint32_t K = M->K;
  sweep_nonzeros(M);
  /* B: hgaprec.cc:1370-1378 */
  memset(M->tmpK, 0, sizeof(double) * K);
  gp_sum_rows(&M->beta, M->tmpK);
  gp_set_prior_rate(&M->theta, M->xi.Ev, M->xi.Elogv);
  gp_update_rate_next(&M->theta, M->tmpK);
  gp_swap(&M->theta); gp_compute_expectations(&M->theta);
  /* C: hgaprec.cc:1380-1386 */
  memset(M->tmpK, 0, sizeof(double) * K);
  gp_sum_rows(&M->theta, M->tmpK);
  gp_set_prior_rate(&M->beta, M->eta.Ev, M->eta.Elogv);
  gp_update_rate_next(&M->beta, M->tmpK);
  gp_swap(&M->beta); gp_compute_expectations(&M->beta);
  /* D */
  if (M->bias) bias_sweeps(M);
  /* E: hgaprec.cc:1398-1405 */
  memset(M->tmpN, 0, sizeof(double) * M->n);
  gp_sum_cols(&M->theta, M->tmpN);
  gp_array_update_shape_next(&M->xi, K * M->xi.sprior);          /* _k * _thetarate.sprior() */
  gp_array_update_rate_next(&M->xi, M->tmpN);
  gp_swap(&M->xi); gp_compute_expectations(&M->xi);
  /* F: hgaprec.cc:1407-1414 */
  memset(M->tmpN, 0, sizeof(double) * M->m);
  gp_sum_cols(&M->beta, M->tmpN);
  gp_array_update_shape_next(&M->eta, K * M->eta.sprior);
  gp_array_update_rate_next(&M->eta, M->tmpN);
  gp_swap(&M->eta); gp_compute_expectations(&M->eta);
}

/* vb_bias() with -novb: hgaprec.cc:1276-1297.  Both rates are built from the
   expectations of the previous iteration -- _theta.sum_rows() runs before
   _theta.swap() / compute_expectations() -- then everything is swapped. */
static void iterate_flat_bias_novb(orc_model *M)
{
  const uint32_t K = M->K;
  sweep_nonzeros(M);
  memset(M->tmpK, 0, sizeof(double) * K);
  gp_sum_rows(&M->beta, M->tmpK);                                   /* 1278-1280 */
  gp_update_rate_next(&M->theta, M->tmpK);
  memset(M->tmpK, 0, sizeof(double) * K);
  gp_sum_rows(&M->theta, M->tmpK);                                  /* 1281-1283: the OLD E[theta] */
  gp_update_rate_next(&M->beta, M->tmpK);
  gp_update_rate_next_all(&M->ubias, M->m);                         /* 1285-1286 */
  gp_update_rate_next_all(&M->ibias, M->n);
  gp_swap(&M->theta); gp_swap(&M->beta); gp_swap(&M->ubias); gp_swap(&M->ibias);   /* 1288-1291 */
  gp_compute_expectations(&M->theta); gp_compute_expectations(&M->beta);            /* 1293-1296 */
  gp_compute_expectations(&M->ubias); gp_compute_expectations(&M->ibias);
}

static void iterate_flat(orc_model *M)  /* vb(): hgaprec.cc:927-956 ; vb_bias(): 1226-1272 */
{
  const uint32_t K = M->K;
  if (M->bias && M->novb) { iterate_flat_bias_novb(M); return; }     /* if (_env.vb) ... else, hgaprec.cc:1250 */
  sweep_nonzeros(M);
  memset(M->tmpK, 0, sizeof(double) * K);
  gp_sum_rows(&M->beta, M->tmpK);
  gp_update_rate_next(&M->theta, M->tmpK);                           /* gpbase.hh:558-562 */
  gp_swap(&M->theta); gp_compute_expectations(&M->theta);
  memset(M->tmpK, 0, sizeof(double) * K);
  gp_sum_rows(&M->theta, M->tmpK);
  gp_update_rate_next(&M->beta, M->tmpK);
  gp_swap(&M->beta); gp_compute_expectations(&M->beta);
  if (M->bias) bias_sweeps(M);
}

/* ---- all-cores variant of step A: OpenMP over users,
   atomic adds on the shared item rows.  Used only for the extra
   "cpu_baseline_all_cores" figure of bench.py; results differ from the serial
   sweep by summation order only. ---- */
#ifdef _OPENMP
#include <omp.h>
static void sweep_nonzeros_omp(orc_model *M)
{
  const uint32_t K = M->K, x = M->bias ? K + 2 : K;
#pragma omp parallel
  {
    double *phi = (double *)malloc(sizeof(double) * (x + 2));
#pragma omp for schedule(dynamic, 64)
    for (int64_t n = 0; n < (int64_t)M->n; ++n) {
      const double *elt = M->theta.Elogv + (size_t)n * K;
      double *st = M->theta.snext + (size_t)n * K;
      for (int64_t j = M->rowptr[n]; j < M->rowptr[n + 1]; ++j) {
        uint32_t m = M->col[j];
        uint8_t y = M->val ? M->val[j] : 1;
        const double *elb = M->beta.Elogv + (size_t)m * K;
        for (uint32_t k = 0; k < K; ++k) phi[k] = elt[k] + elb[k];
        if (M->bias) { phi[K] = M->ubias.Elogv[n]; phi[K + 1] = M->ibias.Elogv[m]; }
        orc_lognormalize(phi, x);
        if (y > 1) for (uint32_t k = 0; k < x; ++k) phi[k] *= y;
        double *sb = M->beta.snext + (size_t)m * K;
        for (uint32_t k = 0; k < K; ++k) st[k] += phi[k];
        for (uint32_t k = 0; k < K; ++k) {
#pragma omp atomic
          sb[k] += phi[k];
        }
        if (M->bias) {
          M->ubias.snext[n] += phi[K];
#pragma omp atomic
          M->ibias.snext[m] += phi[K + 1];
        }
      }
    }
    free(phi);
  }
}
int orc_omp_threads(void) { return omp_get_max_threads(); }
#else
static void sweep_nonzeros_omp(orc_model *M) { sweep_nonzeros(M); }
int orc_omp_threads(void) { return 1; }
#endif

/* one iteration with the parallel step A (the row sweeps stay serial: they are
   a few percent of the time) */
void orc_model_iterate_all_cores(orc_model *M)
{
  void (*serial)(orc_model *) = sweep_nonzeros; (void)serial;
  /* run step A in parallel, then the rest of the iteration as usual by
     temporarily making the serial step A a no-op */
  sweep_nonzeros_omp(M);
  M->skip_step_a = 1; g_parallel_sweeps = 1;
  if (M->hier) iterate_hier(M); else iterate_flat(M);
  M->skip_step_a = 0; g_parallel_sweeps = 0;
}

void orc_model_iterate(orc_model *M, int n_iters)
{
  for (int t = 0; t < n_iters; ++t) {
    if (M->hier) iterate_hier(M); else iterate_flat(M);
  }
}

/* hgaprec.cc:1563-1570 */
static double log_factorial(uint32_t n)
{
  double v = log(1);
  for (uint32_t i = 2; i <= n; ++i) v += log(i);
  return v;
}

/* rating_likelihood_hier hgaprec.cc:1538-1560 ; rating_likelihood 1503-1536 */
static double rating_likelihood(const orc_model *M, uint32_t p, uint32_t q, uint8_t y)
{
  const double *et = M->theta.Ev + (size_t)p * M->K, *eb = M->beta.Ev + (size_t)q * M->K;
  double s = .0;
  for (uint32_t k = 0; k < M->K; ++k) s += et[k] * eb[k];
  if (M->bias) s += M->ubias.Ev[p] + M->ibias.Ev[q];
  if (s < 1e-30) s = 1e-30;
  if (M->binary) return y == 0 ? -s : log(1 - exp(-s));
  return y * log(s) - s - log_factorial(y);
}

double orc_model_heldout_sum(const orc_model *M, const uint32_t *u,
                             const uint32_t *i, const int32_t *y, uint64_t cnt)
{
  double s = .0;
  for (uint64_t a = 0; a < cnt; ++a)       /* yval_t r = i->second: uint8 wrap */
    s += rating_likelihood(M, u[a], i[a], (uint8_t)y[a]);
  return s;
}

/* compute_elbo_term_helper: GPMatrix gpbase.hh:360-387, GPMatrixGR 717-741,
   GPArray 951-969 (k == 1 objects built as GPArray: is_array) */
static double gp_elbo_term(const gp *g, int is_array)
{
  double s = .0;
  if (is_array) {
    for (uint32_t n = 0; n < g->n; ++n) {
      double a, b;
      make_nonzero(g->scurr[n], g->rcurr[n], &a, &b);
      s += g->sprior * log(g->rprior) + (g->sprior - 1) * g->Elogv[n];
      s -= g->rprior * g->Ev[n] + lgamma(g->sprior);
      s -= a * log(b) + (a - 1) * g->Elogv[n];
      s += b * g->Ev[n] + lgamma(a);
    }
    return s;
  }
  for (uint32_t n = 0; n < g->n; ++n) {
    const double *ev = g->Ev + (size_t)n * g->k, *el = g->Elogv + (size_t)n * g->k;
    for (uint32_t k = 0; k < g->k; ++k) {
      if (g->hier && !g->global_rate) {
        s += g->sprior * g->hier_log_rprior[n] + (g->sprior - 1) * el[k];
        s -= g->hier_rprior[n] * ev[k] + lgamma(g->sprior);
      } else {
        s += g->sprior * log(g->rprior) + (g->sprior - 1) * el[k];
        s -= g->rprior * ev[k] + lgamma(g->sprior);
      }
    }
    for (uint32_t k = 0; k < g->k; ++k) {
      double a, b;
      make_nonzero(g->scurr[(size_t)n * g->k + k],
                   g->global_rate ? g->rcurr[k] : g->rcurr[(size_t)n * g->k + k], &a, &b);
      s -= a * log(b) + (a - 1) * el[k];
      s += b * ev[k] + lgamma(a);
    }
  }
  return s;
}

/* HGAPRec::logl hgaprec.cc:2160-2255 (gsl_sf_lngamma == lgamma for x > 0) */
double orc_model_elbo(orc_model *M)
{
  const uint32_t K = M->K, x = M->bias ? K + 2 : K;
  double *phi = M->phi;
  double s = .0;
  for (uint32_t n = 0; n < M->n; ++n) {
    const double *elt = M->theta.Elogv + (size_t)n * K, *et = M->theta.Ev + (size_t)n * K;
    for (int64_t j = M->rowptr[n]; j < M->rowptr[n + 1]; ++j) {
      uint32_t m = M->col[j];
      uint8_t y = M->val ? M->val[j] : 1;
      const double *elb = M->beta.Elogv + (size_t)m * K, *eb = M->beta.Ev + (size_t)m * K;
      for (uint32_t k = 0; k < K; ++k) phi[k] = elt[k] + elb[k];
      if (M->bias) { phi[K] = M->ubias.Elogv[n]; phi[K + 1] = M->ibias.Elogv[m]; }
      orc_lognormalize(phi, x);
      if (y > 1)
        for (uint32_t k = 0; k < x; ++k) phi[k] *= y;
      double v = .0;
      for (uint32_t k = 0; k < K; ++k)
        v += y * phi[k] * (elt[k] + elb[k] - log(phi[k]));
      s += v;
      if (M->bias) {
        s += y * phi[K] * (M->ubias.Elogv[n] - log(phi[K]));
        s += y * phi[K + 1] * (M->ibias.Elogv[m] - log(phi[K + 1]));
      }
      for (uint32_t k = 0; k < K; ++k) s -= et[k] * eb[k];
      if (M->bias) { s -= M->ubias.Ev[n]; s -= M->ibias.Ev[m]; }
    }
  }
  s += gp_elbo_term(&M->theta, 0);
  s += gp_elbo_term(&M->beta, 0);
  if (M->hier) {
    s += gp_elbo_term(&M->xi, 1);
    s += gp_elbo_term(&M->eta, 1);
  }
  if (M->bias) {                      /* n x 1 GPMatrix objects, _hier never set */
    s += gp_elbo_term(&M->ubias, 0);
    s += gp_elbo_term(&M->ibias, 0);
  }
  return s;
}

static gp *state_gp(const orc_model *M, int which)
{
  switch (which / 4) {
    case 0: return (gp *)&M->theta;
    case 1: return (gp *)&M->beta;
    case 2: return M->hier ? (gp *)&M->xi : NULL;
    case 3: return M->hier ? (gp *)&M->eta : NULL;
    case 4: return M->bias ? (gp *)&M->ubias : NULL;
    case 5: return M->bias ? (gp *)&M->ibias : NULL;
  }
  return NULL;
}

size_t orc_model_state(const orc_model *M, int which, const double **ptr)
{
  gp *g = state_gp(M, which);
  if (!g) { *ptr = NULL; return 0; }
  size_t ns = (size_t)g->n * g->k;
  switch (which % 4) {
    case 0: *ptr = g->scurr; return ns;
    case 1: *ptr = g->rcurr; return g->rsize;
    case 2: *ptr = g->Ev; return ns;
    default: *ptr = g->Elogv; return ns;
  }
}

int orc_model_set_state(orc_model *M, int which, const double *src, size_t count)
{
  gp *g = state_gp(M, which);
  if (!g) return -1;
  size_t ns = (size_t)g->n * g->k;
  double *dst; size_t want;
  switch (which % 4) {
    case 0: dst = g->scurr; want = ns; break;
    case 1: dst = g->rcurr; want = g->rsize; break;
    case 2: dst = g->Ev; want = ns; break;
    default: dst = g->Elogv; want = ns; break;
  }
  if (count != want) return -1;
  memcpy(dst, src, count * sizeof(double));
  return 0;
}

/* ------------------------------------------------------------------ */
/* TSV writers: D2Array<double>::save matrix.hh:1140-1166,              */
/*              D1Array<double>::save matrix.hh:725-744                 */
/* id column = IDMap lookup of the row index, else the index itself     */
/* ------------------------------------------------------------------ */
int orc_save_matrix(const char *path, const double *a, uint32_t rows,
                    uint32_t cols, const uint32_t *seq2id, uint32_t nids)
{
  FILE *tf = fopen(path, "w");
  if (!tf) return -1;
  for (uint32_t i = 0; i < rows; ++i) {
    uint32_t id = (seq2id && i < nids) ? seq2id[i] : i;
    fprintf(tf, "%d\t", i);
    fprintf(tf, "%d\t", id);
    for (uint32_t k = 0; k < cols; ++k) {
      if (k == cols - 1) fprintf(tf, "%.8f\n", a[(size_t)i * cols + k]);
      else               fprintf(tf, "%.8f\t", a[(size_t)i * cols + k]);
    }
  }
  fclose(tf);
  return 0;
}

int orc_save_vector(const char *path, const double *a, uint32_t rows,
                    const uint32_t *seq2id, uint32_t nids)
{
  FILE *tf = fopen(path, "w");
  if (!tf) return -1;
  for (uint32_t i = 0; i < rows; ++i) {
    uint32_t id = (seq2id && i < nids) ? seq2id[i] : i;
    fprintf(tf, "%d\t", i);
    fprintf(tf, "%d\t", id);
    fprintf(tf, "%.8f\n", a[i]);
  }
  fclose(tf);
  return 0;
}

/* GP*::save_state gpbase.hh:389-398,743-752,971-980 */
static void save_state(const char *outdir, const char *name, const gp *g,
                       int is_array, const uint32_t *seq2id, uint32_t nids)
{
  char p[4096];
  snprintf(p, sizeof p, "%s/%s_shape.tsv", outdir, name);
  if (is_array) orc_save_vector(p, g->scurr, g->n, seq2id, nids);
  else orc_save_matrix(p, g->scurr, g->n, g->k, seq2id, nids);
  snprintf(p, sizeof p, "%s/%s_rate.tsv", outdir, name);
  if (is_array) orc_save_vector(p, g->rcurr, g->n, seq2id, nids);
  else if (g->global_rate) orc_save_vector(p, g->rcurr, g->k, seq2id, nids);
  else orc_save_matrix(p, g->rcurr, g->n, g->k, seq2id, nids);
  snprintf(p, sizeof p, "%s/%s.tsv", outdir, name);
  if (is_array) orc_save_vector(p, g->Ev, g->n, seq2id, nids);
  else orc_save_matrix(p, g->Ev, g->n, g->k, seq2id, nids);
}

/* hgaprec.cc:2137-2158 */
static void save_model(const orc_model *M, const orc_ratings *R, const char *outdir)
{
  const uint32_t *s2u = R->seq2user, *s2i = R->seq2item;
  if (M->hier) {
    save_state(outdir, "hbeta", &M->beta, 0, s2i, R->nitems);
    save_state(outdir, "betarate", &M->eta, 1, s2i, R->nitems);
    save_state(outdir, "htheta", &M->theta, 0, s2u, R->nusers);
    save_state(outdir, "thetarate", &M->xi, 1, s2u, R->nusers);
  } else {
    save_state(outdir, "beta", &M->beta, 0, s2i, R->nitems);
    save_state(outdir, "theta", &M->theta, 0, s2u, R->nusers);
  }
  if (M->bias) {
    /* bias objects are n x 1 GPMatrix: D2Array::save with one column */
    save_state(outdir, "betabias", &M->ibias, 0, s2i, R->nitems);
    save_state(outdir, "thetabias", &M->ubias, 0, s2u, R->nusers);
  }
}


/* ------------------------------------------------------------------ */
/* Ranking evaluation at report steps: compute_precision               */
/* (hgaprec.cc:1703-1848), compute_itemrank (1606-1701),               */
/* gen_ranking_for_users (2087-2112), prediction_score[_hier]          */
/* (1850-1877, 1966-1991; _use_rate_as_score is true, hgaprec.cc:31)    */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t first; double second; } orc_kv;   /* std::pair<uint32_t,double>: 16 bytes */

static int cmppairval(const void *p1, const void *p2)        /* matrix.hh:288-293 */
{
  const orc_kv *u = (const orc_kv *)p1, *v = (const orc_kv *)p2;
  return u->second < v->second ? 1 : u->second == v->second ? 0 : -1;
}

typedef struct {
  uint32_t *sampled; uint32_t nsampled, cap;    /* _sampled_users (a std::map: sorted keys) */
  /* per-user scratch rows over the items */
  uint8_t *train_r;  uint8_t *is_valid;  int32_t *test_v;  uint8_t *has_test;
  uint32_t *item_deg;
  orc_kv *mlist;
} orc_eval;

static int cmp_u32(const void *a, const void *b)
{ uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }

static orc_eval *eval_new(const orc_ratings *R)
{
  orc_eval *E = (orc_eval *)calloc(1, sizeof(*E));
  uint32_t m = R->nitems ? R->nitems : 1;
  E->train_r = (uint8_t *)calloc(m, 1); E->is_valid = (uint8_t *)calloc(m, 1);
  E->test_v = (int32_t *)calloc(m, 4); E->has_test = (uint8_t *)calloc(m, 1);
  E->item_deg = (uint32_t *)calloc(m, 4);
  for (uint64_t j = 0; j < R->ntr; ++j) E->item_deg[R->col[j]]++;     /* _movies[m]->size() */
  E->mlist = (orc_kv *)malloc(sizeof(orc_kv) * m);
  return E;
}
static void eval_free(orc_eval *E)
{
  if (!E) return;
  free(E->sampled); free(E->train_r); free(E->is_valid); free(E->test_v); free(E->has_test);
  free(E->item_deg); free(E->mlist); free(E);
}

/* first index of user n in a (u,i)-sorted held-out list */
static uint64_t ho_lower(const orc_ratings *R, int w, uint32_t n)
{
  uint64_t lo = 0, hi = R->nho[w];
  while (lo < hi) { uint64_t mid = (lo + hi) / 2; if (R->ho_u[w][mid] < n) lo = mid + 1; else hi = mid; }
  return lo;
}

static void eval_load_user(orc_eval *E, const orc_ratings *R, uint32_t n, int set)
{
  for (int64_t j = R->rowptr[n]; j < R->rowptr[n + 1]; ++j) E->train_r[R->col[j]] = set ? R->val[j] : 0;
  for (uint64_t a = ho_lower(R, 0, n); a < R->nho[0] && R->ho_u[0][a] == n; ++a)
    E->is_valid[R->ho_i[0][a]] = (uint8_t)set;
  for (uint64_t a = ho_lower(R, 1, n); a < R->nho[1] && R->ho_u[1][a] == n; ++a) {
    E->has_test[R->ho_i[1][a]] = (uint8_t)set; E->test_v[R->ho_i[1][a]] = set ? R->ho_y[1][a] : 0;
  }
}

static double prediction_score(const orc_model *M, uint32_t user, uint32_t movie)
{
  const double *et = M->theta.Ev + (size_t)user * M->K, *eb = M->beta.Ev + (size_t)movie * M->K;
  double s = .0;
  for (uint32_t k = 0; k < M->K; ++k) s += et[k] * eb[k];
  if (M->bias) s += M->ubias.Ev[user] + M->ibias.Ev[movie];
  return s;                                   /* _use_rate_as_score */
}

/* ratings.hh:183-189 */
static int test_hit(const orc_ratings *R, int v)
{ return R->binary ? v >= 1 : (uint32_t)v >= R->thr; }

static void eval_score_and_sort(orc_eval *E, const orc_model *M, uint32_t n)
{
  for (uint32_t m = 0; m < M->m; ++m) {
    E->mlist[m].first = m;
    E->mlist[m].second = (E->train_r[m] > 0 || E->is_valid[m]) ? .0 : prediction_score(M, n, m);
  }
  qsort(E->mlist, M->m, sizeof(orc_kv), cmppairval);       /* D1Array<KV>::sort_by_value */
}

static void compute_precision(orc_eval *E, orc_model *M, const orc_ratings *R, uint32_t iter,
                              int save_ranking_file, FILE *pf, const char *outdir)
{
  char p[4096];
  if (iter % 100 == 0 && iter > 0) save_ranking_file = 1;
  double mhits10 = 0, mhits100 = 0;
  uint32_t total_users = 0;
  FILE *f = NULL;
  if (save_ranking_file) { snprintf(p, sizeof p, "%s/ranking.tsv", outdir); f = fopen(p, "w"); }
  if (!save_ranking_file) {                     /* hgaprec.cc:1714-1721 */
    uint32_t cnt = 0;
    if (!E->sampled) { E->cap = 1024; E->sampled = (uint32_t *)malloc(4 * E->cap); }
    do {
      uint32_t n = (uint32_t)orc_rng_uniform_int(&M->rng, M->n);
      int seen = 0;
      for (uint32_t a = 0; a < cnt; ++a) if (E->sampled[a] == n) { seen = 1; break; }
      if (!seen) E->sampled[cnt++] = n;
    } while (cnt < 1000 && cnt < M->n / 2);
    qsort(E->sampled, cnt, 4, cmp_u32);
    E->nsampled = cnt;
  }
  for (uint32_t a = 0; a < E->nsampled; ++a) {
    uint32_t n = E->sampled[a];
    eval_load_user(E, R, n, 1);
    eval_score_and_sort(E, M, n);
    uint32_t hits10 = 0, hits100 = 0;
    for (uint32_t j = 0; j < M->m && j < 100; ++j) {          /* _topN_by_user = 100 */
      uint32_t m = E->mlist[j].first; double pred = E->mlist[j].second;
      int v = 0;
      if (E->has_test[m]) {
        v = test_hit(R, E->test_v[m]) ? 1 : 0;
        if (j < 10) { if (v > 0) { hits10++; hits100++; } }
        else if (j < 100) { if (v > 0) hits100++; }
      }
      if (save_ranking_file && f && E->train_r[m] == 0)
        fprintf(f, "%d\t%d\t%.5f\t%d\n", R->seq2user[n], R->seq2item[m], pred, v);
    }
    mhits10 += (double)hits10 / 10;
    mhits100 += (double)hits100 / 100;
    total_users++;
    eval_load_user(E, R, n, 0);
  }
  if (f) fclose(f);
  fprintf(pf, "%d\t%.5f\t%.5f\n", total_users, (double)mhits10 / total_users, (double)mhits100 / total_users);
  fflush(pf);
}

static void compute_itemrank(orc_eval *E, orc_model *M, const orc_ratings *R, uint32_t iter,
                             int final, const char *outdir)
{
  char p[4096];
  if (iter % 100 == 0 && iter > 0) final = 1;
  if (!final) return;
  uint32_t total_users = 0;
  snprintf(p, sizeof p, "%s/itemrank.tsv", outdir); FILE *f = fopen(p, "w");
  snprintf(p, sizeof p, "%s/meanrank.txt", outdir); FILE *itemf = fopen(p, "w");
  double sum_rank = .0, sum_reciprocal_rank = .0;
  for (uint32_t a = 0; a < E->nsampled; ++a) {
    uint32_t n = E->sampled[a];
    eval_load_user(E, R, n, 1);
    eval_score_and_sort(E, M, n);
    double rank_ui = .0, reciprocal_rank_ui = .0;
    uint32_t ntestitems = 0, nranked = 0;
    for (uint32_t j = 0; j < M->m; ++j) {
      uint32_t m = E->mlist[j].first; double pred = E->mlist[j].second;
      if (E->train_r[m] == 0) nranked++;
      if (E->has_test[m] && test_hit(R, E->test_v[m])) {
        ntestitems++;
        fprintf(f, "%d\t%d\t%.5f\t%d\t%d\n", n, m, pred, j, E->item_deg[m]);
        rank_ui += (j + 1);
        reciprocal_rank_ui += 1 / (j + 1);                  /* integer division, as written */
      }
    }
    if (ntestitems > 0 && nranked > 0) {
      sum_rank += (rank_ui / nranked) / ntestitems;
      sum_reciprocal_rank += reciprocal_rank_ui / ntestitems;
      total_users++;
    }
    eval_load_user(E, R, n, 0);
  }
  fclose(f);
  fprintf(itemf, "%d\t%.5f\t%.5f\n", total_users, (double)sum_rank / total_users,
          (double)sum_reciprocal_rank / total_users);
  fclose(itemf);
}

/* gen_ranking_for_users(false) hgaprec.cc:2087-2112 + read_test_users ratings.cc:273-292 */
static void gen_ranking_for_users(orc_eval *E, orc_model *M, const orc_ratings *R, uint32_t iter,
                                  FILE *pf, const char *datadir, const char *outdir)
{
  char p[4096];
  snprintf(p, sizeof p, "%s/test_users.tsv", datadir);
  FILE *f = fopen(p, "r");
  if (!f) return;
  uint32_t cnt = 0, uid = 0;
  if (!E->sampled) { E->cap = 1024; E->sampled = (uint32_t *)malloc(4 * E->cap); }
  while (!feof(f)) {
    if (fscanf(f, "%u\n", &uid) < 0) break;
    uint32_t n;
    if (!idmap_find(&R->user2seq, uid, &n)) continue;
    int seen = 0;
    for (uint32_t a = 0; a < cnt; ++a) if (E->sampled[a] == n) { seen = 1; break; }
    if (seen) continue;
    if (cnt == E->cap) { E->cap *= 2; E->sampled = (uint32_t *)realloc(E->sampled, 4 * E->cap); }
    E->sampled[cnt++] = n;
  }
  fclose(f);
  qsort(E->sampled, cnt, 4, cmp_u32);
  E->nsampled = cnt;
  compute_precision(E, M, R, iter, 1, pf, outdir);
  compute_itemrank(E, M, R, iter, 1, outdir);
}

/* main.cc:234-361 + HGAPRec::vb_hier / vb / vb_bias report logic +
   compute_likelihood hgaprec.cc:1439-1501 (stop rule) */
int orc_run(const orc_run_args *a)
{
  char p[4096];
  orc_ratings *R = orc_ratings_new(a->n, a->m, a->binary, a->rating_threshold);
  snprintf(p, sizeof p, "%s/train.tsv", a->datadir);
  if (orc_ratings_read_train(R, p)) { orc_ratings_free(R); return -1; }
  { char q[4096];
    snprintf(p, sizeof p, "%s/byusers.tsv", a->outdir);
    snprintf(q, sizeof q, "%s/byitems.tsv", a->outdir);
    orc_ratings_write_marginals(R, p, q); }
  snprintf(p, sizeof p, "%s/validation.tsv", a->datadir);
  if (orc_ratings_read_heldout(R, p, 0)) { orc_ratings_free(R); return -1; }
  snprintf(p, sizeof p, "%s/test.tsv", a->datadir);
  if (orc_ratings_read_heldout(R, p, 1)) { orc_ratings_free(R); return -1; }

  orc_model *M = orc_model_new(R->nusers, R->nitems, a->k, a->hier, a->bias, a->binary);
  orc_model_set_csr(M, R->rowptr, R->col, R->val);
  orc_model_set_novb(M, a->novb);
  snprintf(p, sizeof p, "%s/validation.txt", a->outdir); FILE *vf = fopen(p, "w");
  snprintf(p, sizeof p, "%s/test.txt", a->outdir);       FILE *tf = fopen(p, "w");
  if (!vf || !tf) return -1;
  snprintf(p, sizeof p, "%s/logl.txt", a->outdir); FILE *af = fopen(p, "w");
  snprintf(p, sizeof p, "%s/precision.txt", a->outdir); FILE *pf = fopen(p, "w");
  orc_eval *E = eval_new(R);
  time_t start = time(0);
  orc_model_initialize(M, a->seed);

  double prev_h = .0; uint32_t nh = 0; uint32_t iter = 0; int stopped = 0;
  while (1) {
    if (a->hier && iter > a->max_iterations) break;     /* hgaprec.cc:1337-1339 */
    if (!a->hier && iter > 100000u) break;              /* vb()/vb_bias() have no cap */
    orc_model_iterate(M, 1);
    if (iter % a->rfreq == 0) {
      for (int w = 0; w < 2 && !stopped; ++w) {
        uint64_t k = R->nho[w];
        double s = orc_model_heldout_sum(M, R->ho_u[w], R->ho_i[w], R->ho_y[w], k);
        FILE *ff = w == 0 ? vf : tf;
        fprintf(ff, "%d\t%d\t%.9f\t%d\n", iter, (int)(time(0) - start), s / k, (int)k);
        fflush(ff);
        if (w != 0) continue;
        double av = s / k; int stop = 0, why = -1;
        if (iter > 30) {
          if (av > prev_h && prev_h != 0 && fabs((av - prev_h) / prev_h) < 0.000001) {
            stop = 1; why = 0;
          } else if (av < prev_h) nh++;
          else if (av > prev_h) nh = 0;
          if (nh > 2) { why = 1; stop = 1; }
        }
        prev_h = av;
        snprintf(p, sizeof p, "%s/max.txt", a->outdir);
        FILE *f = fopen(p, "w");
        fprintf(f, "%d\t%d\t%.5f\t%d\n", iter, (int)(time(0) - start), av, why);
        fclose(f);
        if (stop) {                                /* do_on_stop(); exit(0) */
          save_model(M, R, a->outdir);
          gen_ranking_for_users(E, M, R, iter, pf, a->datadir, a->outdir);
          stopped = 1;
        }
      }
      if (stopped) break;
      save_model(M, R, a->outdir);
      compute_precision(E, M, R, iter, 0, pf, a->outdir);
      if (a->hier || !a->bias)                   /* vb_bias() has no compute_itemrank call */
        compute_itemrank(E, M, R, iter, 0, a->outdir);
      if (a->logl && af) { fprintf(af, "%.5f\n", orc_model_elbo(M)); fflush(af); }  /* hgaprec.cc:1426-1427 */
    }
    iter++;
  }
  fclose(vf); fclose(tf); if (af) fclose(af); if (pf) fclose(pf);
  eval_free(E);
  orc_model_free(M); orc_ratings_free(R);
  return (int)iter;
}

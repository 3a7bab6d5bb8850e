/*
 * hpf_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Single-thread fp64 CPU restatement of the reference's CAVI hot path
 * (premgopalan/hgaprec: src/hgaprec.cc, src/gpbase.hh, src/matrix.hh,
 * src/ratings.{hh,cc}).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product
 * (libhpf_hip.so, libhgaprec_host.so, the hgaprec CLI) never links it.
 *
 * PARITY STATUS: "parity unpinned" for the end-to-end path.  The complete
 * reference cannot be built in this image (it needs GSL, which is absent,
 * and no stand-in is allowed), and the reference ships no tests or golden
 * vectors.  What IS pinned, by tests/test_oracle_*.py:
 *   - the per-nonzero softmax (D1Array::logsum/lognormalize/scale), the
 *     row add (D2Array::add_slice), the TSV number format (D2Array::save,
 *     D1Array::save) and the output-directory name + param.txt head
 *     (Env::Env) -- bit-exact against the reference's own GSL-free
 *     translation units compiled in place (oracle/_ref, see oracle/Makefile);
 *   - MT19937 (GSL gsl_rng_mt19937, 2002 seeding, default seed 0 -> 4357)
 *     against the published first output 4293858116 and numpy RandomState;
 *   - digamma against mpmath (50 digits) / scipy.
 * See DESIGN.md section "Oracle".
 */
#ifndef HPF_ORACLE_H
#define HPF_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- MT19937 as used through gsl_rng_default (hgaprec.cc:34-38) ---- */
typedef struct { uint32_t mt[624]; int mti; } orc_rng;
void     orc_rng_seed(orc_rng *r, unsigned long seed);
uint32_t orc_rng_u32(orc_rng *r);
double   orc_rng_uniform(orc_rng *r);
unsigned long orc_rng_uniform_int(orc_rng *r, unsigned long n);

/* digamma for x > 0 (stands where the reference calls gsl_sf_psi) */
double orc_psi(double x);

/* matrix.hh:367-389,399-406 : sequential log-add-exp softmax, in place */
double orc_logsum(const double *x, uint32_t n);
/* D1Array<T>::sum (matrix.hh:327-335): the left-to-right sum the row / column sums are made of */
double orc_sum_strided(const double *d, uint32_t n, size_t stride);
/* test entry: the model's own sweep steps on caller arrays (see hpf_oracle.c) */
void orc_test_sweep_steps(int mode, uint32_t rows, uint32_t k, const double *snext_in, const double *ev,
                          const double *u, double v, double *scurr, double *rcurr, double *snext, double *rnext);
void   orc_lognormalize(double *x, uint32_t n);

/* ---- ratings store (ratings.cc:63-119) ---- */
typedef struct orc_ratings orc_ratings;
/* cap_n / cap_m are the CLI -n / -m capacities */
orc_ratings *orc_ratings_new(uint32_t cap_n, uint32_t cap_m, int binary,
                             uint32_t rating_threshold);
void  orc_ratings_free(orc_ratings *r);
/* returns 0, or -1 on the reference's "unexpected lines" exit / open failure */
int   orc_ratings_read_train(orc_ratings *r, const char *path);
/* which: 0 = validation, 1 = test */
int   orc_ratings_read_heldout(orc_ratings *r, const char *path, int which);
uint32_t orc_ratings_n(const orc_ratings *r);
uint32_t orc_ratings_m(const orc_ratings *r);
uint64_t orc_ratings_nnz(const orc_ratings *r);
/* CSR in the reference's visiting order (user seq, then file order);
 * val already carries the uint8 wrap and the "last duplicate wins" rule */
const int64_t  *orc_ratings_rowptr(const orc_ratings *r);
const uint32_t *orc_ratings_col(const orc_ratings *r);
const uint8_t  *orc_ratings_val(const orc_ratings *r);
const uint32_t *orc_ratings_seq2user(const orc_ratings *r);
const uint32_t *orc_ratings_seq2item(const orc_ratings *r);
/* held-out maps sorted by (user seq, item seq) like std::map<Rating,int> */
uint64_t orc_ratings_heldout_count(const orc_ratings *r, int which);
const uint32_t *orc_ratings_heldout_u(const orc_ratings *r, int which);
const uint32_t *orc_ratings_heldout_i(const orc_ratings *r, int which);
const int32_t  *orc_ratings_heldout_y(const orc_ratings *r, int which);
/* ratings.cc:217-271 */
int orc_ratings_write_marginals(const orc_ratings *r, const char *byusers,
                                const char *byitems);

/* ---- model ---- */
typedef struct orc_model orc_model;
enum {
  ORC_THETA_SHAPE = 0, ORC_THETA_RATE, ORC_THETA_E, ORC_THETA_ELOG,
  ORC_BETA_SHAPE, ORC_BETA_RATE, ORC_BETA_E, ORC_BETA_ELOG,
  ORC_XI_SHAPE, ORC_XI_RATE, ORC_XI_E, ORC_XI_ELOG,
  ORC_ETA_SHAPE, ORC_ETA_RATE, ORC_ETA_E, ORC_ETA_ELOG,
  ORC_UBIAS_SHAPE, ORC_UBIAS_RATE, ORC_UBIAS_E, ORC_UBIAS_ELOG,
  ORC_IBIAS_SHAPE, ORC_IBIAS_RATE, ORC_IBIAS_E, ORC_IBIAS_ELOG,
  ORC_NUM_STATE
};
orc_model *orc_model_new(uint32_t n, uint32_t m, uint32_t K, int hier, int bias,
                         int binary);
void orc_model_free(orc_model *M);
/* Env::vb = !novb (main.cc:64,194); only vb_bias() reads it (hgaprec.cc:1250) */
void orc_model_set_novb(orc_model *M, int novb);
/* borrowed pointers; must outlive the model */
void orc_model_set_csr(orc_model *M, const int64_t *rowptr, const uint32_t *col,
                       const uint8_t *val);
/* hgaprec.cc:153-204 with gsl_rng seeded as hgaprec.cc:34-38 */
void orc_model_initialize(orc_model *M, double seed);
/* one full sweep A-F (hier) / vb() / vb_bias() body, n_iters times */
void orc_model_iterate(orc_model *M, int n_iters);
/* hgaprec.cc:1439-1465 ; y is the int stored in the CountMap */
double orc_model_heldout_sum(const orc_model *M, const uint32_t *u,
                             const uint32_t *i, const int32_t *y, uint64_t cnt);
/* all-cores variant (OpenMP over users, atomics on item rows),
 * for bench.py's extra cpu_baseline_all_cores figure only */
void orc_model_iterate_all_cores(orc_model *M);
int  orc_omp_threads(void);
/* HGAPRec::logl hgaprec.cc:2160-2255: the variational bound written to logl.txt */
double orc_model_elbo(orc_model *M);
/* state access: row-major doubles; returns element count (0 if absent).
 * For the non-hier model *_RATE is the K-vector. */
size_t orc_model_state(const orc_model *M, int which, const double **ptr);
/* overwrite E/Elog/shape state (used to start oracle and device from the
 * same arbitrary point) */
int orc_model_set_state(orc_model *M, int which, const double *src, size_t count);
orc_rng *orc_model_rng(orc_model *M);

/* ---- end-to-end run mirroring main.cc + HGAPRec::vb*() ----
 * Writes validation.txt, test.txt, max.txt, factor TSVs, byusers/byitems
 * into outdir (must exist).  Returns the last iteration executed. */
typedef struct {
  const char *datadir;
  const char *outdir;
  uint32_t n, m, k;
  int hier, bias, binary;
  uint32_t rating_threshold;
  uint32_t rfreq;
  uint32_t max_iterations;
  double seed;
  int logl;                 /* -logl: append the bound to logl.txt at every report */
  int novb;                 /* -novb */
} orc_run_args;
int orc_run(const orc_run_args *a);

/* TSV writers (matrix.hh:725-744,1140-1166) */
int orc_save_matrix(const char *path, const double *a, uint32_t rows,
                    uint32_t cols, const uint32_t *seq2id, uint32_t nids);
int orc_save_vector(const char *path, const double *a, uint32_t rows,
                    const uint32_t *seq2id, uint32_t nids);

#ifdef __cplusplus
}
#endif
#endif

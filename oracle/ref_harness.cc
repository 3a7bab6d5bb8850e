// ref_harness.cc -- TEST INFRASTRUCTURE ONLY.
//
// Driver for the GSL-free translation units of the reference, compiled IN
// PLACE from /root/reference/src (never copied): matrix.hh (D1Array /
// D2Array), env.hh (Env: output-directory naming, param.txt), log.cc.
// The rest of the reference (gpbase.hh, ratings.hh, hgaprec.cc) needs GSL,
// which this image lacks, so it is unbuildable here and is NOT part of
// this binary.  Outputs go to oracle/_ref/ only (git-ignored).
//
// Used by tools/make_golden.py to generate tests/golden/*.json and by
// tests/test_oracle_vs_ref.py (skipped where oracle/_ref/refpart is absent).
//
//   refpart softmax <in.bin> <out.bin>
//       in : u32 nrec ; per record u32 n, u32 y, f64 x[n]
//       out: per record f64 logsum, f64 phi[n]   (lognormalize, scale(y) if y>1)
//   refpart accumulate <in.bin> <out.bin>
//       in : u32 rows, u32 K, u32 width(K or K+2), u32 nrec ;
//            per record u32 row, u32 y, f64 x[width]
//       out: f64 M[rows*K]   (M starts at 0.3; add_slice of each phi)
//   refpart save <in.bin> <matrix.tsv> <vector.tsv>
//       in : u32 rows, u32 cols, u32 nids, u32 ids[nids], f64 a[rows*cols], f64 v[rows]
//   refpart env <cli flags as for hgaprec>      (runs Env::Env in the cwd; also prints file_str("/x.tsv"))
//   refpart arrays <in.bin> <out.bin>
//       in : u32 n, u32 maxn, f64 fill, f64 x[n]
//       out: f64 D1Array::sum(), f64 D1Array::sum(maxn), then a D2Array(3, n) after
//            set_elements(fill) (3n f64), then x after zero() (n f64)
//   refpart rows <in.bin> <out.bin>
//       the matrix.hh calls the sweep methods of gpbase.hh are made of (gpbase.hh itself needs GSL and is not
//       compiled: the ORDER of the calls below is this driver's restatement of gpbase.hh:149-246,560-578,
//       858-903, the arithmetic is the reference's own D2Array / D1Array code)
//       in : u32 mode, u32 rows, u32 k, f64 sprior, f64 rprior, f64 v,
//            f64 snext[rows*k], f64 ev[rows], f64 u[mode == 2 ? rows : k]
//       mode 0 (GPMatrix): rnext.set_elements(n, ev[n]) for every n; rnext.add_slice(i, u) for every i;
//                          scurr.swap(snext); rcurr.swap(rnext); snext/rnext.set_elements(prior)
//       mode 1 (GPMatrixGR): D1Array rnext += u; swaps; set_elements(prior)
//       mode 2 (GPArray):  snext[n] += v; D1Array rnext += u; swaps; set_elements(prior)
//       mode 3 (bias GPMatrix, k == 1): rnext[i][0] += v; swaps; set_elements(prior)
//       out: scurr, rcurr, snext, rnext (row-major; modes 1: rcurr / rnext are k long)
#include <map>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "env.hh"

string Env::prefix = "";
Logger::Level Env::level = Logger::DEBUG;
FILE *Env::_plogf = NULL;

typedef D1Array<double> Array;

static void rd(FILE *f, void *p, size_t n) {
  if (fread(p, 1, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

static int cmd_softmax(const char *in, const char *out) {
  FILE *f = fopen(in, "rb"), *g = fopen(out, "wb");
  uint32_t nrec; rd(f, &nrec, 4);
  for (uint32_t r = 0; r < nrec; ++r) {
    uint32_t n, y; rd(f, &n, 4); rd(f, &y, 4);
    Array phi(n);
    rd(f, phi.data(), 8 * (size_t)n);
    double ls = phi.logsum();
    phi.lognormalize();
    if (y > 1) phi.scale(y);
    fwrite(&ls, 8, 1, g);
    fwrite(phi.data(), 8, n, g);
  }
  fclose(f); fclose(g);
  return 0;
}

static int cmd_accumulate(const char *in, const char *out) {
  FILE *f = fopen(in, "rb"), *g = fopen(out, "wb");
  uint32_t rows, K, width, nrec;
  rd(f, &rows, 4); rd(f, &K, 4); rd(f, &width, 4); rd(f, &nrec, 4);
  D2Array<double> M(rows, K);
  M.set_elements(0.3);
  Array phi(width);
  for (uint32_t r = 0; r < nrec; ++r) {
    uint32_t row, y; rd(f, &row, 4); rd(f, &y, 4);
    rd(f, phi.data(), 8 * (size_t)width);
    phi.lognormalize();
    if (y > 1) phi.scale(y);
    M.add_slice(row, phi);
  }
  const double **d = M.const_data();
  for (uint32_t i = 0; i < rows; ++i) fwrite(d[i], 8, K, g);
  fclose(f); fclose(g);
  return 0;
}

static int cmd_arrays(const char *in, const char *out) {
  FILE *f = fopen(in, "rb"), *g = fopen(out, "wb");
  uint32_t n, maxn; double fill;
  rd(f, &n, 4); rd(f, &maxn, 4); rd(f, &fill, 8);
  Array x(n);
  rd(f, x.data(), 8 * (size_t)n);
  double s0 = x.sum(), s1 = x.sum(maxn);
  fwrite(&s0, 8, 1, g); fwrite(&s1, 8, 1, g);
  D2Array<double> M(3, n);
  M.set_elements(fill);
  const double **d = M.const_data();
  for (uint32_t i = 0; i < 3; ++i) fwrite(d[i], 8, n, g);
  x.zero();
  fwrite(x.data(), 8, n, g);
  fclose(f); fclose(g);
  return 0;
}

static void wr2(FILE *g, D2Array<double> &M, uint32_t rows, uint32_t k) {
  const double **d = M.const_data();
  for (uint32_t i = 0; i < rows; ++i) fwrite(d[i], 8, k, g);
}

static int cmd_rows(const char *in, const char *out) {
  FILE *f = fopen(in, "rb"), *g = fopen(out, "wb");
  uint32_t mode, rows, k; double sprior, rprior, v;
  rd(f, &mode, 4); rd(f, &rows, 4); rd(f, &k, 4); rd(f, &sprior, 8); rd(f, &rprior, 8); rd(f, &v, 8);
  if (mode == 0 || mode == 3) {
    D2Array<double> scurr(rows, k), snext(rows, k), rcurr(rows, k), rnext(rows, k);
    scurr.set_elements(sprior); rcurr.set_elements(rprior);
    snext.set_elements(sprior); rnext.set_elements(rprior);             // set_to_prior
    double **sd = snext.data();
    for (uint32_t i = 0; i < rows; ++i) rd(f, sd[i], 8 * (size_t)k);    // the phi sums already added
    Array ev(rows); rd(f, ev.data(), 8 * (size_t)rows);
    Array u(k); rd(f, u.data(), 8 * (size_t)k);
    if (mode == 0) {
      for (uint32_t n = 0; n < rows; ++n) rnext.set_elements(n, ev[n]);
      for (uint32_t i = 0; i < rows; ++i) rnext.add_slice(i, u);
    } else {
      double **r = rnext.data();
      for (uint32_t i = 0; i < rows; ++i) r[i][0] += v;
    }
    scurr.swap(snext); rcurr.swap(rnext);
    snext.set_elements(sprior); rnext.set_elements(rprior);
    wr2(g, scurr, rows, k); wr2(g, rcurr, rows, k); wr2(g, snext, rows, k); wr2(g, rnext, rows, k);
  } else if (mode == 1) {
    D2Array<double> scurr(rows, k), snext(rows, k);
    Array rcurr(k), rnext(k);
    scurr.set_elements(sprior); rcurr.set_elements(rprior);
    snext.set_elements(sprior); rnext.set_elements(rprior);
    double **sd = snext.data();
    for (uint32_t i = 0; i < rows; ++i) rd(f, sd[i], 8 * (size_t)k);
    Array ev(rows); rd(f, ev.data(), 8 * (size_t)rows);
    Array u(k); rd(f, u.data(), 8 * (size_t)k);
    rnext += u;
    scurr.swap(snext); rcurr.swap(rnext);
    snext.set_elements(sprior); rnext.set_elements(rprior);
    wr2(g, scurr, rows, k); fwrite(rcurr.data(), 8, k, g); wr2(g, snext, rows, k); fwrite(rnext.data(), 8, k, g);
  } else {
    Array scurr(rows), snext(rows), rcurr(rows), rnext(rows);
    scurr.set_elements(sprior); rcurr.set_elements(rprior);
    snext.set_elements(sprior); rnext.set_elements(rprior);
    rd(f, snext.data(), 8 * (size_t)rows);
    Array ev(rows); rd(f, ev.data(), 8 * (size_t)rows);
    Array u(rows); rd(f, u.data(), 8 * (size_t)rows);
    for (uint32_t n = 0; n < rows; ++n) snext[n] += v;
    rnext += u;
    scurr.swap(snext); rcurr.swap(rnext);
    snext.set_elements(sprior); rnext.set_elements(rprior);
    fwrite(scurr.data(), 8, rows, g); fwrite(rcurr.data(), 8, rows, g);
    fwrite(snext.data(), 8, rows, g); fwrite(rnext.data(), 8, rows, g);
  }
  fclose(f); fclose(g);
  return 0;
}

static int cmd_save(const char *in, const char *mt, const char *vt) {
  FILE *f = fopen(in, "rb");
  uint32_t rows, cols, nids;
  rd(f, &rows, 4); rd(f, &cols, 4); rd(f, &nids, 4);
  IDMap m;
  for (uint32_t i = 0; i < nids; ++i) { uint32_t id; rd(f, &id, 4); m[i] = id; }
  D2Array<double> A(rows, cols);
  double **d = A.data();
  for (uint32_t i = 0; i < rows; ++i) rd(f, d[i], 8 * (size_t)cols);
  Array v(rows);
  rd(f, v.data(), 8 * (size_t)rows);
  fclose(f);
  A.save(mt, m);
  v.save(vt, m);
  return 0;
}

static int cmd_env(int argc, char **argv) {
  string fname, label;
  uint32_t n = 0, m = 0, k = 0, rfreq = 10, max_iterations = 1000, rating_threshold = 1;
  double seed = 0, a = 0.3, b = 0.3, c = 0.3, d = 0.3;
  bool logl = false, binary = false, bias = false, hier = false;
  for (int i = 0; i < argc; ++i) {
    if (!strcmp(argv[i], "-dir")) fname = argv[++i];
    else if (!strcmp(argv[i], "-n")) n = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-m")) m = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-k")) k = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-rfreq")) rfreq = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-label")) label = argv[++i];
    else if (!strcmp(argv[i], "-logl")) logl = true;
    else if (!strcmp(argv[i], "-max-iterations")) max_iterations = atoi(argv[++i]);
    else if (!strcmp(argv[i], "-seed")) seed = atof(argv[++i]);
    else if (!strcmp(argv[i], "-a")) a = atof(argv[++i]);
    else if (!strcmp(argv[i], "-b")) b = atof(argv[++i]);
    else if (!strcmp(argv[i], "-c")) c = atof(argv[++i]);
    else if (!strcmp(argv[i], "-d")) d = atof(argv[++i]);
    else if (!strcmp(argv[i], "-binary-data")) binary = true;
    else if (!strcmp(argv[i], "-bias")) bias = true;
    else if (!strcmp(argv[i], "-hier")) hier = true;
    else if (!strcmp(argv[i], "-rating-threshold")) rating_threshold = atoi(argv[++i]);
    else { fprintf(stderr, "unknown flag %s\n", argv[i]); return 2; }
  }
  // argument order of main.cc:234-243
  Env env(n, m, k, fname, false, "", rfreq, false, label, logl, seed,
          max_iterations, false, "", false, a, b, c, d, Env::MENDELEY,
          true, binary, bias, hier, false, true, false, false, false, false,
          false, rating_threshold, false, false, 0.1, 10,
          false, false, false, false, false, false, false);
  printf("%s\n", Env::file_str("/x.tsv").c_str());
  printf("%s\n", Env::prefix.c_str());
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 4 && !strcmp(argv[1], "softmax")) return cmd_softmax(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "accumulate")) return cmd_accumulate(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "arrays")) return cmd_arrays(argv[2], argv[3]);
  if (argc >= 4 && !strcmp(argv[1], "rows")) return cmd_rows(argv[2], argv[3]);
  if (argc >= 5 && !strcmp(argv[1], "save")) return cmd_save(argv[2], argv[3], argv[4]);
  if (argc >= 2 && !strcmp(argv[1], "env")) return cmd_env(argc - 2, argv + 2);
  fprintf(stderr, "usage: refpart softmax|accumulate|arrays|rows|save|env ...\n");
  return 2;
}

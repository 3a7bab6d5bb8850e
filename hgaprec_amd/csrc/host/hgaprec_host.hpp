// hgaprec_host.hpp -- host side of the reference interface (no HIP here).
//
// Mirrors, for the hot path only, what the reference keeps on the host:
//   Env        CLI flags, output-directory name, param.txt   (src/main.cc:99-243, src/env.hh:216-408)
//   Ratings    train/validation/test.tsv -> CSR + held-out lists (src/ratings.cc:63-119,217-271)
//   Mt19937    the gsl_rng_default stream                    (hgaprec.cc:34-38)
//   GammaInit  HGAPRec::initialize draw order                (hgaprec.cc:153-204, gpbase.hh:292-340,939-949)
//   save_*     factor TSV writers                            (matrix.hh:725-744,1140-1166)
// The device side is reached only through include/hpf.h.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>
#include <memory>
#include <utility>
#include <vector>

namespace hgaprec {

// ---------------------------------------------------------------- Env -----
struct Env {
  // values as main.cc:41-97 initialises them
  std::string datfname, label;
  uint32_t n = 0, m = 0, k = 0;
  uint32_t rfreq = 10, max_iterations = 1000, rating_threshold = 1;
  double seed = 0, a = 0.3, b = 0.3, c = 0.3, d = 0.3;
  bool logl = false, batch = true, binary_data = false, bias = false, hier = false, vb = true;
  // flags that are parsed (they are part of the CLI) but select code that is
  // outside the hot path; `unsupported` names the first one seen
  std::string unsupported;
  // not part of the reference CLI: device ordinal for the HIP side, number of
  // GPUs (one process each) and how the per-iteration all-reduce is carried
  int device = 0; bool device_set = false;
  bool no_tiles = false;                // -no-tiles: never tile the phi passes (hpf_config.tiling = 1)
  bool w48 = false;                     // -w48: OPT-IN, lossy -- W kept as the top 48 bits of its fp64 value (hpf_config.w_storage = 2): both phi passes ~17 % faster at
                                        // K = 100; arithmetic and every exported number stay fp64.  Never the default: measured drift vs the exact path 1e-6 after 150 sweeps
                                        // (contract 1e-4); tests/test_gpu_cli.py runs C1 to its stop rule both ways
  bool plain_rows = false;              // -plain-rows: W as plain doubles from the start (hpf_config.w_storage = 3; a packed handle moves there by itself when a state does not fit)
  int ngpus = 1;
  std::string comm_mode = "rccl";      // "rccl" | "host" (host-staged, for tests)
  bool single_allreduce = false;        // -single-allreduce: ONE all-reduce of [m x ld | ld] per iteration after the user half, as
                                        // BASELINE.json words it, instead of the overlapped pair (item sums underneath the user half + tail)
  // extension (the reference has no training resume: its -load is inert,
  // main.cc:137-140): -checkpoint N writes <outdir>/checkpoint.r<rank>of<world>.bin
  // every N iterations, -resume continues from it
  uint32_t checkpoint_every = 0; bool resume = false;
  // extension (the reference reads TSV only): -cache keeps a
  // binary image of the parsed dataset (CSR + id maps + held-out sets) in
  // <dir>/hgaprec.cache.bin and loads it instead of the three TSVs while their
  // sizes / mtimes and -n -m -binary-data -rating-threshold are unchanged
  bool data_cache = false;

  std::string prefix;          // output directory (Env::prefix)
  FILE *plogf = nullptr;       // param.txt
  FILE *logf = nullptr;        // infer.log

  // main.cc:99-232.  Returns 0, or 1 for "unknown option" (the reference
  // asserts there); echo = print the "+ n = ..." lines like the reference.
  int parse(int argc, char **argv, bool echo, std::string *bad_option);
  // env.hh:283-369: the directory name
  std::string make_prefix() const;
  // env.hh:371-402: mkdir (log.cc:97-118), infer.log, param.txt head.
  // Returns 0 or -1.
  int open_output();
  void close_output();
  std::string file_str(const std::string &f) const { return prefix + f; }
  void plog(const std::string &key, double v);
  void plog(const std::string &key, bool v);
  void plog(const std::string &key, uint32_t v);
  void plog(const std::string &key, uint64_t v);
  void plog(const std::string &key, const std::string &v);
  void lerr(const char *fmt, ...);      // log.hh:49 (the only live log level)
};

// ------------------------------------------------------------ Ratings -----
struct HeldOut {          // std::map<Rating,int> flattened in key order
  std::vector<uint32_t> u, i;
  std::vector<int32_t> y;
};

struct Ratings {
  uint32_t cap_n = 0, cap_m = 0;      // env.n / env.m at read time
  bool binary = false;
  uint32_t rating_threshold = 1;
  uint32_t n = 0, m = 0;              // registered users / items
  uint64_t nratings = 0;
  std::vector<uint32_t> seq2user, seq2item;
  // CSR in visiting order: users by seq id, items in file order
  std::vector<int64_t> rowptr;
  std::vector<uint32_t> col;
  std::vector<uint8_t> val;           // last-duplicate-wins, uint8 wrap
  HeldOut validation, test;

  // ratings.cc:42-61 + 63-119.  0 / -1 (cannot open) / -2 (the reference's
  // "unexpected lines" exit(-1))
  int read_train(const std::string &path);
  int read_heldout(const std::string &path, HeldOut *out);
  // ratings.cc:273-292: seq ids (sorted, unique) of the listed users that
  // appear in the training set.  -1 if the file cannot be opened
  int read_test_users(const std::string &path, std::vector<uint32_t> *out) const;
  // ratings.cc:217-271
  int write_marginals(const std::string &byusers, const std::string &byitems,
                      uint32_t *longest_users, uint32_t *longest_items) const;

  // binary dataset image (extension): everything read_train + both
  // read_heldout calls produce.  dir = the -dir argument (source TSVs are
  // fingerprinted by size and mtime).  load: 0 = loaded, 1 = absent / stale /
  // other parameters / truncated (parse the TSVs instead).  save: 0 or -1.
  bool heldout_loaded = false;
  // `image`: where the image lives (default <dir>/hgaprec.cache.bin); the fingerprint is always
  // that of the TSV files in `dir`.  `-ngpus N` uses an image in the output directory to hand the
  // parsed data set from rank 0 to the other ranks.
  int save_cache(const std::string &dir, const std::string &image = "") const;
  int load_cache(const std::string &dir, const std::string &image = "");

 // open-addressing id -> seq maps (std::map in the reference; only lookups
  // and insertion order matter)
  struct IdMap {
    // one 8-byte slot per entry (key, seq + 1; 0 = empty): a probe touches one cache line
    std::vector<uint64_t> slots; uint32_t cnt = 0;
    IdMap() { rehash(1024); }
    void rehash(uint32_t cap);
    bool find(uint32_t key, uint32_t *val) const;
    void put(uint32_t key, uint32_t val);
  } user2seq, item2seq;

 private:
  int read_generic(FILE *f, HeldOut *out);
  int read_generic_parallel(FILE *f, HeldOut *out);   // 0 done, 1 not for this file (small, not well-formed, one thread)
  struct Consumer;
  uint32_t input_rating_class(uint32_t v) const;
  std::vector<uint32_t> tr_u_, tr_i_, tr_y_;   // training triples in file order
};

// ------------------------------------------------------------ MT19937 -----
// gsl_rng_mt19937: 2002 init_genrand seeding, seed 0 -> 4357,
// gsl_rng_uniform = u32 / 2^32
struct Mt19937 {
  uint32_t mt[624]; int mti;
  explicit Mt19937(unsigned long seed = 0) { set(seed); }
  void set(unsigned long seed);
  uint32_t next_u32();
  void refill();
  // out[j] = base + scale * uniform() for the next cnt draws / cnt draws taken and dropped: the stream
  // position afterwards is that of cnt calls of next_u32()
  void fill_affine(double *out, size_t cnt, double base, double scale);
  void skip(size_t cnt);
  double uniform() { return next_u32() / 4294967296.0; }
  unsigned long uniform_int(unsigned long n);
};

double digamma(double x);      // x > 0, |err| ~ 1e-15 (stands for gsl_sf_psi)

// ---------------------------------------------------------- GammaInit -----
// Host arrays of the start state, in the layout of hpf_set_state.
// resize() of these arrays does not write zeros first: every element is assigned by initialize_state, and at
// C2 the four user-side matrices are 3.5 GB that would otherwise be touched twice
template <typename T> struct NoInitAlloc : std::allocator<T> {
  template <typename U> struct rebind { using other = NoInitAlloc<U>; };
  template <typename U> void construct(U *p) noexcept { ::new ((void *)p) U; }
  template <typename U, typename... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
using StateArray = std::vector<double, NoInitAlloc<double>>;
struct GammaState {
  uint32_t n = 0, m = 0, k = 0; bool hier = false, bias = false;
  StateArray theta_shape, theta_rate, theta_E, theta_Elog;
  StateArray beta_shape, beta_rate, beta_E, beta_Elog;
  StateArray xi_shape, xi_rate, xi_E, xi_Elog;
  StateArray eta_shape, eta_rate, eta_E, eta_Elog;
  StateArray ubias_shape, ubias_rate, ubias_E, ubias_Elog;
  StateArray ibias_shape, ibias_rate, ibias_E, ibias_Elog;
};
// hgaprec.cc:153-204 with the RNG already seeded as hgaprec.cc:34-38.
// [user_lo, user_hi): keep only that range of the user-side arrays (a rank's
// shard); the whole MT19937 stream is still consumed in the reference's
// order, so every rank ends with the same generator state and the same
// item-side arrays.  out->n is the number of users kept.
void initialize_state(Mt19937 &rng, uint32_t n, uint32_t m, uint32_t k, bool hier,
                      bool bias, GammaState *out, uint32_t user_lo = 0,
                      uint32_t user_hi = 0xffffffffu);
// the seed rule of hgaprec.cc:34-38 (+ GSL_RNG_SEED like gsl_rng_env_setup)
Mt19937 make_rng(double env_seed);

// ------------------------------------------------------------ writers -----
// D2Array<double>::save / D1Array<double>::save: "seq\tid\tv...\n", %.8f;
// id = seq2id[row] when row < nids else the row index itself
// "%.8f" exactly as printf rounds it, ~10x faster; returns the length (no NUL)
size_t format_fixed8(double v, char *out);
// row0: sequence number of the first row (a rank writing its shard of a matrix)
// threads: how many host threads format the rows (0 = all of them)
// a model file is rewritten in place (save_matrix / save_vector, and the part files put together by rank 0): while that
// is going on "<path>.writing" lies beside it; a marker that outlives the run says the file is not whole
std::string rewrite_marker(const std::string &path);
void rewrite_begin(const std::string &path);
void rewrite_end(const std::string &path);
int save_matrix(const std::string &path, const double *a, uint32_t rows, uint32_t cols,
                const uint32_t *seq2id, uint32_t nids, uint32_t row0 = 0, unsigned threads = 0);
int save_vector(const std::string &path, const double *a, uint32_t rows,
                const uint32_t *seq2id, uint32_t nids, uint32_t row0 = 0);

// ------------------------------------------- held-out series / stopping ---
// HGAPRec::compute_likelihood bookkeeping (hgaprec.cc:1466-1500)
struct StopRule {
  double prev_h = 0.0; uint32_t nh = 0;
  // returns true when the run must stop; *why as written to max.txt
  bool update(uint32_t iter, double a, int *why);
};

// ------------------------------------------------------------- Comm --------
// Host-side collectives of a multi-process run (one process per GPU): a TCP
// star through rank 0 on MASTER_ADDR:MASTER_PORT.  Used for the bootstrap (the
// RCCL unique id), the few scalars per report step, and -- with `-comm host`
// -- a host-staged stand-in for the device all-reduce (tests on one GPU).
// Sums are formed on rank 0 in rank order, so every rank sees the same bits.
struct Comm {
  int rank = 0, world = 1;
  std::vector<int> fds;       // rank 0: one socket per peer (index = peer rank); others: fds[0]
  // nonce: shared secret of the job (0 when the launcher gave none); a peer that does not
  // present it is dropped.  0 / -1
  int init(int rank_, int world_, const std::string &addr, int port, uint64_t nonce = 0);
  void close_all();
  int allreduce_sum(double *v, size_t n);
  int allreduce_max(double *v, size_t n);
  int bcast(void *p, size_t bytes);          // from rank 0
  int barrier();
 private:
  int reduce_impl(double *v, size_t n, bool is_max);
};

// contiguous user ranges balanced on the nnz prefix sum
std::vector<std::pair<uint32_t, uint32_t>> partition_users(const std::vector<int64_t> &rowptr, int world);

}  // namespace hgaprec

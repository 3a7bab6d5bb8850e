/*
 * hpf.h -- C-ABI of libhpf_hip.so: the MI355X (gfx950) device side of the
 * hgaprec CAVI inner loop (hierarchical / Bayesian Poisson factorization).
 *
 * The reference (premgopalan/hgaprec) is one C++ executable with no plugin or
 * FFI seam; this ABI is the seam a maintainer would cut where the inference
 * driver (src/hgaprec.cc, class HGAPRec) meets the Gamma containers
 * (src/gpbase.hh) and the ratings store (src/ratings.hh).  Each entry point
 * names the reference code it replaces (file:line, relative to the reference's
 * src/).  INTEGRATION.md shows the reference-side patch.
 *
 * Conventions
 *  - plain C types only; the caller owns every host pointer and may free it
 *    as soon as the call returns; the handle owns all device memory.
 *  - every function returns HPF_OK (0) or a negative hpf_status and never
 *    throws or aborts; hpf_last_error(h) gives the text of the last failure.
 *  - one host thread per handle; distinct handles are independent.
 *  - all real-valued state is fp64, like the reference (env.hh / gpbase.hh use
 *    double throughout); ids are uint32, ratings uint8 (yval_t, env.hh:20).
 *  - multi-GPU: one handle per rank holding a contiguous range of users (and
 *    their nonzeros); item-side state is replicated.  The only exchange per
 *    iteration is a sum-all-reduce of hpf_exchange_buffer() between
 *    hpf_iterate_local() and hpf_iterate_global().
 */
#ifndef HPF_H
#define HPF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HPF_ABI_VERSION 8

typedef struct hpf_handle hpf_handle;

typedef enum {
  HPF_OK = 0,
  HPF_ERR_INVALID = -1,      /* bad argument / call order                    */
  HPF_ERR_NO_DEVICE = -2,    /* no usable gfx950 device / HIP runtime error   */
  HPF_ERR_OOM = -3,          /* device or host allocation failed              */
  HPF_ERR_HIP = -4,          /* a HIP call or kernel launch failed            */
  HPF_ERR_UNSUPPORTED = -5,  /* e.g. K + 2*bias > HPF_MAX_COLUMNS             */
  HPF_ERR_STATE = -6         /* state not initialised (no CSR / no E, Elog)   */
} hpf_status;

#define HPF_MAX_COLUMNS 1024

/* Gamma objects of the model (HGAPRec members, hgaprec.hh:94-112) x the four
 * per-object arrays of gpbase.hh (shape_curr, rate_curr, expected_v,
 * expected_logv).  Host layout is dense row-major doubles:
 *   THETA_* : n_users x K   (htheta, or theta without -hier)
 *   BETA_*  : n_items x K   (hbeta / beta)
 *   XI_*    : n_users       (thetarate; -hier only)
 *   ETA_*   : n_items       (betarate;  -hier only)
 *   UBIAS_* : n_users       (thetabias; -bias only)
 *   IBIAS_* : n_items       (betabias;  -bias only)
 * Without -hier, THETA_RATE / BETA_RATE are K-vectors (GPMatrixGR::_rcurr,
 * gpbase.hh:520). */
typedef enum {
  HPF_THETA_SHAPE = 0, HPF_THETA_RATE, HPF_THETA_E, HPF_THETA_ELOG,
  HPF_BETA_SHAPE, HPF_BETA_RATE, HPF_BETA_E, HPF_BETA_ELOG,
  HPF_XI_SHAPE, HPF_XI_RATE, HPF_XI_E, HPF_XI_ELOG,
  HPF_ETA_SHAPE, HPF_ETA_RATE, HPF_ETA_E, HPF_ETA_ELOG,
  HPF_UBIAS_SHAPE, HPF_UBIAS_RATE, HPF_UBIAS_E, HPF_UBIAS_ELOG,
  HPF_IBIAS_SHAPE, HPF_IBIAS_RATE, HPF_IBIAS_E, HPF_IBIAS_ELOG,
  HPF_NUM_STATE
} hpf_state;

typedef struct {
  uint32_t struct_size;    /* sizeof(hpf_config), for ABI growth             */
  uint32_t n_users;        /* users owned by THIS handle (its shard)         */
  uint32_t n_items;        /* all items (replicated)                         */
  uint32_t K;              /* factors (-k)                                   */
  uint32_t hier;           /* -hier : vb_hier()  else vb() / vb_bias()       */
  uint32_t bias;           /* -bias                                          */
  uint32_t binary;         /* -binary-data (held-out likelihood form)        */
  uint32_t n_users_total;  /* users over all ranks (item-bias rate 0.3 + n,  */
                           /* hgaprec.cc:1393); 0 => n_users                 */
  int32_t  device;         /* HIP device ordinal                             */
  uint32_t n_ranks;        /* 1 => hpf_iterate() needs no exchange           */
  uint32_t rank;
  uint32_t w_storage;      /* how W = exp(Elog - rowmax), the matrix the phi      */
                           /* passes gather, is stored (arithmetic and             */
                           /* accumulators are fp64 in every mode):                */
                           /* 0: fp64, exact (default; the only mode chosen        */
                           /*    implicitly).  The library may PACK the rows       */
                           /*    losslessly (59 bits per element: every W is a     */
                           /*    positive double in [2^-126, 2) or zero, so the    */
                           /*    sign and four exponent bits carry nothing) when   */
                           /*    that saves a 128-byte line per row -- K = 100:    */
                           /*    six lines instead of seven.  A W below 2^-126 of  */
                           /*    its row maximum (Elog spread > 88 inside a row;   */
                           /*    HPF states have 10-20) cannot be packed: the      */
                           /*    library then moves the rows to plain doubles by   */
                           /*    itself and goes on with the bits a handle made    */
                           /*    with 3 has (hpf_work_info.w_fallbacks).           */
                           /* 3: fp64 as plain doubles, never packed (where 0      */
                           /*    would pack: in the packed shape's pieces).        */
                           /* 1: EXPERIMENTAL fp32 -- drifts out of the 1e-4       */
                           /*    parity contract after ~30 iterations. */
                           /* 2: OPT-IN 48 bits: the top 48 bits of the fp64 value */
                           /*    (36 mantissa bits, rounded to nearest even); rows */
                           /*    are 25-30 % shorter, both phi passes faster by    */
                           /*    about that; measured drift vs the fp64 oracle     */
                           /*    ~1e-11 per early sweep, inside 1e-4 after         */
                           /*    hundreds (tests/w32_error_growth.py).             */
  void    *stream;         /* hipStream_t to run on, NULL => own stream      */
  double   s_prior;        /* 0.3 (hgaprec.cc:13-20 hard-codes both)         */
  double   r_prior;        /* 0.3                                            */
  uint32_t novb;           /* -novb (Env::vb == false).  Read only where the  */
                           /* reference reads it: vb_bias(), i.e. bias without */
                           /* hier (hgaprec.cc:1250,1276-1297) -- both rates   */
                           /* are then built from the PREVIOUS iteration's     */
                           /* expectations before anything is swapped (the     */
                           /* item rate takes the old sum_u E[theta]).  On     */
                           /* several ranks see hpf_start_sums.                */
  uint32_t tiling;         /* 0: the library decides per side whether the phi  */
                           /* pass is tiled (cache blocking of the gathered    */
                           /* rows, one tile per XCD L2) */
                           /* 1: never -- row-major work lists only.  Tiling   */
                           /* changes the ORDER in which a row's nonzeros are  */
                           /* summed (same for every run of one build), not    */
                           /* what is summed.                                  */
} hpf_config;

/* per-kernel device time, milliseconds, from hipEvents recorded on the
 * handle's stream around each launch group */
typedef struct {
  float phi_user_ms;     /* K1a: phi_pass_kernel<..,0>, user-major (theta sums) */
  float combine_user_ms; /*      combine of long user rows                      */
  float phi_item_ms;     /* K1b: phi_pass_kernel<..,1>, item-major (beta sums)  */
  float combine_item_ms; /*      combine of long item rows                      */
  float sweep_user_ms;   /* K2 + K4(xi) + K5(user bias) + K7                    */
  float sweep_item_ms;   /* K3 + K4(eta) + K5(item bias) + K7                   */
  float iteration_ms;    /* first launch -> last launch of the iteration        */
  uint32_t iterations;   /* iterations executed so far                          */
  float exchange_wait_ms;/* end of the user sweep -> start of the item sweep:   */
                         /* the part of the all-reduce the stream had to wait   */
                         /* for (n_ranks > 1; ~0 on one rank)                   */
} hpf_timing;

int  hpf_abi_version(void);
const char *hpf_strerror(int status);
const char *hpf_last_error(const hpf_handle *h);

/* replaces: HGAPRec::HGAPRec allocation of the 8 Gamma objects
 * (hgaprec.cc:8-33) */
int  hpf_create(const hpf_config *cfg, hpf_handle **out);
void hpf_destroy(hpf_handle *h);

/* replaces: the per-user adjacency + rating maps built by
 * Ratings::read_generic (ratings.cc:63-119) and walked by
 * Ratings::get_movies / Ratings::r (ratings.hh:175-181,153-165).
 * CSR over this handle's users in the reference's visiting order; col = item
 * seq id; val = rating as stored by the reference (uint8, already wrapped,
 * last duplicate wins), NULL => every rating is 1 (-binary-data).
 * Host pointers (staged through pinned buffers).  The item-major (CSC) view is
 * built in HBM by a stable radix sort on the item id: inside an item the users
 * stay ascending -- the order in which the reference's serial loop
 * (hgaprec.cc:1340-1345) reaches them. */
int  hpf_upload_csr(hpf_handle *h, const int64_t *rowptr, const uint32_t *col,
                    const uint8_t *val);
/* the same for a CSR that already lives in HBM (DEVICE pointers on the handle's
 * device; the library copies what it keeps, the caller may free on return).
 * The producer's work must have completed before the call. */
int  hpf_upload_csr_device(hpf_handle *h, const int64_t *d_rowptr, const uint32_t *d_col,
                           const uint8_t *d_val);
/* replaces: the per-item user lists Ratings keeps beside the per-user ones
 * (Ratings::get_users, ratings.hh:167-173): host copies of the item-major view
 * (colptr[n_items + 1], users[nnz] ascending inside an item, vals[nnz]); any
 * pointer may be NULL. */
int  hpf_get_csc(hpf_handle *h, int64_t *colptr, uint32_t *users, uint8_t *vals);

/* replaces: the result of HGAPRec::initialize (hgaprec.cc:153-204) -- the host
 * draws the MT19937 stream and hands over E / Elog (and shapes, for export).
 * count must equal the element count of that array. */
int  hpf_set_state(hpf_handle *h, hpf_state which, const double *host, size_t count);
/* replaces: reads of shape_curr()/rate_curr()/expected_v()/expected_logv()
 * by save_model (hgaprec.cc:2137-2158) */
int  hpf_get_state(hpf_handle *h, hpf_state which, double *host, size_t count);
/* the same with DEVICE pointers (dense row-major doubles in HBM, same element
 * counts): no PCIe round trip for callers whose state is already resident */
int  hpf_set_state_device(hpf_handle *h, hpf_state which, const double *dev, size_t count);
int  hpf_get_state_device(hpf_handle *h, hpf_state which, double *dev, size_t count);

/* Snapshot (extension; the reference has no training resume: `-load` is parsed
 * and never consulted, main.cc:137-140): the loop's whole device state as one
 * opaque host blob -- the gathered matrices W, raw sums, expectations, xi/eta
 * vectors, column sums and the bookkeeping flags, verbatim.  A handle of the same
 * shape that loads it continues with IDENTICAL bits (hpf_set_state(ELOG) would
 * re-derive W and round differently).  Call between iterations. */
int  hpf_snapshot_size(hpf_handle *h, size_t *bytes);
int  hpf_snapshot_save(hpf_handle *h, void *host, size_t bytes);
int  hpf_snapshot_load(hpf_handle *h, const void *host, size_t bytes);

/* replaces: n_iters passes of steps A-F of HGAPRec::vb_hier
 * (hgaprec.cc:1340-1414), or of vb (927-956) / vb_bias (1226-1272) without
 * -hier.  With n_ranks > 1 it needs hpf_comm_init and runs the overlapped
 * exchange (see below) itself; every rank must make the same call.
 * Asynchronous on the handle's stream.
 * If a sweep meets an element the packed rows of W cannot hold (hpf_config.w_storage),
 * the passes launched after it return at once; the next synchronising call moves the rows
 * to plain doubles and runs what was skipped (hpf_work_info.w_fallbacks) -- results are
 * those of a handle that never packed, only the timing of that call is not representative.
 * Launch-bound problems (nnz <= 4 Mi, or HPF_GRAPH=1; HPF_GRAPH=0 disables)
 * replay one captured iteration as a hipGraph: same kernels in the same
 * order, identical bits; such iterations report only iteration_ms in
 * hpf_timing (the per-kernel fields read 0).  With n_ranks > 1 (or a communicator)
 * the iteration is cut by its collectives; HPF_GRAPH=1 (experimental, opt-in: measured, it does
 * not pay) replays it as THREE graphs -- item pass | user pass + user sweep | item sweep --
 * (v8; hpf_work_info.graph_replay = 2). */
int  hpf_iterate(hpf_handle *h, int n_iters);

/* n_ranks > 1: step A for the local users, the local user sweep (B, D-user,
 * E) and the local partial sums ... */
int  hpf_iterate_local(hpf_handle *h);
/* hpf_iterate_local in pieces, for callers that overlap communication.  Step A
 * is two independent passes over the same W: the item-major one runs first.
 *   _items : item phi pass -- after it the first n_items*ld doubles of the
 *            exchange buffer (the item shape sums) are final and their
 *            all-reduce may start;
 *   _users : user phi pass + user sweep (B, D-user, E); only writes the last
 *            ld doubles (sum_u E[theta_u,:]), reduced in a second, tiny
 *            all-reduce.  (_phi = _items + the user phi pass, _sweep = the
 *            user sweep alone: the same work cut after step A instead.)
 * Order per iteration: _items, _users (or _phi, _sweep), hpf_iterate_global;
 * anything else returns HPF_ERR_STATE. */
int  hpf_iterate_local_items(hpf_handle *h);
int  hpf_iterate_local_users(hpf_handle *h);
int  hpf_iterate_local_phi(hpf_handle *h);
int  hpf_iterate_local_sweep(hpf_handle *h);
/* ... the caller sum-all-reduces this device buffer of `count` doubles in
 * place (RCCL ncclAllReduce(ncclDouble, ncclSum) / torch.distributed) on the
 * handle's stream ... */
int  hpf_exchange_buffer(hpf_handle *h, void **device_ptr, size_t *count);
/* optional: make the handle use caller-owned device memory (>= count doubles,
 * 16-byte aligned) as the exchange buffer; call before hpf_upload_csr.  The
 * buffer is cleared on the handle's stream: no work queued on another stream
 * may still be using that memory (a caching allocator can hand out a block
 * that earlier kernels of ITS stream are not done with). */
int  hpf_bind_exchange_buffer(hpf_handle *h, void *device_ptr, size_t count);
/* ... then the replicated item sweep (C, D-item, F) on every rank. */
int  hpf_iterate_global(hpf_handle *h);
/* -novb (hpf_config.novb with bias, without hier) on SEVERAL ranks: the first item rate is built
 * from sum_u E[theta_u,:] of the START state (hgaprec.cc:1281-1282 reads _theta.sum_rows() before
 * _theta.swap()), which then has to be summed over the ranks once, before the first iteration.
 * hpf_start_sums derives what the first iteration needs from the state handed in and leaves this
 * rank's part of that sum in the tail of the exchange buffer (its last `ld` doubles,
 * hpf_work_info.ld).  With hpf_comm_init done it also all-reduces the tail itself (and hpf_iterate
 * calls it on its own); otherwise the caller sum-all-reduces those ld doubles in place, like the
 * per-iteration exchange -- but ONLY when hpf_work_info.start_sums_pending reads 1: after
 * hpf_snapshot_load the tail is the reduced sum already (v7).  From v8 on the field may be read before OR after the call:
 * it keeps reading 1 while the tail holds this rank's part only, until the first pass of the next iteration.  Call it after the last hpf_set_state /
 * hpf_snapshot_load and before the first hpf_iterate_local_*: iterating without it returns
 * HPF_ERR_STATE.  A no-op elsewhere. */
int  hpf_start_sums(hpf_handle *h);

/* The library can run that all-reduce itself: RCCL is dlopen'ed on first use
 * (no link-time dependency).  One rank calls hpf_comm_unique_id and ships the
 * HPF_COMM_ID_BYTES bytes to the others by any means; every rank then calls
 * hpf_comm_init (collective) once, and hpf_allreduce_exchange between
 * iterate_local and iterate_global.  = ncclGetUniqueId / ncclCommInitRank /
 * ncclAllReduce(ncclDouble, ncclSum) in place on the handle's stream.
 * Overlapped form: hpf_iterate_local_items, hpf_allreduce_items_begin (the
 * item part goes out on a second, library-owned stream), hpf_iterate_local_users
 * (runs meanwhile), hpf_allreduce_exchange (adds the tail and makes the
 * handle's stream wait for both), hpf_iterate_global. */
#define HPF_COMM_ID_BYTES 128
int  hpf_comm_unique_id(void *id_out);
int  hpf_comm_init(hpf_handle *h, const void *id);
int  hpf_allreduce_items_begin(hpf_handle *h);
int  hpf_allreduce_exchange(hpf_handle *h);
/* host copies of the exchange buffer (host-staged reduction, tests) */
int  hpf_exchange_read(hpf_handle *h, double *host, size_t count);
int  hpf_exchange_write(hpf_handle *h, const double *host, size_t count);

/* replaces: the per-pair loop of HGAPRec::compute_likelihood
 * (hgaprec.cc:1455-1465) with rating_likelihood_hier / rating_likelihood
 * (1538-1560 / 1503-1536).  u is a LOCAL user index, y the int stored in the
 * CountMap (wrapped to uint8 like `yval_t r = i->second`).  Pairs are summed
 * in the order given (the caller passes them sorted by (user, item) like the
 * std::map).  Synchronous. */
int  hpf_heldout_ll(hpf_handle *h, const uint32_t *u, const uint32_t *i,
                    const int32_t *y, size_t cnt, double *sum_out,
                    uint64_t *cnt_out);
/* ABI v8.  A report step evaluates the same validation and test pairs every time (hgaprec.cc:1439-1470 walks the same
 * two maps): hpf_heldout_bind validates and uploads a set ONCE into one of HPF_HELDOUT_SLOTS slots (binding again
 * replaces it; cnt = 0 binds an empty set); hpf_heldout_ll_bound is then the kernel, one DMA of the per-pair values
 * into a page-locked buffer the handle keeps, and the same serial sum in the order the pairs were bound -- the value
 * hpf_heldout_ll returns for the same pairs, bit for bit.  The caller's arrays may be freed after the bind. */
#define HPF_HELDOUT_SLOTS 4
int  hpf_heldout_bind(hpf_handle *h, int slot, const uint32_t *u, const uint32_t *i, const int32_t *y, size_t cnt);
int  hpf_heldout_ll_bound(hpf_handle *h, int slot, double *sum_out, uint64_t *cnt_out);

/* replaces: HGAPRec::logl (hgaprec.cc:2160-2255), the bound written to
 * logl.txt with -logl: the per-nonzero term over this handle's nonzeros plus
 * the Gamma terms (gpbase.hh:360-387,717-741,951-969) of the user-side
 * objects and -- on rank 0 only -- of the replicated item-side objects, so
 * that the sum over ranks is the reference's value.  Needs >= 1 iteration.
 * Synchronous. */
int  hpf_elbo(hpf_handle *h, double *out);

/* ---- ranking evaluation (report steps) ------------------ */
/* replaces: prediction_score_hier / prediction_score (hgaprec.cc:1966-1991,
 * 1850-1877; _use_rate_as_score) for every item: out[n_sel x n_items] =
 * E_theta[users] . E_beta^T (+ biases), on the fp64 matrix cores.  Host out. */
int  hpf_scores(hpf_handle *h, const uint32_t *users, uint32_t n_sel, double *out);
/* replaces: the scoring loop, qsort and top-N walk input of
 * HGAPRec::compute_precision (hgaprec.cc:1722-1765): per selected (local)
 * user the topn best items in the reference's order -- score descending, ties
 * by ascending item (glibc's stable qsort) -- after zeroing the user's
 * training items with a stored rating > 0 and the items of the caller's mask
 * list (CSR over the selected users; the validation items).  topn <= 1024;
 * entries beyond n_items are (0xffffffff, 0). */
int  hpf_rank_topn(hpf_handle *h, const uint32_t *users, uint32_t n_sel,
                   const uint64_t *mask_ptr, const uint32_t *mask_items, uint32_t topn,
                   uint32_t *out_items, double *out_scores);
/* replaces: the position j of a test item in the fully sorted list of
 * HGAPRec::compute_itemrank (hgaprec.cc:1628-1680): query q asks for item
 * q_item[q] in the list of selected user q_sel[q] (an index into users). */
int  hpf_item_ranks(hpf_handle *h, const uint32_t *users, uint32_t n_sel,
                    const uint64_t *mask_ptr, const uint32_t *mask_items,
                    const uint32_t *q_sel, const uint32_t *q_item, uint32_t nq,
                    uint32_t *out_rank, double *out_score);

/* how the uploaded matrix was cut into work (diagnostics, tests, bench):
 * a "segment" is <= 512 consecutive nonzeros of one row; rows longer than that
 * are "long" (their segment sums are combined by a second kernel) and rows with
 * more than 256 segments "huge" (combined in two levels).  On a TILED side
 * (tiles_* > 0) a segment is a run of one row inside one tile of the gathered
 * matrix, and the long rows also count the rows without any nonzero (the
 * combine zeroes them). */
typedef struct {
  uint64_t nnz;
  uint32_t user_segments, user_long_rows, user_huge_rows;
  uint32_t item_segments, item_long_rows, item_huge_rows;
  uint32_t phi_G, phi_R, phi_V;      /* lanes per nonzero, loads per lane, doubles per load */
  uint32_t sweep_G, sweep_R;         /* row sweep: lanes per row, columns per lane */
  uint32_t ld;                       /* row stride of the device matrices, doubles */
  uint32_t graph_replay;             /* 1: hpf_iterate replays a captured hipGraph; 2 (v8, opt-in): a rank of several replays the */
                                     /* iteration as three graphs cut by its collectives (hpf_iterate_local_items,        */
                                     /* hpf_iterate_local_users, hpf_iterate_global; hpf_timing then holds the item half   */
                                     /* under phi_item_ms, the user half under phi_user_ms, combine_* / sweep_user_ms 0)   */
  uint32_t w_layout;                 /* rows of W: 0 plain (phi_V elements per load), 3 packed 59-bit (lossless),  */
                                     /* 2 packed 48-bit (w_storage = 2), 4 plain doubles in the packed shape's     */
                                     /* 16-byte pieces (w_storage = 3, or after a fallback); 2-4: phi_R = pieces per lane */
  uint32_t tiles_user, tiles_item;   /* tiled phi pass: tiles of the gathered matrix (0: the side is row-major)   */
  /* ABI v5 */
  uint32_t tile_rows_user, tile_rows_item;   /* gathered rows per tile of that side's pass (0: row-major)          */
  uint64_t heavy_min_nnz_user, heavy_min_nnz_item; /* an owner row with at least this many nonzeros is regrouped  */
                                     /* tile by tile ("heavy"); the others stay row-major in the same launch       */
  uint32_t w_fallbacks;              /* how often the rows of W moved from the packed form to plain doubles because a    */
                                     /* state turned up that p59 cannot hold (w_layout then reads 4); 0 or 1            */
  uint32_t notes;                    /* bit 0 / 1: the user / item side was left row-major because the device is too     */
                                     /* small for the tiling's temporaries; bit 2 / 3: because their allocation failed   */
  /* ABI v7 */
  uint32_t start_sums_pending;       /* 1: -novb on several ranks and the tail of the exchange buffer does not hold the   */
                                     /* all-reduced sum_u E[theta] of the start state yet -- hpf_start_sums will leave    */
                                     /* this rank's PART there and a caller that owns the exchange has to sum-all-reduce  */
                                     /* it.  0 after hpf_snapshot_load of a state saved between iterations: the tail came */
                                     /* with the snapshot, already reduced; reducing it again would multiply it by the    */
                                     /* number of ranks.  v8: stays 1 after hpf_start_sums (no hpf_comm_init) until the   */
                                     /* next iteration begins -- read it before or after that call.                       */
  /* ABI v8 */
  uint32_t tile_chunk_user;          /* segments per workgroup of that side's tiled pass (0: row-major): two per wave     */
  uint32_t tile_chunk_item;          /* unless the list is too long for one launch of such chunks                         */
  uint32_t reserved0;
} hpf_work_info;
int  hpf_get_work_info(hpf_handle *h, hpf_work_info *out);

/* measurement: one phi pass with the arithmetic taken out -- same work list, index stream and
 * rows, the gathered bytes only folded into a register, nothing written.  The mean time of
 * `reps` launches (side 0: user-major pass, 1: item-major pass) is what the memory system needs
 * for that access pattern: the ceiling bench.py prints beside the pass itself.  Does not touch
 * the model state. */
int  hpf_gather_only(hpf_handle *h, int side, int reps, float *ms_out);

/* TEST HOOK (tests/test_gpu_fullsize.py): overwrite ONE entry of the index stream a phi pass
 * walks (side 0: user-major, 1: item-major; `pos` indexes the pass's own array -- the tiled copy
 * when the side is tiled) with `value` (< rows of the gathered side).  Returns the old value and
 * the owner row whose sum the entry feeds.  A pass over the damaged list still conserves mass
 * (every phi sums to its rating whichever row was gathered); the sampled-row checks must not
 * pass.  Never called by the product path. */
int  hpf_debug_poke_index(hpf_handle *h, int side, uint64_t pos, uint32_t value,
                          uint32_t *old_value, uint32_t *owner_row);

/* Page-locked host memory (ABI v6).  hpf_get_state / hpf_set_state / hpf_upload_csr / the snapshot calls
 * accept any host pointer; ordinary (pageable) memory goes through the library's two pinned staging
 * buffers and a threaded host copy -- on the MI355X boxes measured 44 GB/s out of the device into memory
 * that has been written before, 11-14 GB/s into freshly allocated pages (their first touch is most of
 * the time), 30-44 GB/s into the device -- while a buffer from hpf_host_alloc is the target of the DMA
 * itself (54 GB/s either way; tools/d2h_probe.hip, bench.py --host-handover).  Worth it for buffers that
 * are used again and again (page-locking costs ~0.15 s per GB, and as much again to undo): the CLI's
 * copies of the factor matrices, fetched at every report step (hgaprec.cc:1422 save_model).  No handle
 * is needed; HPF_ERR_OOM when the memory cannot be had (the caller falls back to malloc),
 * HPF_ERR_NO_DEVICE without a HIP device. */
int  hpf_host_alloc(void **ptr, size_t bytes);
int  hpf_host_free(void *ptr);

int  hpf_synchronize(hpf_handle *h);
int  hpf_last_timing(hpf_handle *h, hpf_timing *out);
/* mean over the last n_last iterations (at most 64 are kept); synchronises */
int  hpf_mean_timing(hpf_handle *h, uint32_t n_last, hpf_timing *out);
/* iteration_ms of each of the last n_last iterations, oldest first (hipEvents on the
 * handle's stream: first launch -> last launch of that iteration); *n_out = how many
 * were written (<= n_last, <= 64).  For a median instead of a mean. */
int  hpf_iteration_times(hpf_handle *h, uint32_t n_last, float *ms_out, uint32_t *n_out);

/* algorithmic bytes (SURVEY.md section 8d / DESIGN.md) moved by one launch of
 * the two phi passes and of the row sweeps for the uploaded matrix */
int  hpf_algorithmic_bytes(hpf_handle *h, uint64_t *phi_user, uint64_t *phi_item,
                           uint64_t *rows);

#ifdef __cplusplus
}
#endif
#endif /* HPF_H */

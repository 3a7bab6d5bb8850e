// hpf_capi.hip -- implementation of include/hpf.h (libhpf_hip.so).
// Host orchestration of the gfx950 kernels in hpf_kernels.hpp: device state,
// CSR -> (CSR, CSC, segment lists), the per-iteration launch sequence, the
// exchange buffer for the multi-GPU all-reduce, hipEvent timing.
//
// Device layout (all fp64; row stride ld = G*R*V of the phi kernel shape >= K + 2*bias,
// the columns beyond the live ones hold zeros):
//   user side  theta: S,E,L,W [n x ld]   column K   = user bias (thetabias)
//                                         column K+1 = 0 ("junk": Elog 0)
//   item side  beta : S,E,L,W [m x ld]   column K   = junk, column K+1 = item bias
//   xi/eta     prior_E[rows] (+ prior_used, prior_rate for export)
//   exchange   [m x ld | ld]: item S rows followed by sum_u E[theta_u,:]
#include "../../include/hpf.h"
#include "hpf_kernels.hpp"
#include "hpf_build.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace hpf;

namespace {

struct Side {
  uint32_t rows = 0;
  double *S = nullptr, *E = nullptr, *L = nullptr;
  void *W = nullptr;                   // [rows x ld] double, or float in the f32-storage mode; rows of pieces: [rows x pk.row_bytes]
  bool w_from_sweep = false;           // W was written by a sweep (a W-only repeat of it restores it), not derived from Elog
  double *prior_E = nullptr, *prior_used = nullptr, *prior_rate = nullptr;
  double *prior_elog = nullptr, *prior_elog_used = nullptr;   // Elog xi/eta now / as used by the last rate
  double *colsum = nullptr;       // [ld] sum over this side's rows of E
  double *colsum_used = nullptr;  // [ld] other side's colsum used in the last rate
  double *colsum_part = nullptr;  // [sweep_blocks x ld]
  double *rate_set = nullptr;     // rate handed in by hpf_set_state (export before iter 0)
  size_t  rate_set_count = 0;
  double *prior_shape_set = nullptr, *prior_elog_set = nullptr;  // xi/eta extras
  // phi pass work list: segments of <= seg_max nonzeros of one row; rows of several
  // segments ("long") have their segment sums combined by a second kernel
  Seg *segs = nullptr; uint32_t nseg = 0;
  LongRow *longrows = nullptr; uint32_t nlong = 0;
  uint32_t nlong_wave = 0;             // the first nlong_wave of them: a wave each; the rest (a tiled side's rows with more than 64 partials) a workgroup each
  double *partial = nullptr; uint32_t npartial = 0;
  // rows with more than HUGE_SLOTS segments are combined in two levels so that
  // no wave walks a chain of thousands of partials: groups of GROUP_SLOTS
  // consecutive partials are summed into partial2 (grouprows), then the group
  // sums into S (hugerows, reading partial2)
  LongRow *grouprows = nullptr; uint32_t ngroup = 0;
  LongRow *hugerows = nullptr; uint32_t nhuge = 0;
  double *partial2 = nullptr; uint32_t npartial2 = 0;
  uint32_t *idx = nullptr; uint8_t *val = nullptr;
  // tiled pass (hpf_build.hpp): the nonzeros regrouped by (tile of the gathered row, owner row); segs then
  // index p_idx / p_val, and workgroup b takes the segments chunks[b]
  uint32_t *p_idx = nullptr; uint8_t *p_val = nullptr;
  uint2 *chunks = nullptr; uint32_t nchunk_blocks = 0;
  uint32_t tiles = 0, tile_rows = 0, chunk_segs = 0; uint64_t tiled_nnz = 0, light_below = 0;
  const uint32_t *pass_idx() const { return p_idx ? p_idx : idx; }
  const uint8_t *pass_val() const { return p_idx ? p_val : val; }
  int32_t bias_col = -1, junk_col = -1;
  double bias_rate_add = 0.0;
  uint32_t sweep_blocks = 0;
  bool have_E = false, have_L = false, have_prior = false;
  bool w_dirty = false;      // L was handed in by hpf_set_state: W must be derived from it
  bool l_stale = false;      // a sweep ran since L was last valid: rebuild L on export
  bool es_stale = false;     // S holds raw phi sums and E predates the last sweep
};

}  // namespace

struct hpf_handle {
  hpf_config cfg;
  uint32_t K = 0, C = 0, ld = 0;
  bool w32 = false;                     // W stored as float (hpf_config.w_storage = 1)
  int wl = WL_PLAIN;                    // layout of W rows: plain, WL_P59 (lossless packing, default where it shortens
                                        // the row), WL_F48 (w_storage = 2) or WL_F64 (w_storage = 3, or after a fall-back); rows in pieces: phiR = 16-byte pieces per lane
  PackedRow pk = {0, 0, 0, 0, 0};
  PackedRow pks = {0, 0, 0, 0, 0};      // plain-fp64 rows in pieces for the same columns (codec_f64): what the rows become when p59
                                        // cannot hold a state (recover_flush)
  int sw_mode = SW_PLAIN;               // how the sweep writes W (row_sweep_kernel MODE)
  uint32_t fallbacks = 0;               // automatic moves from p59 rows to plain fp64 rows so far
  uint32_t notes = 0;                   // hpf_work_info.notes
  uint64_t iters_counted = 0;           // iterations launched since the device counter flags[1] was last reset
  bool in_recovery = false;
  uint32_t *flags = nullptr;            // device words: [0] bit 0 = a softmax denominator underflowed, bit 1 = p59 flushed an entry,
                                        // bit 2 = passes and sweeps are skipping; [1] = iterations begun (PhiArgs::flags)
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Side u, it;
  double *exch = nullptr; size_t exch_count = 0; bool exch_external = false;
  // -novb with -bias and without -hier (vb_bias()'s else-branch, hgaprec.cc:1276-1297): the
  // item rate is built from the sum_u E[theta] of BEFORE this iteration's user sweep
  bool jacobi = false;
  double *u_colsum_prev = nullptr;      // [ld]
  bool start_sums_done = false;         // jacobi on several ranks: the start state's sum_u E[theta] has been handed to the exchange
  bool tail_partial = false;            // hpf_start_sums left THIS RANK'S PART of that sum in the tail for a caller that owns the exchange;
                                        // hpf_work_info.start_sums_pending keeps reading 1 until the first pass of the next iteration, so that a
                                        // caller may look at it before or after hpf_start_sums
  double *logfact = nullptr;
  int64_t *rowptr_dev = nullptr;   // user CSR row pointers (ranking mask, CSC build)
  int64_t *colptr_dev = nullptr;   // item-major (CSC) column pointers, built on device
  uint64_t nnz = 0;
  // pinned double buffer for host <-> device hand-over (hpf_upload_csr, hpf_set_state,
  // hpf_get_state): pageable hipMemcpy runs at ~4 GB/s, the staged pipeline at the
  // host memcpy rate.  HPF_H2D=plain|staged|register picks the carrier.
  void *stage[2] = {nullptr, nullptr};
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  int xfer_mode = 1;               // 0 plain, 1 staged, 2 hipHostRegister
  unsigned xfer_threads = 4;
  bool have_csr = false, derived_dirty = true;
  bool sums_dirty = true;               // a state array was handed in since the start state's column sums were taken: prepare_derived
                                        // takes them again.  derived_dirty without it = only W has to be written again (recover_flush):
                                        // the sums in place -- on several ranks the all-reduced ones -- stay
  uint32_t iterations = 0;
  int phiG = 0, phiR = 0, phiV = 0, swG = 0, swR = 0;
  uint32_t sweep_blocks_max = 2048;     // 8 waves per SIMD (HPF_SWEEP_BLOCKS); 1024 -> 2048: C2 user sweep 0.587 -> 0.544 ms
  uint32_t seg_max = 512;
  uint32_t huge_slots = 256, group_slots = 64;  // two-level combine above huge_slots segments (HPF_HUGE_SLOTS)
  // tiled pass: 2 auto, 0 never, 1 forced with every row regrouped (HPF_TILE); bytes of gathered rows per tile
  // (HPF_TILE_BYTES), segments per workgroup (HPF_TILE_CHUNK), the mean run a heavy row must reach in a
  // tile (HPF_TILE_RUN), the share of the nonzeros the heavy rows must hold (HPF_TILE_SHARE, per cent)
  // order of an XCD's queue (HPF_TILE_ORDER): 1 = the row-major rest in front on the even XCDs and behind on the
  // odd ones, so that half the chip pulls over the fabric while the other half runs from its L2 (C2 item pass
  // 4.39 ms; 0 = in front everywhere 4.75; 2 = dealt between the tiles 4.58)
  int tile_order = 1;
  uint32_t tile_split_below = 8;       // with fewer tiles than this EVERY tile is cut eight ways, a piece per XCD queue (round 4: below 32 tiles; C4's seven tiles of
                                       // items are 2 % faster that way than levelled); from here on whole tiles are dealt eight at a time and the remainder levels
                                       // the queues (build_tiled_side)
  int tile_sides = 3;                   // HPF_TILE_SIDES: bit 0 the user pass, bit 1 the item pass (experiments)
  int tile_mode = 2; uint64_t tile_bytes = 4u << 20; uint32_t tile_chunk = 0 /* 0: two segments per wave of the workgroup */, tile_min_run = 0; double tile_min_share = 0.15;
  uint32_t phi_blocks = 65536;      // ~one wave per few segments; the dispatcher balances
  // Threads per workgroup of a packed phi pass.  0 (default): 256 -- four waves, grid-stride -- for a ROW-MAJOR side, 64 -- one
  // wave, a chunk of two segments -- for a TILED side (round 5).  A workgroup's wave slots come back one SIMD at a time but a
  // new workgroup needs one on each of the four SIMDs at once: with the uneven runs of a tiled list a third of the slots stood
  // empty (SQ_WAVE_CYCLES: ~2 of 3 waves per SIMD resident on average) -- which a pass that lives on its L2 hits and on issue
  // pays for (C4 18.65 -> 17.45 ms, a C5 shard 44.8 -> 41.5; experiments.md) and a pass bound by the fabric does not (C2's
  // user pass, a C3 shard: unchanged).  Same segments, same order inside each: the same bits.  HPF_PHI_WG forces 64 | 128 | 256.
  uint32_t phi_wg = 0;
  uint32_t wg_of(bool tiled) const { return wl == WL_PLAIN ? 256u : phi_wg ? phi_wg : (tiled ? 64u : 256u); }   // (plain rows: phi_pass_kernel, always 256)
  uint32_t nz_per_batch() const { return phiG > 0 ? 64u / (uint32_t)phiG : 8u; }
  static constexpr uint32_t RING = 64;          // timed iterations kept
  hipEvent_t evr[RING][8] = {};
  hipEvent_t *ev = evr[0];                      // events of the iteration in flight
  uint32_t ev_count = 0;                        // iterations recorded so far
  void *comm = nullptr;                 // ncclComm_t once hpf_comm_init succeeded
  // the item part of the exchange buffer is reduced on its own stream so that
  // the user-side half of the iteration overlaps it (hpf_allreduce_items_begin)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_reduced = nullptr;
  bool items_reduce_pending = false;
  // hipGraph replay of one whole iteration for launch-bound problems (hpf_iterate,
  // n_ranks == 1).  HPF_GRAPH=0/1 forces it off/on; default: on when the phi passes
  // are short enough that the ~10 launches per iteration dominate (nnz <= graph_nnz_max).
  int graph_mode = -1;                  // -1 auto, 0 off, 1 on
  uint64_t graph_nnz_max = 4u << 20;
  hipGraphExec_t graph_exec = nullptr;
  // the same for a rank of several (round 6): the iteration is cut by its two collectives, so it is THREE graphs -- the item
  // pass | the user pass + the user sweep | the item sweep -- replayed by hpf_iterate_local_items, hpf_iterate_local_users and
  // hpf_iterate_global (and so by hpf_iterate with a communicator) around whatever exchange the caller or the library runs
  hipGraphExec_t split_exec[3] = {nullptr, nullptr, nullptr};
  bool capturing = false;               // inside stream capture: no events, no counters
  int phase = 0;                        // 0 idle | 1 items pass done | 2 users pass done | 3 user sweep done; as graph replays: 4 item piece done | 5 user piece done
  bool ring_graphed[RING] = {};         // slot was a graph replay: only events 0 and 6 exist
  // held-out sets bound once (hpf_heldout_bind): indices and ratings on the device, the per-pair values in a kept
  // page-locked buffer the DMA writes directly
  struct HeldSet { uint32_t *du = nullptr, *di = nullptr; int32_t *dy = nullptr; double *dout = nullptr, *hout = nullptr; size_t cnt = 0; bool bound = false; };
  HeldSet held[HPF_HELDOUT_SLOTS];
  hipEvent_t held_ev[8] = {};
  std::string err;
};

// ---- RCCL, loaded at run time (one process per GPU; the library must not
// depend on librccl at link time: single-GPU users never load it) -------------
namespace {
struct IdByValue { char internal[HPF_COMM_ID_BYTES]; };   // layout of ncclUniqueId (rccl.h:43)
struct RcclApi {
  void *lib = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, IdByValue /* ncclUniqueId by value */, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, void *) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  int (*CommGetAsyncError)(void *, int *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

const char *load_rccl()
{
  if (g_rccl.lib) return nullptr;
  // candidates, in order: HPF_RCCL_LIB (a full path: the operator's word wins), the soname as the loader finds it,
  // $TORCH_LIB_DIR/librccl.so (a box whose only copy ships inside a torch wheel: the C++ CLI has no torch to have
  // mapped it already), $ROCM_PATH/lib, /opt/rocm/lib
  std::vector<std::string> names;
  if (const char *e = getenv("HPF_RCCL_LIB")) if (*e) names.push_back(e);
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  if (const char *e = getenv("TORCH_LIB_DIR")) if (*e) { names.push_back(std::string(e) + "/librccl.so.1"); names.push_back(std::string(e) + "/librccl.so"); }
  if (const char *e = getenv("ROCM_PATH")) if (*e) { names.push_back(std::string(e) + "/lib/librccl.so.1"); names.push_back(std::string(e) + "/lib/librccl.so"); }
  names.push_back("/opt/rocm/lib/librccl.so.1");
  names.push_back("/opt/rocm/lib/librccl.so");
  void *lib = nullptr;
  // a copy this process has already mapped (a host application's, torch's) is THE copy:
  // two RCCL builds in one process do not survive each other's teardown
  for (const std::string &n : names) if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) break;
  if (!lib) for (const std::string &n : names) if ((lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL))) break;
  if (!lib) {
    static std::string why;
    const char *de = dlerror();
    why = std::string("cannot dlopen librccl.so (set HPF_RCCL_LIB to its full path, or TORCH_LIB_DIR to a torch/lib that holds it)") +
          (de ? std::string(": ") + de : std::string());
    return why.c_str();
  }
  RcclApi a; a.lib = lib;
  a.GetUniqueId = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (int (*)(void **, int, IdByValue, int))dlsym(lib, "ncclCommInitRank");
  a.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, void *))dlsym(lib, "ncclAllReduce");
  a.CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
  a.GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
  a.CommGetAsyncError = (int (*)(void *, int *))dlsym(lib, "ncclCommGetAsyncError");   // optional
  if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy || !a.GetErrorString)
    return "librccl.so lacks an expected symbol";
  g_rccl = a;
  return nullptr;
}
}  // namespace

namespace {

// an asynchronous RCCL failure (a peer died, a link error) surfaces here, once
// per iteration, instead of as a hang in a later collective
int check_comm_async(hpf_handle *h);

// inside a do { ... } while (0) body that ends in the function's clean-up: sets rc and leaves the body
#define HIPBRK(h, expr)                                                        \
  {                                                                            \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);            \
      rc = (e_ == hipErrorOutOfMemory) ? HPF_ERR_OOM : HPF_ERR_HIP;            \
      break;                                                                   \
    }                                                                          \
  }

#define HIPCHK(h, expr)                                                        \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);            \
      return (e_ == hipErrorOutOfMemory) ? HPF_ERR_OOM : HPF_ERR_HIP;          \
    }                                                                          \
  } while (0)

template <typename T>
int dalloc(hpf_handle *h, T **p, size_t n)
{
  if (n == 0) n = 1;
  HIPCHK(h, hipMalloc((void **)p, n * sizeof(T)));
  HIPCHK(h, hipMemsetAsync(*p, 0, n * sizeof(T), h->stream));
  return HPF_OK;
}

void dfree(void *p) { if (p) (void)hipFree(p); }

void free_held(hpf_handle::HeldSet &hs)
{
  dfree(hs.du); dfree(hs.di); dfree(hs.dy); dfree(hs.dout);
  if (hs.hout) (void)hipHostFree(hs.hout);
  hs = hpf_handle::HeldSet();
}

void free_side(Side &s, bool S_external)
{
  if (!S_external) dfree(s.S);
  dfree(s.E); dfree(s.L); dfree(s.W);
  dfree(s.prior_E); dfree(s.prior_used); dfree(s.prior_rate);
  dfree(s.prior_elog); dfree(s.prior_elog_used);
  dfree(s.colsum_used); dfree(s.colsum_part);
  dfree(s.rate_set); dfree(s.prior_shape_set); dfree(s.prior_elog_set);
  dfree(s.segs); dfree(s.longrows); dfree(s.grouprows); dfree(s.hugerows);
  dfree(s.partial); dfree(s.partial2); dfree(s.idx); dfree(s.val);
  dfree(s.p_idx); dfree(s.p_val); dfree(s.chunks);
  s = Side();
}

// bytes of a side's W: plain rows of ld doubles (floats use half), or rows of 16-byte pieces; never less than one whole
// row (the packed passes read row 0 of the gathered side even for a segment without nonzeros)
size_t w_bytes(const hpf_handle *h, uint32_t rows)
{
  const size_t row = std::max<size_t>((size_t)h->ld * 8, h->wl != WL_PLAIN ? (size_t)h->pk.row_bytes : 0);
  return (size_t)std::max<uint32_t>(rows, 1u) * row;
}

// pick (G,R[,V]) with G*R*V >= ld, R <= 8.  Cost = padded row length, +20 % when a lane
// group spans less than one 128-byte line per load although the row is at least two
// lines long (K=50: (4,7,2) loses to the wider, more padded (8,4,2): user pass 2.58
// vs 2.29 ms).  Ties (row lengths that several shapes cover exactly, e.g. K = 64,
// 128) are decided by what the K sweep on MI355X showed: keep
// 2..7 loads per lane in flight (R = 1 has no ILP: K=64 user pass 3.30 ms vs 2.35 ms
// at R = 4; R = 8 costs registers), span a line, and among equals be narrow (more
// nonzeros per wave).
bool choose_cfg(uint32_t ld, int V, int *G, int *R, int rmax64 = 8, long *cost_out = nullptr, bool gather = true)
{
  int bestG = 0, bestR = 0; long bestc = -1; int bestp = 0;
  const int Gs[5] = {4, 8, 16, 32, 64};
  for (int gi = 0; gi < 5; ++gi) {
    const int g = Gs[gi];
    const int r = (int)((ld + (uint32_t)(g * V) - 1) / (uint32_t)(g * V));
    if (r < 1 || r > (g == 64 ? rmax64 : 8)) continue;
    const bool narrow = g * V * 8 < 128;
    const long c = (long)g * r * V * ((gather && narrow && ld * 8 >= 256) ? 12 : 10);   // the sweep streams: padding only
    const int p = (r == 1 ? 3 : 0) + (r > 7 ? 2 : 0) + (narrow ? 1 : 0);
    if (bestc < 0 || c < bestc || (c == bestc && p < bestp)) { bestc = c; bestG = g; bestR = r; bestp = p; }
  }
  if (bestc < 0) return false;
  *G = bestG; *R = bestR;
  if (cost_out) *cost_out = bestc;
  return true;
}

// ---- kernel dispatch over the template grid -------------------------------
template <typename WT, int G, int R, int V>
void launch_phi_t(int side, const PhiArgs &a, uint32_t blocks, hipStream_t st)
{
  if (side & 1) hipLaunchKernelGGL((phi_pass_kernel<WT, G, R, V, 1>), dim3(blocks), dim3(256), 0, st, a);
  else          hipLaunchKernelGGL((phi_pass_kernel<WT, G, R, V, 0>), dim3(blocks), dim3(256), 0, st, a);
}
template <typename WT, int G, int V>
bool launch_phi_r(int R, int side, const PhiArgs &a, uint32_t blocks, hipStream_t st)
{
  switch (R) {
    case 1: launch_phi_t<WT, G, 1, V>(side, a, blocks, st); return true;
    case 2: launch_phi_t<WT, G, 2, V>(side, a, blocks, st); return true;
    case 3: launch_phi_t<WT, G, 3, V>(side, a, blocks, st); return true;
    case 4: launch_phi_t<WT, G, 4, V>(side, a, blocks, st); return true;
    case 5: launch_phi_t<WT, G, 5, V>(side, a, blocks, st); return true;
    case 6: launch_phi_t<WT, G, 6, V>(side, a, blocks, st); return true;
    case 7: launch_phi_t<WT, G, 7, V>(side, a, blocks, st); return true;
    case 8: launch_phi_t<WT, G, 8, V>(side, a, blocks, st); return true;
  }
  return false;
}
template <typename WT, int V>
bool launch_phi_g(int G, int R, int side, const PhiArgs &a, uint32_t blocks, hipStream_t st)
{
  switch (G) {
    case 4:  return launch_phi_r<WT, 4, V>(R, side, a, blocks, st);
    case 8:  return launch_phi_r<WT, 8, V>(R, side, a, blocks, st);
    case 16: return launch_phi_r<WT, 16, V>(R, side, a, blocks, st);
    case 32: return launch_phi_r<WT, 32, V>(R, side, a, blocks, st);
    case 64: return launch_phi_r<WT, 64, V>(R, side, a, blocks, st);
  }
  return false;
}
// V = elements per load: 1 or 2 doubles, 2 or 4 floats (8- or 16-byte accesses)
bool launch_phi(bool w32, int G, int R, int V, int side, const PhiArgs &a, uint32_t blocks, hipStream_t st)
{
  if (w32) return V == 4 ? launch_phi_g<float, 4>(G, R, side, a, blocks, st)
                         : launch_phi_g<float, 2>(G, R, side, a, blocks, st);
  return V == 2 ? launch_phi_g<double, 2>(G, R, side, a, blocks, st)
                : launch_phi_g<double, 1>(G, R, side, a, blocks, st);
}

template <int G, int MODE>
bool launch_sweep_r(int R, const SweepArgs &a, uint32_t blocks, hipStream_t st)
{
#define SW(RR) case RR: hipLaunchKernelGGL((row_sweep_kernel<G, RR, MODE>), dim3(blocks), dim3(256), 0, st, a); return true;
  if (MODE == SW_REG_P59) {                  // the slot counts p59 shapes have (p59_of_slots): 6 is none; G = 2 x the pass's lanes <= 64
    switch (R) { SW(1) SW(2) SW(3) SW(4) SW(5) SW(7) SW(8) SW(9) }
    return false;
  }
  switch (R) { SW(1) SW(2) SW(3) SW(4) SW(5) SW(6) SW(7) SW(8) }
  if (MODE == SW_F64 && R == 9) { hipLaunchKernelGGL((row_sweep_kernel<G, 9, MODE>), dim3(blocks), dim3(256), 0, st, a); return true; }
  if (G == 64) switch (R) { SW(9) SW(10) SW(11) SW(12) SW(13) SW(14) SW(15) SW(16) }   // 513..1024 columns
#undef SW
  return false;
}
template <int MODE>
bool launch_sweep_g(int G, int R, const SweepArgs &a, uint32_t blocks, hipStream_t st)
{
  switch (G) {
    case 4:  return MODE == SW_REG_P59 ? false : launch_sweep_r<4, MODE == SW_REG_P59 ? SW_PLAIN : MODE>(R, a, blocks, st);
    case 8:  return launch_sweep_r<8, MODE>(R, a, blocks, st);
    case 16: return launch_sweep_r<16, MODE>(R, a, blocks, st);
    case 32: return launch_sweep_r<32, MODE>(R, a, blocks, st);
    case 64: return launch_sweep_r<64, MODE>(R, a, blocks, st);
  }
  return false;
}
bool launch_sweep(int mode, int G, int R, const SweepArgs &a, uint32_t blocks, hipStream_t st)
{
  switch (mode) {
    case SW_REG_P59: return launch_sweep_g<SW_REG_P59>(G, R, a, blocks, st);
    case SW_LDS_P59: return G == 64 ? launch_sweep_r<64, SW_LDS_P59>(R, a, blocks, st) : false;   // narrower groups build in registers
    case SW_LDS_F48: return launch_sweep_g<SW_LDS_F48>(G, R, a, blocks, st);
    case SW_F64:     return launch_sweep_g<SW_F64>(G, R, a, blocks, st);
  }
  return launch_sweep_g<SW_PLAIN>(G, R, a, blocks, st);
}

// how a packed phi pass is launched: workgroups x threads (256 = four waves that share a chunk of segments, 64 = one wave: a
// tiled side) on a stream -- handed down explicitly (until round 5 the thread count travelled in a thread_local)
struct PhiLaunch { uint32_t blocks, wg; hipStream_t st; };

// packed W rows: G lanes per nonzero, L 16-byte pieces per lane (phi_pass_packed_kernel)
template <template <int> class C, int G, int L>
void launch_phipk_t(int side, const PhiArgs &a, const PhiLaunch &pl)
{
  if (side & 1) hipLaunchKernelGGL((phi_pass_packed_kernel<C, G, L, 1>), dim3(pl.blocks), dim3(pl.wg), 0, pl.st, a);
  else          hipLaunchKernelGGL((phi_pass_packed_kernel<C, G, L, 0>), dim3(pl.blocks), dim3(pl.wg), 0, pl.st, a);
}
template <template <int> class C, int G>
bool launch_phipk_l(int L, int side, const PhiArgs &a, const PhiLaunch &pl)
{
  switch (L) {
    case 1: launch_phipk_t<C, G, 1>(side, a, pl); return true;
    case 2: launch_phipk_t<C, G, 2>(side, a, pl); return true;
    case 3: launch_phipk_t<C, G, 3>(side, a, pl); return true;
    case 4: launch_phipk_t<C, G, 4>(side, a, pl); return true;
    case 5: launch_phipk_t<C, G, 5>(side, a, pl); return true;
    case 6: launch_phipk_t<C, G, 6>(side, a, pl); return true;
    case 7: launch_phipk_t<C, G, 7>(side, a, pl); return true;
    case 8: launch_phipk_t<C, G, 8>(side, a, pl); return true;
  }
  return false;
}
template <template <int> class C>
bool launch_phipk_g(int G, int L, int side, const PhiArgs &a, const PhiLaunch &pl)
{
  switch (G) {
    case 4:  return launch_phipk_l<C, 4>(L, side, a, pl);
    case 8:  return launch_phipk_l<C, 8>(L, side, a, pl);
    case 16: return launch_phipk_l<C, 16>(L, side, a, pl);
    case 32: return launch_phipk_l<C, 32>(L, side, a, pl);
    case 64: return launch_phipk_l<C, 64>(L, side, a, pl);
  }
  return false;
}
template <int G>
bool launch_phif64_l(int L, int side, const PhiArgs &a, const PhiLaunch &pl)
{
  if (L == 9) { launch_phipk_t<codec_f64, G, 9>(side, a, pl); return true; }      // stands in for p59 rows of 17 elements
  return launch_phipk_l<codec_f64, G>(L, side, a, pl);
}
bool launch_phi_packed(int wl, int G, int L, int side, const PhiArgs &a, const PhiLaunch &pl)
{
  if (wl == WL_F64) {
    switch (G) {
      case 4:  return launch_phif64_l<4>(L, side, a, pl);
      case 8:  return launch_phif64_l<8>(L, side, a, pl);
      case 16: return launch_phif64_l<16>(L, side, a, pl);
      case 32: return launch_phif64_l<32>(L, side, a, pl);
      case 64: return launch_phif64_l<64>(L, side, a, pl);
    }
    return false;
  }
  return wl == WL_P59 ? launch_phipk_g<codec_p59>(G, L, side, a, pl) : launch_phipk_g<codec_f48>(G, L, side, a, pl);
}

template <int G>
bool launch_gather_only_l(int L, const PhiArgs &a, uint32_t *sink, const PhiLaunch &pl)
{
#define GO(LL) case LL: hipLaunchKernelGGL((gather_only_kernel<G, LL>), dim3(pl.blocks), dim3(pl.wg), 0, pl.st, a, sink); return true;
  switch (L) { GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8) GO(9) }
#undef GO
  return false;
}
bool launch_gather_only(int G, int L, const PhiArgs &a, uint32_t *sink, const PhiLaunch &pl)
{
  switch (G) {
    case 4:  return launch_gather_only_l<4>(L, a, sink, pl);
    case 8:  return launch_gather_only_l<8>(L, a, sink, pl);
    case 16: return launch_gather_only_l<16>(L, a, sink, pl);
    case 32: return launch_gather_only_l<32>(L, a, sink, pl);
    case 64: return launch_gather_only_l<64>(L, a, sink, pl);
  }
  return false;
}

int recover_flush(hpf_handle *h, uint32_t fl0, uint32_t begun);

// A sweep (or derive_w) met an element the p59 rows cannot hold: move the handle to plain fp64 rows and repeat
// whatever the passes skipped since (recover_flush).  Synchronises the stream.  Every consumer outside the hot
// loop comes through here (or through check_flags) before it touches S, E or W.
int recover_if_flushed(hpf_handle *h)
{
  if (h->in_recovery || h->capturing) return HPF_OK;
  uint32_t f[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(f, h->flags, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!(f[0] & 2u)) return HPF_OK;
  return recover_flush(h, f[0], f[1]);
}

// surfaces a numerical breakdown the kernels flagged (synchronises the stream)
int check_flags(hpf_handle *h)
{
  int rc;
  if ((rc = recover_if_flushed(h))) return rc;
  uint32_t f = 0;
  HIPCHK(h, hipMemcpyAsync(&f, h->flags, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (f & 1u) {
    h->err = h->w32 ? "a softmax denominator underflowed to zero: the Elog spread is too wide for f32-stored W; use w_storage = 0"
                    : "a softmax denominator underflowed to zero (Elog spread > ~700): the state is not a valid HPF state";
    return HPF_ERR_STATE;
  }
  return HPF_OK;
}

int check_launch(hpf_handle *h, const char *what)
{
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    h->err = std::string(what) + ": " + hipGetErrorString(e);
    return HPF_ERR_HIP;
  }
  return HPF_OK;
}

int check_comm_async(hpf_handle *h)
{
  if (!h->comm || !g_rccl.CommGetAsyncError) return HPF_OK;
  int st = 0;
  const int rc = g_rccl.CommGetAsyncError(h->comm, &st);
  if (rc != 0 || (st != 0 && st != 7 /* ncclInProgress */)) {
    h->err = std::string("RCCL asynchronous error: ") + g_rccl.GetErrorString(rc != 0 ? rc : st);
    return HPF_ERR_HIP;
  }
  return HPF_OK;
}

// ---- host <-> device hand-over ----------------------------------------------
// Pageable hipMemcpy is staged by the runtime at ~4 GB/s; here the bytes go
// through two pinned buffers filled / drained by a few host threads while the
// other buffer is on the wire (HPF_H2D=staged, default), or the caller's pages
// are pinned for the duration of the copy (HPF_H2D=register), or it is left to
// the runtime (HPF_H2D=plain).
constexpr size_t STAGE_BYTES = 64u << 20;

int ensure_stage(hpf_handle *h)
{
  if (h->stage[0]) return HPF_OK;
  for (int k = 0; k < 2; ++k) {
    HIPCHK(h, hipHostMalloc(&h->stage[k], STAGE_BYTES, hipHostMallocDefault));
    HIPCHK(h, hipEventCreateWithFlags(&h->stage_ev[k], hipEventDisableTiming));
  }
  h->stage_bytes = STAGE_BYTES;
  return HPF_OK;
}

void par_memcpy(void *dst, const void *src, size_t bytes, unsigned T)
{
  if (T <= 1 || bytes < (8u << 20)) { memcpy(dst, src, bytes); return; }
  const size_t per = ((bytes + T - 1) / T + 4095) & ~(size_t)4095;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < T; ++t) {
    const size_t o = (size_t)t * per;
    if (o >= bytes) break;
    const size_t len = std::min(per, bytes - o);
    th.emplace_back([=] { memcpy((char *)dst + o, (const char *)src + o, len); });
  }
  for (auto &x : th) x.join();
}

// is this host pointer page-locked memory HIP knows (hpf_host_alloc, hipHostMalloc, hipHostRegister)?  Then it is the
// DMA's own source / target: no staging, no host copy
bool host_is_pinned(const void *p)
{
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

// contiguous host -> device; returns after the bytes have left `src`
int h2d(hpf_handle *h, void *dst, const void *src, size_t bytes)
{
  if (!bytes) return HPF_OK;
  if (bytes >= (1u << 20) && host_is_pinned(src) && host_is_pinned((const char *)src + bytes - 1)) {
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  }
  if (h->xfer_mode == 2 && bytes >= (1u << 20)) {
    if (hipHostRegister(const_cast<void *>(src), bytes, hipHostRegisterDefault) == hipSuccess) {
      hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      (void)hipHostUnregister(const_cast<void *>(src));
      if (e != hipSuccess) { h->err = std::string("h2d: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
      return HPF_OK;
    }
    (void)hipGetLastError();             // not registrable (e.g. read-only mapping): stage it
  }
  if (h->xfer_mode == 0 || bytes < (1u << 20)) {
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  }
  int rc;
  if ((rc = ensure_stage(h))) return rc;
  size_t off = 0; int k = 0;
  bool used[2] = {false, false};
  while (off < bytes) {
    const size_t len = std::min(h->stage_bytes, bytes - off);
    if (used[k]) HIPCHK(h, hipEventSynchronize(h->stage_ev[k]));
    par_memcpy(h->stage[k], (const char *)src + off, len, h->xfer_threads);
    HIPCHK(h, hipMemcpyAsync((char *)dst + off, h->stage[k], len, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->stage_ev[k], h->stream));
    used[k] = true; k ^= 1; off += len;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

// contiguous device -> host (synchronous)
int d2h(hpf_handle *h, void *dst, const void *src, size_t bytes)
{
  if (!bytes) return HPF_OK;
  if (bytes >= (1u << 20) && host_is_pinned(dst) && host_is_pinned((char *)dst + bytes - 1)) {
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  }
  if (h->xfer_mode == 2 && bytes >= (1u << 20)) {
    if (hipHostRegister(dst, bytes, hipHostRegisterDefault) == hipSuccess) {
      hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
      (void)hipHostUnregister(dst);
      if (e != hipSuccess) { h->err = std::string("d2h: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
      return HPF_OK;
    }
    (void)hipGetLastError();
  }
  if (h->xfer_mode == 0 || bytes < (1u << 20)) {
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  }
  int rc;
  if ((rc = ensure_stage(h))) return rc;
  // chunk c is copied out of its pinned buffer while chunk c+1 is on the wire
  size_t off = 0, prev_off = 0, prev_len = 0; int k = 0;
  bool have_prev = false;
  while (off < bytes || have_prev) {
    size_t len = 0;
    if (off < bytes) {
      len = std::min(h->stage_bytes, bytes - off);
      HIPCHK(h, hipMemcpyAsync(h->stage[k], (const char *)src + off, len, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipEventRecord(h->stage_ev[k], h->stream));
    }
    if (have_prev) {
      HIPCHK(h, hipEventSynchronize(h->stage_ev[k ^ 1]));
      par_memcpy((char *)dst + prev_off, h->stage[k ^ 1], prev_len, h->xfer_threads);
    }
    have_prev = len > 0; prev_off = off; prev_len = len; off += len; k ^= 1;
  }
  return HPF_OK;
}

uint32_t grid_for(uint64_t n) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>((n + 255) / 256, 1), 16384); }

// dense [rows x cols] host <-> columns [col0, col0+cols) of the padded device
// matrix.  Full-width blocks (cols == ld) move straight; anything else goes
// through a contiguous device chunk and a repack kernel.
int copy_in(hpf_handle *h, double *dev, uint32_t ld, uint32_t col0, const double *host,
            uint32_t rows, uint32_t cols)
{
  if (!rows || !cols) return HPF_OK;
  if (cols == ld && col0 == 0) return h2d(h, dev, host, (size_t)rows * cols * 8);
  const uint64_t rows_per = std::max<uint64_t>(1, (256u << 20) / ((uint64_t)cols * 8));
  double *tmp = nullptr; int rc;
  if ((rc = dalloc(h, &tmp, (size_t)std::min<uint64_t>(rows, rows_per) * cols))) return rc;
  for (uint64_t r0 = 0; r0 < rows && !rc; r0 += rows_per) {
    const uint64_t nr = std::min<uint64_t>(rows_per, rows - r0);
    if ((rc = h2d(h, tmp, host + r0 * cols, (size_t)nr * cols * 8))) break;
    hipLaunchKernelGGL(repack_in_kernel, dim3(grid_for(nr * cols)), dim3(256), 0, h->stream, tmp,
                       dev + r0 * ld, nr, cols, ld, col0);
    rc = check_launch(h, "repack_in_kernel");
  }
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "copy_in: stream error"; rc = HPF_ERR_HIP; }
  dfree(tmp);
  return rc;
}
int copy_out(hpf_handle *h, const double *dev, uint32_t ld, uint32_t col0, double *host,
             uint32_t rows, uint32_t cols)
{
  if (!rows || !cols) return HPF_OK;
  if (cols == ld && col0 == 0) return d2h(h, host, dev, (size_t)rows * cols * 8);
  const uint64_t rows_per = std::max<uint64_t>(1, (256u << 20) / ((uint64_t)cols * 8));
  double *tmp = nullptr; int rc;
  if ((rc = dalloc(h, &tmp, (size_t)std::min<uint64_t>(rows, rows_per) * cols))) return rc;
  for (uint64_t r0 = 0; r0 < rows && !rc; r0 += rows_per) {
    const uint64_t nr = std::min<uint64_t>(rows_per, rows - r0);
    hipLaunchKernelGGL(repack_out_kernel, dim3(grid_for(nr * cols)), dim3(256), 0, h->stream,
                       dev + r0 * ld, tmp, nr, cols, ld, col0);
    if ((rc = check_launch(h, "repack_out_kernel"))) break;
    rc = d2h(h, host + r0 * cols, tmp, (size_t)nr * cols * 8);
  }
  dfree(tmp);
  return rc;
}
// the same for a caller's DEVICE buffer (hpf_set_state_device / hpf_get_state_device)
int copy_in_dev(hpf_handle *h, double *dev, uint32_t ld, uint32_t col0, const double *src,
                uint32_t rows, uint32_t cols)
{
  if (!rows || !cols) return HPF_OK;
  if (cols == ld && col0 == 0)
    HIPCHK(h, hipMemcpyAsync(dev, src, (size_t)rows * cols * 8, hipMemcpyDeviceToDevice, h->stream));
  else {
    hipLaunchKernelGGL(repack_in_kernel, dim3(grid_for((uint64_t)rows * cols)), dim3(256), 0, h->stream, src,
                       dev, (uint64_t)rows, cols, ld, col0);
    int rc = check_launch(h, "repack_in_kernel");
    if (rc) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}
int copy_out_dev(hpf_handle *h, const double *dev, uint32_t ld, uint32_t col0, double *dst,
                 uint32_t rows, uint32_t cols)
{
  if (!rows || !cols) return HPF_OK;
  if (cols == ld && col0 == 0)
    HIPCHK(h, hipMemcpyAsync(dst, dev, (size_t)rows * cols * 8, hipMemcpyDeviceToDevice, h->stream));
  else {
    hipLaunchKernelGGL(repack_out_kernel, dim3(grid_for((uint64_t)rows * cols)), dim3(256), 0, h->stream, dev,
                       dst, (uint64_t)rows, cols, ld, col0);
    int rc = check_launch(h, "repack_out_kernel");
    if (rc) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

// exclusive scan of n counters (uint32 or uint64) into uint64 out[0..n) (+ out[n] = total)
template <typename IN>
int device_scan(hpf_handle *h, const IN *in, uint64_t n, uint64_t *out, bool write_total)
{
  if (n == 0) {
    if (write_total) HIPCHK(h, hipMemsetAsync(out, 0, 8, h->stream));
    return HPF_OK;
  }
  const uint64_t nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  uint64_t *bsum = nullptr; int rc;
  if ((rc = dalloc(h, &bsum, (size_t)nb))) return rc;
  hipLaunchKernelGGL((scan_reduce_kernel<IN>), dim3((uint32_t)nb), dim3(256), 0, h->stream, in, n, bsum);
  hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(256), 0, h->stream, bsum, nb);
  hipLaunchKernelGGL((scan_apply_kernel<IN>), dim3((uint32_t)nb), dim3(256), 0, h->stream, in, n, bsum, out,
                     write_total ? 1 : 0);
  rc = check_launch(h, "device_scan");
  hipError_t e = hipStreamSynchronize(h->stream);
  dfree(bsum);
  if (!rc && e != hipSuccess) { h->err = std::string("device_scan: ") + hipGetErrorString(e); rc = HPF_ERR_HIP; }
  return rc;
}

// Work lists of one side cut on the device from its row pointers (dptr, in HBM): nothing of
// size O(rows) or O(nnz) crosses PCIe.
int device_side_work(hpf_handle *h, Side &s, const int64_t *dptr, uint32_t rows)
{
  dfree(s.segs); dfree(s.longrows); dfree(s.grouprows); dfree(s.hugerows);
  s.segs = nullptr; s.longrows = s.grouprows = s.hugerows = nullptr;
  s.nseg = s.nlong = s.nlong_wave = s.ngroup = s.nhuge = 0;
  dfree(s.partial); dfree(s.partial2);
  s.partial = nullptr; s.partial2 = nullptr;
  s.npartial = 0; s.npartial2 = 0;
  dfree(s.p_idx); dfree(s.p_val); dfree(s.chunks);
  s.p_idx = nullptr; s.p_val = nullptr; s.chunks = nullptr;
  s.nchunk_blocks = 0; s.tiles = 0; s.tile_rows = 0; s.chunk_segs = 0; s.tiled_nnz = 0; s.light_below = 0;
  if (rows == 0) return HPF_OK;
  int rc = HPF_OK;
  uint64_t *cnt[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t *bad = nullptr;
  do {
    for (int k = 0; k < 5 && !rc; ++k) rc = dalloc(h, &cnt[k], (size_t)rows + 1);
    if (rc || (rc = dalloc(h, &bad, 1))) break;
    SegPlan pl = {cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]};
    hipLaunchKernelGGL(seg_plan_kernel, dim3(grid_for(rows)), dim3(256), 0, h->stream, dptr, rows, h->seg_max,
                       h->huge_slots, h->group_slots, pl, bad);
    if ((rc = check_launch(h, "seg_plan_kernel"))) break;
    for (int k = 0; k < 5 && !rc; ++k) rc = device_scan<uint64_t>(h, cnt[k], rows, cnt[k], true);
    if (rc) break;
    uint64_t tot[5]; uint32_t hb = 0;
    hipError_t e = hipSuccess;
    for (int k = 0; k < 5 && e == hipSuccess; ++k)
      e = hipMemcpyAsync(&tot[k], cnt[k] + rows, 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = std::string("seg_plan_kernel: ") + hipGetErrorString(e); rc = HPF_ERR_HIP; break; }
    if (hb) { h->err = "row pointers must start at 0 and be monotone"; rc = HPF_ERR_INVALID; break; }
    if (tot[0] > 0xffffffffull || tot[1] > 0x7fffffffull) { h->err = "too many segments for 32-bit work lists"; rc = HPF_ERR_UNSUPPORTED; break; }
    s.nseg = (uint32_t)tot[0]; s.npartial = (uint32_t)tot[1]; s.nlong = s.nlong_wave = (uint32_t)tot[2];
    s.nhuge = (uint32_t)tot[3]; s.ngroup = s.npartial2 = (uint32_t)tot[4];
    if ((rc = dalloc(h, &s.segs, s.nseg)) || (rc = dalloc(h, &s.longrows, s.nlong))) break;
    if (s.ngroup && ((rc = dalloc(h, &s.grouprows, s.ngroup)) || (rc = dalloc(h, &s.hugerows, s.nhuge)))) break;
    hipLaunchKernelGGL(seg_fill_kernel, dim3(grid_for(rows)), dim3(256), 0, h->stream, dptr, rows, h->seg_max,
                       h->huge_slots, h->group_slots, pl, s.segs, s.longrows, s.hugerows, s.grouprows);
    if ((rc = check_launch(h, "seg_fill_kernel"))) break;
    if ((rc = dalloc(h, &s.partial, (size_t)s.npartial * h->ld))) break;
    if (s.npartial2 && (rc = dalloc(h, &s.partial2, (size_t)s.npartial2 * h->ld))) break;
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "work-list build failed on the device"; rc = HPF_ERR_HIP; }
  } while (0);
  for (int k = 0; k < 5; ++k) dfree(cnt[k]);
  dfree(bad);
  return rc;
}

// ---- stable LSD radix sort of n records on the low `bits` bits of a 32-bit key, carrying a 32-bit
// payload u (NULL at the start: the row of the record, by bisection in rowptr), an optional second
// one x, and an optional byte v.  Buffers: the pass p writes set p & 1 of (K, U, X, V); *res = the
// set that holds the result.  in_* are only read.
struct SortSets { uint32_t *K[2], *U[2], *X[2]; uint8_t *V[2]; };
int radix_sort_records(hpf_handle *h, uint64_t n, uint32_t bits, const uint32_t *in_k, const uint32_t *in_u,
                       const uint32_t *in_x, const uint8_t *in_v, const int64_t *rowptr, uint32_t n_rows,
                       const SortSets &b, int *res)
{
  const uint32_t P = std::max<uint32_t>(1, (bits + RADIX_BITS - 1) / RADIX_BITS);
  const uint64_t ntiles = (n + RADIX_TILE - 1) / RADIX_TILE;
  const uint32_t nblk = (uint32_t)((ntiles + 3) / 4);
  uint64_t *counts = nullptr; uint32_t *bad = nullptr; int rc;
  if ((rc = dalloc(h, &counts, (size_t)ntiles * RADIX_DIGITS)) || (rc = dalloc(h, &bad, 1))) { dfree(counts); dfree(bad); return rc; }
  for (uint32_t p = 0; p < P && !rc; ++p) {
    const int o = (int)(p & 1), i = o ^ 1;
    RadixArgs a;
    a.keys_in = p == 0 ? in_k : b.K[i];
    a.users_in = p == 0 ? in_u : b.U[i];
    a.extra_in = p == 0 ? in_x : b.X[i];
    a.vals_in = p == 0 ? in_v : (in_v ? b.V[i] : nullptr);
    a.rowptr = rowptr; a.n_rows = n_rows;
    a.keys_out = b.K[o]; a.users_out = b.U[o];
    a.extra_out = in_x ? b.X[o] : nullptr;
    a.vals_out = in_v ? b.V[o] : nullptr;
    a.offsets = counts; a.nnz = n; a.ntiles = ntiles; a.shift = p * RADIX_BITS;
    hipLaunchKernelGGL(radix_count_kernel, dim3(nblk), dim3(256), 0, h->stream, a.keys_in, n, a.shift, ntiles, counts,
                       0xffffffffu, bad);
    if ((rc = check_launch(h, "radix_count_kernel"))) break;
    if ((rc = device_scan<uint64_t>(h, counts, ntiles * RADIX_DIGITS, counts, false))) break;
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(256), 0, h->stream, a);
    rc = check_launch(h, "radix_scatter_kernel");
  }
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "record sort failed on the device"; rc = HPF_ERR_HIP; }
  dfree(counts); dfree(bad);
  *res = (int)((P - 1) & 1);
  return rc;
}

uint32_t bits_for(uint64_t count) { uint32_t b = 0; while (b < 32 && ((uint64_t)1 << b) < count) ++b; return std::max<uint32_t>(b, 1); }

// Tiled work list of one side (hpf_build.hpp "Tiled phi pass").  Called after
// device_side_work: when the policy finds the side worth tiling, the plain work list is replaced.
//   ptr       the side's own row pointers (rows + 1), idx / val its nonzeros in row order
//   ptr_oth   row pointers of the gathered side (rows_oth + 1): degrees for the hot-set policy
// Policy (auto):  the gathered matrix must be larger than twice a tile.
//   "ranges"  tiles of T consecutive gathered rows; owner rows with at least tiles * min_run nonzeros are
//             heavy (their nonzeros are regrouped), taken when those hold >= 15 % of the nonzeros
// Forced by HPF_TILE (experimental knob): 0 never, 1 ranges whenever there are two tiles.
int build_tiled_side(hpf_handle *h, Side &s, const int64_t *ptr, uint32_t rows_oth, uint64_t nnz, size_t row_bytes)
{
  if (!((h->tile_sides >> (&s == &h->it ? 1 : 0)) & 1)) return HPF_OK;
  if (h->tile_mode == 0 || h->cfg.tiling == 1 || nnz == 0 || s.rows == 0) return HPF_OK;      // (>= 2^32 nonzeros: positions are 64-bit throughout;
                                                                                              //  segments and partial slots must number < 2^31, checked below)
  const uint32_t T = (uint32_t)std::max<uint64_t>(h->tile_bytes / row_bytes, 1);
  const uint32_t tiles = (rows_oth + T - 1) / T;
  if (tiles < 2 || tiles > 65534) return HPF_OK;
  if (h->tile_mode == 2 && (uint64_t)rows_oth * row_bytes < 2 * h->tile_bytes + h->tile_bytes / 2) return HPF_OK;
  int rc = HPF_OK;
  // a gathered matrix of a few tiles is already served largely from L2 when the ratings are skewed (the popular
  // rows stay resident): regrouping then pays only for rows that meet a tile many times (size sweep, m = 20 000:
  // four tiles, user pass 3.19 ms row-major, 3.64 ms with runs of 16)
  // ... and a run is worth its fixed work from two batches on: a batch is 64 / G nonzeros (K = 50: G = 4, sixteen
  // per batch -- with runs of 16 its ten-tile user side went 2.31 -> 2.68 ms)
  // Round 5: with one wave per workgroup on a tiled side (hpf_handle::phi_wg) a run costs less, and the bar of 16 came down
  // to 12 where a batch holds eight nonzeros or fewer (C2 8.42 -> 8.34 ms, a C3 shard 24.3 -> 23.9; 10 is better still at C2
  // and worse on the shard; C4 is flat from 10 to 16; K = 50's two batches of sixteen stay: 24 loses 1 %; experiments.md)
  const uint32_t per_batch = h->nz_per_batch();
  const uint32_t min_run = h->tile_min_run ? h->tile_min_run                    // HPF_TILE_RUN: as given
                                           : (per_batch >= 16u ? 2u * per_batch : 12u);
  uint64_t light_below = (uint64_t)tiles * min_run * (tiles < 8 ? 4u : 1u);
  if (h->tile_mode == 1) light_below = 0;                         // forced: every row is regrouped
  unsigned long long *stat = nullptr;
  uint64_t heavy_rows = 0, heavy_nnz = nnz;
  if (light_below) {
    if ((rc = dalloc(h, &stat, 2))) return rc;
    hipLaunchKernelGGL(deg_ge_kernel, dim3(grid_for(s.rows)), dim3(256), 0, h->stream, ptr, s.rows, light_below, stat);
    unsigned long long hv[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(hv, stat, 16, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dfree(stat);
    if (e != hipSuccess) { h->err = std::string("deg_ge_kernel: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
    heavy_rows = hv[0]; heavy_nnz = hv[1];
    if ((double)heavy_nnz < h->tile_min_share * (double)nnz) return HPF_OK;
  }
  (void)heavy_rows;
  {
    // Room for the temporaries (26 bytes per nonzero)?  Decided from what this handle holds and the size of the device, not
    // from the free memory of the moment: tiling changes the order of a row's sum, and whether a side is tiled
    // must be a function of the job, not of what else happens to be allocated.  If the allocations fail all the same the
    // side stays row-major and hpf_work_info.notes says so.
    size_t fr = 0, tot = 0;
    const double by = h->u.val ? 1.0 : 0.0;
    const double resident = 5.0 * 8.0 * (double)h->ld * ((double)h->u.rows + (double)h->it.rows)        // S, E, L, W + vectors of both sides
                          + (double)nnz * (3.0 * (4.0 + by));                                          // CSR, CSC and the other side's tiled copy
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && resident + (double)nnz * (26.0 + 4.0 + by) + (double)(1ull << 30) > 0.9 * (double)tot) {
      h->notes |= (&s == &h->it ? 2u : 1u);
      return HPF_OK;
    }
  }

  const uint32_t nkeys = tiles + 1;
  uint32_t *tilemap = nullptr, *first_seg = nullptr, *seg_row = nullptr, *seg_key = nullptr, *iota = nullptr;
  uint64_t *cnt = nullptr, *pl[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int64_t *segptr = nullptr;
  SortSets b = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  SortSets sb = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  uint32_t *key0 = nullptr, *row0 = nullptr;
  Seg *segs = nullptr; LongRow *longs = nullptr, *huges = nullptr, *groups = nullptr;
  double *partial = nullptr, *partial2 = nullptr; uint2 *chunks_dev = nullptr;
  uint32_t *keep_idx = nullptr; uint8_t *keep_val = nullptr;
  bool done = false;
  do {
    // ---- keys and the sort
    if ((rc = dalloc(h, &tilemap, rows_oth)) || (rc = dalloc(h, &key0, (size_t)nnz)) || (rc = dalloc(h, &row0, (size_t)nnz))) break;
    hipLaunchKernelGGL(tile_map_kernel, dim3(grid_for(rows_oth)), dim3(256), 0, h->stream, tilemap, rows_oth, T);
    const uint64_t nwt = (nnz + RADIX_TILE - 1) / RADIX_TILE;
    const uint32_t wblk = (uint32_t)((nwt + 3) / 4);
    hipLaunchKernelGGL(tile_key_kernel, dim3(wblk), dim3(256), 0, h->stream, ptr, s.rows, s.idx, nnz, tilemap, light_below,
                       key0, row0);
    if ((rc = check_launch(h, "tile_key_kernel"))) break;
    const uint32_t kbits = bits_for(nkeys), KP = (kbits + RADIX_BITS - 1) / RADIX_BITS;
    for (int k = 0; k < (KP > 1 ? 2 : 1) && !rc; ++k) {
      if ((rc = dalloc(h, &b.K[k], (size_t)nnz)) || (rc = dalloc(h, &b.U[k], (size_t)nnz)) || (rc = dalloc(h, &b.X[k], (size_t)nnz))) break;
      if (s.val) rc = dalloc(h, &b.V[k], (size_t)nnz);
    }
    if (rc) break;
    int r = 0;
    if ((rc = radix_sort_records(h, nnz, kbits, key0, row0, s.idx, s.val, nullptr, 0, b, &r))) break;
    dfree(key0); key0 = nullptr; dfree(row0); row0 = nullptr; dfree(tilemap); tilemap = nullptr;
    const uint32_t *skey = b.K[r], *srow = b.U[r];
    keep_idx = b.X[r]; keep_val = b.V[r]; b.X[r] = nullptr; b.V[r] = nullptr;
    for (int k = 0; k < 2; ++k) if (k != r) { dfree(b.K[k]); dfree(b.U[k]); dfree(b.X[k]); dfree(b.V[k]); b.K[k] = b.U[k] = b.X[k] = nullptr; b.V[k] = nullptr; }

    // ---- segments
    if ((rc = dalloc(h, &cnt, (size_t)nwt + 1))) break;
    hipLaunchKernelGGL(seg_count_kernel, dim3(wblk), dim3(256), 0, h->stream, skey, srow, nnz, h->seg_max, cnt);
    if ((rc = check_launch(h, "seg_count_kernel"))) break;
    if ((rc = device_scan<uint64_t>(h, cnt, nwt, cnt, true))) break;
    uint64_t nseg64 = 0;
    HIPBRK(h, hipMemcpyAsync(&nseg64, cnt + nwt, 8, hipMemcpyDeviceToHost, h->stream));
    HIPBRK(h, hipStreamSynchronize(h->stream));
    if (nseg64 == 0 || nseg64 > 0x7fffffffull) break;              // leaves the plain list in place
    const uint32_t nseg = (uint32_t)nseg64;
    if ((rc = dalloc(h, &segs, nseg)) || (rc = dalloc(h, &seg_row, nseg)) || (rc = dalloc(h, &seg_key, nseg)) ||
        (rc = dalloc(h, &first_seg, nkeys))) break;
    HIPBRK(h, hipMemsetAsync(first_seg, 0xff, (size_t)nkeys * 4, h->stream));
    hipLaunchKernelGGL(seg_emit_kernel, dim3(wblk), dim3(256), 0, h->stream, skey, srow, nnz, h->seg_max, cnt, segs, seg_row, seg_key);
    hipLaunchKernelGGL(seg_len_kernel, dim3(grid_for(nseg)), dim3(256), 0, h->stream, segs, nseg, nnz, seg_key, first_seg);
    if ((rc = check_launch(h, "seg_emit_kernel"))) break;
    std::vector<uint32_t> fs(nkeys);
    HIPBRK(h, hipMemcpyAsync(fs.data(), first_seg, (size_t)nkeys * 4, hipMemcpyDeviceToHost, h->stream));
    HIPBRK(h, hipStreamSynchronize(h->stream));
    dfree(b.K[r]); dfree(b.U[r]); b.K[r] = b.U[r] = nullptr;
    dfree(cnt); cnt = nullptr; dfree(seg_key); seg_key = nullptr; dfree(first_seg); first_seg = nullptr;

    // ---- per owner row: its segments in key order -> partial slots and combine lists
    if ((rc = dalloc(h, &iota, nseg))) break;
    hipLaunchKernelGGL(iota_kernel, dim3(grid_for(nseg)), dim3(256), 0, h->stream, iota, nseg);
    for (int k = 0; k < 2 && !rc; ++k) { if ((rc = dalloc(h, &sb.K[k], nseg))) break; rc = dalloc(h, &sb.U[k], nseg); }
    if (rc) break;
    int sr = 0;
    if ((rc = radix_sort_records(h, nseg, bits_for(s.rows), seg_row, iota, nullptr, nullptr, nullptr, 0, sb, &sr))) break;
    if ((rc = dalloc(h, &segptr, (size_t)s.rows + 1))) break;
    hipLaunchKernelGGL(colptr_from_sorted_kernel, dim3(grid_for(nseg)), dim3(256), 0, h->stream, sb.K[sr], (uint64_t)nseg, s.rows, segptr);
    for (int k = 0; k < 5 && !rc; ++k) rc = dalloc(h, &pl[k], (size_t)s.rows + 1);
    if (rc) break;
    SegPlan plan = {pl[0], pl[1], pl[2], pl[3], pl[4]};
    hipLaunchKernelGGL(slot_plan_kernel, dim3(grid_for(s.rows)), dim3(256), 0, h->stream, segptr, s.rows, h->huge_slots,
                       h->group_slots, plan);
    if ((rc = check_launch(h, "slot_plan_kernel"))) break;
    for (int k = 0; k < 5 && !rc; ++k) rc = device_scan<uint64_t>(h, pl[k], s.rows, pl[k], true);
    if (rc) break;
    uint64_t tot[5];
    {
      hipError_t e = hipSuccess;
      for (int k = 0; k < 5 && e == hipSuccess; ++k) e = hipMemcpyAsync(&tot[k], pl[k] + s.rows, 8, hipMemcpyDeviceToHost, h->stream);
      HIPBRK(h, e);
    }
    HIPBRK(h, hipStreamSynchronize(h->stream));
    if (tot[1] > 0x7fffffffull) break;
    const uint32_t npartial = (uint32_t)tot[1], nlong = (uint32_t)tot[2], nhuge = (uint32_t)tot[3], ngroup = (uint32_t)tot[4];
    if ((rc = dalloc(h, &longs, nlong))) break;
    if (ngroup && ((rc = dalloc(h, &groups, ngroup)) || (rc = dalloc(h, &huges, nhuge)))) break;
    hipLaunchKernelGGL(slot_fill_kernel, dim3(grid_for(s.rows)), dim3(256), 0, h->stream, segptr, s.rows, h->huge_slots,
                       h->group_slots, plan, sb.U[sr], segs, longs, huges, groups);
    if ((rc = check_launch(h, "slot_fill_kernel"))) break;
    if ((rc = dalloc(h, &partial, (size_t)npartial * h->ld))) break;
    if (ngroup && (rc = dalloc(h, &partial2, (size_t)ngroup * h->ld))) break;
    // the combine gives the rows with more than COMBINE_SPLIT partials a workgroup each: they go behind the others (a stable
    // partition of the row-ordered list: a function of the matrix alone)
    // (the rows WITHOUT any nonzero stay in the wave kernel's part: taking them out -- cleared once, and again only after
    // an export or a hand-over had written S -- was built in round 5 and bought nothing measurable: C2's 48 000 of them
    // are 40 MB of zeros, ~10 us of a 45 us launch that the multi-segment light rows' chains set; experiments.md)
    uint32_t nlong_wave = nlong;
    if (nlong) {
      std::vector<LongRow> lr(nlong), big;
      HIPBRK(h, hipMemcpyAsync(lr.data(), longs, (size_t)nlong * sizeof(LongRow), hipMemcpyDeviceToHost, h->stream));
      HIPBRK(h, hipStreamSynchronize(h->stream));
      size_t w = 0;
      for (const LongRow &x : lr) { if (x.nslots > COMBINE_SPLIT) big.push_back(x); else lr[w++] = x; }
      if (!big.empty()) {
        std::copy(big.begin(), big.end(), lr.begin() + (ptrdiff_t)w);
        nlong_wave = (uint32_t)w;
        HIPBRK(h, hipMemcpyAsync(longs, lr.data(), (size_t)nlong * sizeof(LongRow), hipMemcpyHostToDevice, h->stream));
        HIPBRK(h, hipStreamSynchronize(h->stream));
      }
    }

    // ---- chunks: eight queues, one per XCD (workgroup b runs on XCD b % 8).  The first 8 * floor(t / 8) of the t tiles
    // that hold segments go WHOLE to the queue with the least work so far (a tile's segments stay together and in order:
    // its rows are fetched into one L2, once); the remaining t mod 8 tiles level the queues: their segments are poured, in
    // order, into the queues up to the common fill mark, so that a levelling tile is shared by two or three XCDs instead of
    // all eight and every queue ends at the same count (round 5; until round 4 EVERY tile was cut eight ways below 32
    // tiles, so that each XCD fetched every tile: an eighth of C4 -- 22 tiles of users -- ran its item pass in 1.36 ms,
    // now 1.10; 9 and 12 tiles: -4 %; from 32 tiles on nothing changes).  Below tile_split_below = 8 tiles the old rule
    // stays (HPF_TILE_SPLIT_BELOW=N: for fewer than N tiles).  The row-major rest (key 0) is cut eight ways.
    std::vector<std::pair<uint32_t, uint32_t>> q[8], qt[8];
    uint64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto key_end = [&](uint32_t k) { for (uint32_t j = k + 1; j < nkeys; ++j) if (fs[j] != 0xffffffffu) return fs[j]; return nseg; };
    uint64_t tiled_segs = 0;
    uint32_t tiles_present = 0;
    for (uint32_t k = 1; k < nkeys; ++k) if (fs[k] != 0xffffffffu) { tiles_present++; tiled_segs += key_end(k) - fs[k]; }
    const bool old_rule = tiles < h->tile_split_below;
    const uint32_t whole_tiles = old_rule ? 0u : (tiles_present / 8u) * 8u;
    uint32_t seen = 0;
    int fill = 0;                                                  // the queue the levelling tiles are being poured into
    const uint64_t mark = (tiled_segs + 7) / 8;                    // the common fill mark
    for (uint32_t k = 1; k < nkeys; ++k) {
      if (fs[k] == 0xffffffffu) continue;
      const uint32_t a0 = fs[k], a1 = key_end(k);
      if (seen++ < whole_tiles) {
        int best = 0;
        for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
        qt[best].push_back({a0, a1}); load[best] += a1 - a0;
        continue;
      }
      if (old_rule) {
        const uint64_t n = a1 - a0;
        for (int x = 0; x < 8; ++x) {
          const uint32_t lo = a0 + (uint32_t)(n * x / 8), hi = a0 + (uint32_t)(n * (x + 1) / 8);
          if (hi > lo) { qt[x].push_back({lo, hi}); load[x] += hi - lo; }
        }
        continue;
      }
      for (uint32_t p0 = a0; p0 < a1;) {
        while (fill < 7 && load[fill] >= mark) ++fill;
        const uint64_t room = fill < 7 ? mark - load[fill] : (uint64_t)(a1 - p0);
        const uint32_t take = (uint32_t)std::min<uint64_t>(room, a1 - p0);
        qt[fill].push_back({p0, p0 + take}); load[fill] += take;
        p0 += take;
      }
    }
    // the row-major rest: an eighth per queue.  tile_order 0: in front of the tiles; 1: in front on the even
    // XCDs, behind on the odd ones (half the chip pulls over the fabric while the other half runs from L2);
    // 2: dealt between the tiles in equal pieces
    const uint32_t c0 = fs[0] != 0xffffffffu ? fs[0] : 0u, c1 = fs[0] != 0xffffffffu ? key_end(0) : 0u;
    for (int x = 0; x < 8; ++x) {
      const uint64_t n = c1 - c0;
      const uint32_t lo = c0 + (uint32_t)(n * x / 8), hi = c0 + (uint32_t)(n * (x + 1) / 8);
      const bool front = h->tile_order == 0 || (h->tile_order == 1 && (x & 1) == 0);
      if (h->tile_order == 2 && !qt[x].empty()) {
        const uint64_t nt = qt[x].size(), nc = hi - lo;
        for (uint64_t t = 0; t < nt; ++t) {
          const uint32_t l2 = lo + (uint32_t)(nc * t / nt), h2 = lo + (uint32_t)(nc * (t + 1) / nt);
          if (h2 > l2) q[x].push_back({l2, h2});
          q[x].push_back(qt[x][t]);
        }
        continue;
      }
      if (front && hi > lo) q[x].push_back({lo, hi});
      for (auto &rg : qt[x]) q[x].push_back(rg);
      if (!front && hi > lo) q[x].push_back({lo, hi});
    }
    // a launch holds fewer than 2^32 work-items (the AQL packet counts them in 32 bits): at most 2^28 / wg workgroups
    // (2^20 of four waves, 2^22 of one -- round 5 kept 2^20 for the one-wave groups as well, and the longest lists, whole
    // C3 or C5 on one GPU, got chunks of 8-16 segments instead of two), so a list too long for chunks of
    // tile_chunk segments gets longer chunks; hpf_work_info.tile_chunk_user / _item say what was used
    const uint32_t wg = h->wg_of(true);
    const size_t max_wgs = ((size_t)1 << 28) / wg;
    uint32_t CH = h->tile_chunk ? h->tile_chunk : 2 * (wg / 64);      // two segments per wave
    // the row-major segments of a side with short rows (users: a few batches each) come in chunks of ~4096 nonzeros:
    // a workgroup that lives for two 40-nonzero rows costs more to dispatch than to run (K = 50, 10^6 light users
    // in chunks of 8: user pass 2.27 -> 2.54 ms)
    uint32_t CHc = CH;
    if (c1 > c0) {
      int64_t cold_nnz = (int64_t)nnz;
      if (c1 < nseg) {
        Seg first_tiled;
        HIPBRK(h, hipMemcpy(&first_tiled, segs + c1, sizeof(Seg), hipMemcpyDeviceToHost));
        cold_nnz = first_tiled.start;
      }
      const uint64_t avg = std::max<uint64_t>((uint64_t)cold_nnz / (c1 - c0), 1);
      CHc = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(4096ull * wg / 256 / avg, CH), 512);
    }
    auto chunk_of = [&](const std::pair<uint32_t, uint32_t> &rg, uint32_t ch, uint32_t chc) { return (rg.first >= c0 && rg.second <= c1) ? chc : ch; };
    for (;; CH *= 2, CHc *= 2) {
      size_t worst = 0;
      for (int x = 0; x < 8; ++x) {
        size_t c = 0;
        for (auto &rg : q[x]) { const uint32_t ch = chunk_of(rg, CH, CHc); c += (rg.second - rg.first + ch - 1) / ch; }
        worst = std::max(worst, c);
      }
      if (worst * 8 <= max_wgs || CH >= (1u << 30)) break;
    }
    std::vector<uint2> qc[8];
    size_t longest = 0;
    for (int x = 0; x < 8; ++x) {
      for (auto &rg : q[x]) {
        const uint32_t ch = chunk_of(rg, CH, CHc);
        for (uint32_t p0 = rg.first; p0 < rg.second; p0 += std::min(ch, rg.second - p0)) qc[x].push_back(make_uint2(p0, p0 + std::min(ch, rg.second - p0)));
      }
      longest = std::max(longest, qc[x].size());
    }
    if (longest == 0 || longest * 8 > max_wgs) break;
    std::vector<uint2> chunks(longest * 8, make_uint2(0u, 0u));
    for (int x = 0; x < 8; ++x) for (size_t j = 0; j < qc[x].size(); ++j) chunks[j * 8 + x] = qc[x][j];
    if ((rc = dalloc(h, &chunks_dev, chunks.size()))) break;
    HIPBRK(h, hipMemcpyAsync(chunks_dev, chunks.data(), chunks.size() * sizeof(uint2), hipMemcpyHostToDevice, h->stream));
    HIPBRK(h, hipStreamSynchronize(h->stream));

    // ---- swap the side's work list
    dfree(s.segs); dfree(s.longrows); dfree(s.grouprows); dfree(s.hugerows); dfree(s.partial); dfree(s.partial2);
    s.segs = segs; s.nseg = nseg; segs = nullptr;
    s.longrows = longs; s.nlong = nlong; s.nlong_wave = nlong_wave; longs = nullptr;
    s.grouprows = groups; s.ngroup = ngroup; groups = nullptr;
    s.hugerows = huges; s.nhuge = nhuge; huges = nullptr;
    s.partial = partial; s.npartial = npartial; partial = nullptr;
    s.partial2 = partial2; s.npartial2 = ngroup; partial2 = nullptr;
    s.p_idx = keep_idx; s.p_val = keep_val; keep_idx = nullptr; keep_val = nullptr;
    s.chunks = chunks_dev; s.nchunk_blocks = (uint32_t)chunks.size(); chunks_dev = nullptr;
    s.tiles = tiles; s.tile_rows = T; s.tiled_nnz = heavy_nnz; s.light_below = light_below; s.chunk_segs = CH;
    (void)tiled_segs;
    done = true;
  } while (0);
  dfree(tilemap); dfree(first_seg); dfree(seg_row); dfree(seg_key); dfree(iota); dfree(cnt); dfree(segptr);
  for (int k = 0; k < 5; ++k) dfree(pl[k]);
  for (int k = 0; k < 2; ++k) { dfree(b.K[k]); dfree(b.U[k]); dfree(b.X[k]); dfree(b.V[k]); dfree(sb.K[k]); dfree(sb.U[k]); }
  dfree(key0); dfree(row0); dfree(segs); dfree(longs); dfree(huges); dfree(groups); dfree(partial); dfree(partial2);
  dfree(chunks_dev); dfree(keep_idx); dfree(keep_val);
  if (rc == HPF_ERR_OOM && !done) {       // tiling is optional: without the room for it the side stays row-major -- and says so
    (void)hipGetLastError();
    h->err.clear();
    h->notes |= (&s == &h->it ? 8u : 4u);
    rc = HPF_OK;
  }
  return rc;
}

// the tiled lists of both sides (after device_side_work); a tile is 4 MiB of the rows the pass gathers
int build_work_lists_tiled(hpf_handle *h, uint64_t nnz)
{
  const uint32_t n = h->u.rows, m = h->it.rows;
  int rc;
  const size_t rowb = h->wl != WL_PLAIN ? (size_t)h->pk.row_bytes : (size_t)h->ld * (h->w32 ? 4 : 8);
  h->notes &= ~15u;
  if ((rc = build_tiled_side(h, h->u, h->rowptr_dev, m, nnz, rowb))) return rc;     // the user pass gathers item rows
  return build_tiled_side(h, h->it, h->colptr_dev, n, nnz, rowb);
}

// Item-major view of the ratings, built in HBM: h->u.idx / h->u.val (CSR order)
// and h->rowptr_dev are in place; fills h->colptr_dev, h->it.idx, h->it.val.
// Stable LSD radix sort on the item id => users ascending inside an item.
int build_csc_device(hpf_handle *h, uint32_t n, uint32_t m, uint64_t nnz)
{
  int rc;
  uint32_t *bad = nullptr;
  uint32_t *kbuf[2] = {nullptr, nullptr}, *ut = nullptr; uint8_t *vt = nullptr; uint64_t *counts = nullptr;
  unsigned char *pool = nullptr;
  dfree(h->colptr_dev); h->colptr_dev = nullptr;
  dfree(h->it.idx); dfree(h->it.val); h->it.idx = nullptr; h->it.val = nullptr;
  do {
    if ((rc = dalloc(h, &h->colptr_dev, (size_t)m + 1))) break;          // zeroed: the answer for nnz == 0
    if ((rc = dalloc(h, &h->it.idx, (size_t)nnz))) break;
    if (h->u.val && (rc = dalloc(h, &h->it.val, (size_t)nnz))) break;
    if (nnz == 0) break;
    if ((rc = dalloc(h, &bad, 1))) break;
    uint32_t bits = 0;
    while (bits < 32 && ((uint64_t)1 << bits) < (uint64_t)m) ++bits;
    const uint32_t P = std::max<uint32_t>(1, (bits + RADIX_BITS - 1) / RADIX_BITS);
    const uint64_t ntiles = (nnz + RADIX_TILE - 1) / RADIX_TILE;
    const uint32_t nblk = (uint32_t)((ntiles + 3) / 4);
    // one allocation for every temporary of the sort (mapping ~100 GB piecewise was most of
    // whole C5's upload time): [counts | keys A | keys B | users | ratings], 256-byte aligned parts
    {
      auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
      const size_t b_counts = up((size_t)ntiles * RADIX_DIGITS * 8), b_k = up((size_t)nnz * 4);
      const size_t b_k1 = P >= 2 ? b_k : 0, b_ut = P >= 2 ? b_k : 0, b_vt = (P >= 2 && h->u.val) ? up((size_t)nnz) : 0;
      if ((rc = dalloc(h, &pool, b_counts + b_k + b_k1 + b_ut + b_vt))) break;
      char *q = (char *)pool;
      counts = (uint64_t *)q; q += b_counts;
      kbuf[0] = (uint32_t *)q; q += b_k;
      if (b_k1) { kbuf[1] = (uint32_t *)q; q += b_k1; }
      if (b_ut) { ut = (uint32_t *)q; q += b_ut; }
      if (b_vt) { vt = (uint8_t *)q; q += b_vt; }
    }
    for (uint32_t p = 0; p < P && !rc; ++p) {
      RadixArgs a;
      a.keys_in = p == 0 ? h->u.idx : kbuf[(p - 1) & 1];
      const bool out_final = ((P - 1 - p) & 1u) == 0;      // the last pass lands in it.idx / it.val
      a.users_in = p == 0 ? nullptr : (out_final ? ut : h->it.idx);
      a.vals_in = !h->u.val ? nullptr : p == 0 ? h->u.val : (out_final ? vt : h->it.val);
      a.rowptr = h->rowptr_dev; a.n_rows = n;
      a.extra_in = nullptr; a.extra_out = nullptr;
      a.keys_out = kbuf[p & 1];
      a.users_out = out_final ? h->it.idx : ut;
      a.vals_out = !h->u.val ? nullptr : (out_final ? h->it.val : vt);
      a.offsets = counts; a.nnz = nnz; a.ntiles = ntiles; a.shift = p * RADIX_BITS;
      hipLaunchKernelGGL(radix_count_kernel, dim3(nblk), dim3(256), 0, h->stream, a.keys_in, nnz, a.shift, ntiles, counts,
                         m, bad);
      if ((rc = check_launch(h, "radix_count_kernel"))) break;
      if (p == 0) {                                        // every item id was range-checked on the way
        uint32_t hb = 0;
        hipError_t e = hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { h->err = std::string("radix_count_kernel: ") + hipGetErrorString(e); rc = HPF_ERR_HIP; break; }
        if (hb) { h->err = "item index out of range"; rc = HPF_ERR_INVALID; break; }
      }
      if ((rc = device_scan<uint64_t>(h, counts, ntiles * RADIX_DIGITS, counts, false))) break;
      hipLaunchKernelGGL(radix_scatter_kernel, dim3(nblk), dim3(256), 0, h->stream, a);
      rc = check_launch(h, "radix_scatter_kernel");
    }
    if (rc) break;
    hipLaunchKernelGGL(colptr_from_sorted_kernel, dim3(grid_for(nnz)), dim3(256), 0, h->stream, kbuf[(P - 1) & 1], nnz, m,
                       h->colptr_dev);
    if ((rc = check_launch(h, "colptr_from_sorted_kernel"))) break;
    if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "CSC build failed on the device"; rc = HPF_ERR_HIP; }
  } while (0);
  dfree(bad); dfree(pool);
  return rc;
}

// host digamma for the xi/eta Elog export (same series as the device one)
double host_digamma(double x)
{
  double acc = 0.0;
  while (x < 10.0) { acc -= 1.0 / x; x += 1.0; }
  const double xi = 1.0 / x, x2 = xi * xi;
  double s = x2 * (1.0 / 12 - x2 * (1.0 / 120 - x2 * (1.0 / 252 - x2 * (1.0 / 240 -
             x2 * (1.0 / 132 - x2 * (691.0 / 32760 - x2 * (1.0 / 12)))))));
  return acc + std::log(x) - 0.5 * xi - s;
}

// shape (prior added, in place) and E from the raw sums and the rate the last
// sweep used; needed by every consumer outside the hot loop
int refresh_es(hpf_handle *h, Side &s)
{
  if (!s.es_stale || !s.rows) { s.es_stale = false; return HPF_OK; }
  { int rc0 = recover_if_flushed(h); if (rc0) return rc0; }       // a repeat of the last sweep needs S as the pass left it
  const size_t ne = (size_t)s.rows * h->ld;
  const uint32_t blocks = (uint32_t)std::min<size_t>((ne + 255) / 256, 8192);
  hipLaunchKernelGGL(materialize_es_kernel, dim3(blocks), dim3(256), 0, h->stream, s.S, s.E, s.prior_used,
                     s.colsum_used, s.rows, h->ld, h->K, s.bias_col, s.bias_rate_add, h->cfg.s_prior,
                     h->cfg.r_prior, h->cfg.hier);
  int rc = check_launch(h, "materialize_es_kernel");
  if (!rc) s.es_stale = false;
  return rc;
}

// rebuild Elog from shape and the rate the last sweep used (export only)
int refresh_elog(hpf_handle *h, Side &s)
{
  { int rc0 = refresh_es(h, s); if (rc0) return rc0; }
  if (!s.l_stale || !s.rows) { s.l_stale = false; return HPF_OK; }
  const size_t ne = (size_t)s.rows * h->ld;
  const uint32_t blocks = (uint32_t)std::min<size_t>((ne + 255) / 256, 8192);
  hipLaunchKernelGGL(elog_kernel, dim3(blocks), dim3(256), 0, h->stream, s.S, s.prior_used,
                     s.colsum_used, s.L, s.rows, h->ld, h->K, s.bias_col, s.bias_rate_add,
                     h->cfg.r_prior, h->cfg.hier);
  int rc = check_launch(h, "elog_kernel");
  if (!rc) s.l_stale = false;
  return rc;
}

int prepare_derived(hpf_handle *h)
{
  if (!h->derived_dirty) return HPF_OK;
  if (!(h->u.have_L && h->it.have_L && h->it.have_E)) {
    h->err = "state not initialised: set THETA_ELOG, BETA_ELOG and BETA_E before iterating";
    return HPF_ERR_STATE;
  }
  if (h->cfg.hier && !(h->u.have_prior && h->it.have_prior)) {
    h->err = "state not initialised: set XI_E and ETA_E (-hier)";
    return HPF_ERR_STATE;
  }
  Side *sides[2] = {&h->u, &h->it};
  for (int attempt = 0; attempt < 2; ++attempt) {
    bool derived = false;
    for (Side *s : sides) {
      if (!s->rows || !s->w_dirty) continue;
      s->w_dirty = false; s->w_from_sweep = false;
      derived = true;
      const uint32_t blocks = std::min<uint32_t>((s->rows + 3) / 4, 4096);
      hipLaunchKernelGGL(derive_w_kernel, dim3(blocks), dim3(256), 0, h->stream, s->L, s->W,
                         h->wl != WL_PLAIN ? (uint32_t)h->wl : (uint32_t)h->w32,
                         s->rows, h->ld, h->K, s->bias_col, s->junk_col, h->pk, h->flags);
    }
    if (!derived || h->wl != WL_P59 || h->capturing) break;
    // an Elog spread above 88 inside a row: p59 cannot hold the state -- plain rows then, derived again (recover_flush)
    uint32_t f[2] = {0, 0};
    HIPCHK(h, hipMemcpyAsync(f, h->flags, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!(f[0] & 2u)) break;
    { int rc0 = recover_flush(h, f[0], f[1]); if (rc0) return rc0; }
  }
  // c[k] = sum_i E[beta_ik]: consumed by the first user sweep
  if (h->sums_dirty) {
    Side &s = h->it;
    { int rc0 = refresh_es(h, s); if (rc0) return rc0; }
    const uint32_t nb = s.sweep_blocks;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, h->stream, s.E, s.rows,
                       h->ld, h->K, s.colsum_part);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(h->ld), dim3(256), 0, h->stream,
                       s.colsum_part, nb, h->ld, s.colsum, h->flags);
  }
  if (h->jacobi && h->sums_dirty) {      // sum_u E[theta] of the start state: the first item rate uses it
    h->start_sums_done = false;           // several ranks: this rank's part only, until hpf_start_sums hands it to the exchange
    Side &s = h->u;
    if (!s.have_E) { h->err = "state not initialised: -novb needs THETA_E (the first item rate is built from it)"; return HPF_ERR_STATE; }
    { int rc0 = refresh_es(h, s); if (rc0) return rc0; }
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(s.sweep_blocks), dim3(256), 0, h->stream, s.E, s.rows,
                       h->ld, h->K, s.colsum_part);
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(h->ld), dim3(256), 0, h->stream,
                       s.colsum_part, s.sweep_blocks, h->ld, s.colsum, h->flags);
  }
  int rc = check_launch(h, "prepare_derived");
  if (rc) return rc;
  h->derived_dirty = false;
  h->sums_dirty = false;
  return HPF_OK;
}

// rows of W in 16-byte pieces?
bool rows_in_pieces(const hpf_handle *h) { return h->wl != WL_PLAIN; }
int run_phi(hpf_handle *h, Side &own, Side &oth, hipEvent_t after_kernel)
{
  const int side = &own == &h->it ? 1 : 0;
  hipStream_t st = h->stream;
  PhiArgs a;
  a.segs = own.segs; a.nseg = own.nseg; a.idx = own.pass_idx(); a.val = own.pass_val();
  a.W_own = own.W; a.W_oth = oth.W; a.S_own = own.S; a.partial = own.partial; a.flags = h->flags;
  a.chunks = own.chunks; a.ld = h->ld;
  if (a.nseg) {
    const uint32_t wpb = rows_in_pieces(h) ? h->wg_of(own.chunks != nullptr) / 64 : 4;      // waves per workgroup
    const uint32_t blocks = own.chunks ? own.nchunk_blocks : std::min<uint32_t>((a.nseg + wpb - 1) / wpb, h->phi_blocks * (4 / wpb));
    const PhiLaunch pl = {blocks, wpb * 64, st};
    const bool ok = rows_in_pieces(h) ? launch_phi_packed(h->wl, h->phiG, h->phiR, side, a, pl)
                                      : launch_phi(h->w32, h->phiG, h->phiR, h->phiV, side, a, blocks, st);
    if (!ok) { h->err = "no phi kernel for this configuration"; return HPF_ERR_UNSUPPORTED; }
  } else if (side == 1) {
    // the item-major pass opens an iteration (phi_pass_skips keeps the books of the fallback protocol): an empty one still does
    PhiArgs e = a; e.segs = nullptr; e.nseg = 0; e.chunks = nullptr;
    hipLaunchKernelGGL(phi_open_kernel, dim3(1), dim3(64), 0, st, e);
  }
  // the event separates the phi kernel from the combine that follows it
  if (!h->capturing) HIPCHK(h, hipEventRecord(after_kernel, st));
  auto combine = [&](const LongRow *rows, uint32_t nrows, const double *src, double *dst) {
    hipLaunchKernelGGL(combine_partials_kernel, dim3(std::min<uint32_t>((nrows + 3) / 4, 16384)), dim3(256), 0, st,
                       rows, nrows, src, dst, h->ld, h->flags);
  };
  if (own.ngroup) combine(own.grouprows, own.ngroup, own.partial, own.partial2);     // level 1 of the very long rows: partial -> partial2
  if (own.nlong_wave) combine(own.longrows, own.nlong_wave, own.partial, own.S);
  if (own.nlong > own.nlong_wave)          // a tiled side's heavy rows (a partial per tile they meet): a workgroup each
    hipLaunchKernelGGL(combine_partials_wg_kernel, dim3(std::min<uint32_t>(own.nlong - own.nlong_wave, 65536)), dim3(256), (size_t)4 * h->ld * 8,
                       st, own.longrows + own.nlong_wave, own.nlong - own.nlong_wave, own.partial, own.S, h->ld, h->flags);
  if (own.nhuge) combine(own.hugerows, own.nhuge, own.partial2, own.S);              // level 2: partial2 -> S
  return check_launch(h, "phi pass");
}

// how the sweep of this handle writes W, and the arguments that do not change from launch to launch
void sweep_args(hpf_handle *h, Side &s, SweepArgs &a)
{
  a.S = s.S; a.W = s.W; a.w32 = h->w32;
  a.pk = h->pk; a.flags = h->flags;
  a.prior_E = s.prior_E; a.prior_rate = s.prior_rate;
  a.psi_prior_shape = host_digamma(h->cfg.s_prior + (double)h->K * h->cfg.s_prior);
  a.colsum_part = s.colsum_part; a.colsum_used = s.colsum_used;
  a.rows = s.rows; a.ld = h->ld; a.K = h->K;
  a.bias_col = s.bias_col; a.junk_col = s.junk_col; a.bias_rate_add = s.bias_rate_add;
  a.s_prior = h->cfg.s_prior; a.r_prior = h->cfg.r_prior; a.hier = h->cfg.hier;
}

int run_sweep(hpf_handle *h, Side &s, const double *colsum_oth, double *colsum_out)
{
  hipStream_t st = h->stream;
  SweepArgs a;
  sweep_args(h, s, a);
  a.colsum_oth = colsum_oth;                 // the kernel also copies it to colsum_used: what the rate was built from (export of *_rate.tsv)
  s.l_stale = true; s.es_stale = true; s.w_from_sweep = true;
  if (!launch_sweep(h->sw_mode, h->swG, h->swR, a, s.sweep_blocks, st)) {
    h->err = "no sweep kernel for this configuration"; return HPF_ERR_UNSUPPORTED;
  }
  if (h->cfg.hier && s.rows)                // xi / eta: E and Elog from the rate the sweep just wrote
    hipLaunchKernelGGL(prior_update_kernel, dim3(std::min<uint32_t>((s.rows + 255) / 256, 4096)), dim3(256), 0, st,
                       s.prior_E, s.prior_used, s.prior_rate, s.prior_elog, s.prior_elog_used, s.rows,
                       h->cfg.s_prior + (double)h->K * h->cfg.s_prior, a.psi_prior_shape, h->flags);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(h->ld), dim3(256), 0, st,
                     s.colsum_part, s.sweep_blocks, h->ld, colsum_out, h->flags);
  return check_launch(h, "row sweep");
}

// Step A is two independent gather passes over the same W_theta / W_beta: the
// item-major one (beta shape sums, straight into the exchange buffer) runs
// first so that, with several ranks, its all-reduce can overlap the whole
// user-side half of the iteration.
// events: 0 start | 1 phi_item kernel done | 2 its combine done |
//         3 phi_user kernel done | 4 its combine done | 5 user sweep | 6 item sweep
//         7 start of the replicated half (after whatever exchange the stream waited for)
bool want_graph(const hpf_handle *h);
void drop_graph(hpf_handle *h);
int sweep_users(hpf_handle *h);
int iterate_global(hpf_handle *h);

// the three pieces of a sharded iteration, each captured once (the kernels' arguments stay fixed until the CSR, the rows' layout
// or the exchange buffer change: drop_graph there).  Piece 1 carries the -novb copy of the old column sums like the eager path.
int build_split_graphs(hpf_handle *h)
{
  for (int k = 0; k < 3; ++k) {
    hipGraph_t g = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    h->capturing = true;
    int rc = k == 0 ? run_phi(h, h->it, h->u, nullptr) : k == 1 ? run_phi(h, h->u, h->it, nullptr) : iterate_global(h);
    if (!rc && k == 1) rc = sweep_users(h);
    h->capturing = false;
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (!rc && e != hipSuccess) { h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); rc = HPF_ERR_HIP; }
    if (!rc) {
      e = hipGraphInstantiate(&h->split_exec[k], g, nullptr, nullptr, 0);
      if (e != hipSuccess) { h->split_exec[k] = nullptr; h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); rc = HPF_ERR_HIP; }
    }
    if (g) (void)hipGraphDestroy(g);
    if (rc) { drop_graph(h); return rc; }
  }
  return HPF_OK;
}
// OPT-IN (HPF_GRAPH=1 under HPF_EXPERIMENTAL), never chosen by the library: measured on MI355X (profiles/r06/experiments.md 4)
// the three replays buy nothing where the pieces are long -- an eighth of C4, 2.50 ms eager against 2.51 -- and LOSE where they
// are short -- C1 through the pieces: 0.101 ms eager, 0.127 replayed: a replay costs the host 10-16 us, three of them more than
// the eleven eager launches they stand for, which the host issues ahead of the device anyway.  (One graph for a whole
// iteration -- one rank -- does pay: 0.085 ms.)
bool split_graph_on(const hpf_handle *h) { return !h->in_recovery && h->graph_mode == 1; }

// allow_graph: the caller goes on with hpf_iterate_local_users and hpf_iterate_global -- the cut the graphs are captured along
int phi_items(hpf_handle *h, bool allow_graph = false)
{
  int rc;
  if (h->capturing) return run_phi(h, h->it, h->u, nullptr);
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  // several ranks: a pass that skipped would leave this rank's sums out of the all-reduce, so the host looks at the flag
  // before every iteration (one stream synchronisation) and repairs the rows first; one rank lets the passes skip and
  // catches up at its next synchronisation point (recover_flush)
  if (h->cfg.n_ranks > 1 && h->wl == WL_P59 && (rc = recover_if_flushed(h))) return rc;
  // (sums_dirty, not derived_dirty: a rank that only has to write its W again after a fall-back from the packed rows keeps
  // the all-reduced sums it holds and must not enter a collective the other ranks never issue -- ADVICE r4)
  if (h->jacobi && h->cfg.n_ranks > 1 && (h->sums_dirty || !h->start_sums_done)) {
    if (h->comm) { if ((rc = hpf_start_sums(h))) return rc; }
    else {
      h->err = "-novb on several ranks: call hpf_start_sums and sum-all-reduce the last ld doubles of the exchange buffer before the first iteration";
      return HPF_ERR_STATE;
    }
  }
  if ((rc = prepare_derived(h))) return rc;
  h->tail_partial = false;
  h->ev = h->evr[h->ev_count % hpf_handle::RING];
  h->ring_graphed[h->ev_count % hpf_handle::RING] = false;
  const bool replay = allow_graph && split_graph_on(h);
  if (replay && !h->split_exec[0] && (rc = build_split_graphs(h))) return rc;
  HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
  if (replay) {
    // one launch: pass and combines; the per-kernel events coincide (phi_item_ms then holds the whole item half)
    HIPCHK(h, hipGraphLaunch(h->split_exec[0], h->stream));
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    h->phase = 4;                                            // 4: the pieces of this iteration are graph replays
  } else {
    if ((rc = run_phi(h, h->it, h->u, h->ev[1]))) return rc;   // step A, beta shape sums
    h->phase = 1;
  }
  HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
  return HPF_OK;
}

int phi_users(hpf_handle *h)
{
  int rc;
  if (h->capturing) return run_phi(h, h->u, h->it, nullptr);
  if (h->phase != 1) { h->err = "call order: items pass, users pass, user sweep, iterate_global"; return HPF_ERR_STATE; }
  if ((rc = run_phi(h, h->u, h->it, h->ev[3]))) return rc;   // step A, theta shape sums
  HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
  h->phase = 2;
  return HPF_OK;
}

// steps B (+D user, E): theta rate uses c = sum_i E[beta]; emits
// d = sum_u E[theta] into the tail of the exchange buffer.  Does not touch the
// item part of the exchange buffer, so its all-reduce may already be running.
int sweep_users(hpf_handle *h)
{
  int rc;
  if (!h->capturing && h->phase != 2) { h->err = "call order: items pass, users pass, user sweep, iterate_global"; return HPF_ERR_STATE; }
  if (h->jacobi)        // keep what _theta.sum_rows() still returns before _theta.swap() (hgaprec.cc:1281-1282)
    HIPCHK(h, hipMemcpyAsync(h->u_colsum_prev, h->u.colsum, (size_t)h->ld * 8, hipMemcpyDeviceToDevice, h->stream));
  if ((rc = run_sweep(h, h->u, h->it.colsum, h->u.colsum))) return rc;
  if (h->capturing) return HPF_OK;
  HIPCHK(h, hipEventRecord(h->ev[5], h->stream));
  h->phase = 3;
  return HPF_OK;
}

int iterate_local_phi(hpf_handle *h)
{
  int rc;
  if ((rc = phi_items(h))) return rc;
  return phi_users(h);
}

int iterate_local_users(hpf_handle *h)
{
  int rc;
  if (h->phase == 4) {                                       // the user half as one graph replay
    HIPCHK(h, hipGraphLaunch(h->split_exec[1], h->stream));
    for (int j = 3; j <= 5; ++j) HIPCHK(h, hipEventRecord(h->ev[j], h->stream));
    h->u.l_stale = h->u.es_stale = true; h->u.w_from_sweep = true;
    h->phase = 5;
    return HPF_OK;
  }
  if ((rc = phi_users(h))) return rc;
  return sweep_users(h);
}

int iterate_local(hpf_handle *h)
{
  int rc;
  if ((rc = iterate_local_phi(h))) return rc;
  return sweep_users(h);
}

int iterate_global(hpf_handle *h)
{
  int rc;
  // steps C (+D item, F): beta rate uses d (all-reduced when n_ranks > 1)
  if (!h->capturing && h->phase != 3 && h->phase != 5) { h->err = "call order: items pass, users pass, user sweep, iterate_global"; return HPF_ERR_STATE; }
  if (!h->capturing) HIPCHK(h, hipEventRecord(h->ev[7], h->stream));
  if (!h->capturing && h->phase == 5) {
    HIPCHK(h, hipGraphLaunch(h->split_exec[2], h->stream));
    h->it.l_stale = h->it.es_stale = true; h->it.w_from_sweep = true;
  } else if ((rc = run_sweep(h, h->it, h->jacobi ? h->u_colsum_prev : h->u.colsum, h->it.colsum))) return rc;
  if (h->capturing) return HPF_OK;
  HIPCHK(h, hipEventRecord(h->ev[6], h->stream));
  h->phase = 0;
  h->ev_count++;
  h->iterations++;
  h->iters_counted++;
  return HPF_OK;
}

void drop_graph(hpf_handle *h);

// p59 rows -> plain doubles in the same shape (codec_f64): W reallocated, the work lists cut again
// for the new row size -- exactly what a handle created with w_storage = 3 holds
int set_rows_f64(hpf_handle *h)
{
  int rc;
  drop_graph(h);
  h->wl = WL_F64; h->pk = h->pks; h->phiR = (int)h->pks.L;
  h->sw_mode = SW_F64;
  Side *sides[2] = {&h->u, &h->it};
  for (Side *s : sides) {
    dfree(s->W); s->W = nullptr;
    unsigned char *w = nullptr;
    if ((rc = dalloc(h, &w, w_bytes(h, s->rows)))) return rc;
    s->W = w;
  }
  if (h->have_csr) {
    if ((rc = device_side_work(h, h->u, h->rowptr_dev, h->u.rows))) return rc;
    if ((rc = device_side_work(h, h->it, h->colptr_dev, h->it.rows))) return rc;
    if ((rc = build_work_lists_tiled(h, h->nnz))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

// Flag bit 1: an element of W fell below 2^-126 of its row maximum (an Elog spread above 88 inside a row) and the p59
// rows cannot hold it.  The layout is lossless by contract, and the library holds everything it takes to go on without
// it: the rows become plain doubles, W is written again -- by a W-ONLY repeat of the side's last sweep (same S, the prior
// and the column sums that sweep used: the same arithmetic, so the same bits a w_storage = 3 handle has) or by derive_w
// where W came from a handed-in Elog -- and the iterations the passes skipped in the meantime are run.
// fl0 / begun: the device's flag word and its count of iterations begun; the stream is synchronised.
int recover_flush(hpf_handle *h, uint32_t fl0, uint32_t begun)
{
  if (h->wl != WL_P59) { h->err = "internal: an entry of W was reported flushed, but the rows are not packed"; return HPF_ERR_STATE; }
  int rc;
  h->in_recovery = true;
  const uint64_t redo = h->iters_counted > (uint64_t)begun ? h->iters_counted - (uint64_t)begun : 0;   // several ranks: none (the host looks every iteration)
  do {
    if ((rc = set_rows_f64(h))) break;
    const uint32_t clear[2] = {fl0 & 1u, 0u};
    if (hipMemcpyAsync(h->flags, clear, 8, hipMemcpyHostToDevice, h->stream) != hipSuccess) { h->err = "recover_flush: cannot reset the flags"; rc = HPF_ERR_HIP; break; }
    h->iters_counted = 0;
    Side *sides[2] = {&h->u, &h->it};
    for (Side *s : sides) {
      if (!s->rows) continue;
      if (!s->w_from_sweep || s->w_dirty) { s->w_dirty = true; h->derived_dirty = true; continue; }     // from Elog: prepare_derived
      SweepArgs a;
      sweep_args(h, *s, a);
      a.prior_E = s->prior_used; a.colsum_oth = s->colsum_used; a.colsum_used = nullptr;             // what that sweep read
      if (!s->es_stale) a.s_prior = 0.0;      // an export has turned S into the shape (prior added, in place) since: the same sum, not taken twice
      if (!launch_sweep(h->sw_mode, h->swG, h->swR, a, s->sweep_blocks, h->stream)) { h->err = "no sweep kernel for this configuration"; rc = HPF_ERR_UNSUPPORTED; break; }
    }
    if (rc || (rc = check_launch(h, "repeat of the sweeps in plain rows"))) break;
    h->fallbacks++;
    if (redo) {
      if (h->cfg.n_ranks != 1) { h->err = "internal: iterations were skipped on a rank of several"; rc = HPF_ERR_STATE; break; }
      h->iterations -= (uint32_t)std::min<uint64_t>(redo, h->iterations);
      for (uint64_t t = 0; t < redo && !rc; ++t) { if (!(rc = iterate_local(h))) rc = iterate_global(h); }
    }
    if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "recover_flush: stream error"; rc = HPF_ERR_HIP; }
  } while (0);
  h->in_recovery = false;
  return rc;
}

void drop_graph(hpf_handle *h)
{
  if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
  for (int k = 0; k < 3; ++k) if (h->split_exec[k]) { (void)hipGraphExecDestroy(h->split_exec[k]); h->split_exec[k] = nullptr; }
}

bool want_graph(const hpf_handle *h)
{
  if (h->graph_mode >= 0) return h->graph_mode == 1;
  return h->nnz <= h->graph_nnz_max;
}

// one iteration (both phi passes, both sweeps) captured once; every kernel
// argument is a pointer or scalar that stays fixed until the CSR or the
// exchange buffer is replaced (drop_graph there)
int build_graph(hpf_handle *h)
{
  hipGraph_t g = nullptr;
  HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
  h->capturing = true;
  int rc = iterate_local(h);
  if (!rc) rc = iterate_global(h);
  h->capturing = false;
  hipError_t e = hipStreamEndCapture(h->stream, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) { h->err = std::string("hipStreamEndCapture: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
  e = hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) { h->graph_exec = nullptr; h->err = std::string("hipGraphInstantiate: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
  return HPF_OK;
}

int iterate_graph(hpf_handle *h, int n_iters)
{
  int rc;
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  if ((rc = prepare_derived(h))) return rc;
  if (!h->graph_exec && (rc = build_graph(h))) return rc;
  for (int t = 0; t < n_iters; ++t) {
    const uint32_t slot = h->ev_count % hpf_handle::RING;
    h->ev = h->evr[slot];
    h->ring_graphed[slot] = true;
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    HIPCHK(h, hipGraphLaunch(h->graph_exec, h->stream));
    HIPCHK(h, hipEventRecord(h->ev[6], h->stream));
    h->u.l_stale = h->u.es_stale = h->it.l_stale = h->it.es_stale = true;
    h->u.w_from_sweep = h->it.w_from_sweep = true;
    h->ev_count++;
    h->iterations++;
    h->iters_counted++;
  }
  return HPF_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
extern "C" {

int hpf_abi_version(void) { return HPF_ABI_VERSION; }

const char *hpf_strerror(int st)
{
  switch (st) {
    case HPF_OK: return "ok";
    case HPF_ERR_INVALID: return "invalid argument or call order";
    case HPF_ERR_NO_DEVICE: return "no usable HIP device";
    case HPF_ERR_OOM: return "out of memory";
    case HPF_ERR_HIP: return "HIP runtime error";
    case HPF_ERR_UNSUPPORTED: return "unsupported configuration";
    case HPF_ERR_STATE: return "model state not initialised";
  }
  return "unknown status";
}

const char *hpf_last_error(const hpf_handle *h) { return h ? h->err.c_str() : "null handle"; }

int hpf_create(const hpf_config *cfg, hpf_handle **out)
{
  if (!cfg || !out) return HPF_ERR_INVALID;
  *out = nullptr;
  if (cfg->struct_size != sizeof(hpf_config)) return HPF_ERR_INVALID;
  if (cfg->K == 0 || cfg->n_items == 0) return HPF_ERR_INVALID;
  if (cfg->n_ranks == 0 || cfg->rank >= cfg->n_ranks) return HPF_ERR_INVALID;
  const bool jacobi = cfg->novb && cfg->bias && !cfg->hier;       // the only place the reference reads Env::vb
  const uint32_t C = cfg->K + (cfg->bias ? 2u : 0u);
  if (C > HPF_MAX_COLUMNS) return HPF_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return HPF_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return HPF_ERR_NO_DEVICE;
  if (hipSetDevice(cfg->device) != hipSuccess) return HPF_ERR_NO_DEVICE;

  hpf_handle *h = new (std::nothrow) hpf_handle();
  if (!h) return HPF_ERR_OOM;
  h->cfg = *cfg;
  if (h->cfg.n_users_total == 0) h->cfg.n_users_total = cfg->n_users;
  if (h->cfg.s_prior <= 0) h->cfg.s_prior = 0.3;
  if (h->cfg.r_prior <= 0) h->cfg.r_prior = 0.3;
  h->w32 = cfg->w_storage == 1;          // only ever chosen by the caller's hpf_config
  if (cfg->w_storage > 3 || cfg->tiling > 1) { delete h; return HPF_ERR_INVALID; }
  h->K = cfg->K; h->C = C;
  // Tuning knobs are read from the environment ONLY under HPF_EXPERIMENTAL=1 (tests, tools/):
  // a stray variable must not change the layout or the summation order of a production run.
  const bool experimental = [] { const char *e = getenv("HPF_EXPERIMENTAL"); return e && atoi(e) == 1; }();
  auto knob = [&](const char *name) -> const char * { return experimental ? getenv(name) : nullptr; };
  if (const char *e = knob("HPF_H2D")) h->xfer_mode = !strcmp(e, "plain") ? 0 : !strcmp(e, "register") ? 2 : 1;
  if (const char *e = knob("HPF_H2D_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) h->xfer_threads = (unsigned)v; }

  auto fail = [&](int rc) { hpf_destroy(h); return rc; };
  if (cfg->stream) h->stream = (hipStream_t)cfg->stream;
  else {
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return fail(HPF_ERR_HIP);
    h->own_stream = true;
  }
  for (uint32_t r = 0; r < hpf_handle::RING; ++r)
    for (int e = 0; e < 8; ++e)
      if (hipEventCreate(&h->evr[r][e]) != hipSuccess) return fail(HPF_ERR_HIP);

  // Kernel shape and row stride.  The phi pass gives G lanes to a nonzero, each with R loads
  // of V elements; the row stride of every device matrix is EXACTLY ld = G*R*V elements
  // (round 3): the live columns K + 2*bias are padded with zero columns, so the kernels carry
  // no column test (K=100: 800-byte rows become 896 = seven whole 128-byte lines, the
  // number of lines a gather of an unaligned 800-byte row touched anyway).
  {
    // 16-byte loads when they pad no worse than 8-byte ones (measured: C2
    // phi_user 4.36 ms vs 4.52 ms).  Elements per load: doubles 1|2, floats 2|4.
    bool phi_cfg_forced = false;
    const int Vs = h->w32 ? 2 : 1, Vl = 2 * Vs;
    int g1 = 0, r1 = 0, g2 = 0, r2 = 0;
    long w1 = 1L << 40, w2 = 1L << 40;
    const bool ok1 = choose_cfg(C, Vs, &g1, &r1, 8, &w1), ok2 = choose_cfg(C, Vl, &g2, &r2, 8, &w2);
    if (!ok1 && !ok2) return fail(HPF_ERR_UNSUPPORTED);
    if (!ok1) w1 = 1L << 40;
    if (!ok2) w2 = 1L << 40;
    if (w2 <= w1) { h->phiG = g2; h->phiR = r2; h->phiV = Vl; }
    else { h->phiG = g1; h->phiR = r1; h->phiV = Vs; }
    if (h->w32) {
      // f32 rows are half as long: measured at C2 (K=100) the passes want 256
      // contiguous bytes per nonzero-group -- (G,R,V) = (16,2,4): 3.7 + 3.4 ms
      // against 5.0 + 3.8 ms for (8,4,4) and 9.8 + 5.3 ms for the least-padding (4,7,4)
      const int g = C > 512 ? 32 : C > 32 ? 16 : C > 16 ? 8 : 4;
      h->phiG = g; h->phiV = 4; h->phiR = (int)((C + (uint32_t)(4 * g) - 1) / (uint32_t)(4 * g));
    }
    if (const char *e = knob("HPF_PHI_CFG")) {           // "G,R,V"
      int g = 0, r = 0, v = 0;
      if (sscanf(e, "%d,%d,%d", &g, &r, &v) == 3 && (h->w32 ? (v == 2 || v == 4) : (v == 1 || v == 2)) && r >= 1 && r <= 8 &&
          (g == 4 || g == 8 || g == 16 || g == 32 || g == 64) && (uint32_t)(g * r * v) >= C && g * r * v <= 2048) {
        h->phiG = g; h->phiR = r; h->phiV = v;
        phi_cfg_forced = true;                             // an explicit plain shape: no packing
      }
    }
    h->ld = (uint32_t)(h->phiG * h->phiR * h->phiV);
    {
      // Packed rows: G lanes x L 16-byte pieces, E elements per lane (p59: 128L/59, f48: 8L/3); fewest
      // row bytes G*L*16 with G*E >= C.  w_storage 0 takes the lossless p59 packing when it saves
      // at least one 128-byte line per row against the plain fp64 row (K = 100: 6 instead of 7,
      // K = 50: 3 instead of 4 -- measured on a C5 shard: 65.6 -> 51.7 ms); 2 asks for f48; 3 keeps
      // plain rows.  HPF_W_PACK=1 (experimental) packs whenever a shape exists.
      const bool force_pack = knob("HPF_W_PACK") && atoi(knob("HPF_W_PACK")) == 1;
      // w_storage 3 asks for plain doubles: where the default would pack, the rows keep the packed SHAPE (lanes per nonzero,
      // columns, row stride of S) and hold plain doubles in its pieces (WL_F64) -- the layout a packed handle falls back to
      // when a state turns up that p59 cannot hold, so that the two give the same bits
      const int want = cfg->w_storage == 2 ? WL_F48 : (cfg->w_storage == 0 || cfg->w_storage == 3 || force_pack) && cfg->w_storage != 1 ? WL_P59 : WL_PLAIN;
      if (want != WL_PLAIN && !(phi_cfg_forced && want == WL_P59)) {
        long bestb = -1; int bg = 0, bl = 0;
        const int Gq[5] = {8, 16, 32, 64, 4};
        for (int g : Gq)
          for (int l = 1; l <= 8; ++l) {
            const int e = want == WL_F48 ? (8 * l) / 3 : (128 * l) / 59;
            if ((uint32_t)(g * e) < C) continue;
            const long b = (long)g * l * 16;
            if (bestb < 0 || b < bestb) { bestb = b; bg = g; bl = l; }
            break;                                                                   // larger l only adds bytes
          }
        const long plain_lines = ((long)h->ld * 8 + 127) / 128, packed_lines = (bestb + 127) / 128;
        bool take = bestb > 0 && (want == WL_F48 || force_pack || packed_lines < plain_lines);
        if (want == WL_F48 && bestb < 0) return fail(HPF_ERR_UNSUPPORTED);
        // the sweep must have a shape for the packed stride too
        if (take) {
          const int e = want == WL_F48 ? (8 * bl) / 3 : (128 * bl) / 59;
          const uint32_t pld = (uint32_t)(bg * e);
          bool fits = false;
          if (want == WL_F48) for (int g : {64, 32, 16, 8, 4}) { const uint32_t r = (pld + (uint32_t)g - 1) / (uint32_t)g; fits |= r >= 1 && r <= (g == 64 ? 16u : 8u); }
          else fits = bg <= 32 || e <= 16;      // p59: groups of twice the pass's lanes (<= 9 slots), or 64 lanes with a slot per element
          if (!fits && want == WL_F48) return fail(HPF_ERR_UNSUPPORTED);
          if (!fits) take = false;                         // e.g. 961..1024 columns: 1088 packed columns have none; rows stay plain
        }
        if (take) {
          const int e = want == WL_F48 ? (8 * bl) / 3 : (128 * bl) / 59;
          h->wl = want;
          h->phiG = bg; h->phiR = bl; h->phiV = 0;
          h->pk.G = (uint32_t)bg; h->pk.L = (uint32_t)bl; h->pk.E = (uint32_t)e; h->pk.row_bytes = (uint32_t)(bg * bl) * 16u;
          h->pk.lgG = 0; while ((1u << h->pk.lgG) < (uint32_t)bg) ++h->pk.lgG;
          h->ld = (uint32_t)(bg * e);
          if (want == WL_P59) {
            const uint32_t ls = ((uint32_t)e + 1u) / 2u;
            h->pks = h->pk; h->pks.L = ls; h->pks.E = 2u * ls; h->pks.row_bytes = (uint32_t)bg * ls * 16u;
            if (cfg->w_storage == 3) { h->wl = WL_F64; h->pk = h->pks; h->phiR = (int)ls; }
          }
        }
      }
    }
    // The row sweep gives G' lanes to a row with R' columns each.  Plain rows: G'*R' == ld exactly.  p59 rows (and the plain
    // doubles in their shape): G' is TWICE the pass's lanes -- the two lanes that share a packed lane swap their halves and
    // build the row in registers (row_sweep_kernel, SW_REG_P59) -- or the pass's 64.  48-bit rows: G'*R' >= ld, built in LDS.
    h->swG = h->swR = 0;
    if (h->wl == WL_P59 || h->wl == WL_F64) {
      const uint32_t ep = h->ld / (uint32_t)h->phiG;                 // elements per lane of the p59 shape
      if (h->phiG <= 32) { h->swG = 2 * h->phiG; h->swR = (int)((ep + 1) / 2); h->sw_mode = h->wl == WL_P59 ? SW_REG_P59 : SW_F64; }
      else { h->swG = 64; h->swR = (int)ep; h->sw_mode = h->wl == WL_P59 ? SW_LDS_P59 : SW_F64; }
    } else {
      h->sw_mode = h->wl == WL_F48 ? SW_LDS_F48 : SW_PLAIN;
      const int Gs[5] = {64, 32, 16, 8, 4};
      int best = 1 << 30;
      for (int g : Gs) {
        if (h->wl == WL_PLAIN && h->ld % (uint32_t)g) continue;
        const int r = (int)((h->ld + (uint32_t)g - 1) / (uint32_t)g);
        if (r < 1 || r > (g == 64 ? 16 : 8)) continue;
        const int p = (r == 1 ? 3 : 0) + (r > 7 ? 2 : 0) + (g * 8 < 128 ? 1 : 0);   // same preferences as round 1's K sweep
        if (p < best || (p == best && g < h->swG)) { best = p; h->swG = g; h->swR = r; }
      }
      if (!h->swG) return fail(HPF_ERR_UNSUPPORTED);
      if (const char *e = knob("HPF_SWEEP_CFG")) {         // "G,R" with G*R == ld (plain rows)
        int g = 0, r = 0;
        if (h->wl == WL_PLAIN && sscanf(e, "%d,%d", &g, &r) == 2 && r >= 1 && r <= (g == 64 ? 16 : 8) && (g == 4 || g == 8 || g == 16 || g == 32 || g == 64) &&
            (uint32_t)(g * r) == h->ld) { h->swG = g; h->swR = r; }
      }
    }
  }
  if (const char *e = knob("HPF_SWEEP_BLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 65536) h->sweep_blocks_max = (uint32_t)v; }
  if (const char *e = knob("HPF_GRAPH")) h->graph_mode = atoi(e) != 0;
  if (const char *e = knob("HPF_SEG_MAX")) { int v = atoi(e); if (v >= 16) h->seg_max = (uint32_t)v; }
  if (const char *e = knob("HPF_HUGE_SLOTS")) { int v = atoi(e); if (v >= 2) { h->huge_slots = (uint32_t)v; h->group_slots = std::max<uint32_t>(2, std::min<uint32_t>(64, (uint32_t)v / 2)); } }
  if (const char *e = knob("HPF_TILE")) { int v = atoi(e); if (v >= 0 && v <= 2) h->tile_mode = v; }
  if (const char *e = knob("HPF_TILE_SIDES")) { int v = atoi(e); if (v >= 0 && v <= 3) h->tile_sides = v; }
  if (const char *e = knob("HPF_TILE_SPLIT_BELOW")) { int v = atoi(e); if (v >= 0) h->tile_split_below = (uint32_t)v; }
  if (const char *e = knob("HPF_TILE_ORDER")) { int v = atoi(e); if (v >= 0 && v <= 2) h->tile_order = v; }
  if (const char *e = knob("HPF_TILE_BYTES")) { long long v = atoll(e); if (v >= 1024) h->tile_bytes = (uint64_t)v; }
  if (const char *e = knob("HPF_TILE_CHUNK")) { int v = atoi(e); if (v >= 1) h->tile_chunk = (uint32_t)v; }       // default: two per wave of the workgroup
  if (const char *e = knob("HPF_TILE_RUN")) { int v = atoi(e); if (v >= 1) h->tile_min_run = (uint32_t)v; }
  if (const char *e = knob("HPF_TILE_SHARE")) { int v = atoi(e); if (v >= 0 && v <= 100) h->tile_min_share = v / 100.0; }
  if (const char *e = knob("HPF_PHI_BLOCKS")) { int v = atoi(e); if (v >= 1) h->phi_blocks = (uint32_t)v; }
  if (const char *e = knob("HPF_PHI_WG")) { int v = atoi(e); if (v == 64 || v == 128 || v == 256) h->phi_wg = (uint32_t)v; }

  const uint32_t n = cfg->n_users, m = cfg->n_items, ld = h->ld;
  h->u.rows = n; h->it.rows = m;
  if (cfg->bias) {
    h->u.bias_col = (int32_t)h->K;      h->u.junk_col = (int32_t)h->K + 1;
    h->it.bias_col = (int32_t)h->K + 1; h->it.junk_col = (int32_t)h->K;
    h->u.bias_rate_add = (double)m;                       // hgaprec.cc:1389 (0.3 + m)
    h->it.bias_rate_add = (double)h->cfg.n_users_total;   // hgaprec.cc:1393 (0.3 + n)
  }
  int rc;
  Side *sides[2] = {&h->u, &h->it};
  for (Side *s : sides) {
    const size_t ne = (size_t)s->rows * ld;
    const uint32_t gpb = 256u / (uint32_t)h->swG;         // groups (rows) per block
    s->sweep_blocks = std::max<uint32_t>(1, std::min<uint32_t>((s->rows + gpb - 1) / gpb, h->sweep_blocks_max));
    if (s == &h->u) { if ((rc = dalloc(h, &s->S, ne))) return fail(rc); }
    if ((rc = dalloc(h, &s->E, ne))) return fail(rc);
    if ((rc = dalloc(h, &s->L, ne))) return fail(rc);
    { unsigned char *w = nullptr; if ((rc = dalloc(h, &w, w_bytes(h, s->rows)))) return fail(rc); s->W = w; }
    if ((rc = dalloc(h, &s->prior_E, s->rows))) return fail(rc);
    if ((rc = dalloc(h, &s->prior_used, s->rows))) return fail(rc);
    if ((rc = dalloc(h, &s->prior_rate, s->rows))) return fail(rc);
    if ((rc = dalloc(h, &s->prior_elog, s->rows))) return fail(rc);
    if ((rc = dalloc(h, &s->prior_elog_used, s->rows))) return fail(rc);
    if ((rc = dalloc(h, &s->colsum_used, ld))) return fail(rc);
    if ((rc = dalloc(h, &s->colsum_part, (size_t)s->sweep_blocks * ld))) return fail(rc);
  }
  // exchange buffer: item S rows | sum_u E[theta]   (+ item colsum kept apart)
  h->exch_count = (size_t)m * ld + ld;
  if ((rc = dalloc(h, &h->exch, h->exch_count))) return fail(rc);
  h->it.S = h->exch; h->u.colsum = h->exch + (size_t)m * ld;
  if ((rc = dalloc(h, &h->it.colsum, ld))) return fail(rc);
  h->jacobi = jacobi;
  if ((rc = dalloc(h, &h->u_colsum_prev, ld))) return fail(rc);
  // log y! as HGAPRec::log_factorial does it (hgaprec.cc:1563-1570)
  {
    double lf[256]; lf[0] = std::log(1.0); lf[1] = lf[0];
    for (uint32_t y = 2; y < 256; ++y) lf[y] = lf[y - 1] + std::log((double)y);
    if ((rc = dalloc(h, &h->logfact, 256))) return fail(rc);
    if ((rc = dalloc(h, &h->flags, 4))) return fail(rc);
    if (hipMemcpyAsync(h->logfact, lf, sizeof lf, hipMemcpyHostToDevice, h->stream) != hipSuccess) return fail(HPF_ERR_HIP);
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(HPF_ERR_HIP);
  *out = h;
  return HPF_OK;
}

void hpf_destroy(hpf_handle *h)
{
  if (!h) return;
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
  drop_graph(h);
  if (h->comm_stream) {
    (void)hipStreamSynchronize(h->comm_stream);
    (void)hipEventDestroy(h->ev_ready); (void)hipEventDestroy(h->ev_reduced);
    (void)hipStreamDestroy(h->comm_stream);
  }
  double *ucol = h->u.colsum;  (void)ucol;      // lives inside exch
  h->u.colsum = nullptr;
  double *icol = h->it.colsum; h->it.colsum = nullptr;
  free_side(h->u, false);
  free_side(h->it, true);
  dfree(icol);
  if (!h->exch_external) dfree(h->exch);
  dfree(h->logfact); dfree(h->rowptr_dev); dfree(h->colptr_dev); dfree(h->flags); dfree(h->u_colsum_prev);
  for (int k = 0; k < HPF_HELDOUT_SLOTS; ++k) free_held(h->held[k]);
  for (int k = 0; k < 8; ++k) if (h->held_ev[k]) (void)hipEventDestroy(h->held_ev[k]);
  for (int k = 0; k < 2; ++k) {
    if (h->stage[k]) (void)hipHostFree(h->stage[k]);
    if (h->stage_ev[k]) (void)hipEventDestroy(h->stage_ev[k]);
  }
  for (uint32_t r = 0; r < hpf_handle::RING; ++r)
    for (int e = 0; e < 8; ++e) if (h->evr[r][e]) (void)hipEventDestroy(h->evr[r][e]);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int hpf_bind_exchange_buffer(hpf_handle *h, void *dev, size_t count)
{
  if (!h || !dev) return HPF_ERR_INVALID;
  if (count < h->exch_count || ((uintptr_t)dev & 15u)) { h->err = "exchange buffer too small or misaligned"; return HPF_ERR_INVALID; }
  if (h->have_csr || h->iterations) { h->err = "bind the exchange buffer before hpf_upload_csr"; return HPF_ERR_INVALID; }
  HIPCHK(h, hipMemsetAsync(dev, 0, h->exch_count * 8, h->stream));
  // keep any item shapes already handed in
  HIPCHK(h, hipMemcpyAsync(dev, h->exch, h->exch_count * 8, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (!h->exch_external) dfree(h->exch);
  h->exch = (double *)dev; h->exch_external = true;
  h->it.S = h->exch; h->u.colsum = h->exch + (size_t)h->it.rows * h->ld;
  return HPF_OK;
}

int hpf_comm_unique_id(void *id_out)
{
  if (!id_out) return HPF_ERR_INVALID;
  if (load_rccl()) return HPF_ERR_UNSUPPORTED;
  return g_rccl.GetUniqueId(id_out) == 0 ? HPF_OK : HPF_ERR_HIP;
}

int hpf_comm_init(hpf_handle *h, const void *id)
{
  if (!h || !id) return HPF_ERR_INVALID;
  if (h->comm) { h->err = "communicator already initialised"; return HPF_ERR_INVALID; }
  if (const char *e = load_rccl()) { h->err = e; return HPF_ERR_UNSUPPORTED; }
  IdByValue v; memcpy(v.internal, id, HPF_COMM_ID_BYTES);
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const int rc = g_rccl.CommInitRank(&h->comm, (int)h->cfg.n_ranks, v, (int)h->cfg.rank);
  if (rc != 0) { h->comm = nullptr; h->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc); return HPF_ERR_HIP; }
  return HPF_OK;
}

int hpf_allreduce_items_begin(hpf_handle *h)
{
  if (!h) return HPF_ERR_INVALID;
  if (!h->comm) { h->err = "hpf_comm_init has not been called"; return HPF_ERR_STATE; }
  if (h->phase != 1 && h->phase != 4) { h->err = "hpf_allreduce_items_begin follows hpf_iterate_local_items"; return HPF_ERR_STATE; }
  if (!h->comm_stream) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_reduced, hipEventDisableTiming));
  }
  HIPCHK(h, hipEventRecord(h->ev_ready, h->stream));                 // item sums are final here
  HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_ready, 0));
  const size_t items = (size_t)h->it.rows * h->ld;
  const int rc = g_rccl.AllReduce(h->exch, h->exch, items, 8, 0, h->comm, (void *)h->comm_stream);
  if (rc != 0) { h->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(rc); return HPF_ERR_HIP; }
  h->items_reduce_pending = true;
  return HPF_OK;
}

int hpf_allreduce_exchange(hpf_handle *h)
{
  if (!h) return HPF_ERR_INVALID;
  if (!h->comm) { h->err = "hpf_comm_init has not been called"; return HPF_ERR_STATE; }
  // ncclDouble = 8, ncclSum = 0 (rccl.h:448,467); in place
  if (!h->items_reduce_pending) {               // everything at once, on the stream the kernels use
    const int rc = g_rccl.AllReduce(h->exch, h->exch, h->exch_count, 8, 0, h->comm, (void *)h->stream);
    if (rc != 0) { h->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(rc); return HPF_ERR_HIP; }
    return check_comm_async(h);
  }
  // the item part is already on its way: add the [ld] tail (sum_u E[theta]), then
  // let the kernels' stream wait for both
  const size_t items = (size_t)h->it.rows * h->ld;
  HIPCHK(h, hipEventRecord(h->ev_ready, h->stream));
  HIPCHK(h, hipStreamWaitEvent(h->comm_stream, h->ev_ready, 0));
  const int rc = g_rccl.AllReduce(h->exch + items, h->exch + items, h->exch_count - items, 8, 0, h->comm,
                                  (void *)h->comm_stream);
  if (rc != 0) { h->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(rc); return HPF_ERR_HIP; }
  HIPCHK(h, hipEventRecord(h->ev_reduced, h->comm_stream));
  HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_reduced, 0));
  h->items_reduce_pending = false;
  return check_comm_async(h);
}

int hpf_start_sums(hpf_handle *h)
{
  if (!h) return HPF_ERR_INVALID;
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  int rc;
  if ((rc = prepare_derived(h))) return rc;
  if (!h->jacobi || h->cfg.n_ranks == 1) return HPF_OK;
  if (h->start_sums_done) return HPF_OK;
  if (h->comm) {                        // the library owns the exchange: the [ld] tail, in place, on the kernels' stream
    const size_t items = (size_t)h->it.rows * h->ld;
    const int rc2 = g_rccl.AllReduce(h->exch + items, h->exch + items, h->exch_count - items, 8, 0, h->comm, (void *)h->stream);
    if (rc2 != 0) { h->err = std::string("ncclAllReduce: ") + g_rccl.GetErrorString(rc2); return HPF_ERR_HIP; }
    if ((rc = check_comm_async(h))) return rc;
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));     // the caller's all-reduce may run on another stream
  h->start_sums_done = true;
  h->tail_partial = !h->comm;
  return HPF_OK;
}

int hpf_exchange_read(hpf_handle *h, double *host, size_t count)
{
  if (!h || !host || count != h->exch_count) return HPF_ERR_INVALID;
  HIPCHK(h, hipMemcpyAsync(host, h->exch, count * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

int hpf_exchange_write(hpf_handle *h, const double *host, size_t count)
{
  if (!h || !host || count != h->exch_count) return HPF_ERR_INVALID;
  HIPCHK(h, hipMemcpyAsync(h->exch, host, count * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

int hpf_exchange_buffer(hpf_handle *h, void **dev, size_t *count)
{
  if (!h || !dev || !count) return HPF_ERR_INVALID;
  *dev = h->exch; *count = h->exch_count;
  return HPF_OK;
}

// common tail of both uploads: h->u.idx / h->u.val / h->rowptr_dev are in place
static int finish_upload(hpf_handle *h, uint64_t nnz)
{
  const uint32_t n = h->u.rows, m = h->it.rows;
  int rc;
  if ((rc = build_csc_device(h, n, m, nnz))) return rc;
  int64_t last = 0;
  HIPCHK(h, hipMemcpyAsync(&last, h->colptr_dev + m, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if ((uint64_t)last != nnz) { h->err = "internal: item-major view lost nonzeros"; return HPF_ERR_HIP; }
  if ((rc = device_side_work(h, h->u, h->rowptr_dev, n))) return rc;
  if ((rc = device_side_work(h, h->it, h->colptr_dev, m))) return rc;
  if ((rc = build_work_lists_tiled(h, nnz))) return rc;
  HIPCHK(h, hipMemsetAsync(h->flags, 0, 8, h->stream));
  h->iters_counted = 0;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->nnz = nnz; h->have_csr = true;
  return HPF_OK;
}

static int check_rowptr(hpf_handle *h, const int64_t *rowptr, uint32_t n)
{
  if (rowptr[0] != 0) { h->err = "rowptr[0] must be 0"; return HPF_ERR_INVALID; }
  for (uint32_t r = 0; r < n; ++r)
    if (rowptr[r + 1] < rowptr[r]) { h->err = "rowptr not monotone"; return HPF_ERR_INVALID; }
  return HPF_OK;
}

static int alloc_user_nonzeros(hpf_handle *h, uint64_t nnz, bool with_val)
{
  int rc;
  drop_graph(h);
  h->have_csr = false;
  dfree(h->u.idx); dfree(h->u.val); h->u.idx = nullptr; h->u.val = nullptr;
  dfree(h->rowptr_dev); h->rowptr_dev = nullptr;
  if ((rc = dalloc(h, &h->u.idx, (size_t)nnz))) return rc;
  if (with_val && (rc = dalloc(h, &h->u.val, (size_t)nnz))) return rc;
  return dalloc(h, &h->rowptr_dev, (size_t)h->u.rows + 1);
}

int hpf_upload_csr(hpf_handle *h, const int64_t *rowptr, const uint32_t *col, const uint8_t *val)
{
  if (!h || !rowptr) return HPF_ERR_INVALID;
  const uint32_t n = h->u.rows;
  int rc;
  if ((rc = check_rowptr(h, rowptr, n))) return rc;
  const uint64_t nnz = (uint64_t)rowptr[n];
  if (nnz && !col) return HPF_ERR_INVALID;
  if ((rc = alloc_user_nonzeros(h, nnz, val != nullptr))) return rc;
  if ((rc = h2d(h, h->u.idx, col, nnz * 4))) return rc;
  if (val && (rc = h2d(h, h->u.val, val, nnz))) return rc;
  if ((rc = h2d(h, h->rowptr_dev, rowptr, ((size_t)n + 1) * 8))) return rc;
  return finish_upload(h, nnz);
}

int hpf_upload_csr_device(hpf_handle *h, const int64_t *d_rowptr, const uint32_t *d_col, const uint8_t *d_val)
{
  if (!h || !d_rowptr) return HPF_ERR_INVALID;
  const uint32_t n = h->u.rows;
  int rc;
  // only the two ends of the row pointers come to the host; monotonicity is checked on the
  // device while the work lists are cut (seg_plan_kernel)
  int64_t ends[2] = {0, 0};
  HIPCHK(h, hipMemcpyAsync(&ends[0], d_rowptr, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&ends[1], d_rowptr + n, 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (ends[0] != 0) { h->err = "rowptr[0] must be 0"; return HPF_ERR_INVALID; }
  if (ends[1] < 0) { h->err = "rowptr not monotone"; return HPF_ERR_INVALID; }
  const uint64_t nnz = (uint64_t)ends[1];
  if (nnz && !d_col) return HPF_ERR_INVALID;
  if ((rc = alloc_user_nonzeros(h, nnz, d_val != nullptr))) return rc;
  if (nnz) HIPCHK(h, hipMemcpyAsync(h->u.idx, d_col, nnz * 4, hipMemcpyDeviceToDevice, h->stream));
  if (nnz && d_val) HIPCHK(h, hipMemcpyAsync(h->u.val, d_val, nnz, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipMemcpyAsync(h->rowptr_dev, d_rowptr, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, h->stream));
  // the sort's first pass bisects the row pointers: they must be monotone BEFORE it runs
  {
    uint32_t *bad = nullptr; uint32_t hb = 0;
    if ((rc = dalloc(h, &bad, 1))) return rc;
    hipLaunchKernelGGL(rowptr_check_kernel, dim3(grid_for(n)), dim3(256), 0, h->stream, h->rowptr_dev, n, bad);
    hipError_t e = hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dfree(bad);
    if (e != hipSuccess) { h->err = std::string("rowptr check: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
    if (hb) { h->err = "rowptr not monotone"; return HPF_ERR_INVALID; }
  }
  return finish_upload(h, nnz);
}

int hpf_get_csc(hpf_handle *h, int64_t *colptr, uint32_t *users, uint8_t *vals)
{
  if (!h) return HPF_ERR_INVALID;
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  int rc;
  if (colptr && (rc = d2h(h, colptr, h->colptr_dev, ((size_t)h->it.rows + 1) * 8))) return rc;
  if (users && (rc = d2h(h, users, h->it.idx, (size_t)h->nnz * 4))) return rc;
  if (vals) {
    if (!h->it.val) { h->err = "no ratings were uploaded (-binary-data)"; return HPF_ERR_INVALID; }
    if ((rc = d2h(h, vals, h->it.val, (size_t)h->nnz))) return rc;
  }
  return HPF_OK;
}

static int state_dims(const hpf_handle *h, int which, Side **side, uint32_t *rows, uint32_t *cols,
                      int *col0, int *kind)
{
  const int obj = which / 4; *kind = which % 4;
  hpf_handle *hh = const_cast<hpf_handle *>(h);
  switch (obj) {
    case 0: *side = &hh->u;  *rows = h->u.rows;  *cols = h->K; *col0 = 0; return 0;
    case 1: *side = &hh->it; *rows = h->it.rows; *cols = h->K; *col0 = 0; return 0;
    case 2: if (!h->cfg.hier) return -1; *side = &hh->u;  *rows = h->u.rows;  *cols = 1; *col0 = -1; return 0;
    case 3: if (!h->cfg.hier) return -1; *side = &hh->it; *rows = h->it.rows; *cols = 1; *col0 = -1; return 0;
    case 4: if (!h->cfg.bias) return -1; *side = &hh->u;  *rows = h->u.rows;  *cols = 1; *col0 = h->u.bias_col; return 0;
    case 5: if (!h->cfg.bias) return -1; *side = &hh->it; *rows = h->it.rows; *cols = 1; *col0 = h->it.bias_col; return 0;
  }
  return -1;
}

// on_device: `host` is a device pointer (hpf_set_state_device)
static int set_state_impl(hpf_handle *h, hpf_state which, const double *host, size_t count, bool on_device)
{
  if (!h || !host || which < 0 || which >= HPF_NUM_STATE) return HPF_ERR_INVALID;
  Side *s; uint32_t rows, cols; int col0, kind;
  if (state_dims(h, which, &s, &rows, &cols, &col0, &kind)) { h->err = "state not part of this model"; return HPF_ERR_INVALID; }
  const int obj = which / 4;
  int rc;
  auto put = [&](void *dst, const void *src, size_t bytes) -> int {
    if (!on_device) return h2d(h, dst, src, bytes);
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  };
  auto put2d = [&](double *dev, uint32_t c0) -> int {
    return on_device ? copy_in_dev(h, dev, h->ld, c0, host, rows, cols) : copy_in(h, dev, h->ld, c0, host, rows, cols);
  };
  // a new state supersedes whatever the kernels flagged about the old one -- but iterations a flushed entry made the
  // passes skip are part of the old one: they are run first (recover_flush)
  if ((rc = recover_if_flushed(h))) return rc;
  HIPCHK(h, hipMemsetAsync(h->flags, 0, 4, h->stream));
  if (obj == 2 || obj == 3) {                 // xi / eta vectors
    if (count != rows) return HPF_ERR_INVALID;
    double **dst = nullptr;
    switch (kind) {
      case 0: dst = &s->prior_shape_set; break;
      case 1: dst = &s->prior_rate; break;
      case 2: dst = &s->prior_E; s->have_prior = true; break;
      default: dst = &s->prior_elog; break;
    }
    if (!*dst && (rc = dalloc(h, dst, rows))) return rc;
    return put(*dst, host, (size_t)rows * 8);
  }
  const bool gr_rate = (obj <= 1 && kind == 1 && !h->cfg.hier);
  if (kind == 1) {                            // rate: kept only for export before iteration 0
    const size_t want = gr_rate ? h->K : (size_t)rows * cols;
    if (count != want) return HPF_ERR_INVALID;
    if (obj >= 4) return HPF_OK;              // bias rate is the constant 0.3 + m / 0.3 + n
    dfree(s->rate_set); s->rate_set = nullptr;
    if ((rc = dalloc(h, &s->rate_set, want))) return rc;
    s->rate_set_count = want;
    return put(s->rate_set, host, want * 8);
  }
  if (count != (size_t)rows * cols) return HPF_ERR_INVALID;
  if ((rc = refresh_es(h, *s))) return rc;     // the rest of S / E must be current before patching
  double *dev = kind == 0 ? s->S : kind == 2 ? s->E : s->L;
  if (kind == 3 && s->l_stale) {
    // the rest of L predates the last sweep: rebuild it before patching in the new part
    if ((rc = refresh_elog(h, *s))) return rc;
  }
  if ((rc = put2d(dev, (uint32_t)col0))) return rc;
  if (obj <= 1) {
    if (kind == 2) s->have_E = true;
    if (kind == 3) s->have_L = true;
  }
  if (kind == 3) s->w_dirty = true;          // Elog of theta/beta or of a bias column
  if (kind != 0) h->derived_dirty = h->sums_dirty = true;
  return HPF_OK;
}

int hpf_set_state(hpf_handle *h, hpf_state which, const double *host, size_t count)
{
  return set_state_impl(h, which, host, count, false);
}
int hpf_set_state_device(hpf_handle *h, hpf_state which, const double *dev, size_t count)
{
  return set_state_impl(h, which, dev, count, true);
}

static int get_state_impl(hpf_handle *h, hpf_state which, double *host, size_t count, bool on_device)
{
  if (!h || !host || which < 0 || which >= HPF_NUM_STATE) return HPF_ERR_INVALID;
  Side *s; uint32_t rows, cols; int col0, kind;
  if (state_dims(h, which, &s, &rows, &cols, &col0, &kind)) { h->err = "state not part of this model"; return HPF_ERR_INVALID; }
  const int obj = which / 4;
  const double s0 = h->cfg.s_prior, r0 = h->cfg.r_prior;
  auto get = [&](void *dst, const void *src, size_t bytes) -> int {
    if (!on_device) return d2h(h, dst, src, bytes);
    HIPCHK(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  };
  auto fill = [&](double v, size_t cnt) -> int {         // a constant vector
    std::vector<double> tmp(cnt, v);
    if (!on_device) { memcpy(host, tmp.data(), cnt * 8); return HPF_OK; }
    HIPCHK(h, hipMemcpyAsync(host, tmp.data(), cnt * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return HPF_OK;
  };
  if (obj == 2 || obj == 3) {
    if (count != rows) return HPF_ERR_INVALID;
    if (kind == 2 || kind == 1) return get(host, kind == 2 ? s->prior_E : s->prior_rate, (size_t)rows * 8);
    if (kind == 3) return get(host, s->prior_elog, (size_t)rows * 8);   // set by the host, then maintained by the sweep
    if (h->iterations == 0) {
      const double *src = s->prior_shape_set;
      if (!src) { h->err = "state was never set"; return HPF_ERR_STATE; }
      return get(host, src, (size_t)rows * 8);
    }
    // after a sweep: shape = s0 + K*s0 (gpbase.hh:877-882)
    return fill(s0 + (double)h->K * s0, rows);
  }
  if (kind == 1) {                            // rate
    const bool gr = (obj <= 1 && !h->cfg.hier);
    const size_t want = gr ? h->K : (size_t)rows * cols;
    if (count != want) return HPF_ERR_INVALID;
    if (obj >= 4) {                           // gpbase.hh:225-231 via hgaprec.cc:1389,1393
      if (h->iterations == 0) { h->err = "bias rate is defined after the first sweep"; return HPF_ERR_STATE; }
      return fill(r0 + s->bias_rate_add, rows);
    }
    if (h->iterations == 0) {
      if (!s->rate_set || s->rate_set_count != want) { h->err = "state was never set"; return HPF_ERR_STATE; }
      return get(host, s->rate_set, want * 8);
    }
    if (gr) {                                 // r_k = 0.3 + colsum_k  (gpbase.hh:558-562)
      std::vector<double> cs(h->ld);
      HIPCHK(h, hipMemcpyAsync(cs.data(), s->colsum_used, (size_t)h->ld * 8, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (uint32_t k = 0; k < h->K; ++k) cs[k] += r0;
      if (!on_device) { memcpy(host, cs.data(), (size_t)h->K * 8); return HPF_OK; }
      HIPCHK(h, hipMemcpyAsync(host, cs.data(), (size_t)h->K * 8, hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      return HPF_OK;
    }
    double *tmp = on_device ? host : nullptr; int rc = HPF_OK;
    if (!on_device && (rc = dalloc(h, &tmp, want))) return rc;
    hipLaunchKernelGGL(build_rate_kernel, dim3(1024), dim3(256), 0, h->stream, s->prior_used,
                       s->colsum_used, rows, h->K, tmp);
    rc = check_launch(h, "build_rate");
    if (!rc) {
      if (on_device) { if (hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "build_rate failed"; rc = HPF_ERR_HIP; } }
      else rc = d2h(h, host, tmp, want * 8);
    }
    if (!on_device) dfree(tmp);
    return rc;
  }
  if (count != (size_t)rows * cols) return HPF_ERR_INVALID;
  { int rc = check_flags(h); if (rc) return rc; }
  { int rc = kind == 3 ? refresh_elog(h, *s) : refresh_es(h, *s); if (rc) return rc; }
  const double *dev = kind == 0 ? s->S : kind == 2 ? s->E : s->L;
  return on_device ? copy_out_dev(h, dev, h->ld, (uint32_t)col0, host, rows, cols)
                   : copy_out(h, dev, h->ld, (uint32_t)col0, host, rows, cols);
}

int hpf_get_state(hpf_handle *h, hpf_state which, double *host, size_t count)
{
  return get_state_impl(h, which, host, count, false);
}
int hpf_get_state_device(hpf_handle *h, hpf_state which, double *dev, size_t count)
{
  return get_state_impl(h, which, dev, count, true);
}

// ---- snapshot: the loop's device state, verbatim -------------------------------
namespace {
constexpr char SNAP_MAGIC[9] = "HPFSNAP4";
struct SnapHeader {
  char magic[8];                        // "HPFSNAP4" (round 6: the order of a p59 lane's dwords changed, and the header carries sums_dirty / start_sums_done)
  uint32_t n_users, n_items, K, ld, hier, bias, w32, iterations;
  // what else the state is only meaningful with: the priors, the job's shape and
  // the ratings it was fitted to -- a snapshot of another data set of the same dimensions,
  // or of another rank's shard, must not load
  uint32_t n_users_total, rank, n_ranks, novb;
  uint64_t nnz;
  double s_prior, r_prior;
  uint32_t side_flags[2];               // bit 0 have_E, 1 have_L, 2 have_prior, 3 w_dirty, 4 l_stale, 5 es_stale,
                                        // 6 rate_set present, 7 prior_shape_set present
  uint32_t derived_dirty;
  uint32_t more_flags;                  // bit 0 sums_dirty, bit 1 start_sums_done (a handle whose W alone was pending must not
                                        // come back as one whose start sums are pending -- on several ranks it would enter a collective alone)
  uint64_t rate_set_count[2];
  uint64_t total_bytes;
};
struct SnapSection { void *ptr; size_t bytes; };

// every array the iteration reads or the exports are rebuilt from, in a fixed order
void snapshot_sections(hpf_handle *h, const uint64_t rate_cnt[2], const uint32_t flags[2], std::vector<SnapSection> *out)
{
  Side *sides[2] = {&h->u, &h->it};
  for (int k = 0; k < 2; ++k) {
    Side &s = *sides[k];
    const size_t mat = (size_t)s.rows * h->ld * 8, vec = (size_t)s.rows * 8, row = (size_t)h->ld * 8;
    for (void *p : {(void *)s.S, (void *)s.E, (void *)s.L}) out->push_back({p, mat});
    out->push_back({s.W, w_bytes(h, s.rows)});
    for (double *p : {s.prior_E, s.prior_used, s.prior_rate, s.prior_elog, s.prior_elog_used}) out->push_back({p, vec});
    out->push_back({s.colsum, row}); out->push_back({s.colsum_used, row});
    if (flags[k] & 64u) out->push_back({s.rate_set, (size_t)rate_cnt[k] * 8});
    if (flags[k] & 128u) out->push_back({s.prior_shape_set, vec});
  }
}
uint32_t side_flag_word(const Side &s)
{
  return (s.have_E ? 1u : 0u) | (s.have_L ? 2u : 0u) | (s.have_prior ? 4u : 0u) | (s.w_dirty ? 8u : 0u) |
         (s.l_stale ? 16u : 0u) | (s.es_stale ? 32u : 0u) | (s.rate_set ? 64u : 0u) | (s.prior_shape_set ? 128u : 0u) |
         (s.w_from_sweep ? 256u : 0u);
}
void fill_snap_header(hpf_handle *h, SnapHeader *hd)
{
  memset(hd, 0, sizeof *hd);
  memcpy(hd->magic, SNAP_MAGIC, 8);
  hd->n_users = h->u.rows; hd->n_items = h->it.rows; hd->K = h->K; hd->ld = h->ld;
  hd->hier = h->cfg.hier; hd->bias = h->cfg.bias; hd->w32 = h->cfg.w_storage | ((uint32_t)h->wl << 8); hd->iterations = h->iterations;   // storage mode and the row layout in use
  hd->n_users_total = h->cfg.n_users_total; hd->rank = h->cfg.rank; hd->n_ranks = h->cfg.n_ranks;
  hd->novb = h->jacobi ? 1u : 0u; hd->nnz = h->have_csr ? h->nnz : 0;
  hd->s_prior = h->cfg.s_prior; hd->r_prior = h->cfg.r_prior;
  hd->side_flags[0] = side_flag_word(h->u); hd->side_flags[1] = side_flag_word(h->it);
  hd->derived_dirty = h->derived_dirty;
  hd->more_flags = (h->sums_dirty ? 1u : 0u) | (h->start_sums_done ? 2u : 0u);
  hd->rate_set_count[0] = h->u.rate_set ? h->u.rate_set_count : 0;
  hd->rate_set_count[1] = h->it.rate_set ? h->it.rate_set_count : 0;
  std::vector<SnapSection> sec;
  snapshot_sections(h, hd->rate_set_count, hd->side_flags, &sec);
  size_t tot = sizeof *hd;
  for (const SnapSection &x : sec) tot += x.bytes;
  hd->total_bytes = tot;
}
}  // namespace

int hpf_snapshot_size(hpf_handle *h, size_t *bytes)
{
  if (!h || !bytes) return HPF_ERR_INVALID;
  SnapHeader hd; fill_snap_header(h, &hd);
  *bytes = (size_t)hd.total_bytes;
  return HPF_OK;
}

int hpf_snapshot_save(hpf_handle *h, void *host, size_t bytes)
{
  if (!h || !host) return HPF_ERR_INVALID;
  if (h->phase != 0) { h->err = "snapshot inside an iteration"; return HPF_ERR_STATE; }
  { int rc = check_flags(h); if (rc) return rc; }
  SnapHeader hd; fill_snap_header(h, &hd);
  if (bytes != hd.total_bytes) { h->err = "snapshot buffer size differs from hpf_snapshot_size"; return HPF_ERR_INVALID; }
  memcpy(host, &hd, sizeof hd);
  std::vector<SnapSection> sec;
  snapshot_sections(h, hd.rate_set_count, hd.side_flags, &sec);
  char *p = (char *)host + sizeof hd;
  for (const SnapSection &x : sec) {
    int rc = d2h(h, p, x.ptr, x.bytes);
    if (rc) return rc;
    p += x.bytes;
  }
  return HPF_OK;
}

int hpf_snapshot_load(hpf_handle *h, const void *host, size_t bytes)
{
  if (!h || !host || bytes < sizeof(SnapHeader)) return HPF_ERR_INVALID;
  if (h->phase != 0 || h->items_reduce_pending) { h->err = "snapshot load inside an iteration"; return HPF_ERR_STATE; }
  // ---- everything is validated before the handle is touched: a rejected blob leaves it as it was
  SnapHeader hd; memcpy(&hd, host, sizeof hd);
  // a snapshot taken after the rows fell back to plain doubles loads into a handle that still packs them: same job, the
  // handle follows (validated below like everything else, but the layout has to be known to size the sections)
  const bool follow = !memcmp(hd.magic, SNAP_MAGIC, 8) && h->wl == WL_P59 && hd.w32 == (h->cfg.w_storage | ((uint32_t)WL_F64 << 8)) &&
                      hd.n_users == h->u.rows && hd.n_items == h->it.rows && hd.K == h->K && hd.ld == h->ld && hd.total_bytes == bytes;
  if (follow) { int rc0 = recover_if_flushed(h); if (!rc0 && h->wl == WL_P59) rc0 = set_rows_f64(h); if (rc0) return rc0; h->fallbacks++; }
  if (memcmp(hd.magic, SNAP_MAGIC, 8) || hd.total_bytes != bytes || hd.n_users != h->u.rows || hd.n_items != h->it.rows ||
      hd.K != h->K || hd.ld != h->ld || hd.hier != h->cfg.hier || hd.bias != h->cfg.bias || hd.w32 != (h->cfg.w_storage | ((uint32_t)h->wl << 8))) {
    h->err = "not a snapshot of this model (shape, flags or storage differ)"; return HPF_ERR_INVALID;
  }
  if (hd.s_prior != h->cfg.s_prior || hd.r_prior != h->cfg.r_prior || hd.n_users_total != h->cfg.n_users_total ||
      hd.rank != h->cfg.rank || hd.n_ranks != h->cfg.n_ranks || hd.novb != (h->jacobi ? 1u : 0u)) {
    h->err = "not a snapshot of this job (priors, rank layout or update order differ)"; return HPF_ERR_INVALID;
  }
  if (hd.nnz != (h->have_csr ? h->nnz : 0)) {
    h->err = "the snapshot was taken on other ratings (nonzero count differs): upload the same CSR first"; return HPF_ERR_INVALID;
  }
  Side *sides[2] = {&h->u, &h->it};
  for (int k = 0; k < 2; ++k) {
    const size_t lim = (size_t)sides[k]->rows * h->K;
    const bool has = hd.side_flags[k] & 64u;
    if (has != (hd.rate_set_count[k] != 0) || hd.rate_set_count[k] > lim) { h->err = "damaged snapshot header"; return HPF_ERR_INVALID; }
  }
  {
    size_t tot = sizeof hd;
    for (int k = 0; k < 2; ++k) {
      const size_t mat = (size_t)sides[k]->rows * h->ld * 8, vec = (size_t)sides[k]->rows * 8, row = (size_t)h->ld * 8;
      tot += 3 * mat + w_bytes(h, sides[k]->rows) + 5 * vec + 2 * row;
      if (hd.side_flags[k] & 64u) tot += (size_t)hd.rate_set_count[k] * 8;
      if (hd.side_flags[k] & 128u) tot += vec;
    }
    if (tot != bytes) { h->err = "damaged snapshot header"; return HPF_ERR_INVALID; }
  }
  int rc;
  double *new_rate[2] = {nullptr, nullptr}, *new_pss[2] = {nullptr, nullptr};
  for (int k = 0; k < 2; ++k) {                           // optional arrays the snapshot carries
    Side &s = *sides[k];
    rc = HPF_OK;
    if ((hd.side_flags[k] & 64u)) rc = dalloc(h, &new_rate[k], (size_t)hd.rate_set_count[k]);
    if (!rc && (hd.side_flags[k] & 128u) && !s.prior_shape_set) rc = dalloc(h, &new_pss[k], s.rows);
    if (rc) { for (int j = 0; j < 2; ++j) { dfree(new_rate[j]); dfree(new_pss[j]); } return rc; }
  }
  for (int k = 0; k < 2; ++k) {
    Side &s = *sides[k];
    dfree(s.rate_set); s.rate_set = new_rate[k]; s.rate_set_count = (size_t)hd.rate_set_count[k];
    if (new_pss[k]) s.prior_shape_set = new_pss[k];
  }
  std::vector<SnapSection> sec;
  snapshot_sections(h, hd.rate_set_count, hd.side_flags, &sec);
  drop_graph(h);
  const char *p = (const char *)host + sizeof hd;
  for (const SnapSection &x : sec) {
    if ((rc = h2d(h, x.ptr, p, x.bytes))) return rc;
    p += x.bytes;
  }
  for (int k = 0; k < 2; ++k) {
    Side &s = *sides[k]; const uint32_t f = hd.side_flags[k];
    s.have_E = f & 1u; s.have_L = f & 2u; s.have_prior = f & 4u; s.w_dirty = f & 8u; s.l_stale = f & 16u; s.es_stale = f & 32u;
    s.w_from_sweep = (f & 256u) != 0;
  }
  h->derived_dirty = hd.derived_dirty != 0;
  h->sums_dirty = (hd.more_flags & 1u) != 0;
  h->start_sums_done = (hd.more_flags & 2u) != 0;      // the tail of the exchange buffer came with the snapshot
  h->iterations = hd.iterations;
  h->phase = 0;
  HIPCHK(h, hipMemsetAsync(h->flags, 0, 8, h->stream));
  h->iters_counted = 0;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

int hpf_iterate(hpf_handle *h, int n_iters)
{
  if (!h || n_iters < 0) return HPF_ERR_INVALID;
  if (h->cfg.n_ranks != 1 || h->comm) {
    // several ranks -- or one with a communicator of its own (a one-rank RCCL communicator: tests, bench.py --comm library on one
    // GPU) --: whole iterations only when the library owns the exchange
    if (!h->comm) { h->err = "hpf_iterate with n_ranks > 1 needs hpf_comm_init (or use iterate_local / your all-reduce / iterate_global)"; return HPF_ERR_INVALID; }
    for (int t = 0; t < n_iters; ++t) {
      int rc;
      if ((rc = phi_items(h, true))) return rc;
      if ((rc = hpf_allreduce_items_begin(h))) return rc;
      if ((rc = iterate_local_users(h))) return rc;
      if ((rc = hpf_allreduce_exchange(h))) return rc;
      if ((rc = iterate_global(h))) return rc;
    }
    return HPF_OK;
  }
  if (n_iters > 0 && want_graph(h)) return iterate_graph(h, n_iters);
  for (int t = 0; t < n_iters; ++t) {
    int rc;
    if ((rc = iterate_local(h))) return rc;
    if ((rc = iterate_global(h))) return rc;
  }
  return HPF_OK;
}

int hpf_iterate_local(hpf_handle *h) { return h ? iterate_local(h) : HPF_ERR_INVALID; }
int hpf_iterate_local_phi(hpf_handle *h) { return h ? iterate_local_phi(h) : HPF_ERR_INVALID; }
int hpf_iterate_local_sweep(hpf_handle *h) { return h ? sweep_users(h) : HPF_ERR_INVALID; }
int hpf_iterate_local_items(hpf_handle *h) { return h ? phi_items(h, true) : HPF_ERR_INVALID; }
int hpf_iterate_local_users(hpf_handle *h) { return h ? iterate_local_users(h) : HPF_ERR_INVALID; }
int hpf_iterate_global(hpf_handle *h) { return h ? iterate_global(h) : HPF_ERR_INVALID; }

namespace {
// per-pair log-likelihoods of cnt pairs whose indices lie on the device, into hout (host; page-locked for a bound set),
// summed serially in the order handed in: the map order of hgaprec.cc:1455-1465
int heldout_run(hpf_handle *h, const uint32_t *du, const uint32_t *di, const int32_t *dy, double *dout, double *hout, size_t cnt,
                double *sum_out, bool pinned)
{
  int rc;
  if ((rc = check_flags(h))) return rc;
  if ((rc = refresh_es(h, h->u)) || (rc = refresh_es(h, h->it))) return rc;
  LLArgs a;
  a.u = du; a.i = di; a.y = dy; a.cnt = cnt; a.Et = h->u.E; a.Eb = h->it.E;
  a.logfact = h->logfact; a.out = dout; a.ld = h->ld; a.K = h->K;
  a.ubias_col = h->cfg.bias ? h->u.bias_col : -1;
  a.ibias_col = h->cfg.bias ? h->it.bias_col : -1;
  a.binary = h->cfg.binary;
  const uint32_t blocks = (uint32_t)std::min<size_t>((cnt + 15) / 16, 4096);
  hipLaunchKernelGGL(heldout_ll_kernel, dim3(blocks), dim3(256), 0, h->stream, a);
  if ((rc = check_launch(h, "heldout_ll"))) return rc;
  // The sum is a chain of dependent adds in the order handed in -- that order is the contract -- and at 10^6 pairs it is what the
  // call costs (~0.8 ns per pair).  Into page-locked memory the values therefore arrive in pieces, and the host adds piece c
  // while piece c + 1 is on the wire: the DMA disappears behind the chain.  (Pageable memory: one copy, as before.)
  constexpr int PIECES = 8;
  const bool piped = pinned && cnt >= ((size_t)1 << 16);
  hipError_t e = hipSuccess;
  if (piped) {
    for (int c = 0; c < PIECES && e == hipSuccess; ++c)
      if (!h->held_ev[c]) e = hipEventCreateWithFlags(&h->held_ev[c], hipEventDisableTiming);
    const size_t per = (cnt + PIECES - 1) / PIECES;
    for (int c = 0; c < PIECES && e == hipSuccess; ++c) {
      const size_t p0 = std::min(cnt, (size_t)c * per), p1 = std::min(cnt, p0 + per);
      if (p1 > p0) e = hipMemcpyAsync(hout + p0, dout + p0, (p1 - p0) * 8, hipMemcpyDeviceToHost, h->stream);
      if (e == hipSuccess) e = hipEventRecord(h->held_ev[c], h->stream);
    }
    double s = .0;
    for (int c = 0; c < PIECES && e == hipSuccess; ++c) {
      e = hipEventSynchronize(h->held_ev[c]);
      const size_t p0 = std::min(cnt, (size_t)c * per), p1 = std::min(cnt, p0 + per);
      for (size_t p = p0; p < p1; ++p) s += hout[p];
    }
    if (e != hipSuccess) { (void)hipStreamSynchronize(h->stream); h->err = hipGetErrorString(e); return HPF_ERR_HIP; }
    *sum_out = s;
    return HPF_OK;
  }
  e = hipMemcpyAsync(hout, dout, cnt * 8, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = hipGetErrorString(e); return HPF_ERR_HIP; }
  double s = .0;
  for (size_t p = 0; p < cnt; ++p) s += hout[p];
  *sum_out = s;
  return HPF_OK;
}
}  // namespace

int hpf_heldout_ll(hpf_handle *h, const uint32_t *u, const uint32_t *i, const int32_t *y,
                   size_t cnt, double *sum_out, uint64_t *cnt_out)
{
  if (!h || !sum_out) return HPF_ERR_INVALID;
  *sum_out = 0.0; if (cnt_out) *cnt_out = cnt;
  if (cnt == 0) return HPF_OK;
  if (!u || !i || !y) return HPF_ERR_INVALID;
  if (!(h->u.have_E && h->it.have_E) && h->iterations == 0) { h->err = "E state not set"; return HPF_ERR_STATE; }
  for (size_t p = 0; p < cnt; ++p)
    if (u[p] >= h->u.rows || i[p] >= h->it.rows) { h->err = "held-out index out of range"; return HPF_ERR_INVALID; }
  uint32_t *du = nullptr, *di = nullptr; int32_t *dy = nullptr; double *dout = nullptr;
  int rc = HPF_OK;
  std::vector<double> out(cnt);
  do {
    if ((rc = dalloc(h, &du, cnt)) || (rc = dalloc(h, &di, cnt)) || (rc = dalloc(h, &dy, cnt)) ||
        (rc = dalloc(h, &dout, cnt))) break;
    hipError_t e = hipMemcpyAsync(du, u, cnt * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(di, i, cnt * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dy, y, cnt * 4, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = HPF_ERR_HIP; break; }
    rc = heldout_run(h, du, di, dy, dout, out.data(), cnt, sum_out, false);
  } while (0);
  dfree(du); dfree(di); dfree(dy); dfree(dout);
  return rc;
}

// A report step evaluates the SAME validation and test pairs every time (hgaprec.cc:1439-1470 walks the same two maps):
// bound once, a set is validated and uploaded once, and hpf_heldout_ll_bound is the kernel, one DMA of the per-pair
// values into a kept page-locked buffer and the ordered host sum -- the same sum, bit for bit, as hpf_heldout_ll's.
int hpf_heldout_bind(hpf_handle *h, int slot, const uint32_t *u, const uint32_t *i, const int32_t *y, size_t cnt)
{
  if (!h || slot < 0 || slot >= HPF_HELDOUT_SLOTS) return HPF_ERR_INVALID;
  if (cnt && (!u || !i || !y)) return HPF_ERR_INVALID;
  for (size_t p = 0; p < cnt; ++p)
    if (u[p] >= h->u.rows || i[p] >= h->it.rows) { h->err = "held-out index out of range"; return HPF_ERR_INVALID; }
  hpf_handle::HeldSet &hs = h->held[slot];
  HIPCHK(h, hipStreamSynchronize(h->stream));
  free_held(hs);
  hs.cnt = cnt; hs.bound = true;
  if (cnt == 0) return HPF_OK;
  int rc;
  if ((rc = dalloc(h, &hs.du, cnt)) || (rc = dalloc(h, &hs.di, cnt)) || (rc = dalloc(h, &hs.dy, cnt)) || (rc = dalloc(h, &hs.dout, cnt))) { free_held(hs); return rc; }
  if (hipHostMalloc((void **)&hs.hout, cnt * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); free_held(hs); h->err = "hpf_heldout_bind: no page-locked memory"; return HPF_ERR_OOM; }
  if ((rc = h2d(h, hs.du, u, cnt * 4)) || (rc = h2d(h, hs.di, i, cnt * 4)) || (rc = h2d(h, hs.dy, y, cnt * 4))) { free_held(hs); return rc; }
  return HPF_OK;
}

int hpf_heldout_ll_bound(hpf_handle *h, int slot, double *sum_out, uint64_t *cnt_out)
{
  if (!h || !sum_out || slot < 0 || slot >= HPF_HELDOUT_SLOTS) return HPF_ERR_INVALID;
  const hpf_handle::HeldSet &hs = h->held[slot];
  if (!hs.bound) { h->err = "hpf_heldout_ll_bound: nothing bound to this slot"; return HPF_ERR_STATE; }
  *sum_out = 0.0; if (cnt_out) *cnt_out = hs.cnt;
  if (hs.cnt == 0) return HPF_OK;
  if (!(h->u.have_E && h->it.have_E) && h->iterations == 0) { h->err = "E state not set"; return HPF_ERR_STATE; }
  return heldout_run(h, hs.du, hs.di, hs.dy, hs.dout, hs.hout, hs.cnt, sum_out, true);
}

int hpf_elbo(hpf_handle *h, double *out)
{
  if (!h || !out) return HPF_ERR_INVALID;
  *out = 0.0;
  if (!h->have_csr || h->iterations == 0) { h->err = "the ELBO is defined after the first iteration"; return HPF_ERR_STATE; }
  int rc;
  if ((rc = refresh_elog(h, h->u))) return rc;
  if ((rc = refresh_elog(h, h->it))) return rc;
  // per-nonzero term: the user-major work list(s) of the phi pass, one wave per segment
  const uint32_t nb_nnz = (uint32_t)std::min<uint64_t>(((uint64_t)h->u.nseg + 3) / 4 + 1, 16384);
  const uint32_t nb_g = 1024;
  double *part = nullptr, *Mt = nullptr, *Mb = nullptr;
  const size_t npart = (size_t)2 * nb_nnz + 2 * nb_g;
  if ((rc = dalloc(h, &part, npart))) return rc;
  std::vector<double> hp(npart, 0.0);
  do {
    // fp64 W: logsumexp from the hot loop's W and the row maxima of Elog (no exp per
    // element); the f32-stored W is not precise enough for that, it takes the Elog form
    const bool from_w = !h->w32 && h->wl == WL_PLAIN && h->nnz;
    if (from_w) {
      if ((rc = prepare_derived(h))) break;                  // W follows a set_state(ELOG), if any
      if ((rc = dalloc(h, &Mt, h->u.rows)) || (rc = dalloc(h, &Mb, h->it.rows))) break;
      hipLaunchKernelGGL(rowmax_elog_kernel, dim3((h->u.rows + 255) / 256), dim3(256), 0, h->stream,
                         h->u.L, Mt, h->u.rows, h->ld, h->K, h->u.bias_col, h->u.junk_col);
      hipLaunchKernelGGL(rowmax_elog_kernel, dim3((h->it.rows + 255) / 256), dim3(256), 0, h->stream,
                         h->it.L, Mb, h->it.rows, h->ld, h->K, h->it.bias_col, h->it.junk_col);
    }
    if (from_w && h->u.nseg) {
      const uint32_t ph = 0;
      ElboNnzWArgs a;
      a.segs = h->u.segs; a.nseg = h->u.nseg; a.col = h->u.pass_idx(); a.val = h->u.pass_val();
      a.Wt = (const double *)h->u.W; a.Wb = (const double *)h->it.W; a.Et = h->u.E; a.Eb = h->it.E;
      a.Mt = Mt; a.Mb = Mb;
      a.partial = part + (size_t)ph * nb_nnz; a.ld = h->ld; a.K = h->K;
      a.ubias_col = h->cfg.bias ? h->u.bias_col : -1; a.ibias_col = h->cfg.bias ? h->it.bias_col : -1;
      hipLaunchKernelGGL(elbo_nnz_w_kernel, dim3(nb_nnz), dim3(256), 0, h->stream, a);
    }
    if (h->nnz && !from_w && h->u.nseg) {
      const uint32_t ph = 0;
      ElboNnzArgs a;
      a.segs = h->u.segs; a.nseg = h->u.nseg; a.col = h->u.pass_idx(); a.val = h->u.pass_val();
      a.Lt = h->u.L; a.Lb = h->it.L; a.Et = h->u.E; a.Eb = h->it.E;
      a.partial = part + (size_t)ph * nb_nnz; a.ld = h->ld; a.K = h->K; a.C = h->C;
      a.ubias_col = h->cfg.bias ? h->u.bias_col : -1; a.ibias_col = h->cfg.bias ? h->it.bias_col : -1;
      hipLaunchKernelGGL(elbo_nnz_kernel, dim3(nb_nnz), dim3(256), 0, h->stream, a);
    }
    const double s0 = h->cfg.s_prior, ps = s0 + (double)h->K * s0;
    Side *sides[2] = {&h->u, &h->it};
    for (int k = 0; k < 2; ++k) {
      Side &s = *sides[k];
      if (k == 1 && h->cfg.rank != 0) continue;      // replicated item side: counted once
      if (!s.rows) continue;
      ElboGammaArgs g;
      g.S = s.S; g.E = s.E; g.L = s.L; g.prior_used = s.prior_used; g.prior_elog_used = s.prior_elog_used;
      g.colsum_used = s.colsum_used; g.prior_E = s.prior_E; g.prior_elog = s.prior_elog; g.prior_rate = s.prior_rate;
      g.partial = part + (size_t)2 * nb_nnz + (size_t)k * nb_g; g.rows = s.rows; g.ld = h->ld; g.K = h->K;
      g.bias_col = s.bias_col; g.bias_rate_add = s.bias_rate_add; g.s_prior = s0; g.r_prior = h->cfg.r_prior;
      g.lg_s_prior = std::lgamma(s0); g.prior_shape = ps; g.lg_prior_shape = std::lgamma(ps); g.hier = h->cfg.hier;
      hipLaunchKernelGGL(elbo_gamma_kernel, dim3(nb_g), dim3(256), 0, h->stream, g);
    }
    if ((rc = check_launch(h, "elbo"))) break;
    hipError_t e = hipMemcpyAsync(hp.data(), part, npart * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = HPF_ERR_HIP; break; }
    double s = 0.0;
    for (size_t k = 0; k < npart; ++k) s += hp[k];
    *out = s;
  } while (0);
  dfree(part); dfree(Mt); dfree(Mb);
  return rc;
}

// ---- ranking evaluation ----------------------------------------------------
namespace {
struct RankCtx {
  uint32_t *d_users = nullptr; uint64_t *d_mptr = nullptr; uint32_t *d_mitems = nullptr;
  double *d_scores = nullptr; uint32_t batch = 0, n_sel = 0;
  ~RankCtx() { dfree(d_users); dfree(d_mptr); dfree(d_mitems); dfree(d_scores); }
};

int rank_prepare(hpf_handle *h, const uint32_t *users, uint32_t n_sel, const uint64_t *mask_ptr,
                 const uint32_t *mask_items, RankCtx &c)
{
  if (h->iterations == 0 && !(h->u.have_E && h->it.have_E)) { h->err = "E state not set"; return HPF_ERR_STATE; }
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  const uint32_t m = h->it.rows;
  for (uint32_t b = 0; b < n_sel; ++b)
    if (users[b] >= h->u.rows) { h->err = "user index out of range"; return HPF_ERR_INVALID; }
  const uint64_t nmask = mask_ptr ? mask_ptr[n_sel] : 0;
  if (mask_ptr) {
    if (mask_ptr[0] != 0) { h->err = "mask_ptr[0] must be 0"; return HPF_ERR_INVALID; }
    for (uint32_t b = 0; b < n_sel; ++b) if (mask_ptr[b + 1] < mask_ptr[b]) { h->err = "mask_ptr not monotone"; return HPF_ERR_INVALID; }
    if (nmask && !mask_items) return HPF_ERR_INVALID;
    for (uint64_t j = 0; j < nmask; ++j) if (mask_items[j] >= m) { h->err = "mask item out of range"; return HPF_ERR_INVALID; }
  }
  int rc;
  if ((rc = refresh_es(h, h->u)) || (rc = refresh_es(h, h->it))) return rc;
  c.n_sel = n_sel;
  // rows of scores kept at once: <= 1 GiB, a multiple of 16
  uint64_t bmax = ((1ull << 30) / (8ull * std::max<uint32_t>(m, 1))) & ~15ull;
  bmax = std::max<uint64_t>(bmax, 16);
  c.batch = (uint32_t)std::min<uint64_t>(bmax, ((uint64_t)n_sel + 15) & ~15ull);
  if ((rc = dalloc(h, &c.d_users, n_sel))) return rc;
  if ((rc = dalloc(h, &c.d_scores, (size_t)c.batch * m))) return rc;
  HIPCHK(h, hipMemcpyAsync(c.d_users, users, (size_t)n_sel * 4, hipMemcpyHostToDevice, h->stream));
  if (mask_ptr) {
    if ((rc = dalloc(h, &c.d_mptr, (size_t)n_sel + 1))) return rc;
    if ((rc = dalloc(h, &c.d_mitems, (size_t)nmask))) return rc;
    HIPCHK(h, hipMemcpyAsync(c.d_mptr, mask_ptr, ((size_t)n_sel + 1) * 8, hipMemcpyHostToDevice, h->stream));
    if (nmask) HIPCHK(h, hipMemcpyAsync(c.d_mitems, mask_items, (size_t)nmask * 4, hipMemcpyHostToDevice, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return HPF_OK;
}

// scores (+ mask) of selected rows [b0, b1) into c.d_scores
int rank_scores(hpf_handle *h, RankCtx &c, uint32_t b0, uint32_t b1, bool mask)
{
  const uint32_t m = h->it.rows, rows = b1 - b0;
  ScoreArgs a;
  a.users = c.d_users + b0; a.Et = h->u.E; a.Eb = h->it.E; a.scores = c.d_scores;
  a.n_sel = rows; a.m = m; a.ld = h->ld; a.K = h->K;
  a.ubias_col = h->cfg.bias ? h->u.bias_col : -1; a.ibias_col = h->cfg.bias ? h->it.bias_col : -1;
  hipLaunchKernelGGL(score_tile_kernel, dim3((m + 255) / 256, (rows + 15) / 16), dim3(256), 0, h->stream, a);
  if (mask)
    hipLaunchKernelGGL(mask_scores_kernel, dim3(std::min<uint32_t>((rows + 3) / 4, 4096)), dim3(256), 0,
                       h->stream, c.d_users + b0, rows, h->rowptr_dev, h->u.idx, h->u.val,
                       c.d_mptr ? c.d_mptr + b0 : nullptr, c.d_mitems, c.d_scores, m);
  return check_launch(h, "score/mask");
}
}  // namespace

int hpf_scores(hpf_handle *h, const uint32_t *users, uint32_t n_sel, double *out)
{
  if (!h || (n_sel && (!users || !out))) return HPF_ERR_INVALID;
  if (!n_sel) return HPF_OK;
  RankCtx c; int rc;
  if ((rc = rank_prepare(h, users, n_sel, nullptr, nullptr, c))) return rc;
  const uint32_t m = h->it.rows;
  for (uint32_t b0 = 0; b0 < n_sel; b0 += c.batch) {
    const uint32_t b1 = std::min(n_sel, b0 + c.batch);
    if ((rc = rank_scores(h, c, b0, b1, false))) return rc;
    HIPCHK(h, hipMemcpyAsync(out + (size_t)b0 * m, c.d_scores, (size_t)(b1 - b0) * m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return HPF_OK;
}

int hpf_rank_topn(hpf_handle *h, const uint32_t *users, uint32_t n_sel, const uint64_t *mask_ptr,
                  const uint32_t *mask_items, uint32_t topn, uint32_t *out_items, double *out_scores)
{
  if (!h || topn == 0 || topn > 1024 || (n_sel && (!users || !out_items || !out_scores))) return HPF_ERR_INVALID;
  if (!n_sel) return HPF_OK;
  RankCtx c; int rc;
  if ((rc = rank_prepare(h, users, n_sel, mask_ptr, mask_items, c))) return rc;
  uint32_t NP = 1; while (NP < topn) NP <<= 1;
  uint32_t *d_items = nullptr; double *d_sc = nullptr;
  if ((rc = dalloc(h, &d_items, (size_t)n_sel * topn)) || (rc = dalloc(h, &d_sc, (size_t)n_sel * topn))) { dfree(d_items); return rc; }
  for (uint32_t b0 = 0; b0 < n_sel && !rc; b0 += c.batch) {
    const uint32_t b1 = std::min(n_sel, b0 + c.batch);
    if ((rc = rank_scores(h, c, b0, b1, true))) break;
    hipLaunchKernelGGL(topn_kernel, dim3(b1 - b0), dim3(256), (size_t)NP * 12, h->stream, c.d_scores, b1 - b0,
                       h->it.rows, topn, NP, d_items + (size_t)b0 * topn, d_sc + (size_t)b0 * topn);
    rc = check_launch(h, "topn_kernel");
  }
  if (!rc) {
    hipError_t e = hipMemcpyAsync(out_items, d_items, (size_t)n_sel * topn * 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_scores, d_sc, (size_t)n_sel * topn * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = HPF_ERR_HIP; }
  }
  dfree(d_items); dfree(d_sc);
  return rc;
}

int hpf_item_ranks(hpf_handle *h, const uint32_t *users, uint32_t n_sel, const uint64_t *mask_ptr,
                   const uint32_t *mask_items, const uint32_t *q_sel, const uint32_t *q_item, uint32_t nq,
                   uint32_t *out_rank, double *out_score)
{
  if (!h || (n_sel && !users) || (nq && (!q_sel || !q_item || !out_rank || !out_score))) return HPF_ERR_INVALID;
  if (!nq) return HPF_OK;
  for (uint32_t q = 0; q < nq; ++q)
    if (q_sel[q] >= n_sel || q_item[q] >= h->it.rows) { h->err = "query out of range"; return HPF_ERR_INVALID; }
  RankCtx c; int rc;
  if ((rc = rank_prepare(h, users, n_sel, mask_ptr, mask_items, c))) return rc;
  uint32_t *d_qs = nullptr, *d_qi = nullptr, *d_rank = nullptr; double *d_sc = nullptr;
  do {
    if ((rc = dalloc(h, &d_qs, nq)) || (rc = dalloc(h, &d_qi, nq)) || (rc = dalloc(h, &d_rank, nq)) || (rc = dalloc(h, &d_sc, nq))) break;
    hipError_t e = hipMemcpyAsync(d_qs, q_sel, (size_t)nq * 4, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_qi, q_item, (size_t)nq * 4, hipMemcpyHostToDevice, h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = HPF_ERR_HIP; break; }
    for (uint32_t b0 = 0; b0 < n_sel && !rc; b0 += c.batch) {
      const uint32_t b1 = std::min(n_sel, b0 + c.batch);
      if ((rc = rank_scores(h, c, b0, b1, true))) break;
      hipLaunchKernelGGL(rank_query_kernel, dim3(std::min<uint32_t>(nq, 16384)), dim3(256), 0, h->stream, c.d_scores,
                         h->it.rows, d_qs, d_qi, nq, b0, b1, d_rank, d_sc);
      rc = check_launch(h, "rank_query_kernel");
    }
    if (rc) break;
    e = hipMemcpyAsync(out_rank, d_rank, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out_score, d_sc, (size_t)nq * 8, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); rc = HPF_ERR_HIP; }
  } while (0);
  dfree(d_qs); dfree(d_qi); dfree(d_rank); dfree(d_sc);
  return rc;
}

int hpf_get_work_info(hpf_handle *h, hpf_work_info *out)
{
  if (!h || !out) return HPF_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->nnz = h->nnz;
  out->user_segments = h->u.nseg;
  out->user_long_rows = h->u.nlong + h->u.nhuge;
  out->user_huge_rows = h->u.nhuge;
  out->item_segments = h->it.nseg;
  out->item_long_rows = h->it.nlong + h->it.nhuge;
  out->item_huge_rows = h->it.nhuge;
  out->phi_G = (uint32_t)h->phiG; out->phi_R = (uint32_t)h->phiR; out->phi_V = (uint32_t)h->phiV;
  out->sweep_G = (uint32_t)h->swG; out->sweep_R = (uint32_t)h->swR;
  out->ld = h->ld;
  out->w_layout = (uint32_t)h->wl;
  out->tiles_user = h->u.tiles; out->tiles_item = h->it.tiles;
  out->tile_rows_user = h->u.tile_rows; out->tile_rows_item = h->it.tile_rows;
  out->heavy_min_nnz_user = h->u.tiles ? h->u.light_below : 0; out->heavy_min_nnz_item = h->it.tiles ? h->it.light_below : 0;
  out->w_fallbacks = h->fallbacks;
  out->notes = h->notes;
  out->start_sums_pending = (h->jacobi && h->cfg.n_ranks > 1 && (h->sums_dirty || !h->start_sums_done || h->tail_partial)) ? 1u : 0u;
  out->graph_replay = (h->cfg.n_ranks == 1 && !h->comm) ? ((h->have_csr && want_graph(h)) ? 1u : 0u) : (h->have_csr && split_graph_on(h) ? 2u : 0u);
  out->tile_chunk_user = h->u.chunks ? h->u.chunk_segs : 0; out->tile_chunk_item = h->it.chunks ? h->it.chunk_segs : 0;
  return HPF_OK;
}

int hpf_gather_only(hpf_handle *h, int side, int reps, float *ms_out)
{
  if (!h || !ms_out || reps < 1 || (side != 0 && side != 1)) return HPF_ERR_INVALID;
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  if (h->w32 || (h->wl == WL_PLAIN && h->phiV != 2)) { h->err = "gather-only probe: rows of 16-byte pieces only"; return HPF_ERR_UNSUPPORTED; }
  { int rc0 = recover_if_flushed(h); if (rc0) return rc0; }
  int rc;
  if ((rc = prepare_derived(h))) return rc;
  Side &own = side ? h->it : h->u, &oth = side ? h->u : h->it;
  PhiArgs a;
  a.segs = own.segs; a.nseg = own.nseg; a.idx = own.pass_idx(); a.val = own.pass_val();
  a.W_own = own.W; a.W_oth = oth.W; a.S_own = nullptr; a.partial = nullptr; a.flags = h->flags;
  a.chunks = own.chunks; a.ld = h->ld;
  *ms_out = 0.0f;
  if (!a.nseg) return HPF_OK;
  uint32_t *sink = nullptr;
  if ((rc = dalloc(h, &sink, 1))) return rc;
  const uint32_t wpb = rows_in_pieces(h) ? h->wg_of(own.chunks != nullptr) / 64 : 4;        // the pass's own workgroups (run_phi)
  const uint32_t blocks = own.chunks ? own.nchunk_blocks : std::min<uint32_t>((a.nseg + wpb - 1) / wpb, h->phi_blocks * (4 / wpb));
  const PhiLaunch pl = {blocks, wpb * 64, h->stream};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  bool ok = e == hipSuccess && launch_gather_only(h->phiG, h->phiR, a, sink, pl);      // warm-up
  if (ok) {
    e = hipEventRecord(e0, h->stream);
    for (int r = 0; r < reps && ok; ++r) ok = launch_gather_only(h->phiG, h->phiR, a, sink, pl);
    if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / (float)reps;
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  dfree(sink);
  if (!ok) { h->err = "no gather-only kernel for this shape"; return HPF_ERR_UNSUPPORTED; }
  if (e != hipSuccess) { h->err = std::string("gather-only probe: ") + hipGetErrorString(e); return HPF_ERR_HIP; }
  return check_launch(h, "gather-only probe");
}

int hpf_host_alloc(void **ptr, size_t bytes)
{
  if (!ptr) return HPF_ERR_INVALID;
  *ptr = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); return HPF_ERR_NO_DEVICE; }
  if (hipHostMalloc(ptr, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; return HPF_ERR_OOM; }
  return HPF_OK;
}

int hpf_host_free(void *ptr)
{
  if (!ptr) return HPF_OK;
  if (hipHostFree(ptr) != hipSuccess) { (void)hipGetLastError(); return HPF_ERR_INVALID; }
  return HPF_OK;
}

int hpf_synchronize(hpf_handle *h)
{
  if (!h) return HPF_ERR_INVALID;
  return check_flags(h);
}

int hpf_mean_timing(hpf_handle *h, uint32_t n_last, hpf_timing *out)
{
  if (!h || !out) return HPF_ERR_INVALID;
  memset(out, 0, sizeof *out);
  out->iterations = h->iterations;
  uint32_t n = std::min<uint32_t>(std::min<uint32_t>(n_last, h->ev_count), hpf_handle::RING);
  if (n == 0) return HPF_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t slot = (h->ev_count - 1 - k) % hpf_handle::RING;
    hipEvent_t *ev = h->evr[slot];
    float ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // a graph-replayed iteration is one launch: only its total is known
    if (!h->ring_graphed[slot]) {
      for (int j = 0; j < 5; ++j) HIPCHK(h, hipEventElapsedTime(&ms[j], ev[j], ev[j + 1]));
      HIPCHK(h, hipEventElapsedTime(&ms[7], ev[5], ev[7]));      // exchange the stream waited for
      HIPCHK(h, hipEventElapsedTime(&ms[5], ev[7], ev[6]));      // the item sweep itself
    }
    HIPCHK(h, hipEventElapsedTime(&ms[6], ev[0], ev[6]));
    for (int j = 0; j < 8; ++j) acc[j] += ms[j];
  }
  out->phi_item_ms = (float)(acc[0] / n);
  out->combine_item_ms = (float)(acc[1] / n);
  out->phi_user_ms = (float)(acc[2] / n);
  out->combine_user_ms = (float)(acc[3] / n);
  out->sweep_user_ms = (float)(acc[4] / n);
  out->sweep_item_ms = (float)(acc[5] / n);
  out->iteration_ms = (float)(acc[6] / n);
  out->exchange_wait_ms = (float)(acc[7] / n);
  return HPF_OK;
}

int hpf_last_timing(hpf_handle *h, hpf_timing *out) { return hpf_mean_timing(h, 1, out); }

int hpf_iteration_times(hpf_handle *h, uint32_t n_last, float *ms_out, uint32_t *n_out)
{
  if (!h || !ms_out || !n_out) return HPF_ERR_INVALID;
  *n_out = 0;
  const uint32_t n = std::min<uint32_t>(std::min<uint32_t>(n_last, h->ev_count), hpf_handle::RING);
  if (n == 0) return HPF_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (uint32_t k = 0; k < n; ++k) {                       // oldest first
    const uint32_t slot = (h->ev_count - n + k) % hpf_handle::RING;
    float ms = 0.0f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->evr[slot][0], h->evr[slot][6]));
    ms_out[k] = ms;
  }
  *n_out = n;
  return HPF_OK;
}

int hpf_debug_poke_index(hpf_handle *h, int side, uint64_t pos, uint32_t value, uint32_t *old_value, uint32_t *owner_row)
{
  if (!h || (side != 0 && side != 1)) return HPF_ERR_INVALID;
  if (!h->have_csr) { h->err = "hpf_upload_csr has not been called"; return HPF_ERR_STATE; }
  Side &own = side ? h->it : h->u, &oth = side ? h->u : h->it;
  if (pos >= h->nnz || value >= oth.rows) { h->err = "poke: position or value out of range"; return HPF_ERR_INVALID; }
  uint32_t *arr = own.p_idx ? own.p_idx : own.idx;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  uint32_t old = 0;
  HIPCHK(h, hipMemcpy(&old, arr + pos, 4, hipMemcpyDeviceToHost));
  HIPCHK(h, hipMemcpy(arr + pos, &value, 4, hipMemcpyHostToDevice));
  if (old_value) *old_value = old;
  if (owner_row) {                          // the segment that holds the position names its owner
    std::vector<Seg> segs(own.nseg);
    if (own.nseg) HIPCHK(h, hipMemcpy(segs.data(), own.segs, (size_t)own.nseg * sizeof(Seg), hipMemcpyDeviceToHost));
    *owner_row = 0xffffffffu;
    for (const Seg &sg : segs)
      if ((uint64_t)sg.start <= pos && pos < (uint64_t)sg.start + sg.len) { *owner_row = sg.row; break; }
  }
  drop_graph(h);
  return HPF_OK;
}

int hpf_algorithmic_bytes(hpf_handle *h, uint64_t *phi_user, uint64_t *phi_item, uint64_t *rows)
{
  if (!h) return HPF_ERR_INVALID;
  // SURVEY.md 8(d) with s_a = 8, K' = K + (bias ? 1 : 0) and s_e = the bytes a stored element of W
  // takes: 8 in plain rows, 59/8 in the lossless packed rows (the default at K = 100), 6 / 4 in
  // the opt-in lossy modes -- so that "algorithmic bytes" never exceed what has to move.  The
  // item-side accumulate term nnz*K'*s_a of that formula is, in this
  // formulation, the gather of the item pass (same byte count); the item pass
  // is credited with that term only (its own index / row traffic is not
  // counted), so phi_user + phi_item == B_phi of SURVEY.md exactly.
  const uint64_t Kp = h->K + (h->cfg.bias ? 1u : 0u), nnz = h->nnz;
  const uint64_t by = h->u.val ? 1u : 0u, n = h->u.rows, m = h->it.rows;
  const uint64_t seb = h->w32 ? 32 : h->wl == WL_F48 ? 48 : h->wl == WL_P59 ? 59 : 64, sa = 8;   // BITS per stored W element; bytes per accumulator
  if (phi_user) *phi_user = nnz * (4 + by) + 8 * (n + 1) + nnz * Kp * seb / 8 + n * Kp * seb / 8 + n * Kp * sa;
  if (phi_item) *phi_item = nnz * Kp * seb / 8;
  if (rows) *rows = (n + m) * Kp * 2 * sa + (n + m) * Kp * 2 * seb / 8 + 64 * (n + m);
  return HPF_OK;
}

}  // extern "C"

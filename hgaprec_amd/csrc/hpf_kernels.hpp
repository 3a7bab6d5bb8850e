// hpf_kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels for the
// hgaprec CAVI inner loop.  No MFMA: the path is gather / elementwise /
// small-group reduction work, HBM- and L2-bound.
//
// Formulation.  The reference computes, per nonzero (u,i,y)
//   phi_k = y * exp(x_k - logsumexp(x)),  x_k = Elog_theta[u,k] + Elog_beta[i,k]
// (hgaprec.cc:206-239 get_phi, matrix.hh:367-389 logsum/lognormalize) and adds
// phi to the shape rows of u and of i (gpbase.hh:175-180).  Because
// exp(a+b) = exp(a)exp(b) and the softmax is invariant to a per-row factor,
// the row sweeps store W = exp(Elog - rowmax(Elog)) once per matrix element
// and the per-nonzero work becomes
//   e_k = Wt[u,k] * Wb[i,k];   phi_k = y * e_k / sum_j e_j
// -- no transcendental in the nnz*K loop.  The two shape scatters become two
// gather passes (user-major over CSR, item-major over CSC) so that every
// shape row is produced by exactly one owner in a fixed order: no atomics,
// bit-reproducible results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

// Build-time switches (A/B builds of tools/ and profiles/ only; the library ships the defaults)
#ifndef HPF_REDUCE_SCATTER
#define HPF_REDUCE_SCATTER 1   // end of a segment in the packed passes: 1 = the groups' accumulators reduced halving the element list
#endif                         // per level (round 6), 0 = every element on every lane at every level (rounds 1-5); the same bits
#ifndef HPF_P59_PAIRED
#define HPF_P59_PAIRED 1       // order of a p59 lane's dwords: 1 = (low word, stream dword) pairs (round 6), 0 = round 5's
#endif

namespace hpf {

// ---------------------------------------------------------------------
// work item of a phi pass: up to seg_max consecutive nonzeros of ONE owner row
// ---------------------------------------------------------------------
struct Seg {
  int64_t  start;   // first nonzero (index into idx/val)
  uint32_t row;     // owner row
  uint32_t len;     // nonzeros in this segment (may be 0 for an empty row)
  int32_t  pslot;   // >= 0: write the partial sum to partial[pslot] (long row)
                    //  < 0: the row is a single segment, write S[row] directly
  uint32_t pad;
};

struct LongRow { uint32_t row; uint32_t first_slot; uint32_t nslots; uint32_t pad; };

// ---------------------------------------------------------------------
// cross-lane helpers (wave64).  DPP moves for spans inside a 16-lane row,
// ds_bpermute (via __shfl_xor) across rows.
// ---------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  // every lane reads a live lane under these controls: bound_ctrl frees the destination
  // from being tied to an "old" value (no v_mov copy in front of each v_mov_dpp)
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// sum over aligned groups of G lanes; every lane of the group gets the total
template <int G>
__device__ __forceinline__ double group_sum(double v)
{
  if (G >= 2)  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  if (G >= 4)  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  if (G >= 8)  v += dpp_mov<0x141>(v);   // row_half_mirror      : quad ^ 1
  if (G >= 16) v += dpp_mov<0x140>(v);   // row_mirror           : half ^ 1
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <int G>
__device__ __forceinline__ double group_max(double v)
{
  if (G >= 2)  v = fmax(v, dpp_mov<0xB1>(v));
  if (G >= 4)  v = fmax(v, dpp_mov<0x4E>(v));
  if (G >= 8)  v = fmax(v, dpp_mov<0x141>(v));
  if (G >= 16) v = fmax(v, dpp_mov<0x140>(v));
  if (G >= 32) v = fmax(v, __shfl_xor(v, 16, 64));
  if (G >= 64) v = fmax(v, __shfl_xor(v, 32, 64));
  return v;
}

// 1/x to ~1 ulp: v_rcp_f64 seed + two Newton steps (5 instructions instead
// of the ~15 of an IEEE-exact fp64 division); x is a positive normal number
__device__ __forceinline__ double fast_rcp(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// ---------------------------------------------------------------------
// K1: phi pass.  Template: G lanes per nonzero, R loads per lane, V elements
// per load (8- or 16-byte loads).  The row stride of every device matrix is
// EXACTLY G*R*V elements (hpf_create pads K + 2*bias up to it; the pad columns
// of W hold zeros), so lane g of a group owns columns
//   (g + G*t)*V + v ,  t < R, v < V
// with no column test anywhere, and the stride is a compile-time constant.
// A wave walks one segment; its 64/G groups take consecutive nonzeros, so a
// "batch" is 64/G nonzeros.
//
// Per batch (round 3, VERDICT r2 #1: 188 -> ~60 VALU instructions at K=100):
//   ssum  = sum_k own_k * x_k          two FMA chains + DPP group sum
//   scale = yy / ssum                  v_rcp_f64 + 2 Newton steps (~1 ulp)
//   acc_k += x_k * scale               ONE fma per element: the owner's factor
//                                      own_k is the same for every nonzero of the
//                                      segment, so it multiplies the finished sum
//                                      once (S_k = own_k * acc_k) instead of every term
// The gathers of batch b+1 are in flight while batch b is worked on, in two
// register sets used alternately (no copies).  Inactive slots of the last
// batch gather row 0 and carry yy = 0.
// ---------------------------------------------------------------------
// V elements of type T moved by one 8- or 16-byte access
template <typename T, int V>
struct __attribute__((aligned((sizeof(T) * V > 16) ? 16 : sizeof(T) * V))) vecw { T x[V]; };

struct PhiArgs {
  const Seg      *segs;
  uint32_t        nseg;
  const uint32_t *idx;      // other-side row of each nonzero
  const uint8_t  *val;      // rating (NULL: all ones)
  const void     *W_own;    // [rows_own x G*R*V] of WT (double, or float in the f32-storage mode)
  const void     *W_oth;    // [rows_oth x G*R*V] of WT
  double         *S_own;    // [rows_own x G*R*V]  raw sums (prior added by sweep)
  double         *partial;  // [npartial x G*R*V]
  uint32_t       *flags;    // [0] bit 0: a live nonzero saw sum_k e_k == 0 (underflow of W); bit 1: a sweep had to flush an
                            // entry the p59 rows cannot hold; bit 2: a pass has seen bit 1 -- everything launched since is a
                            // no-op until the host has switched the row layout (hpf_capi.hip, recover_flush)
                            // [1] iterations begun with bits 1 and 2 clear (counted by the item-major pass, which opens an iteration)
  // tiled pass (hpf_build.hpp): workgroup b works on the segments [chunks[b].x, chunks[b].y), one wave
  // per segment in turn; the host lays the chunks out so that b % 8 -- the XCD a workgroup lands on --
  // walks one tile after another.  NULL: the waves stride over the whole list.
  const uint2    *chunks;
  uint32_t        ld;       // row stride of S_own / partial, columns (layouts whose rows do not fix it: codec_f64)
};

// Entry of every phi-pass kernel.  A sweep that could not store an element in the packed rows raises bit 1;
// the passes that follow must not run on that W: they return at once (and the item-major pass, which opens
// an iteration, turns bit 1 into bit 2, which stops the sweeps as well) until the host -- at its next
// synchronisation point -- has moved the handle to plain fp64 rows and repeats what was skipped.
__device__ __forceinline__ bool phi_pass_skips(const PhiArgs &a, int side)
{
  const uint32_t fl = a.flags[0];
  if (side == 1 && blockIdx.x == 0 && threadIdx.x == 0) {
    if (fl & 2u) atomicOr(a.flags, 4u);
    else if (!(fl & 4u)) atomicAdd(a.flags + 1, 1u);
  }
  return (fl & 6u) != 0u;
}

// the segments of this wave: first, end, stride
struct SegRange { uint32_t s, end, step; };
__device__ __forceinline__ SegRange seg_range(const PhiArgs &a)
{
  SegRange r;
  if (a.chunks) {
    const uint2 c = a.chunks[blockIdx.x];
    r.s = __builtin_amdgcn_readfirstlane(c.x + (threadIdx.x >> 6));
    r.end = __builtin_amdgcn_readfirstlane(c.y);
    r.step = blockDim.x >> 6;
  } else {
    r.s = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    r.end = a.nseg;
    r.step = (gridDim.x * blockDim.x) >> 6;
  }
  return r;
}

// an item-major pass over no nonzeros at all still opens the iteration (phi_pass_skips)
__global__ void phi_open_kernel(PhiArgs a) { (void)phi_pass_skips(a, 1); }

// one batch: x = the gathered rows (one nonzero per group), yf = its rating
// factor as a float (0 for an empty slot)
template <typename WT, int G, int R, int V>
__device__ __forceinline__ void phi_batch(const vecw<WT, V> (&x)[R], const vecw<double, V> (&own)[R],
                                          vecw<double, V> (&acc)[R], float yf, bool &underflow)
{
  double s[2] = {0.0, 0.0};              // two FMA chains: the dependent latency is halved
#pragma unroll
  for (int e = 0; e < R * V; ++e) {
    const double o = own[e / V].x[e % V], xv = (double)x[e / V].x[e % V];
    s[e & 1] = (e < 2) ? o * xv : fma(o, xv, s[e & 1]);
  }
  const double ssum = group_sum<G>((R * V > 1) ? s[0] + s[1] : s[0]);
  const double yy = (double)yf;
  const bool ok = ssum > 0.0;
  underflow |= (yf > 0.0f) && !ok;
  const double scale = ok ? yy * fast_rcp(ssum) : 0.0;
#pragma unroll
  for (int t = 0; t < R; ++t)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[t].x[v] = fma((double)x[t].x[v], scale, acc[t].x[v]);
}

// WT: storage type of W.  Arithmetic and accumulators are fp64 either way.
// SIDE only names the instantiation (0 = user-major pass over CSR, 1 =
// item-major pass over CSC) so that profilers report the passes apart.
template <typename WT, int G, int R, int V, int SIDE>
__global__ __launch_bounds__(256) void phi_pass_kernel(PhiArgs a)
{
  constexpr int NG = 64 / G;             // nonzeros per batch
  constexpr uint32_t LD = G * R * V;     // row stride, elements
  const int lane = threadIdx.x & 63;
  const int g = lane % G;                // column lane
  const int q = lane / G;                // group = nonzero slot in a batch
  if (phi_pass_skips(a, SIDE)) return;
  const SegRange sr = seg_range(a);
  const WT *W_own = (const WT *)a.W_own + (size_t)g * V;
  const WT *W_oth = (const WT *)a.W_oth + (size_t)g * V;
  bool underflow = false;

  for (uint32_t s = sr.s; s < sr.end; s += sr.step) {
    const Seg sg = a.segs[s];            // wave-uniform: scalar loads
    const uint32_t len = sg.len;
    const int64_t start = sg.start;

    vecw<double, V> own[R], acc[R];
    {
      const WT *wo = W_own + (size_t)sg.row * LD;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const vecw<WT, V> raw = *reinterpret_cast<const vecw<WT, V> *>(wo + (size_t)G * t * V);
#pragma unroll
        for (int v = 0; v < V; ++v) { own[t].x[v] = (double)raw.x[v]; acc[t].x[v] = 0.0; }
      }
    }

    if (len > 0) {
      // indices / rating factors of the current and the next chunk of 64 nonzeros, one
      // per lane.  The factor is yy = (y > 1 ? y : 1) -- "if (y > 1) phi.scale(y)",
      // hgaprec.cc:1355-1356: a rating that wrapped to 0 in the reference's uint8 store is
      // not scaled -- as a float (exact), 0 beyond the end of the segment.
      auto load_i = [&](uint32_t o) -> uint32_t { return (o < len) ? a.idx[start + o] : 0u; };
      auto load_y = [&](uint32_t o) -> float {
        if (o >= len) return 0.0f;
        if (!a.val) return 1.0f;
        const uint32_t y = a.val[start + o];
        return (y > 1u) ? (float)y : 1.0f;
      };
      uint32_t cur_i = load_i((uint32_t)lane), nxt_i = load_i(64u + lane);
      float cur_y = load_y((uint32_t)lane), nxt_y = load_y(64u + lane);

      const uint32_t nb = (len + NG - 1) / NG;      // batches
      vecw<WT, V> xa[R], xb[R];
      float ya, yb = 0.0f;
      auto gather = [&](vecw<WT, V> (&x)[R], float &y, uint32_t b) {
        const int src = (int)((b % G) * NG) + q;
        const uint32_t in = (uint32_t)__shfl((int)cur_i, src, 64);
        y = __shfl(cur_y, src, 64);
        const WT *p = W_oth + (size_t)in * LD;
#pragma unroll
        for (int t = 0; t < R; ++t) x[t] = *reinterpret_cast<const vecw<WT, V> *>(p + (size_t)G * t * V);
      };
      auto next_chunk = [&](uint32_t b) {           // batch b opens a new chunk of 64
        cur_i = nxt_i; cur_y = nxt_y;
        const uint32_t o = (b / G + 1) * 64u + lane;
        nxt_i = load_i(o); nxt_y = load_y(o);
      };
      // Two register sets, refilled right after their batch is consumed: 1-2 batches of
      // gathers are in flight behind the one being worked on.  The steady loop has no
      // conditional gather (a conditional one costs a third register set and a copy per
      // round); the last one to three batches are peeled.  G is even, so only even batch
      // numbers open a chunk.
      gather(xa, ya, 0);
      if (nb > 1) gather(xb, yb, 1);
      uint32_t bb = 0;
      for (; bb + 3 < nb; bb += 2) {
        phi_batch<WT, G, R, V>(xa, own, acc, ya, underflow);
        __builtin_amdgcn_sched_barrier(0);   // the refill stays behind the last use of xa
        if (((bb + 2) % G) == 0) next_chunk(bb + 2);
        gather(xa, ya, bb + 2);
        phi_batch<WT, G, R, V>(xb, own, acc, yb, underflow);
        __builtin_amdgcn_sched_barrier(0);
        gather(xb, yb, bb + 3);
      }
      phi_batch<WT, G, R, V>(xa, own, acc, ya, underflow);
      if (bb + 1 < nb) {
        const bool third = bb + 2 < nb;
        __builtin_amdgcn_sched_barrier(0);
        if (third) {
          if (((bb + 2) % G) == 0) next_chunk(bb + 2);
          gather(xa, ya, bb + 2);
        }
        phi_batch<WT, G, R, V>(xb, own, acc, yb, underflow);
        if (third) phi_batch<WT, G, R, V>(xa, own, acc, ya, underflow);
      }
    }

    // ---- reduce the 64/G group accumulators, apply the owner's factor, write the row (or partial)
    double *dst = ((sg.pslot >= 0) ? a.partial + (size_t)sg.pslot * LD : a.S_own + (size_t)sg.row * LD) + (size_t)g * V;
#pragma unroll
    for (int t = 0; t < R; ++t) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        double r = acc[t].x[v];
        if (G <= 32) r += __shfl_xor(r, 32, 64);
        if (G <= 16) r += __shfl_xor(r, 16, 64);
        if (G <= 8)  r += __shfl_xor(r, 8, 64);
        if (G <= 4)  r += __shfl_xor(r, 4, 64);
        acc[t].x[v] = own[t].x[v] * r;
      }
      if (q == 0) *reinterpret_cast<vecw<double, V> *>(dst + (size_t)G * t * V) = acc[t];
    }
  }
  if (__any(underflow) && lane == 0) atomicOr(a.flags, 1u);
}

// ---------------------------------------------------------------------
// K1 over PACKED rows of W.  Both phi passes sit at the rate at which L2 misses are filled
// (DESIGN.md section 6): the only lever left is 128-byte lines per gathered row.
//
//   p59  (the default whenever it shortens the row; LOSSLESS): every W is a positive fp64 in
//        [2^-126, 2) or zero, so its sign bit and the four high exponent bits carry nothing:
//        52 mantissa + 7 exponent bits = 59 bits per element.  K = 100: 104 columns in 768 bytes
//        = six lines instead of seven.  A W below 2^-126 of its row maximum (an Elog spread
//        above 88 inside a row) cannot be stored: the sweep flushes it to zero and raises flag
//        bit 1 (hpf_config.w_storage = 3 keeps plain fp64 rows for such states).
//   f48  (hpf_config.w_storage = 2, opt-in, LOSSY): the top 48 bits of the fp64 value
//        (36 mantissa bits, rounded to nearest even: 2^-37 relative).  K = 100: five lines.
//
// A row is L 16-byte pieces per lane for G lanes, INTERLEAVED: piece t of lane g sits at byte
// (t*G + g)*16, so that one load instruction of a lane group reads G*16 contiguous bytes.  The
// 4L dwords of a lane hold E elements; lane g owns columns e*G + g.  Layout of a lane's dwords
//   p59: d[0..E)  low 32 mantissa bits of each element; then a stream of 27-bit fields
//        (7 exponent bits above 20 mantissa bits; the all-zero element stands for 0 and reads
//        back as 2^-127), element e at bit 27 e of the stream
//   f48: d[0..E)  high dwords; then the 16-bit low parts, two per dword
// Row stride of the fp64 matrices (S, E, Elog) and of the exchange buffer: ld = G*E columns.
// Arithmetic and accumulators are fp64 in every mode.
// ---------------------------------------------------------------------
//   f64  plain fp64 elements in the same interleaved pieces, two per piece (E = 2L).  What a handle falls back to
//        when a state turns up that p59 cannot hold (hpf_capi.hip, recover_flush) and what hpf_config.w_storage = 3
//        asks for from the start.
//        The row stride of S / partial stays that of the packed shape it stands in for (PhiArgs::ld <= G * E).
enum { WL_PLAIN = 0, WL_F48 = 2, WL_P59 = 3, WL_F64 = 4 };      // layout codes (hpf_work_info.w_layout)

template <int L> struct codec_f64 {
  static constexpr int E = 2 * L;
  static constexpr bool fixed_ld = false;
  static __device__ __forceinline__ double get(const uint32_t (&d)[4 * L], int e)
  {
    return __hiloint2double((int)d[2 * e + 1], (int)d[2 * e]);
  }
};

template <int L> struct codec_f48 {
  static constexpr bool fixed_ld = true;
  static constexpr int E = (8 * L) / 3;                   // 2 5 8 10 13 16 18 21
  static_assert(E + (E + 1) / 2 <= 4 * L, "lane dwords overflow");
  static __device__ __forceinline__ double get(const uint32_t (&d)[4 * L], int e)
  {
    const uint32_t lw = d[E + e / 2];
    return __hiloint2double((int)d[e], (int)((e & 1) ? (lw & 0xffff0000u) : (lw << 16)));
  }
};

template <int L> struct codec_p59 {
  static constexpr bool fixed_ld = true;
  static constexpr int E = (128 * L) / 59;                // 2 4 6 8 10 13 15 17
  static constexpr int S = (27 * E + 31) / 32;            // dwords of the field stream
  static_assert(E + S <= 4 * L, "lane dwords overflow");
  // Place of logical dword k among the lane's 4L (k < E: low word of element k; k >= E: dword k - E of the field stream).
  // Round 6: PAIRED -- the low word of element e sits at the even place 2e and stream dword j at the odd place 2j + 1 (the
  // E - 2L low words that find no even place take the odd places the stream leaves free).  A decoded element is the register
  // PAIR (low word, high word) and the high word is computed from the stream: with the low word in an even register and a
  // stream dword that is dead by then beside it, the pair is formed in place -- the elements are decoded from the last to the
  // first, so that stream dword e has been consumed by the elements above e when element e overwrites it.  In the round-5
  // order (all low words, then the stream) every other low word had to be copied into a fresh pair first: 11 v_mov per batch
  // of 93 VALU instructions and 12 more registers (tools/count_isa.py; profiles/r06/experiments.md 1).
  static constexpr __host__ __device__ int pos(int k)
  {
    if (!HPF_P59_PAIRED) return k;
    return k < E ? (k < 2 * L ? 2 * k : 2 * (S + k - 2 * L) + 1) : 2 * (k - E) + 1;
  }
  static __device__ __forceinline__ double get(const uint32_t (&d)[4 * L], int e)
  {
    const int o = 27 * e, i = E + o / 32, sh = o % 32;    // compile-time after unrolling
    uint32_t f;
    if (sh == 0) f = d[pos(i)];
    else if (sh + 27 <= 32) f = d[pos(i)] >> sh;
    else f = __builtin_amdgcn_alignbit(d[pos(i + 1)], d[pos(i)], sh);
    // exponent field + 896.  The all-zero element (padding columns, a flushed entry) decodes to
    // 2^-127 = 5.9e-39 instead of 0 -- no compare and two selects per element: products of two such
    // entries are 3e-77, below any sum they could join by sixty orders.  Mask and bias in ONE
    // instruction: v_and_or_b32 takes no literal on gfx9, so the mask sits in an SGPR and the bias in
    // a VGPR (the compiler left to itself emits v_and + v_or with literals)
    uint32_t hi;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(hi) : "v"(f), "s"(0x7ffffffu), "v"(0x38000000u));
    return __hiloint2double((int)hi, (int)d[pos(e)]);
  }
};

// ---- writing packed rows (row sweep, derive_w): a group of lanes builds the row in LDS -- the
// 27-bit fields of p59 straddle dwords, so they are OR-ed in (ds_or_b32: order-free, hence
// deterministic) -- and copies it out with 16-byte stores.  `buf` has the layout of the row.
struct PackedRow { uint32_t G, E, L, row_bytes, lgG; };  // of the phi kernel shape (G = 1 << lgG: no integer division)

__device__ __forceinline__ uint32_t packed_dword_index(const PackedRow &pk, uint32_t g, uint32_t d)
{
  return ((((d >> 2) << pk.lgG) + g) << 2) + (d & 3u);   // piece (d/4) of lane g, word d%4
}

__device__ __forceinline__ void packed_clear(uint32_t *buf, const PackedRow &pk, uint32_t li, uint32_t nl)
{
  for (uint32_t d = li; d < pk.row_bytes / 4; d += nl) buf[d] = 0u;
}

// codec_p59<L>::pos at run time (the LDS writers take the shape from PackedRow)
__device__ __forceinline__ uint32_t p59_pos(const PackedRow &pk, uint32_t k)
{
  if (!HPF_P59_PAIRED) return k;
  const uint32_t S = (27u * pk.E + 31u) / 32u;
  return k < pk.E ? (k < 2u * pk.L ? 2u * k : 2u * (S + k - 2u * pk.L) + 1u) : 2u * (k - pk.E) + 1u;
}

// returns true when a nonzero w had to be flushed (below 2^-126)
__device__ __forceinline__ bool p59_put(uint32_t *buf, const PackedRow &pk, uint32_t c, double w)
{
  const uint32_t g = c & (pk.G - 1u), e = c >> pk.lgG;
  const uint32_t hi = (uint32_t)__double2hiint(w), lo = (uint32_t)__double2loint(w);
  // representable: exponent field 897..1023, i.e. 2^-126 <= w < 2.  Anything else -- zero, a smaller
  // value, and what a valid state never holds: w >= 2, infinities, NaN, negative numbers -- is stored as
  // the all-zero element; only an exact +0 does so without being reported
  const bool tiny = hi < 0x38100000u || hi >= 0x40000000u;
  const uint32_t f = tiny ? 0u : hi - 0x38000000u;
  buf[packed_dword_index(pk, g, p59_pos(pk, e))] = tiny ? 0u : lo;
  const uint32_t o = 27u * e, i = pk.E + o / 32u, sh = o % 32u;
  if (f) {
    atomicOr(&buf[packed_dword_index(pk, g, p59_pos(pk, i))], f << sh);
    if (sh > 5u) atomicOr(&buf[packed_dword_index(pk, g, p59_pos(pk, i + 1))], f >> (32u - sh));
  }
  return tiny && !(hi == 0u && lo == 0u);
}

__device__ __forceinline__ void f48_put(uint32_t *buf, const PackedRow &pk, uint32_t c, double w)
{
  const uint32_t g = c & (pk.G - 1u), e = c >> pk.lgG;
  unsigned long long b = (unsigned long long)__double_as_longlong(w);
  b += 0x7fffull + ((b >> 16) & 1ull);                    // round to nearest even at bit 16
  buf[packed_dword_index(pk, g, e)] = (uint32_t)(b >> 32);
  atomicOr(&buf[packed_dword_index(pk, g, pk.E + e / 2)], (uint32_t)((b >> 16) & 0xffffull) << (16u * (e & 1u)));
}

__device__ __forceinline__ void packed_copy_out(const uint32_t *buf, void *W, size_t row, const PackedRow &pk,
                                                uint32_t li, uint32_t nl)
{
  uint4 *dst = reinterpret_cast<uint4 *>((unsigned char *)W + row * (size_t)pk.row_bytes);
  const uint4 *src = reinterpret_cast<const uint4 *>(buf);
  for (uint32_t p = li; p < pk.row_bytes / 16; p += nl) dst[p] = src[p];
}

// one batch over E element slots: x = the gathered rows in OthC's layout (LT pieces per lane)
// element c of a row of plain doubles in interleaved pieces (codec_f64): piece e/2 of lane g, half e%2
__device__ __forceinline__ void f64_put(void *W, size_t row, const PackedRow &pk, uint32_t c, double w)
{
  const uint32_t g = c & (pk.G - 1u), e = c >> pk.lgG;
  double *d = reinterpret_cast<double *>((unsigned char *)W + row * (size_t)pk.row_bytes);
  d[((((e >> 1) << pk.lgG) + g) << 1) + (e & 1u)] = w;
}

// ---- reduce-scatter of a wave's 64 / G group accumulators (phi_segments) --------------------------------------------
// element slots a lane holds after the levels at lane distances D, D/2, ..., G
template <int G, int N, int D = 32> struct scatter_len_at { static constexpr int value = D >= G ? scatter_len_at<G, (N + 1) / 2, D / 2>::value : N; };
template <int G, int N> struct scatter_len_at<G, N, 0> { static constexpr int value = N; };
template <int G, int E> struct scatter_len { static constexpr int value = scatter_len_at<G, E, 32>::value; };

// one level at lane distance D over the first N slots of v / w (w: the owner's factors, selected alongside); o, n: the run
// of element slots this lane holds -- [o, o + n) -- before and after
template <int G, int E, int D, int N = E>
__device__ __forceinline__ void reduce_scatter(double (&v)[E], double (&w)[E], int lane, uint32_t &o, uint32_t &n)
{
  if constexpr (D >= G && D >= 1) {
    constexpr int NA = (N + 1) / 2;                       // the half a lane with bit D clear keeps
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool hasB = NA + j < N;
      const double A = v[j], B = hasB ? v[NA + j] : 0.0;
      if constexpr (D == 32 || D == 16) {
        // gfx950's v_permlane32_swap / v_permlane16_swap exchange the upper half (the odd 16-lane rows) of one register with
        // the lower half (the even rows) of another: with A and B as the two registers, afterwards the FIRST holds, on the
        // lanes that keep A, their own A and, on the lanes that keep B, the partner's B -- and the second the other two.
        // Their sum is mine + the partner's on every lane: no select, no LDS crossbar.
        const uint32_t alo = (uint32_t)__double2loint(A), ahi = (uint32_t)__double2hiint(A);
        const uint32_t blo = (uint32_t)__double2loint(B), bhi = (uint32_t)__double2hiint(B);
        double x, y;
        if constexpr (D == 32) {
          const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false), hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
          x = __hiloint2double((int)hi[0], (int)lo[0]); y = __hiloint2double((int)hi[1], (int)lo[1]);
        } else {
          const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false), hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
          x = __hiloint2double((int)hi[0], (int)lo[0]); y = __hiloint2double((int)hi[1], (int)lo[1]);
        }
        v[j] = x + y;
      } else {
        const double mine = up ? B : A, send = up ? A : B;
        v[j] = mine + (D == 8 ? dpp_mov<0x128>(send) /* row_ror:8 = lane ^ 8 inside a 16-lane row */ : __shfl_xor(send, D, 64));
      }
      if (hasB) w[j] = up ? w[NA + j] : w[j];
    }
    o += up ? (uint32_t)NA : 0u;
    n = up ? (n > (uint32_t)NA ? n - (uint32_t)NA : 0u) : (n < (uint32_t)NA ? n : (uint32_t)NA);
    reduce_scatter<G, E, D / 2, NA>(v, w, lane, o, n);
  }
}

template <class OthC, int E, int G, int LT>
__device__ __forceinline__ void phi_batch_packed(const uint32_t (&x)[4 * LT], const double (&own)[E],
                                                 double (&acc)[E], float yf, bool &underflow)
{
  double xv[E];
#pragma unroll
  for (int e = E - 1; e >= 0; --e) xv[e] = OthC::get(x, e);     // last to first: codec_p59::pos
  double s[2] = {0.0, 0.0};
#pragma unroll
  for (int e = 0; e < E; ++e) s[e & 1] = (e < 2) ? own[e] * xv[e] : fma(own[e], xv[e], s[e & 1]);
  const double ssum = group_sum<G>((E > 1) ? s[0] + s[1] : s[0]);
  const double yy = (double)yf;
  const bool ok = ssum > 0.0;
  underflow |= (yf > 0.0f) && !ok;
  const double scale = ok ? yy * fast_rcp(ssum) : 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) acc[e] = fma(xv[e], scale, acc[e]);
}

// The segments of one wave, over rows of 16-byte pieces: the owner's row in OwnC's layout (LO pieces per
// lane), the gathered rows in OthC's (LT pieces).  E = OwnC::E element slots are worked on (OthC::E >= E; the
// library only instantiates OwnC = OthC -- the two-layout form is what the fp64-shadow experiment of round 4
// ran on, profiles/r04/experiments.md 1).  W_own / W_oth already point at this lane's first piece.
template <class OwnC, class OthC, int G, int LO, int LT>
__device__ __forceinline__ void phi_segments(const PhiArgs &a, const SegRange &sr, const unsigned char *W_own,
                                             const unsigned char *W_oth, int lane, bool &underflow)
{
  constexpr int NG = 64 / G;
  constexpr int E = OwnC::E;
  static_assert(OthC::E >= E, "the gathered layout must cover the owner's element slots");
  const uint32_t LD = OwnC::fixed_ld ? (uint32_t)(G * E) : a.ld;       // columns: stride of S / partial
  constexpr uint32_t ROWO = G * LO * 16, ROWT = G * LT * 16;           // bytes of a row of W, either layout
  const int g = lane % G, q = lane / G;

  auto load_own = [&](uint32_t (&d)[4 * LO], const unsigned char *base) {
#pragma unroll
    for (int t = 0; t < LO; ++t) {
      const uint4 v = *reinterpret_cast<const uint4 *>(base + (size_t)t * G * 16);
      d[4 * t] = v.x; d[4 * t + 1] = v.y; d[4 * t + 2] = v.z; d[4 * t + 3] = v.w;
    }
  };
  auto load_row = [&](uint32_t (&d)[4 * LT], const unsigned char *base) {
#pragma unroll
    for (int t = 0; t < LT; ++t) {
      const uint4 v = *reinterpret_cast<const uint4 *>(base + (size_t)t * G * 16);
      d[4 * t] = v.x; d[4 * t + 1] = v.y; d[4 * t + 2] = v.z; d[4 * t + 3] = v.w;
    }
  };
  // 64 indices / ratings of a segment from offset o on (zeros past its end)
  auto load_i = [&](int64_t start, uint32_t len, uint32_t o) -> uint32_t { return (o < len) ? a.idx[start + o] : 0u; };
  auto load_y = [&](int64_t start, uint32_t len, uint32_t o) -> float {
    if (o >= len) return 0.0f;
    if (!a.val) return 1.0f;
    const uint32_t y = a.val[start + o];
    return (y > 1u) ? (float)y : 1.0f;
  };

  // The wave's segments are one stream: the descriptor of the next segment and its first 64
  // indices are fetched while the current one is worked on, so that a segment starts with its
  // owner row and its first two batches of gathers in flight together -- one memory round trip
  // instead of three in a row (descriptor, indices, rows).  Rows of fifty nonzeros (C2's users)
  // and the runs of a tiled pass are a handful of batches long: the round trips were most of them.
  if (sr.s >= sr.end) return;
  Seg sgn = a.segs[sr.s];
  uint32_t nxt_i = load_i(sgn.start, sgn.len, (uint32_t)lane);
  float nxt_y = load_y(sgn.start, sgn.len, (uint32_t)lane);
  for (uint32_t s = sr.s; s < sr.end; s += sr.step) {
    const Seg sg = sgn;
    const bool more = s + sr.step < sr.end;
    if (more) sgn = a.segs[s + sr.step];
    const uint32_t len = sg.len;
    const int64_t start = sg.start;
    uint32_t cur_i = nxt_i;
    float cur_y = nxt_y;
    const uint32_t nb = (len + NG - 1) / NG;
    uint32_t xa[4 * LT], xb[4 * LT];
    float ya = 0.0f, yb = 0.0f;
    auto gather = [&](uint32_t (&x)[4 * LT], float &y, uint32_t b) {
      const int src = (int)((b % G) * NG) + q;
      const uint32_t in = (uint32_t)__shfl((int)cur_i, src, 64);
      y = __shfl(cur_y, src, 64);
      load_row(x, W_oth + (size_t)in * ROWT);
    };
    // chunk c has become the current one: fetch the one after it -- of this segment, or the
    // first one of the next segment
    auto fetch_after = [&](uint32_t c) {
      const uint32_t o = (c + 1) * 64u;
      const bool same = o < len;                        // scalar selects, no branch: one masked load each
      const int64_t st = same ? start : sgn.start;
      const uint32_t ln = same ? len : (more ? sgn.len : 0u), of = same ? o : 0u;
      nxt_i = load_i(st, ln, of + lane); nxt_y = load_y(st, ln, of + lane);
    };
    auto next_chunk = [&](uint32_t b) {
      cur_i = nxt_i; cur_y = nxt_y;
      fetch_after(b / G);
    };
    double own[E], acc[E];
    {
      uint32_t r[4 * LO];
      load_own(r, W_own + (size_t)sg.row * ROWO);
      gather(xa, ya, 0);          // unconditional (a conditional gather costs copies and waits): past the
      gather(xb, yb, 1);          // segment's end the index reads 0 -- row 0, loaded and never used
      fetch_after(0);
#pragma unroll
      for (int e = E - 1; e >= 0; --e) { own[e] = OwnC::get(r, e); acc[e] = 0.0; }
    }
    if (len > 0) {
      // same schedule as phi_pass_kernel: two register sets, peeled tail
      uint32_t bb = 0;
      for (; bb + 3 < nb; bb += 2) {
        phi_batch_packed<OthC, E, G, LT>(xa, own, acc, ya, underflow);
        __builtin_amdgcn_sched_barrier(0);
        if (((bb + 2) % G) == 0) next_chunk(bb + 2);
        gather(xa, ya, bb + 2);
        phi_batch_packed<OthC, E, G, LT>(xb, own, acc, yb, underflow);
        __builtin_amdgcn_sched_barrier(0);
        gather(xb, yb, bb + 3);
      }
      phi_batch_packed<OthC, E, G, LT>(xa, own, acc, ya, underflow);
      if (bb + 1 < nb) {
        const bool third = bb + 2 < nb;
        __builtin_amdgcn_sched_barrier(0);
        if (third) {
          if (((bb + 2) % G) == 0) next_chunk(bb + 2);
          gather(xa, ya, bb + 2);
        }
        phi_batch_packed<OthC, E, G, LT>(xb, own, acc, yb, underflow);
        if (third) phi_batch_packed<OthC, E, G, LT>(xa, own, acc, ya, underflow);
      }
    }
    double *dst = ((sg.pslot >= 0) ? a.partial + (size_t)sg.pslot * LD : a.S_own + (size_t)sg.row * LD) + g;
#if HPF_REDUCE_SCATTER
    // The 64 / G groups' accumulators, reduced HALVING the element list at every level instead of keeping every element on
    // every lane: at lane distance d a lane whose bit d is clear keeps the first half of its list and hands the second half to
    // its partner, and the other way round -- the same pairs are added as in the butterfly below (a + b on one lane is b + a
    // on the other: the same bits), but a level moves half the elements of the one before: 13 -> 7 -> 4 -> 2 (-> 1) shuffles
    // instead of 13 per level, and in the end lane (q, g) holds the totals of a contiguous run of element slots [o, o + n)
    // of column lane g, which ALL lanes then store at once.  The owner's factors follow by selects (no shuffle).
    {
      uint32_t o = 0, n = E;
      reduce_scatter<G, E, 32>(acc, own, lane, o, n);
      constexpr int NF = scatter_len<G, E>::value;          // element slots a lane ends with, at most
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const uint32_t e = o + (uint32_t)j;
        if ((uint32_t)j < n && (OwnC::fixed_ld || e * G + (uint32_t)g < LD)) dst[(size_t)e * G] = own[j] * acc[j];
      }
    }
#else
#pragma unroll
    for (int e = 0; e < E; ++e) {
      double r = acc[e];
      if (G <= 32) r += __shfl_xor(r, 32, 64);
      if (G <= 16) r += __shfl_xor(r, 16, 64);
      if (G <= 8)  r += __shfl_xor(r, 8, 64);
      if (G <= 4)  r += __shfl_xor(r, 4, 64);
      if (q == 0 && (OwnC::fixed_ld || (uint32_t)(e * G + g) < LD)) dst[(size_t)e * G] = own[e] * r;
    }
#endif
  }
}

// Three waves per SIMD (168 VGPRs).  Round 6 built and measured the other ways of spending the registers -- a fourth wave
// with the owner's factors in LDS, rows read by half their lanes (twice the nonzeros per batch, two waves) -- and an LDS pad
// that takes waves away: 3 -> 2 waves cost 7-18 %, every build that spills inside the batch loop loses far more (a scratch
// reload waits behind every gather in flight), and none reaches the one below (profiles/r06/experiments.md 2, 3;
// pass_builds_prototype.diff keeps the code).
template <template <int> class CodecT, int G, int L, int SIDE>
__global__ __launch_bounds__(256, 3) void phi_pass_packed_kernel(PhiArgs a)
{
  if (phi_pass_skips(a, SIDE)) return;
  const int lane = threadIdx.x & 63;
  const SegRange sr = seg_range(a);
  bool underflow = false;
  phi_segments<CodecT<L>, CodecT<L>, G, L, L>(a, sr, (const unsigned char *)a.W_own + (size_t)(lane % G) * 16,
                                              (const unsigned char *)a.W_oth + (size_t)(lane % G) * 16, lane, underflow);
  if (__any(underflow) && lane == 0) atomicOr(a.flags, 1u);
}

// ---------------------------------------------------------------------
// Measurement only (hpf_gather_only): a phi pass with the arithmetic taken out.  Same work
// list, same index stream, same rows, same two register sets -- the gathered pieces are folded
// with XORs into one register and nothing is written.  Its time is what the memory system
// needs for the pass's access pattern; bench.py prints it beside the pass (roofline block).
// Rows of L 16-byte pieces per lane at stride G*16: the packed layouts and plain fp64 rows with
// 16-byte loads (V = 2) alike.
// ---------------------------------------------------------------------
template <int G, int LO, int LT>
__device__ __forceinline__ void gather_only_segments(const PhiArgs &a, const SegRange &sr, const unsigned char *W_own,
                                                     const unsigned char *W_oth, int lane, uint4 &acc)
{
  constexpr int NG = 64 / G;
  constexpr uint32_t ROWO = G * LO * 16, ROWT = G * LT * 16;
  const int q = lane / G;
  auto load_own = [&](uint4 (&d)[LO], const unsigned char *base) {
#pragma unroll
    for (int t = 0; t < LO; ++t) d[t] = *reinterpret_cast<const uint4 *>(base + (size_t)t * G * 16);
  };
  auto load_row = [&](uint4 (&d)[LT], const unsigned char *base) {
#pragma unroll
    for (int t = 0; t < LT; ++t) d[t] = *reinterpret_cast<const uint4 *>(base + (size_t)t * G * 16);
  };
  auto fold = [&](const uint4 (&d)[LT]) {
#pragma unroll
    for (int t = 0; t < LT; ++t) { acc.x ^= d[t].x; acc.y ^= d[t].y; acc.z ^= d[t].z; acc.w ^= d[t].w; }
  };
  auto load_i = [&](int64_t start, uint32_t len, uint32_t o) -> uint32_t { return (o < len) ? a.idx[start + o] : 0u; };
  if (sr.s >= sr.end) return;
  Seg sgn = a.segs[sr.s];                           // the pass's stream of segments (phi_segments)
  uint32_t nxt_i = load_i(sgn.start, sgn.len, (uint32_t)lane);
  for (uint32_t s = sr.s; s < sr.end; s += sr.step) {
    const Seg sg = sgn;
    const bool more = s + sr.step < sr.end;
    if (more) sgn = a.segs[s + sr.step];
    const uint32_t len = sg.len;
    const int64_t start = sg.start;
    uint32_t cur_i = nxt_i;
    const uint32_t nb = (len + NG - 1) / NG;
    uint4 xa[LT], xb[LT];
    auto gather = [&](uint4 (&x)[LT], uint32_t b) {
      const int src = (int)((b % G) * NG) + q;
      const uint32_t in = (uint32_t)__shfl((int)cur_i, src, 64);
      load_row(x, W_oth + (size_t)in * ROWT);
    };
    auto fetch_after = [&](uint32_t c) {
      const uint32_t o = (c + 1) * 64u;
      const bool same = o < len;
      const int64_t st = same ? start : sgn.start;
      const uint32_t ln = same ? len : (more ? sgn.len : 0u), of = same ? o : 0u;
      nxt_i = load_i(st, ln, of + lane);
    };
    auto next_chunk = [&](uint32_t b) { cur_i = nxt_i; fetch_after(b / G); };
    {
      uint4 r[LO];
      load_own(r, W_own + (size_t)sg.row * ROWO); gather(xa, 0); gather(xb, 1); fetch_after(0);
#pragma unroll
      for (int t = 0; t < LO; ++t) { acc.x ^= r[t].x; acc.y ^= r[t].y; acc.z ^= r[t].z; acc.w ^= r[t].w; }
    }
    if (len == 0) continue;
    uint32_t bb = 0;
    for (; bb + 3 < nb; bb += 2) {
      fold(xa);
      __builtin_amdgcn_sched_barrier(0);
      if (((bb + 2) % G) == 0) next_chunk(bb + 2);
      gather(xa, bb + 2);
      fold(xb);
      __builtin_amdgcn_sched_barrier(0);
      gather(xb, bb + 3);
    }
    fold(xa);
    if (bb + 1 < nb) {
      const bool third = bb + 2 < nb;
      __builtin_amdgcn_sched_barrier(0);
      if (third) {
        if (((bb + 2) % G) == 0) next_chunk(bb + 2);
        gather(xa, bb + 2);
      }
      fold(xb);
      if (third) fold(xa);
    }
  }
}

template <int G, int L>
__global__ __launch_bounds__(256) void gather_only_kernel(PhiArgs a, uint32_t *sink)
{
  const int lane = threadIdx.x & 63;
  const SegRange sr = seg_range(a);
  uint4 acc = {0u, 0u, 0u, 0u};
  gather_only_segments<G, L, L>(a, sr, (const unsigned char *)a.W_own + (size_t)(lane % G) * 16,
                                (const unsigned char *)a.W_oth + (size_t)(lane % G) * 16, lane, acc);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1u;      // keeps the loads alive
}

// long rows: S[row] = sum over its segments' partials, in segment order.
// One wave per long row; a lane owns the column pairs (2 lane, 2 lane + 1) and (128 + 2 lane, ...) -- 16-byte loads
// (round 4; ld is even and rows are 16-byte aligned) -- and the slot loop is unrolled 16-fold, so 32 independent loads
// are in flight per lane (the longest row -- thousands of slots for a blockbuster item -- sets the kernel's duration:
// it is a latency chain).  The adds of a column stay in slot order.
struct ColQuad { double a0, a1, b0, b1; };
__device__ __forceinline__ ColQuad combine_columns(const double *partial, uint32_t first_slot, uint32_t nslots, uint32_t ld, uint32_t c)
{
  const bool two = c + 128 < ld;
  const double2 *p0 = reinterpret_cast<const double2 *>(partial + (size_t)first_slot * ld + c);
  const double2 *p1 = two ? p0 + 64 : p0;
  const size_t st = ld / 2;                        // row stride in double2
  ColQuad s = {0.0, 0.0, 0.0, 0.0};
  uint32_t q = 0;
  // N slots' loads in flight, then their adds in slot order
  auto round = [&](auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    double2 v0[N], v1[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { v0[j] = p0[(size_t)(q + j) * st]; v1[j] = p1[(size_t)(q + j) * st]; }
#pragma unroll
    for (int j = 0; j < N; ++j) { s.a0 += v0[j].x; s.a1 += v0[j].y; s.b0 += v1[j].x; s.b1 += v1[j].y; }
    q += N;
  };
  while (q + 16 <= nslots) round(std::integral_constant<int, 16>());
  // the last < 16 slots: a round of eight and one of four where they fit, only the last < 4 one by one (round 6: walked one
  // by one -- load, wait, add -- the last twelve of a 44-slot quarter were a chain of twelve round trips.  The same loads, the
  // same adds in the same order: the same bits.  A first attempt padded the tail to sixteen CLAMPED loads instead: the light
  // rows of two or three slots then paid for sixteen, and everything got slower -- profiles/r06/experiments.md 9)
  if (q + 8 <= nslots) round(std::integral_constant<int, 8>());
  if (q + 4 <= nslots) round(std::integral_constant<int, 4>());
  for (; q < nslots; ++q) { const double2 u = p0[(size_t)q * st], w = p1[(size_t)q * st]; s.a0 += u.x; s.a1 += u.y; s.b0 += w.x; s.b1 += w.y; }
  return s;
}

__global__ __launch_bounds__(256) void combine_partials_kernel(const LongRow *rows, uint32_t nrows,
                                                               const double *partial, double *S, uint32_t ld,
                                                               const uint32_t *flags)
{
  if (flags[0] & 6u) return;                       // the pass before it did not run (phi_pass_skips)
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t r = wave; r < nrows; r += nwaves) {
    const LongRow lr = rows[r];
    for (uint32_t c = 2 * lane; c < ld; c += 256) {
      const ColQuad s = combine_columns(partial, lr.first_slot, lr.nslots, ld, c);
      double2 *d = reinterpret_cast<double2 *>(S + (size_t)lr.row * ld + c);
      d[0] = make_double2(s.a0, s.a1);
      if (c + 128 < ld) d[64] = make_double2(s.b0, s.b1);
    }
  }
}

// The heavy rows of a TILED side carry one partial per tile they meet (C2's items 184, C4's 176): a single wave walking
// that chain is a dozen round trips in a row.  Those rows -- more than COMBINE_SPLIT partials; build_tiled_side puts them
// behind the others in the list -- get a WORKGROUP each (round 4): the four waves take a quarter of the slots each
// (contiguous quarters, cut by the slot count alone) and the four sums are added in wave order.  Which kernel a row gets,
// and hence the order of its sum, is a function of its slot count: the same bits on every run.
constexpr uint32_t COMBINE_SPLIT = 64;
__global__ __launch_bounds__(256) void combine_partials_wg_kernel(const LongRow *rows, uint32_t nrows,
                                                                  const double *partial, double *S, uint32_t ld,
                                                                  const uint32_t *flags)
{
  extern __shared__ double combine_part[];         // [4][ld]: the launch sizes it
  if (flags[0] & 6u) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double *mine = combine_part + (size_t)wv * ld;
  for (uint32_t r = blockIdx.x; r < nrows; r += gridDim.x) {
    const LongRow lr = rows[r];
    const uint32_t per = (lr.nslots + 3) / 4;
    const uint32_t q0 = min(per * wv, lr.nslots), q1 = min(q0 + per, lr.nslots);
    for (uint32_t c = 2 * lane; c < ld; c += 256) {
      const ColQuad s = combine_columns(partial, lr.first_slot + q0, q1 - q0, ld, c);
      mine[c] = s.a0; mine[c + 1] = s.a1;
      if (c + 128 < ld) { mine[c + 128] = s.b0; mine[c + 129] = s.b1; }
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < ld; c += 256)
      S[(size_t)lr.row * ld + c] = ((combine_part[c] + combine_part[ld + c]) + combine_part[2 * ld + c]) + combine_part[3 * ld + c];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------
// digamma for x > 0, in the split form the sweep needs:
//     psi(x) = log(xs) - corr ,   exp(psi(x)) = xs * exp(-corr)
// x < 10 is shifted by exactly 10 with psi(x) = psi(x+10) - P'(x)/P(x),
// P(x) = prod_{j<10}(x+j) (one reciprocal instead of ten), then the asymptotic
// series (A&S 6.3.18) through x^-14 at xs >= 10.  Stands where the reference
// calls gsl_sf_psi (gpbase.hh:260).  |err| <~ 4e-15 abs on [1e-30, inf).
// ---------------------------------------------------------------------
struct PsiParts { double xs, corr; };

__device__ __forceinline__ PsiParts psi_parts(double x)
{
  // sum_{j<10} 1/(x+j): the terms pair up, 1/(x+j) + 1/(x+9-j) = (2x+9)/(t + j(9-j))
  // with t = x(x+9), so P(x) = Q(t) = prod_{j<5}(t + c_j), c = 0, 8, 14, 18, 20, and
  // the sum is (2x+9) Q'(t)/Q(t): half the multiplies of the direct product.
  // Branch-free (round 3): 94-99 % of the shapes are below 10 (tools/shape_distribution.py),
  // so the shift is always computed and dropped by two selects when x >= 10 -- the callers'
  // column loops then stay one straight block the scheduler can interleave.
  const bool small = x < 10.0;
  const double t = x * (x + 9.0);
  double p = t, dp = 1.0;                       // j = 0: f = t
  const double cj[4] = {8.0, 14.0, 18.0, 20.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) { const double f = t + cj[j]; dp = fma(dp, f, p); p *= f; }
  dp *= fma(2.0, x, 9.0);
  // x >= 10: p may overflow to inf for x > ~1e30 (dp too): the select below drops both
  const double shift = small ? dp * fast_rcp(p) : 0.0;
  x = small ? x + 10.0 : x;
  const double xi = fast_rcp(x), x2 = xi * xi;
  double s = 1.0 / 12.0;
  s = fma(-x2, s, 691.0 / 32760.0);
  s = fma(-x2, s, 1.0 / 132.0);
  s = fma(-x2, s, 1.0 / 240.0);
  s = fma(-x2, s, 1.0 / 252.0);
  s = fma(-x2, s, 1.0 / 120.0);
  s = fma(-x2, s, 1.0 / 12.0);
  PsiParts r;
  r.xs = x;
  r.corr = fma(x2, s, fma(0.5, xi, shift));
  return r;
}

// psi_parts for the row sweep, which also needs 1 / rate of the same element: the three reciprocals -- 1/P(x),
// 1/(x + 10), 1/rate -- are ONE v_rcp_f64 (+ two Newton steps) of their product and six multiplies instead of three
// reciprocals of five instructions each (C2 user sweep 0.445 -> 0.425 ms, profiles/r04/experiments.md 8).  P(x) <= 19^10,
// the shifted x <= 20 and any rate a model can take keep the product far inside the range; where x >= 10 (no shift) P is
// replaced by 1, so that an overflowed product never enters.  ri = 1 / rt to ~2 ulp (the exported E is an IEEE
// division, materialize_es_kernel; this one feeds the column sums and W); corr as in psi_parts to a few 1e-16.
struct PsiRate { double xs, corr, ri; };

__device__ __forceinline__ PsiRate psi_parts_rate(double x, double rt)
{
  const bool small = x < 10.0;
  const double t = x * (x + 9.0);
  double p = t, dp = 1.0;
  const double cj[4] = {8.0, 14.0, 18.0, 20.0};
#pragma unroll
  for (int j = 0; j < 4; ++j) { const double f = t + cj[j]; dp = fma(dp, f, p); p *= f; }
  dp *= fma(2.0, x, 9.0);
  const double pp = small ? p : 1.0;
  const double xs = small ? x + 10.0 : x;
  const double pq = pp * xs;
  const double r = fast_rcp(pq * rt);
  PsiRate o;
  o.ri = r * pq;
  const double ipq = r * rt;                    // 1 / (pp * xs)
  const double xi = ipq * pp, x2 = xi * xi;
  const double shift = small ? dp * (ipq * xs) : 0.0;
  double sr = 1.0 / 12.0;
  sr = fma(-x2, sr, 691.0 / 32760.0);
  sr = fma(-x2, sr, 1.0 / 132.0);
  sr = fma(-x2, sr, 1.0 / 240.0);
  sr = fma(-x2, sr, 1.0 / 252.0);
  sr = fma(-x2, sr, 1.0 / 120.0);
  sr = fma(-x2, sr, 1.0 / 12.0);
  o.xs = xs;
  o.corr = fma(x2, sr, fma(0.5, xi, shift));
  return o;
}

// a * b + c with c a wave-uniform constant held in an SGPR pair
__device__ __forceinline__ double fma_uc(double a, double b, double c)
{
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}

// exp(-c) for c >= 0 (the sweep's exp(psi(shape)) = xs * exp(-corr)): n = rint(-c log2 e),
// r = -c - n ln2 in two pieces, a degree-13 Taylor polynomial on |r| <= ln2/2 (remainder
// 4e-18), scaled by 2^n with v_ldexp_f64 (which also delivers the gradual underflow);
// c > 745 gives 0.  ~1 ulp, like the library exp it replaces, without that routine's
// special-case selects for positive arguments, infinities and NaN.
__device__ __forceinline__ double exp_neg(double c)
{
  const double z = -fmin(c, 750.0);
  const double n = rint(z * 1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, z);      // ln2 high part (fdlibm split)
  r = fma(n, -1.90821492927058770002e-10, r);
  // Horner with the coefficient as the (wave-uniform) addend of a three-address v_fma_f64:
  // left to itself the compiler keeps the coefficients in VGPRs and spends a v_mov per step
  // on copying each into the accumulator of a two-address v_fmac
  double p = 1.0 / 6227020800.0;
  p = fma_uc(p, r, 1.0 / 479001600.0);
  p = fma_uc(p, r, 1.0 / 39916800.0);
  p = fma_uc(p, r, 1.0 / 3628800.0);
  p = fma_uc(p, r, 1.0 / 362880.0);
  p = fma_uc(p, r, 1.0 / 40320.0);
  p = fma_uc(p, r, 1.0 / 5040.0);
  p = fma_uc(p, r, 1.0 / 720.0);
  p = fma_uc(p, r, 1.0 / 120.0);
  p = fma_uc(p, r, 1.0 / 24.0);
  p = fma_uc(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

__device__ __forceinline__ double digamma_pos(double x)
{
  const PsiParts r = psi_parts(x);
  return log(r.xs) - r.corr;
}

// ---------------------------------------------------------------------
// K2/K3 (+K4, K5, K7): row sweep.  One G-lane group per row, column
// g + G*t in register slot t.  For every row (reference steps B/C, D, E/F of
// hgaprec.cc:1370-1414; non-hier: 944-956, 1252-1268):
//   shape = s_prior + S            (S = raw phi sums; left raw: shape and E are
//                                   materialised on demand, materialize_es_kernel)
//   rate  = prior_rate(row) + colsum_other[k]        k < K
//           r_prior + n_other_total                   bias column
//   E = shape/rate
//   W = exp(Elog)/rowmax = [xs*exp(-corr)/rate] / rowmax   (no log in the loop;
//       Elog = psi(shape) - log(rate) itself is only exported: elog_kernel)
//   hier: xi/eta update  E_prior(row) = (s0 + K*s0) / (r0 + sum_k E[row,k])
//   block partial column sums of E  (-> colsum kernel, fixed order)
// ---------------------------------------------------------------------
struct SweepArgs {
  const double *S;          // [rows x ld] raw phi sums (read only)
  void         *W;          // [rows x ld] double, or float when w32; packed / f64 rows: [rows x pk.row_bytes]
  uint32_t      w32;
  const double *prior_E;    // [rows] E[xi] / E[eta] the rate uses (hier); updated by prior_update_kernel
  double       *prior_rate; // [rows] out: rate of the xi/eta Gamma, r0 + sum_k E[row,k]
  double        psi_prior_shape; // psi(s0 + K*s0), constant, from the host (prior_update_kernel)
  const double *colsum_oth; // [ld]   sum over the other side's rows of E
  double       *colsum_used;// [ld]   out: copy of colsum_oth -- what this rate was built from (export of *_rate.tsv,
                            //        and what a W-only repeat of this sweep reads); NULL: no copy (the repeat itself)
  double       *colsum_part;// [nblocks x ld]
  uint32_t      rows, ld, K;
  PackedRow     pk;         // rows of W in 16-byte pieces (every mode but SW_PLAIN): the phi kernel's G, E, L and the row bytes
  uint32_t     *flags;      // bit 1: a nonzero W below 2^-126 was flushed by the p59 layout; bit 2: do not run
  int32_t       bias_col;   // column holding this side's bias (-1: none)
  int32_t       junk_col;   // column holding the other side's bias (-1: none)
  double        bias_rate_add;  // n_other_total for the bias column
  double        s_prior, r_prior;
  uint32_t      hier;
};

// how a sweep writes the rows of W
enum { SW_PLAIN = 0,      // double (or float, a.w32) at stride G*R: no column test
       SW_LDS_F48 = 2,    // the group builds the packed row in LDS (any G): the 48-bit layout ...
       SW_LDS_P59 = 3,    // ... and p59 where the register form below has no shape (64-lane groups of the pass)
       SW_F64 = 4,        // plain doubles into interleaved pieces (codec_f64): 8-byte stores, no staging
       SW_REG_P59 = 5 };  // p59 built in REGISTERS (round 4): the sweep's group is twice the pass's

// lane ^ GP for the power-of-two distances a sweep group spans
template <int GP>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v)
{
  if (GP == 8)  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);      // row_ror:8
  if (GP == 4)  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                        // xor 4 (bit mode)
  if (GP == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                        // xor 16
  return (uint32_t)__shfl_xor((int)v, GP, 64);
}

// the p59 shape a register-building sweep of R slots serves: the pass gives a nonzero GP = G/2 lanes with E elements
// each, the sweep's lane (gp, h) owns the elements e = h + 2t.  E <= 10 is even (L <= 5), above that odd: R names it.
template <int R> struct p59_of_slots {
  static constexpr int E = R <= 5 ? 2 * R : 2 * R - 1;
  static constexpr int L = (59 * E + 127) / 128;
  static constexpr bool valid = R != 6 && R <= 9 && codec_p59<L>::E == E;
};

// element e of a lane's dwords D (p59): its low word, and its 27-bit field OR-ed into the stream.  Compile-time places.
template <int E, int L>
__device__ __forceinline__ void p59_place(uint32_t (&D)[4 * L], int e, uint32_t lo, uint32_t f)
{
  using C = codec_p59<L>;
  const int o = 27 * e, i = E + o / 32, sh = o % 32;
  D[C::pos(e)] = lo;
  D[C::pos(i)] |= f << sh;
  if (sh > 5) D[C::pos(i + 1)] |= f >> (32 - sh);
}

// MODE = SW_PLAIN: W is stored as double (or float, a.w32) and the row stride IS G*R (hpf_create): no column test.
// Rows in pieces (every other mode): the stride a.ld = G_phi * E may be smaller than G*R; the slots from tb on then
// also hold columns past the end of the row, whose loads and stores are masked.
template <int G, int R, int MODE>
__global__ __launch_bounds__(256) void row_sweep_kernel(SweepArgs a)
{
  constexpr bool PIECES = MODE != SW_PLAIN;
  constexpr bool LDSPK = MODE == SW_LDS_F48 || MODE == SW_LDS_P59;
  const uint32_t LD = PIECES ? a.ld : (uint32_t)(G * R);
  __shared__ uint32_t pkbuf[LDSPK ? 512 * R : 1];  // 256/G groups x (<= 2*G*R dwords of packed row)
  __shared__ double red[4][G * R];       // per-wave column partials
  if (a.flags[0] & 4u) return;           // a pass has found W unusable (phi_pass_skips): nothing runs until the host has dealt with it
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane % G, q = lane / G;
  const uint32_t K = a.K;
  const uint32_t grp = (blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint32_t ngrp = (gridDim.x * blockDim.x) / G;
  if (a.colsum_used && blockIdx.x == 0)            // what this rate is built from, kept for the export and for a repeat
    for (uint32_t c = threadIdx.x; c < LD; c += blockDim.x) a.colsum_used[c] = a.colsum_oth[c];

  // Slot t of a lane is column g + G*t.  Slots below tb = K / G hold factor columns in
  // every lane: rate = prior(row) + column sum of the other side, E enters the sums -- no
  // column test at all (round 3).  The one or two slots from tb on mix factor columns with
  // this side's bias column (constant rate, E in no sum), the other side's bias slot
  // (W = 1 before the row-max scaling: Elog 0) and padding (W = 0); the branch between
  // the two bodies is wave-uniform.
  const uint32_t tb = K / G;
  double csum[R], cso[R];
#pragma unroll
  for (int t = 0; t < R; ++t) {
    const uint32_t c = g + G * t;
    csum[t] = 0.0;
    cso[t] = (c < K) ? a.colsum_oth[c] : 0.0;
  }
  const double rate_bias = a.r_prior + a.bias_rate_add;
  const double s_prior = a.s_prior;
  const bool hier = a.hier != 0;
  bool flushed = false;

  // software pipeline: the next row's raw sums and prior are loaded before this row is
  // worked on (the ~90 fp64 instructions per element then cover the load latency).
  // HPF_SWEEP_PIPE = 2 (tools/sweep_probe only) holds TWO rows ahead: measured in round 5, no gain (experiments.md)
#ifndef HPF_SWEEP_PIPE
#define HPF_SWEEP_PIPE 1
#endif
  double snx[R]; double prn = a.r_prior;
#if HPF_SWEEP_PIPE == 2
  double sn2[R]; double pr2 = a.r_prior;
#endif
  auto fetch_row = [&](uint32_t r_, double (&dst)[R], double &pdst) {
    if (r_ < a.rows) {
#pragma unroll
      for (int t = 0; t < R; ++t) dst[t] = (!PIECES || (uint32_t)(g + G * t) < LD) ? a.S[(size_t)r_ * LD + g + G * t] : 0.0;
      if (hier) pdst = a.prior_E[r_];
    }
  };
  fetch_row(grp, snx, prn);
#if HPF_SWEEP_PIPE == 2
  fetch_row(grp + ngrp, sn2, pr2);
#endif
  for (uint32_t row = grp; row < a.rows; row += ngrp) {
    const size_t base = (size_t)row * LD;
    double w[R];
#pragma unroll
    for (int t = 0; t < R; ++t) w[t] = snx[t];
    const double pr = prn;
#if HPF_SWEEP_PIPE == 2
#pragma unroll
    for (int t = 0; t < R; ++t) snx[t] = sn2[t];
    prn = pr2;
    fetch_row(row + 2 * ngrp, sn2, pr2);
#else
    fetch_row(row + ngrp, snx, prn);
#endif
    double wmax = 0.0, rsum = 0.0;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const bool mixed = (uint32_t)t >= tb;         // wave-uniform
      const uint32_t c = g + G * t;
      // GPBase::make_nonzero (gpbase.hh:27-44: "if (!(av > .0)) a = 1e-30") as one v_max_f64:
      // the same for zero, negative and NaN; a value in (0, 1e-30) -- which a shape
      // s_prior + sum >= 0.3 or a rate prior + column sum cannot take -- would become 1e-30
      const double sh = fmax(s_prior + w[t], 1e-30);
      double rt = pr + cso[t];
      // (the empty asm keeps these wave-uniform fix-ups real branches: if-converted, their
      // selects would run in every slot)
      if (mixed) { asm volatile(""); rt = (c < K) ? rt : rate_bias; }
      rt = fmax(rt, 1e-30);
      const PsiRate ps = psi_parts_rate(sh, rt);    // psi(shape) in its split form and 1 / rate, one reciprocal for both
      const double ri = ps.ri;                      // ~2 ulp; the exported E is an IEEE sh / rt
      double e = sh * ri;                           //   (materialize_es_kernel); this one feeds sums
      if (mixed) { asm volatile(""); e = (c < K) ? e : 0.0; }
      rsum += e; csum[t] += e;
      double wl = ps.xs * exp_neg(ps.corr) * ri;    // exp(psi(shape) - log(rate))
      if (mixed) { asm volatile(""); wl = (c < K || (int32_t)c == a.bias_col) ? wl : ((int32_t)c == a.junk_col ? 1.0 : 0.0); }
      w[t] = wl;
      wmax = fmax(wmax, wl);
    }
    wmax = group_max<G>(wmax);
    rsum = group_sum<G>(rsum);
    const double inv = (wmax > 0.0) ? fast_rcp(wmax) : 0.0;
    if constexpr (MODE == SW_REG_P59) {
      // The pass gives a nonzero GP = G/2 lanes; packed lane gp holds the elements e = 0..E-1 of the columns
      // e*GP + gp.  This sweep's lane g = gp + GP*h owns the slots t, i.e. the columns g + G*t = (h + 2t)*GP + gp:
      // the elements e = h + 2t of packed lane gp -- the even ones (h = 0) or the odd ones (h = 1).  The two lanes
      // swap their (low word, 27-bit field) pairs, each then holds all E elements and builds the lane's 4L dwords
      // with compile-time shifts; h = 0 stores the even 16-byte pieces, h = 1 the odd ones.  No LDS, no atomics.
      using P = p59_of_slots<R>;
      if constexpr (P::valid) {
        constexpr int E = P::E, L = P::L, GP = G / 2;
        const int gp = g % GP;
        const bool h = g >= GP;
        uint32_t D[4 * L];
#pragma unroll
        for (int i = 0; i < 4 * L; ++i) D[i] = 0u;
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const bool pair = 2 * t + 1 < E;                // element 2t + 1 exists (E odd: the last slot of h = 1 is past the row)
          const double v = w[t] * inv;
          const uint32_t vhi = (uint32_t)__double2hiint(v), vlo = (uint32_t)__double2loint(v);
          const bool live = pair || !h;
          // representable: exponent field 897..1023, i.e. 2^-126 <= w < 2 (see p59_put)
          const bool tiny = vhi < 0x38100000u || vhi >= 0x40000000u;
          flushed |= live && tiny && !(vhi == 0u && vlo == 0u);
          const uint32_t of = (tiny || !live) ? 0u : vhi - 0x38000000u, ol = (tiny || !live) ? 0u : vlo;
          const uint32_t pf = lane_xor<GP>(of), pl = lane_xor<GP>(ol);
          p59_place<E, L>(D, 2 * t, h ? pl : ol, h ? pf : of);
          if (pair) p59_place<E, L>(D, 2 * t + 1, h ? ol : pl, h ? of : pf);
        }
        // h = 0 stores the even pieces, h = 1 the odd ones.  The choice is ARITHMETIC (v_bfi_b32 under a lane mask): written
        // as a select between two elements of D the compiler turns it into a run-time index into the register array --
        // a chain of 24 compares per word
        const uint32_t hm = h ? 0xffffffffu : 0u;
        auto pick = [&](uint32_t odd, uint32_t even) -> uint32_t { return (odd & hm) | (even & ~hm); };
        uint4 *dst = reinterpret_cast<uint4 *>((unsigned char *)a.W + (size_t)row * (size_t)(GP * L * 16)) + gp + (h ? GP : 0);
#pragma unroll
        for (int j = 0; 2 * j < L; ++j) {
          const int p0 = 2 * j, p1 = 2 * j + 1;
          if (p1 < L) {
            dst[(size_t)p0 * GP] = make_uint4(pick(D[4 * p1], D[4 * p0]), pick(D[4 * p1 + 1], D[4 * p0 + 1]),
                                              pick(D[4 * p1 + 2], D[4 * p0 + 2]), pick(D[4 * p1 + 3], D[4 * p0 + 3]));
          } else if (!h) {
            dst[(size_t)p0 * GP] = make_uint4(D[4 * p0], D[4 * p0 + 1], D[4 * p0 + 2], D[4 * p0 + 3]);
          }
        }
      }
    } else if constexpr (LDSPK) {
      uint32_t *buf = pkbuf + (threadIdx.x / G) * (2 * G * R);
      packed_clear(buf, a.pk, (uint32_t)g, (uint32_t)G);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the group's lanes exchange through LDS: clear, put, copy out
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < R; ++t)
        if ((uint32_t)(g + G * t) < LD) {
          if (MODE == SW_LDS_P59) flushed |= p59_put(buf, a.pk, g + G * t, w[t] * inv);
          else f48_put(buf, a.pk, g + G * t, w[t] * inv);
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      packed_copy_out(buf, a.W, row, a.pk, (uint32_t)g, (uint32_t)G);
      __builtin_amdgcn_wave_barrier();                             // the next row's clear stays behind this copy
    } else if constexpr (MODE == SW_F64) {
#pragma unroll
      for (int t = 0; t < R; ++t)
        if ((uint32_t)(g + G * t) < a.pk.G * a.pk.E) f64_put(a.W, row, a.pk, g + G * t, (uint32_t)(g + G * t) < LD ? w[t] * inv : 0.0);
    } else if (a.w32) {
#pragma unroll
      for (int t = 0; t < R; ++t) ((float *)a.W)[base + g + G * t] = (float)(w[t] * inv);
    } else {
#pragma unroll
      for (int t = 0; t < R; ++t) ((double *)a.W)[base + g + G * t] = w[t] * inv;
    }
    // thetarate/betarate (gpbase.hh:877-889,912-925 via hgaprec.cc:1398-1414): the rate goes
    // out here; shape / rate and psi(shape) - log(rate) -- an IEEE division and a log that
    // only one lane in G would work on -- are prior_update_kernel's, over all rows at once
    if (hier && g == 0) a.prior_rate[row] = a.r_prior + rsum;
  }
  if (flushed) atomicOr(a.flags, 2u);

  // block partial column sums, fixed order: groups of a wave, then waves
#pragma unroll
  for (int t = 0; t < R; ++t) {
    double v = csum[t];
    if (G <= 32) v += __shfl_xor(v, 32, 64);
    if (G <= 16) v += __shfl_xor(v, 16, 64);
    if (G <= 8)  v += __shfl_xor(v, 8, 64);
    if (G <= 4)  v += __shfl_xor(v, 4, 64);
    if (q == 0) red[wv][g + G * t] = v;
  }
  __syncthreads();
  for (uint32_t c = threadIdx.x; c < LD; c += blockDim.x)
    a.colsum_part[(size_t)blockIdx.x * LD + c] = (c < K) ? red[0][c] + red[1][c] + red[2][c] + red[3][c] : 0.0;
}

// xi / eta after a sweep (hier): remember what the rate just used (export, ELBO), then
//   E = (s0 + K s0) / rate ,  Elog = psi(s0 + K s0) - log(rate)        gpbase.hh:877-925
// with the rate the sweep left in prior_rate.  One thread per row.
__global__ __launch_bounds__(256) void prior_update_kernel(double *prior_E, double *prior_used,
                                                           const double *prior_rate, double *prior_elog,
                                                           double *prior_elog_used, uint32_t rows,
                                                           double prior_shape, double psi_prior_shape, const uint32_t *flags)
{
  if (flags[0] & 4u) return;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const double rt = prior_rate[r];
    prior_used[r] = prior_E[r];
    prior_elog_used[r] = prior_elog[r];
    prior_E[r] = prior_shape / rt;
    prior_elog[r] = psi_prior_shape - log(rt);
  }
}

// shape = s_prior + S_raw (in place) and E = shape / rate for export, held-out
// likelihood, ranking and ELBO: the hot loop itself only needs W and the
// column sums, so the sweep does not spend 16 B/element of writes on them.
// (The sweep's own E, which only feeds the column / row sums, is sh * rcp(rate): <= 1 ulp apart.)
__global__ void materialize_es_kernel(double *S, double *E, const double *prior_used,
                                      const double *colsum_used, uint32_t rows, uint32_t ld, uint32_t K,
                                      int32_t bias_col, double bias_rate_add, double s_prior,
                                      double r_prior, uint32_t hier)
{
  const size_t n = (size_t)rows * ld;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(e / ld), c = (uint32_t)(e % ld);
    double sh = 0.0, ev = 0.0;
    if (c < K || (int32_t)c == bias_col) {
      sh = s_prior + S[e];
      double rt = (c < K) ? (hier ? prior_used[row] : r_prior) + colsum_used[c] : r_prior + bias_rate_add;
      sh = (sh > 0.0) ? sh : 1e-30;
      rt = (rt > 0.0) ? rt : 1e-30;
      ev = sh / rt;
    }
    S[e] = sh;
    E[e] = ev;
  }
}

// Elog = psi(shape) - log(rate), materialised on demand for export
// (gpbase.hh:248-262): rate rebuilt from what the last sweep used
__global__ void elog_kernel(const double *S, const double *prior_used, const double *colsum_used,
                            double *L, uint32_t rows, uint32_t ld, uint32_t K, int32_t bias_col,
                            double bias_rate_add, double r_prior, uint32_t hier)
{
  const size_t n = (size_t)rows * ld;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(e / ld), c = (uint32_t)(e % ld);
    double l = 0.0;
    if (c < K || (int32_t)c == bias_col) {
      const double sh = S[e];
      double rt = (c < K) ? (hier ? prior_used[row] : r_prior) + colsum_used[c] : r_prior + bias_rate_add;
      rt = (rt > 0.0) ? rt : 1e-30;
      l = digamma_pos(sh > 0.0 ? sh : 1e-30) - log(rt);
    }
    L[e] = l;
  }
}

// out[c] = sum over block partials: one 256-thread block per column, each
// thread sums a fixed strided subset in order, then a fixed-shape LDS tree --
// the order depends only on nblocks, so the result is bit-reproducible
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const double *part, uint32_t nblocks,
                                                              uint32_t ld, double *out, const uint32_t *flags)
{
  __shared__ double red[256];
  if (flags[0] & 4u) return;
  const uint32_t c = blockIdx.x;
  double s = 0.0;
  for (uint32_t b = threadIdx.x; b < nblocks; b += 256) s += part[(size_t)b * ld + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = red[0];
}

// plain column sums of E over rows (used once after hpf_set_state):
// block partials in the same layout as the sweep's
__global__ __launch_bounds__(256) void colsum_partial_kernel(const double *E,
    uint32_t rows, uint32_t ld, uint32_t K, double *part)
{
  // thread c accumulates column c over this block's slice of rows
  const uint32_t per = (rows + gridDim.x - 1) / gridDim.x;
  const uint32_t r0 = blockIdx.x * per;
  const uint32_t r1 = (r0 + per < rows) ? r0 + per : rows;
  for (uint32_t c = threadIdx.x; c < ld; c += blockDim.x) {
    double s = 0.0;
    if (c < K)
      for (uint32_t r = r0; r < r1; ++r) s += E[(size_t)r * ld + c];
    part[(size_t)blockIdx.x * ld + c] = s;
  }
}

// W = exp(L - rowmax(L)) over the live columns (after hpf_set_state(ELOG))
// wmode: 0 double, 1 float, WL_F48 / WL_P59 the packed layouts (built per row in LDS), WL_F64 plain doubles in
// interleaved pieces
__global__ __launch_bounds__(256) void derive_w_kernel(const double *L, void *W, uint32_t wmode, uint32_t rows,
                                uint32_t ld, uint32_t K, int32_t bias_col,
                                int32_t junk_col, PackedRow pk, uint32_t *flags)
{
  __shared__ uint32_t pkbuf[4][2304];            // a packed row per wave (<= 64 lanes x 8 pieces x 16 B + slack)
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  uint32_t *buf = pkbuf[threadIdx.x >> 6];
  const bool packed = wmode == WL_P59 || wmode == WL_F48;
  const uint32_t cols = (wmode == WL_F64) ? pk.G * pk.E : ld;          // f64 rows: the padding slots past ld hold zeros
  bool flushed = false;
  for (uint32_t row = wave; row < rows; row += nwaves) {
    double m = -1.0e308;
    for (uint32_t c = lane; c < ld; c += 64) {
      const bool live = c < K || (int32_t)c == bias_col || (int32_t)c == junk_col;
      if (live) m = fmax(m, L[(size_t)row * ld + c]);
    }
    m = group_max<64>(m);
    if (packed) {
      packed_clear(buf, pk, (uint32_t)lane, 64u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    for (uint32_t c = lane; c < cols; c += 64) {
      const bool live = c < K || (int32_t)c == bias_col || (int32_t)c == junk_col;
      const double wv = (live && c < ld) ? exp(L[(size_t)row * ld + c] - m) : 0.0;
      if (wmode == WL_P59) flushed |= p59_put(buf, pk, c, wv);
      else if (wmode == WL_F48) f48_put(buf, pk, c, wv);
      else if (wmode == WL_F64) f64_put(W, row, pk, c, wv);
      else if (wmode == 1) ((float *)W)[(size_t)row * ld + c] = (float)wv;
      else ((double *)W)[(size_t)row * ld + c] = wv;
    }
    if (packed) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      packed_copy_out(buf, W, row, pk, (uint32_t)lane, 64u);
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (flushed) atomicOr(flags, 2u);
}

// ---------------------------------------------------------------------
// K6: held-out log-likelihood per pair (rating_likelihood_hier,
// hgaprec.cc:1538-1560; rating_likelihood 1503-1536).  One 16-lane group
// per pair; the per-pair values are summed on the host in the given order.
// ---------------------------------------------------------------------
struct LLArgs {
  const uint32_t *u, *i;
  const int32_t  *y;
  uint64_t        cnt;
  const double   *Et, *Eb;     // [n x ld], [m x ld]
  const double   *logfact;     // [256]
  double         *out;         // [cnt]
  uint32_t        ld, K;
  int32_t         ubias_col, ibias_col;   // -1 without -bias
  uint32_t        binary;
};

__global__ __launch_bounds__(256) void heldout_ll_kernel(LLArgs a)
{
  constexpr int G = 16;
  const int lane = threadIdx.x & 63, g = lane % G;
  const uint64_t grp = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint64_t ngrp = ((uint64_t)gridDim.x * blockDim.x) / G;
  for (uint64_t p0 = 0; p0 < a.cnt; p0 += ngrp) {
    const uint64_t p = p0 + grp;
    const bool ok = p < a.cnt;
    const uint32_t u = ok ? a.u[p] : 0u, it = ok ? a.i[p] : 0u;
    const double *et = a.Et + (size_t)u * a.ld, *eb = a.Eb + (size_t)it * a.ld;
    double s = 0.0;
    if (ok)
      for (uint32_t c = g; c < a.K; c += G) s = fma(et[c], eb[c], s);
    s = group_sum<G>(s);
    if (ok && g == 0) {
      if (a.ubias_col >= 0) s += et[a.ubias_col] + eb[a.ibias_col];
      if (s < 1e-30) s = 1e-30;
      const uint32_t y = (uint32_t)a.y[p] & 0xffu;       // yval_t wrap
      double ll;
      if (a.binary) ll = (y == 0) ? -s : log(1.0 - exp(-s));
      else ll = (double)y * log(s) - s - a.logfact[y];
      a.out[p] = ll;
    }
  }
}

// ---------------------------------------------------------------------
// ELBO (HGAPRec::logl, hgaprec.cc:2160-2255), report-time only.
// Per nonzero the reference adds  sum_k y*sphi_k*(x_k - log sphi_k) - sum_k E_t E_b
// with sphi = yy*softmax(x), yy = (y > 1 ? y : 1); since
// x_k - log sphi_k = logsumexp(x) - log yy for every k, the first sum is
// y*yy*(logsumexp(x) - log yy).  One 16-lane group per nonzero; block partials.
// ---------------------------------------------------------------------
struct ElboNnzArgs {
  const Seg      *segs;     // the user-major work list of the phi pass: (row, start, len)
  uint32_t        nseg;
  const uint32_t *col;
  const uint8_t  *val;
  const double   *Lt, *Lb, *Et, *Eb;
  double         *partial;  // [gridDim.x]
  uint32_t        ld, K, C; // C = live columns (K or K+2)
  int32_t         ubias_col, ibias_col;
};

// one wave per segment (its user is known: no search for the owner of a
// nonzero), the wave's four 16-lane groups take the segment's nonzeros in turn
__global__ __launch_bounds__(256) void elbo_nnz_kernel(ElboNnzArgs a)
{
  constexpr int G = 16;
  __shared__ double red[256 / G];
  const int lane = threadIdx.x & 63, g = lane % G, q = lane / G;
  const uint32_t grp_in_block = threadIdx.x / G;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  for (uint32_t s = wave; s < a.nseg; s += nwaves) {
    const Seg sg = a.segs[s];
    const double *lt = a.Lt + (size_t)sg.row * a.ld, *et = a.Et + (size_t)sg.row * a.ld;
    for (uint32_t o = 0; o < sg.len; o += 64 / G) {
      const bool ok = o + q < sg.len;
      const int64_t j = sg.start + o + q;
      const uint32_t it = ok ? a.col[j] : 0u;
      const double y = (ok && a.val) ? (double)a.val[j] : 1.0;
      const double *lb = a.Lb + (size_t)it * a.ld, *eb = a.Eb + (size_t)it * a.ld;
      double mx = -1.0e308, dot = 0.0;
      for (uint32_t c = g; c < a.C; c += G) mx = fmax(mx, lt[c] + lb[c]);
      mx = group_max<G>(mx);
      double se = 0.0;
      for (uint32_t c = g; c < a.C; c += G) se += exp(lt[c] + lb[c] - mx);
      for (uint32_t c = g; c < a.K; c += G) dot = fma(et[c], eb[c], dot);
      se = group_sum<G>(se);
      dot = group_sum<G>(dot);
      if (ok && g == 0) {
        const double yy = (y > 1.0) ? y : 1.0;
        double v = y * yy * (mx + log(se) - log(yy)) - dot;
        if (a.ubias_col >= 0) v -= et[a.ubias_col] + eb[a.ibias_col];
        acc += v;
      }
    }
  }
  if (g == 0) red[grp_in_block] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (uint32_t k = 0; k < blockDim.x / G; ++k) s += red[k];
    a.partial[blockIdx.x] = s;
  }
}

// The same term from the hot loop's own arrays: W = exp(L - M) with M = the row
// maximum of L over the live columns (0 in the other side's bias slot), so
//   logsumexp(x) = log(sum_k Wt[u,k] * Wb[i,k]) + Mt[u] + Mb[i]
// -- no exp per element, one pass over the row pair instead of two.  fp64 W only.
struct ElboNnzWArgs {
  const Seg      *segs;
  uint32_t        nseg;
  const uint32_t *col;
  const uint8_t  *val;
  const double   *Wt, *Wb, *Et, *Eb;   // [rows x ld]
  const double   *Mt, *Mb;             // [rows] row maxima of Elog
  double         *partial;             // [gridDim.x]
  uint32_t        ld, K;
  int32_t         ubias_col, ibias_col;
};

__global__ __launch_bounds__(256) void elbo_nnz_w_kernel(ElboNnzWArgs a)
{
  constexpr int G = 16;
  __shared__ double red[256 / G];
  const int lane = threadIdx.x & 63, g = lane % G, q = lane / G;
  const uint32_t grp_in_block = threadIdx.x / G;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  double acc = 0.0;
  for (uint32_t s = wave; s < a.nseg; s += nwaves) {
    const Seg sg = a.segs[s];
    const double *wt = a.Wt + (size_t)sg.row * a.ld, *et = a.Et + (size_t)sg.row * a.ld;
    const double mt = a.Mt[sg.row];
    for (uint32_t o = 0; o < sg.len; o += 64 / G) {
      const bool ok = o + q < sg.len;
      const int64_t j = sg.start + o + q;
      const uint32_t it = ok ? a.col[j] : 0u;
      const double y = (ok && a.val) ? (double)a.val[j] : 1.0;
      const double *wb = a.Wb + (size_t)it * a.ld, *eb = a.Eb + (size_t)it * a.ld;
      double z = 0.0, dot = 0.0;
      for (uint32_t c = g; c < a.ld; c += G) z = fma(wt[c], wb[c], z);       // padding columns hold 0
      for (uint32_t c = g; c < a.K; c += G) dot = fma(et[c], eb[c], dot);
      z = group_sum<G>(z);
      dot = group_sum<G>(dot);
      if (ok && g == 0) {
        const double yy = (y > 1.0) ? y : 1.0;
        double v = y * yy * (log(z) + mt + a.Mb[it] - log(yy)) - dot;
        if (a.ubias_col >= 0) v -= et[a.ubias_col] + eb[a.ibias_col];
        acc += v;
      }
    }
  }
  if (g == 0) red[grp_in_block] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (uint32_t k = 0; k < blockDim.x / G; ++k) s += red[k];
    a.partial[blockIdx.x] = s;
  }
}

// M[row] = max over the live columns of Elog (the other side's bias slot counts as 0)
__global__ void rowmax_elog_kernel(const double *L, double *M, uint32_t rows, uint32_t ld, uint32_t K,
                                   int32_t bias_col, int32_t junk_col)
{
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const double *l = L + (size_t)row * ld;
  double m = -1.0e308;
  for (uint32_t c = 0; c < K; ++c) m = fmax(m, l[c]);
  if (bias_col >= 0) m = fmax(m, l[bias_col]);
  if (junk_col >= 0) m = fmax(m, 0.0);
  M[row] = m;
}

// Gamma terms of one side (compute_elbo_term_helper, gpbase.hh:360-387,717-741)
// plus, with hier, of its xi/eta array (gpbase.hh:951-969); block partials.
struct ElboGammaArgs {
  const double *S, *E, *L;               // [rows x ld] shape, E, Elog
  const double *prior_used, *prior_elog_used;   // hier: E / Elog of xi (eta) the rate used
  const double *colsum_used;             // [ld]
  const double *prior_E, *prior_elog, *prior_rate;   // the xi / eta array itself
  double       *partial;                 // [gridDim.x]
  uint32_t      rows, ld, K;
  int32_t       bias_col;
  double        bias_rate_add, s_prior, r_prior, lg_s_prior, prior_shape, lg_prior_shape;
  uint32_t      hier;
};

__global__ __launch_bounds__(256) void elbo_gamma_kernel(ElboGammaArgs a)
{
  __shared__ double red[256];
  const double s0 = a.s_prior, r0 = a.r_prior, lr0 = log(r0);
  double acc = 0.0;
  const size_t n = (size_t)a.rows * a.ld;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(e / a.ld), c = (uint32_t)(e % a.ld);
    const bool real = c < a.K, isb = (int32_t)c == a.bias_col;
    if (real || isb) {
      const double sh = a.S[e], ev = a.E[e], el = a.L[e];
      const bool h = real && a.hier;
      double b = real ? (a.hier ? a.prior_used[row] : r0) + a.colsum_used[c] : r0 + a.bias_rate_add;
      const double av = (sh > 0.0) ? sh : 1e-30;
      b = (b > 0.0) ? b : 1e-30;
      double t = s0 * (h ? a.prior_elog_used[row] : lr0) + (s0 - 1.0) * el;
      t -= (h ? a.prior_used[row] : r0) * ev + a.lg_s_prior;
      t -= av * log(b) + (av - 1.0) * el;
      t += b * ev + lgamma(av);
      acc += t;
    }
    if (a.hier && c == 0) {            // the GPArray element of this row
      const double av = a.prior_shape, b = a.prior_rate[row], ev = a.prior_E[row], el = a.prior_elog[row];
      double t = s0 * lr0 + (s0 - 1.0) * el;
      t -= r0 * ev + a.lg_s_prior;
      t -= av * log(b) + (av - 1.0) * el;
      t += b * ev + a.lg_prior_shape;
      acc += t;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) a.partial[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------
// Ranking evaluation (report steps): compute_precision / compute_itemrank
// (hgaprec.cc:1703-1848, 1606-1701).  The reference scores EVERY item for each
// sampled user (prediction_score_hier, hgaprec.cc:1966-1991), zeroes training
// and validation items, qsorts all m scores and walks the top 100.  Here:
//   score_tile_kernel   dense E_theta[sel] x E_beta^T on the fp64 matrix cores
//                       (v_mfma_f64_16x16x4_f64) -- the one GEMM-shaped piece
//   mask_scores_kernel  training (rating > 0) and validation items -> 0.0
//   topn_kernel         exact top-N per user: 8-pass radix select on the
//                       64-bit score pattern, ties by ascending item index
//                       (= glibc's stable merge-sort qsort), bitonic sort of N
//   rank_query_kernel   position of one item in the full order (itemrank)
// ---------------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

struct ScoreArgs {
  const uint32_t *users;    // [n_sel] selected user rows
  const double   *Et, *Eb;  // [n x ld], [m x ld]
  double         *scores;   // [n_sel x m]
  uint32_t        n_sel, m, ld, K;
  int32_t         ubias_col, ibias_col;
};

// block = 4 waves; wave w: 16 users x 64 items (4 MFMA tiles of 16x16)
__global__ __launch_bounds__(256) void score_tile_kernel(ScoreArgs a)
{
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const uint32_t u0 = blockIdx.y * 16, i0 = (blockIdx.x * 4 + wv) * 64;
  if (i0 >= a.m) return;
  const uint32_t usel = u0 + r16;
  const uint32_t urow = usel < a.n_sel ? a.users[usel] : 0u;
  const double *pa = a.Et + (size_t)urow * a.ld;
  const double *pb[4]; bool okb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t it = i0 + 16 * t + r16;
    okb[t] = it < a.m;
    pb[t] = a.Eb + (size_t)(okb[t] ? it : 0u) * a.ld;
  }
  double4_t acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (uint32_t k0 = 0; k0 < a.K; k0 += 4) {
    const uint32_t k = k0 + kq;
    const bool okk = k < a.K;
    const double av = (okk && usel < a.n_sel) ? pa[k] : 0.0;       // A[i = lane%16][k = lane/16]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double bv = (okk && okb[t]) ? pb[t][k] : 0.0;         // B[k = lane/16][j = lane%16]
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
    }
  }
  // D (4 doubles per lane): register r holds D[i = lane/16 + 4*r][j = lane%16]
  // (verified against a dense product in tests/test_gpu_ranking.py)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint32_t it = i0 + 16 * t + r16;
    if (it >= a.m) continue;
    const double bi = a.ibias_col >= 0 ? a.Eb[(size_t)it * a.ld + a.ibias_col] : 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t us = u0 + kq + 4 * r;
      if (us >= a.n_sel) continue;
      double v = acc[t][r];
      if (a.ubias_col >= 0) v += a.Et[(size_t)a.users[us] * a.ld + a.ubias_col] + bi;   // s += Eb_u + Eb_i
      a.scores[(size_t)us * a.m + it] = v;
    }
  }
}

// one wave per selected user: zero the scores of training items with a stored
// rating > 0 (a uint8-wrapped 0 is NOT skipped: "_ratings.r(n,m) > 0") and of
// the caller's mask list (validation items)
__global__ void mask_scores_kernel(const uint32_t *users, uint32_t n_sel, const int64_t *rowptr,
                                   const uint32_t *col, const uint8_t *val, const uint64_t *mask_ptr,
                                   const uint32_t *mask_items, double *scores, uint32_t m)
{
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t b = wave; b < n_sel; b += nwaves) {
    const uint32_t u = users[b];
    double *row = scores + (size_t)b * m;
    for (int64_t j = rowptr[u] + lane; j < rowptr[u + 1]; j += 64)
      if (!val || val[j] > 0) row[col[j]] = 0.0;
    if (mask_ptr)
      for (uint64_t j = mask_ptr[b] + lane; j < mask_ptr[b + 1]; j += 64) row[mask_items[j]] = 0.0;
  }
}

__device__ __forceinline__ unsigned long long score_key(double v)
{
  return (unsigned long long)__double_as_longlong(v);     // scores are >= +0.0: bit order = value order
}

// exact top-N of one row of scores per block (256 threads); NP = pow2 >= N
__global__ __launch_bounds__(256) void topn_kernel(const double *scores, uint32_t n_sel, uint32_t m,
                                                   uint32_t N, uint32_t NP, uint32_t *out_items,
                                                   double *out_scores)
{
  extern __shared__ unsigned char smem[];
  unsigned long long *ck = (unsigned long long *)smem;            // [NP] candidate keys
  uint32_t *ci = (uint32_t *)(ck + NP);                          // [NP] candidate items
  __shared__ uint32_t hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ uint32_t s_remaining, s_cnt, s_wtot[4], s_eq_taken;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t b = blockIdx.x;
  if (b >= n_sel) return;
  const double *row = scores + (size_t)b * m;
  const uint32_t Ne = N < m ? N : m;

  for (uint32_t k = tid; k < NP; k += 256) { ck[k] = 0ull; ci[k] = 0xffffffffu; }
  if (tid == 0) { s_prefix = 0ull; s_remaining = Ne; s_cnt = 0; s_eq_taken = 0; }
  __syncthreads();
  if (Ne > 0) {
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      hist[tid] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      for (uint32_t i = tid; i < m; i += 256) {
        const unsigned long long key = score_key(row[i]);
        if (pass == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(key >> shift) & 255ull], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t cum = 0, rem = s_remaining; int d = 255;
        for (; d > 0; --d) { if (cum + hist[d] >= rem) break; cum += hist[d]; }
        s_remaining = rem - cum;
        s_prefix = (prefix << 8) | (unsigned long long)d;
      }
      __syncthreads();
    }
    const unsigned long long T = s_prefix;       // key of the Ne-th largest score
    const uint32_t need_eq = s_remaining;         // how many of the == T to take (lowest indices)
    for (uint32_t i = tid; i < m; i += 256) {
      const unsigned long long key = score_key(row[i]);
      if (key > T) { const uint32_t p = atomicAdd(&s_cnt, 1u); ck[p] = key; ci[p] = i; }
    }
    __syncthreads();
    const uint32_t base = s_cnt;                  // = Ne - need_eq
    for (uint32_t c0 = 0; c0 < m; c0 += 256) {    // ties in ascending item order
      const uint32_t i = c0 + tid;
      const bool eq = i < m && score_key(row[i]) == T;
      const unsigned long long bal = __ballot(eq);
      const uint32_t before = __popcll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) s_wtot[wv] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t woff = 0, tot = 0;
      for (uint32_t w = 0; w < 4; ++w) { if (w < wv) woff += s_wtot[w]; tot += s_wtot[w]; }
      const uint32_t taken = s_eq_taken;
      const uint32_t pos = taken + woff + before;
      if (eq && pos < need_eq) { ck[base + pos] = T; ci[base + pos] = i; }
      __syncthreads();
      if (tid == 0) s_eq_taken = taken + tot;
      __syncthreads();
      if (s_eq_taken >= need_eq) break;
    }
    __syncthreads();
    // bitonic sort of NP entries: key descending, item ascending
    for (uint32_t k = 2; k <= NP; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        for (uint32_t t = tid; t < NP; t += 256) {
          const uint32_t x = t ^ j;
          if (x > t) {
            const bool up = (t & k) == 0;         // "up" block: best entries first
            const unsigned long long ka = ck[t], kb = ck[x]; const uint32_t ia = ci[t], ib = ci[x];
            const bool a_first = ka > kb || (ka == kb && ia < ib);
            if (a_first != up) { ck[t] = kb; ck[x] = ka; ci[t] = ib; ci[x] = ia; }
          }
        }
        __syncthreads();
      }
  }
  for (uint32_t k = tid; k < N; k += 256) {
    out_items[(size_t)b * N + k] = k < Ne ? ci[k] : 0xffffffffu;
    out_scores[(size_t)b * N + k] = k < Ne ? __longlong_as_double((long long)ck[k]) : 0.0;
  }
}

// rank position (0-based) of item q_item[q] in the full descending order of
// row q_sel[q] (ties by ascending item index); one block per query
__global__ __launch_bounds__(256) void rank_query_kernel(const double *scores, uint32_t m,
                                                         const uint32_t *q_sel, const uint32_t *q_item,
                                                         uint32_t nq, uint32_t sel0, uint32_t sel1,
                                                         uint32_t *out_rank, double *out_score)
{
  __shared__ uint32_t red[256];
  for (uint32_t q = blockIdx.x; q < nq; q += gridDim.x) {
    const uint32_t b = q_sel[q];
    if (b < sel0 || b >= sel1) continue;           // not in this batch of rows (uniform per block)
    const double *row = scores + (size_t)(b - sel0) * m;
    const uint32_t it = q_item[q];
    const unsigned long long kq = score_key(row[it]);
    uint32_t c = 0;
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
      const unsigned long long k = score_key(row[i]);
      c += (k > kq || (k == kq && i < it)) ? 1u : 0u;
    }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) { out_rank[q] = red[0]; out_score[q] = row[it]; }
    __syncthreads();
  }
}

// materialise the per-element rate matrix for export (htheta_rate.tsv):
// rate[row,k] = prior_used[row] + colsum[k]   (gpbase.hh:163-173,218-223)
__global__ void build_rate_kernel(const double *prior_used, const double *colsum,
                                  uint32_t rows, uint32_t K, double *out)
{
  const size_t n = (size_t)rows * K;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (size_t)gridDim.x * blockDim.x)
    out[e] = prior_used[e / K] + colsum[e % K];
}

// single-column staging for the bias objects (host <-> padded layout)
__global__ void column_scatter_kernel(const double *src, double *dst_col, uint32_t rows, uint32_t ld)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x)
    dst_col[(size_t)r * ld] = src[r];
}
__global__ void column_gather_kernel(const double *src_col, double *dst, uint32_t rows, uint32_t ld)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x)
    dst[r] = src_col[(size_t)r * ld];
}

}  // namespace hpf

// hpf_build.hpp -- gfx950 kernels of the device-side hand-over (hpf_upload_csr /
// hpf_upload_csr_device): the item-major (CSC) view of the ratings is built in
// HBM from the user-major CSR by a hand-written stable LSD radix sort on the
// item id, so that inside an item the users stay ascending -- the order in
// which the reference's serial loop (hgaprec.cc:1340-1345) reaches them, and
// bit for bit the order of a serial counting sort.
//
//   colptr_from_sorted    column pointers from the sorted item ids (no atomics)
//   scan_*_kernel         exclusive scan (3 phases, fixed shape -> same result every run)
//   radix_count_kernel    per wave-tile digit histogram
//   radix_scatter_kernel  stable scatter of (key, user, rating): ranks by wave ballots,
//                         per-wave running digit counters in LDS, no atomics on order
//   seg_plan / seg_fill   the phi passes' work lists (segments, long rows, two-level groups)
//   repack_*_kernel       dense [rows x cols] <-> padded [rows x ld] column block
//
// All integer / byte work, HBM-bound; nothing here is shaped for MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hpf_kernels.hpp"      // Seg, LongRow

namespace hpf {

// ---------------------------------------------------------------------
// column pointers from the SORTED item ids: colptr[i] = first position whose
// key is >= i.  No atomics: a blockbuster item (C5: one item holds 12 % of all
// nonzeros) would serialise hundreds of millions of adds on one address --
// item degrees by atomicAdd took 7.2 ms at C2, longer than the three sort passes.
// Every position that starts a run of equal keys fills the pointers of its own
// item and of the empty items before it; the last position closes the array.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void colptr_from_sorted_kernel(const uint32_t *keys, uint64_t nnz, uint32_t m,
                                                                 int64_t *colptr)
{
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz;
       j += (uint64_t)gridDim.x * blockDim.x) {
    const int64_t k = (int64_t)keys[j];
    const int64_t kp = j ? (int64_t)keys[j - 1] : -1;
    for (int64_t i = kp + 1; i <= k; ++i) colptr[i] = (int64_t)j;
    if (j + 1 == nnz)
      for (int64_t i = k + 1; i <= (int64_t)m; ++i) colptr[i] = (int64_t)nnz;
  }
}

// ---------------------------------------------------------------------
// exclusive scan of n values (IN = uint32_t or uint64_t) into uint64_t.
// Chunk = SCAN_CHUNK consecutive elements per 256-thread block.
//   1. scan_reduce_kernel : block sums
//   2. scan_spine_kernel  : one block turns the block sums into exclusive prefixes
//   3. scan_apply_kernel  : block-local exclusive scan + its prefix; out[n] = total
// in == out is allowed for IN = uint64_t (in place).
// ---------------------------------------------------------------------
constexpr int SCAN_PER_THREAD = 16;
constexpr int SCAN_CHUNK = 256 * SCAN_PER_THREAD;

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, int lane)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// inclusive scan over the 256 threads of a block; *total = block sum
__device__ __forceinline__ uint64_t block_incl_scan_u64(uint64_t v, uint64_t *total)
{
  __shared__ uint64_t wsum[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  v = wave_incl_scan_u64(v, lane);
  __syncthreads();                       // wsum may still be read by a previous call
  if (lane == 63) wsum[wv] = v;
  __syncthreads();
  uint64_t off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { if (w < wv) off += wsum[w]; tot += wsum[w]; }
  *total = tot;
  return v + off;
}

template <typename IN>
__global__ __launch_bounds__(256) void scan_reduce_kernel(const IN *in, uint64_t n, uint64_t *bsum)
{
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
  uint64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; ++k) if (base + k < n) s += (uint64_t)in[base + k];
  uint64_t tot;
  (void)block_incl_scan_u64(s, &tot);
  if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scan_spine_kernel(uint64_t *bsum, uint64_t nb)
{
  uint64_t carry = 0;
  for (uint64_t b0 = 0; b0 < nb; b0 += 256) {
    const uint64_t i = b0 + threadIdx.x;
    const uint64_t v = i < nb ? bsum[i] : 0;
    uint64_t tot;
    const uint64_t inc = block_incl_scan_u64(v, &tot);
    if (i < nb) bsum[i] = carry + inc - v;
    carry += tot;
  }
}

template <typename IN>
__global__ __launch_bounds__(256) void scan_apply_kernel(const IN *in, uint64_t n, const uint64_t *bsum,
                                                         uint64_t *out, int write_total)
{
  const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_PER_THREAD;
  uint64_t v[SCAN_PER_THREAD], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; ++k) { v[k] = (base + k < n) ? (uint64_t)in[base + k] : 0; s += v[k]; }
  uint64_t tot;
  const uint64_t inc = block_incl_scan_u64(s, &tot);
  uint64_t run = bsum[blockIdx.x] + inc - s;
#pragma unroll
  for (int k = 0; k < SCAN_PER_THREAD; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
  // the element that closes the last chunk carries the grand total
  if (write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = bsum[blockIdx.x] + tot;
}

// ---------------------------------------------------------------------
// stable LSD radix sort pass on 8-bit digits of the item id.
// A wave owns RADIX_ROUNDS * 64 consecutive elements (a tile); the element
// order inside a digit is (tile, round, lane) = the input order.
//   counts / offsets layout: [digit][tile]  (scan over it gives the global
//   position of the first element of that digit in that tile)
// ---------------------------------------------------------------------
constexpr int RADIX_BITS = 8;
constexpr int RADIX_DIGITS = 1 << RADIX_BITS;
constexpr int RADIX_ROUNDS = 64;
constexpr int RADIX_TILE = RADIX_ROUNDS * 64;

// key_limit / bad: the first pass also checks every item id (< key_limit) on its way
__global__ __launch_bounds__(256) void radix_count_kernel(const uint32_t *keys, uint64_t nnz, uint32_t shift,
                                                          uint64_t ntiles, uint64_t *counts, uint32_t key_limit,
                                                          uint32_t *bad)
{
  __shared__ uint32_t hist[4][RADIX_DIGITS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t tile = (uint64_t)blockIdx.x * 4 + wv;
  for (int d = lane; d < RADIX_DIGITS; d += 64) hist[wv][d] = 0;
  __syncthreads();
  bool oob = false;
  if (tile < ntiles) {
    const uint64_t base = tile * RADIX_TILE;
#pragma unroll 4
    for (int r = 0; r < RADIX_ROUNDS; ++r) {
      const uint64_t j = base + (uint64_t)r * 64 + lane;
      if (j < nnz) {
        const uint32_t key = keys[j];
        oob |= key >= key_limit;
        atomicAdd(&hist[wv][(key >> shift) & (RADIX_DIGITS - 1)], 1u);      // LDS, per wave: order-free integers
      }
    }
  }
  if (oob) atomicOr(bad, 1u);
  __syncthreads();
  if (tile < ntiles)
    for (int d = lane; d < RADIX_DIGITS; d += 64) counts[(uint64_t)d * ntiles + tile] = hist[wv][d];
}

// rowptr: user CSR row pointers (first pass: the user of nonzero j is found by
// bisection inside the tile's row window); users_in: payload of later passes.
struct RadixArgs {
  const uint32_t *keys_in;    // item ids in input order
  const uint32_t *users_in;   // NULL on the first pass
  const uint8_t  *vals_in;    // NULL with -binary-data
  const uint32_t *extra_in;   // second 32-bit payload (the tile sort carries the gathered index here), or NULL
  uint32_t       *extra_out;
  const int64_t  *rowptr;     // [n+1], first pass only
  uint32_t        n_rows;
  uint32_t       *keys_out;   // always written: colptr_from_sorted_kernel reads the last pass's keys
  uint32_t       *users_out;
  uint8_t        *vals_out;
  const uint64_t *offsets;    // [digit][tile] exclusive scan of the counts
  uint64_t        nnz, ntiles;
  uint32_t        shift;
};

// last row r with rowptr[r] <= j, r in [lo, hi]
__device__ __forceinline__ uint32_t row_of(const int64_t *rowptr, uint32_t lo, uint32_t hi, int64_t j)
{
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (rowptr[mid] <= j) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void radix_scatter_kernel(RadixArgs a)
{
  __shared__ uint64_t pos[4][RADIX_DIGITS];      // next output position of each digit, per wave
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t tile = (uint64_t)blockIdx.x * 4 + wv;
  if (tile >= a.ntiles) return;                  // whole wave leaves: no block-wide barrier below
  for (int d = lane; d < RADIX_DIGITS; d += 64) pos[wv][d] = a.offsets[(uint64_t)d * a.ntiles + tile];
  const uint64_t base = tile * RADIX_TILE;
  const uint64_t end = (base + RADIX_TILE < a.nnz) ? base + RADIX_TILE : a.nnz;
  uint32_t row_lo = 0, row_hi = 0;
  if (!a.users_in) {
    row_lo = row_of(a.rowptr, 0, a.n_rows - 1, (int64_t)base);
    row_hi = row_of(a.rowptr, row_lo, a.n_rows - 1, (int64_t)(end - 1));
  }
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int r = 0; r < RADIX_ROUNDS; ++r) {
    const uint64_t j = base + (uint64_t)r * 64 + lane;
    const bool ok = j < end;
    uint32_t key = 0, user = 0, val = 0, extra = 0;
    if (ok) {
      key = a.keys_in[j];
      user = a.users_in ? a.users_in[j] : row_of(a.rowptr, row_lo, row_hi, (int64_t)j);
      if (a.vals_in) val = a.vals_in[j];
      if (a.extra_in) extra = a.extra_in[j];
    }
    const uint32_t d = (key >> a.shift) & (RADIX_DIGITS - 1);
    // lanes holding the same digit (inactive lanes match nobody)
    uint64_t same = __ballot(ok);
#pragma unroll
    for (int b = 0; b < RADIX_BITS; ++b) {
      const uint64_t bal = __ballot((d >> b) & 1u);
      same &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const uint32_t before = (uint32_t)__popcll(same & lt), cnt = (uint32_t)__popcll(same);
    uint64_t p = 0;
    if (ok) p = pos[wv][d];
    // every lane has read its digit's position before the leaders advance it
    // (LDS operations of one wave complete in issue order)
    __builtin_amdgcn_wave_barrier();
    if (ok && before == 0) pos[wv][d] = p + cnt;
    __builtin_amdgcn_wave_barrier();
    if (ok) {
      p += before;
      if (a.keys_out) a.keys_out[p] = key;
      a.users_out[p] = user;
      if (a.vals_out) a.vals_out[p] = (uint8_t)val;
      if (a.extra_out) a.extra_out[p] = extra;
    }
  }
}

// ---------------------------------------------------------------------
// Work lists of a phi pass, cut on the device from a row-pointer array:
// a row of deg <= seg_max nonzeros is one segment that writes its shape row
// itself; a longer row is cut into ceil(deg / seg_max) segments that write
// partial slots; rows with more than huge_slots segments are combined in two
// levels (groups of group_slots partials, then the group sums).
//   seg_plan_kernel   per row: the five counters whose exclusive scans place its output
//   seg_fill_kernel   per row: its Seg / LongRow records at the scanned offsets
// Also validates the row pointers (monotone, first one 0).
// ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowptr_check_kernel(const int64_t *ptr, uint32_t rows, uint32_t *bad)
{
  bool b = false;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x)
    b |= ptr[r + 1] < ptr[r];
  if (b) atomicOr(bad, 1u);
}

struct SegPlan {            // per-row counters (scanned in place into offsets)
  uint64_t *nseg, *nslot, *nlong, *nhuge, *ngroup;
};

__global__ __launch_bounds__(256) void seg_plan_kernel(const int64_t *ptr, uint32_t rows, uint32_t seg_max,
                                                       uint32_t huge_slots, uint32_t group_slots, SegPlan p,
                                                       uint32_t *bad)
{
  bool b = false;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const int64_t a = ptr[r], e = ptr[r + 1];
    b |= e < a || (r == 0 && a != 0);
    const uint64_t deg = e > a ? (uint64_t)(e - a) : 0;
    const bool is_long = deg > seg_max;
    const uint64_t ns = is_long ? (deg + seg_max - 1) / seg_max : 1;
    const bool is_huge = is_long && ns > huge_slots;
    p.nseg[r] = ns;
    p.nslot[r] = is_long ? ns : 0;
    p.nlong[r] = (is_long && !is_huge) ? 1 : 0;
    p.nhuge[r] = is_huge ? 1 : 0;
    p.ngroup[r] = is_huge ? (ns + group_slots - 1) / group_slots : 0;
  }
  if (b) atomicOr(bad, 1u);
}

__global__ __launch_bounds__(256) void seg_fill_kernel(const int64_t *ptr, uint32_t rows, uint32_t seg_max,
                                                       uint32_t huge_slots, uint32_t group_slots, SegPlan off,
                                                       Seg *segs, LongRow *longs, LongRow *huges, LongRow *groups)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const int64_t a = ptr[r];
    const uint64_t deg = (uint64_t)(ptr[r + 1] - a);
    const bool is_long = deg > seg_max;
    const uint64_t ns = is_long ? (deg + seg_max - 1) / seg_max : 1;
    const uint64_t so = off.nseg[r], po = off.nslot[r];
    for (uint64_t k = 0; k < ns; ++k) {
      Seg s;
      s.start = a + (int64_t)(k * seg_max);
      s.row = r;
      s.len = is_long ? (uint32_t)((deg - k * seg_max < seg_max) ? deg - k * seg_max : seg_max) : (uint32_t)deg;
      s.pslot = is_long ? (int32_t)(po + k) : -1;
      s.pad = 0;
      segs[so + k] = s;
    }
    if (!is_long) continue;
    if (ns <= huge_slots) {
      LongRow lr; lr.row = r; lr.first_slot = (uint32_t)po; lr.nslots = (uint32_t)ns; lr.pad = 0;
      longs[off.nlong[r]] = lr;
    } else {
      const uint64_t go = off.ngroup[r], ng = (ns + group_slots - 1) / group_slots;
      LongRow top; top.row = r; top.first_slot = (uint32_t)go; top.nslots = (uint32_t)ng; top.pad = 0;
      huges[off.nhuge[r]] = top;
      for (uint64_t g = 0; g < ng; ++g) {
        LongRow gr; gr.row = (uint32_t)(go + g); gr.first_slot = (uint32_t)(po + g * group_slots);
        gr.nslots = (uint32_t)((ns - g * group_slots < group_slots) ? ns - g * group_slots : group_slots); gr.pad = 0;
        groups[go + g] = gr;
      }
    }
  }
}

// ---------------------------------------------------------------------
// Tiled phi pass (round 3): cache blocking of the gathered side.
// A phi pass whose gathered rows sit in the XCD's own 4 MiB L2 runs three times faster than one
// that has every row brought over the fabric (tools/l2probe.sh).  So the rows of the gathered
// matrix are cut into tiles an L2 holds, and the nonzeros of the owner rows that meet every tile
// often enough ("heavy" rows) are regrouped tile by tile: a work item is then a run of nonzeros of
// one owner row inside one tile, the segments of a tile are handed to ONE XCD's workgroups in
// order, and every gathered row crosses the fabric once per pass instead of once per nonzero.
// The nonzeros of the light rows keep key 0: they stay row-major and are gathered as before.
//
//   tile_map_kernel       gathered row -> tile id (1 + row / T; a table, so that another grouping is one kernel away)
//   tile_key_kernel       nonzero -> sort key: 0 for a light owner row, else the tile of its gathered row
//   (radix sort on the key, stable: inside a key the order stays owner row, then as uploaded)
//   seg_count / seg_emit  segments = maximal runs of one (key, owner row), cut at multiples of seg_max
//   seg_len_kernel        lengths from the next segment's start; first segment of every key
//   slot_plan / slot_fill per owner row: its segments in key order get consecutive partial slots
//                         (rows with one segment write S themselves, rows with none are zeroed by
//                         the combine), plus the LongRow lists of the combine kernels
// All integer work, deterministic: no atomics decide an order.
// ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_map_kernel(uint32_t *tilemap, uint32_t rows, uint32_t T)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x)
    tilemap[r] = 1u + r / T;
}

// sum of the degrees >= cutoff and how many rows have one (two integers: order-free)
__global__ __launch_bounds__(256) void deg_ge_kernel(const int64_t *ptr, uint32_t rows, uint64_t cutoff,
                                                     unsigned long long *out /* [2]: rows, nonzeros */)
{
  unsigned long long c = 0, d = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const uint64_t deg = (uint64_t)(ptr[r + 1] - ptr[r]);
    if (deg >= cutoff) { c += 1; d += deg; }
  }
  for (int o = 32; o > 0; o >>= 1) { c += __shfl_xor(c, o, 64); d += __shfl_xor(d, o, 64); }
  if ((threadIdx.x & 63) == 0 && c) { atomicAdd(out, c); atomicAdd(out + 1, d); }
}

// a wave owns RADIX_TILE consecutive nonzeros (the owner row of each by bisection in the wave's window)
__global__ __launch_bounds__(256) void tile_key_kernel(const int64_t *ptr, uint32_t rows, const uint32_t *idx, uint64_t nnz,
                                                       const uint32_t *tilemap, uint64_t light_below, uint32_t *key,
                                                       uint32_t *row_out)
{
  const int lane = threadIdx.x & 63;
  const uint64_t wt = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t base = wt * RADIX_TILE;
  if (base >= nnz) return;
  const uint64_t end = (base + RADIX_TILE < nnz) ? base + RADIX_TILE : nnz;
  const uint32_t row_lo = row_of(ptr, 0, rows - 1, (int64_t)base);
  const uint32_t row_hi = row_of(ptr, row_lo, rows - 1, (int64_t)(end - 1));
  for (uint64_t j = base + lane; j < end; j += 64) {
    const uint32_t r = row_of(ptr, row_lo, row_hi, (int64_t)j);
    const bool light = (uint64_t)(ptr[r + 1] - ptr[r]) < light_below;
    key[j] = light ? 0u : tilemap[idx[j]];
    row_out[j] = r;
  }
}

__device__ __forceinline__ bool seg_starts_at(const uint32_t *key, const uint32_t *row, uint64_t j, uint32_t seg_max)
{
  return j == 0 || key[j] != key[j - 1] || row[j] != row[j - 1] || (j % seg_max) == 0;
}

__global__ __launch_bounds__(256) void seg_count_kernel(const uint32_t *key, const uint32_t *row, uint64_t nnz,
                                                        uint32_t seg_max, uint64_t *cnt)
{
  const int lane = threadIdx.x & 63;
  const uint64_t wt = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t base = wt * RADIX_TILE;
  if (base >= nnz) return;
  const uint64_t end = (base + RADIX_TILE < nnz) ? base + RADIX_TILE : nnz;
  uint32_t c = 0;
  for (uint64_t j = base + lane; j < end; j += 64) c += seg_starts_at(key, row, j, seg_max) ? 1u : 0u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if (lane == 0) cnt[wt] = c;
}

__global__ __launch_bounds__(256) void seg_emit_kernel(const uint32_t *key, const uint32_t *row, uint64_t nnz,
                                                       uint32_t seg_max, const uint64_t *off, Seg *segs,
                                                       uint32_t *seg_row, uint32_t *seg_key)
{
  const int lane = threadIdx.x & 63;
  const uint64_t wt = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t base = wt * RADIX_TILE;
  if (base >= nnz) return;
  const uint64_t end = (base + RADIX_TILE < nnz) ? base + RADIX_TILE : nnz;
  const uint64_t lt = (1ull << lane) - 1ull;
  uint64_t next = off[wt];
  for (uint64_t j0 = base; j0 < end; j0 += 64) {
    const uint64_t j = j0 + lane;
    const bool st = j < end && seg_starts_at(key, row, j, seg_max);
    const uint64_t m = __ballot(st);
    if (st) {
      const uint64_t s = next + (uint64_t)__popcll(m & lt);
      Seg sg; sg.start = (int64_t)j; sg.row = row[j]; sg.len = 0; sg.pslot = -1; sg.pad = 0;
      segs[s] = sg;
      seg_row[s] = sg.row;
      seg_key[s] = key[j];
    }
    next += (uint64_t)__popcll(m);
  }
}

// first_seg[k] = first segment of key k (preset to 0xffffffff: the key has none)
__global__ __launch_bounds__(256) void seg_len_kernel(Seg *segs, uint32_t nseg, uint64_t nnz, const uint32_t *seg_key,
                                                      uint32_t *first_seg)
{
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < nseg; s += gridDim.x * blockDim.x) {
    const int64_t e = (s + 1 < nseg) ? segs[s + 1].start : (int64_t)nnz;
    segs[s].len = (uint32_t)(e - segs[s].start);
    if (s == 0 || seg_key[s] != seg_key[s - 1]) first_seg[seg_key[s]] = s;
  }
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t *v, uint32_t n)
{
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}

// segptr: [rows + 1] offsets into the row-sorted segment list
__global__ __launch_bounds__(256) void slot_plan_kernel(const int64_t *segptr, uint32_t rows, uint32_t huge_slots,
                                                        uint32_t group_slots, SegPlan p)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const uint64_t cnt = (uint64_t)(segptr[r + 1] - segptr[r]);
    const bool combined = cnt != 1;                        // none: the combine zeroes the row
    const bool is_huge = cnt > huge_slots;
    p.nseg[r] = cnt;
    p.nslot[r] = combined ? cnt : 0;
    p.nlong[r] = (combined && !is_huge) ? 1 : 0;
    p.nhuge[r] = is_huge ? 1 : 0;
    p.ngroup[r] = is_huge ? (cnt + group_slots - 1) / group_slots : 0;
  }
}

__global__ __launch_bounds__(256) void slot_fill_kernel(const int64_t *segptr, uint32_t rows, uint32_t huge_slots,
                                                        uint32_t group_slots, SegPlan off, const uint32_t *seg_list,
                                                        Seg *segs, LongRow *longs, LongRow *huges, LongRow *groups)
{
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const int64_t a = segptr[r];
    const uint64_t cnt = (uint64_t)(segptr[r + 1] - a);
    if (cnt == 1) { segs[seg_list[a]].pslot = -1; continue; }
    const uint64_t po = off.nslot[r];
    for (uint64_t k = 0; k < cnt; ++k) segs[seg_list[a + (int64_t)k]].pslot = (int32_t)(po + k);
    if (cnt <= huge_slots) {
      LongRow lr; lr.row = r; lr.first_slot = (uint32_t)po; lr.nslots = (uint32_t)cnt; lr.pad = 0;
      longs[off.nlong[r]] = lr;
    } else {
      const uint64_t go = off.ngroup[r], ng = (cnt + group_slots - 1) / group_slots;
      LongRow top; top.row = r; top.first_slot = (uint32_t)go; top.nslots = (uint32_t)ng; top.pad = 0;
      huges[off.nhuge[r]] = top;
      for (uint64_t g = 0; g < ng; ++g) {
        LongRow gr; gr.row = (uint32_t)(go + g); gr.first_slot = (uint32_t)(po + g * group_slots);
        gr.nslots = (uint32_t)((cnt - g * group_slots < group_slots) ? cnt - g * group_slots : group_slots); gr.pad = 0;
        groups[go + g] = gr;
      }
    }
  }
}

// ---------------------------------------------------------------------
// dense [rows x cols] block  <->  columns [col0, col0 + cols) of a padded
// [rows x ld] matrix, rows [r0, r0 + rows).  dense is contiguous.
// ---------------------------------------------------------------------
__global__ void repack_in_kernel(const double *dense, double *padded, uint64_t rows, uint32_t cols,
                                 uint32_t ld, uint32_t col0)
{
  const uint64_t n = rows * cols;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = e / cols; const uint32_t c = (uint32_t)(e % cols);
    padded[r * ld + col0 + c] = dense[e];
  }
}
__global__ void repack_out_kernel(const double *padded, double *dense, uint64_t rows, uint32_t cols,
                                  uint32_t ld, uint32_t col0)
{
  const uint64_t n = rows * cols;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
       e += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = e / cols; const uint32_t c = (uint32_t)(e % cols);
    dense[e] = padded[r * ld + col0 + c];
  }
}

}  // namespace hpf

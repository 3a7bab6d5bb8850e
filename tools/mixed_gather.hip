// mixed_gather.hip -- what the machine gives the access pattern of a TILED phi pass: whole-row gathers of which a share
// `f` hits the L2 of the XCD the wave runs on (a hot set of rows that fits it) and the rest comes over the fabric (a cold
// matrix far larger than the Infinity Cache).  Everything else as in tools/gather_ceiling.hip: G lanes per row, L 16-byte
// pieces per lane, two row-loads in flight per wave, rows folded into one register, indices from a counter hash.
//
// Two ways of mixing: "per gather" -- every gather of every lane group is hot with probability f (what a batch of a tiled
// pass looks like: a few of its rows miss) -- and "per wave" -- a share f of the WAVES reads nothing but hot rows, the
// others nothing but cold ones (hot and cold work kept apart in time, side by side on every CU).
//
// Two models to hold the result against, with B_hot / B_cold the bytes of either kind and the two pure rates measured by
// the same kernel (f = 1 and f = 0):   overlap  T = max(B_hot / R_hot, B_cold / R_cold)     the two paths run side by side
//                                      serial   T = B_hot / R_hot + B_cold / R_cold         one pipeline serves both
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/mixed_gather tools/mixed_gather.hip && tools/mixed_gather
//
// profiles/r06/experiments.md holds the figures next to the tiled passes' own L2-side rates.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (uint32_t)x;
}

// MODE 0: per gather, 1: per wave.  thr = f * 2^32 (hot when the hash falls below it); f = 1 is thr = 0xffffffff.
template <int G, int L, int MODE>
__global__ __launch_bounds__(256) void mixed_kernel(const unsigned char *hot, uint32_t hot_rows, const unsigned char *cold, uint32_t cold_rows,
                                                    uint32_t thr, uint64_t gathers_per_group, uint32_t *sink)
{
  constexpr uint32_t ROWB = G * L * 16;
  const int lane = threadIdx.x & 63, g = lane % G;
  const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
  const bool wave_hot = mix(wave * 0xD6E8FEB86659FD93ull + 17) <= thr;
  uint4 acc = {0, 0, 0, 0};
  uint4 a[L], b[L];
  auto load = [&](uint4 (&x)[L], uint64_t k) {
    const uint32_t h = mix(group * 0x9E3779B97F4A7C15ull + k);
    const bool is_hot = MODE == 1 ? wave_hot : (mix(h ^ 0xA5A5A5A5u) <= thr);
    const unsigned char *p = (is_hot ? hot + (size_t)(h % hot_rows) * ROWB : cold + (size_t)(h % cold_rows) * ROWB) + (size_t)g * 16;
#pragma unroll
    for (int t = 0; t < L; ++t) x[t] = *reinterpret_cast<const uint4 *>(p + (size_t)t * G * 16);
  };
  auto fold = [&](const uint4 (&x)[L]) {
#pragma unroll
    for (int t = 0; t < L; ++t) { acc.x ^= x[t].x; acc.y ^= x[t].y; acc.z ^= x[t].z; acc.w ^= x[t].w; }
  };
  load(a, 0); load(b, 1);
  for (uint64_t k = 0; k + 3 < gathers_per_group; k += 2) {
    fold(a); __builtin_amdgcn_sched_barrier(0); load(a, k + 2);
    fold(b); __builtin_amdgcn_sched_barrier(0); load(b, k + 3);
  }
  fold(a); fold(b);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int G, int L, int MODE>
double run(const unsigned char *hot, uint32_t hot_rows, const unsigned char *cold, uint32_t cold_rows, double f, uint64_t total, uint32_t blocks,
           uint32_t *sink)
{
  const uint64_t groups = (uint64_t)blocks * 256 / G, per = total / groups;
  const uint32_t thr = f >= 1.0 ? 0xffffffffu : (uint32_t)(f * 4294967296.0);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((mixed_kernel<G, L, MODE>), dim3(blocks), dim3(256), 0, 0, hot, hot_rows, cold, cold_rows, thr, per, sink);
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mixed_kernel<G, L, MODE>), dim3(blocks), dim3(256), 0, 0, hot, hot_rows, cold, cold_rows, thr, per, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return 3.0 * (double)per * (double)groups * G * L * 16 / (ms * 1e-3) / 1e9;
}

template <int G, int L>
void sweep(const char *name, const unsigned char *hot, const unsigned char *cold, size_t hot_bytes, size_t cold_bytes, uint32_t *sink, bool &first)
{
  const uint32_t hot_rows = (uint32_t)(hot_bytes / (G * L * 16)), cold_rows = (uint32_t)(cold_bytes / (G * L * 16));
  const uint64_t total = 400000000ull * 128 / (G * L * 16);       // ~51 GB of rows per launch
  const double fs[8] = {0.0, 0.25, 0.5, 0.66, 0.73, 0.85, 0.95, 1.0};
  double r_cold = 0, r_hot = 0;
  for (int pass = 0; pass < 2; ++pass)          // the two pure rates first
    for (uint32_t blocks : {3072u})
      (pass ? r_hot : r_cold) = run<G, L, 0>(hot, hot_rows, cold, cold_rows, pass ? 1.0 : 0.0, total, blocks, sink);
  for (double f : fs) {
    const double g0 = run<G, L, 0>(hot, hot_rows, cold, cold_rows, f, total, 3072, sink);
    const double g1 = run<G, L, 1>(hot, hot_rows, cold, cold_rows, f, total, 3072, sink);
    const double g0_6 = run<G, L, 0>(hot, hot_rows, cold, cold_rows, f, total, 6144, sink);
    const double serial = 1.0 / (f / r_hot + (1.0 - f) / r_cold);
    const double overlap = 1.0 / (f / r_hot > (1.0 - f) / r_cold ? f / r_hot : (1.0 - f) / r_cold);
    printf("%s {\"rows\": \"%s\", \"hot_share\": %.2f, \"per_gather_GBps\": %.0f, \"per_gather_6waves_GBps\": %.0f, \"per_wave_GBps\": %.0f, "
           "\"model_serial_GBps\": %.0f, \"model_overlap_GBps\": %.0f}", first ? " " : ",\n ", name, f, g0, g0_6, g1, serial, overlap);
    first = false;
    fflush(stdout);
  }
}

int main()
{
  uint32_t *sink; CHECK(hipMalloc(&sink, 4));
  const size_t hot_bytes = (size_t)3 << 20, cold_bytes = (size_t)1536 << 20;      // 3 MiB: inside one 4 MiB L2; 1.5 GiB: HBM
  unsigned char *hot, *cold;
  CHECK(hipMalloc(&hot, hot_bytes)); CHECK(hipMemset(hot, 1, hot_bytes));
  CHECK(hipMalloc(&cold, cold_bytes)); CHECK(hipMemset(cold, 2, cold_bytes));
  printf("{\"what\": \"whole-row gathers, a share from a 3 MiB hot set (every XCD's L2 holds it), the rest from 1.5 GiB (HBM); tools/mixed_gather.hip\", "
         "\"results\": [\n");
  bool first = true;
  sweep<16, 6>("1536 B (K = 200)", hot, cold, hot_bytes, cold_bytes, sink, first);
  sweep<8, 6>("768 B (K = 100)", hot, cold, hot_bytes, cold_bytes, sink, first);
  sweep<4, 6>("384 B (K = 50)", hot, cold, hot_bytes, cold_bytes, sink, first);
  printf("\n]}\n");
  return 0;
}

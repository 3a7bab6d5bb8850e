// gather_ceiling.hip -- what the machine gives a kernel that does NOTHING but the phi pass's
// access pattern: groups of G lanes read whole rows of `row_bytes` (L 16-byte pieces per lane,
// interleaved like the packed W rows) at random row indices, two row-loads in flight per wave,
// and fold them into one register.  No arithmetic worth the name, no index stream from memory
// (indices come from a counter hash), 3 or 6 waves per SIMD.  Prints GB/s of row bytes for
// matrices inside one L2, inside the Infinity Cache and in HBM.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_ceiling tools/gather_ceiling.hip && tools/gather_ceiling
//
// DESIGN.md section 6 quotes it next to the phi passes' 7.2-7.5 TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (uint32_t)x;
}

template <int G, int L>
__global__ __launch_bounds__(256) void gather_kernel(const unsigned char *W, uint32_t rows, uint64_t gathers_per_group,
                                                     uint32_t *sink)
{
  constexpr uint32_t ROWB = G * L * 16;
  const int lane = threadIdx.x & 63, g = lane % G;
  const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const unsigned char *base = W + (size_t)g * 16;
  uint4 acc = {0, 0, 0, 0};
  uint4 a[L], b[L];
  auto load = [&](uint4 (&x)[L], uint64_t k) {
    const uint32_t r = mix(group * 0x9E3779B97F4A7C15ull + k) % rows;
    const unsigned char *p = base + (size_t)r * ROWB;
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
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;      // keep the loads alive
}

template <int G, int L>
double run(const unsigned char *W, uint32_t rows, uint64_t total_gathers, uint32_t blocks, uint32_t *sink)
{
  const uint64_t groups = (uint64_t)blocks * 256 / G;
  const uint64_t per = total_gathers / groups;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((gather_kernel<G, L>), dim3(blocks), dim3(256), 0, 0, W, rows, per, sink);   // warm-up
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gather_kernel<G, L>), dim3(blocks), dim3(256), 0, 0, W, rows, per, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 3.0 * (double)per * (double)groups * G * L * 16;
  return bytes / (ms * 1e-3) / 1e9;
}

int main()
{
  uint32_t *sink; CHECK(hipMalloc(&sink, 4));
  const size_t cap = (size_t)2 << 30;                      // 2 GiB arena
  unsigned char *W; CHECK(hipMalloc(&W, cap)); CHECK(hipMemset(W, 1, cap));
  const uint64_t total = 50000000ull;                      // gathers per launch, like C2's nonzeros
  printf("{\"what\": \"random whole-row gathers, nothing else (tools/gather_ceiling.hip)\", \"gathers_per_launch\": %llu, \"results\": [\n",
         (unsigned long long)total);
  const uint32_t blocks_list[2] = {3072, 6144};            // 3 and 6 waves per SIMD resident
  struct Case { const char *name; int L; size_t bytes; } cases[] = {
    {"packed rows 768 B, 3.6 MB matrix (one L2)", 6, (size_t)3600000},
    {"packed rows 768 B, 77 MB matrix (Infinity Cache)", 6, (size_t)77000000},
    {"packed rows 768 B, 768 MB matrix (HBM)", 6, (size_t)768000000},
    {"plain rows 896 B, 90 MB matrix (Infinity Cache)", 7, (size_t)90000000},
    {"plain rows 896 B, 896 MB matrix (HBM)", 7, (size_t)896000000},
    {"48-bit rows 640 B, 640 MB matrix (HBM)", 5, (size_t)640000000},
  };
  bool first = true;
  for (const Case &c : cases)
    for (uint32_t blocks : blocks_list) {
      const uint32_t rows = (uint32_t)(c.bytes / (8 * c.L * 16));
      double gbs = 0;
      if (c.L == 6) gbs = run<8, 6>(W, rows, total, blocks, sink);
      else if (c.L == 7) gbs = run<8, 7>(W, rows, total, blocks, sink);
      else gbs = run<8, 5>(W, rows, total, blocks, sink);
      printf("%s {\"case\": \"%s\", \"waves_per_simd\": %u, \"row_GBps\": %.0f}", first ? " " : ",\n ", c.name, blocks / 1024, gbs);
      first = false;
    }
  printf("\n]}\n");
  return 0;
}

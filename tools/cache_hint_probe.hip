// cache_hint_probe.hip -- do the load cache-policy bits of gfx950 let a kernel keep a hot set in L2
// while it streams cold rows past it?  Half of the waves gather whole 768-byte rows from a HOT
// matrix (3 MB: fits every XCD's 4 MiB L2), the other half from a COLD one (768 MB, HBM); the cold
// waves load with one of: plain, nt, sc1, sc0 sc1, nt sc0 sc1 (buffer loads, aux bits 0 = sc0,
// 1 = nt, 4 = sc1).  If a policy protected the hot set, the hot waves' rate would go up and the
// total with it.  (It does not: see profiles/r03/experiments.md.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/cache_hint_probe tools/cache_hint_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint64_t x)
{
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (uint32_t)x;
}

template <int AUX>
__global__ __launch_bounds__(256) void probe(const unsigned char *hot, uint32_t hot_rows, const unsigned char *cold,
                                             uint32_t cold_rows, uint32_t per_group, unsigned long long *clk, uint32_t *sink,
                                             uint32_t hot_every)
{
  constexpr int G = 8, L = 6; constexpr uint32_t ROWB = G * L * 16;
  const int lane = threadIdx.x & 63, g = lane % G;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const bool is_cold = hot_every == 0 || (wave % hot_every) != 0;      // one wave in hot_every gathers from the hot matrix
  const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(is_cold ? cold : hot), 0,
                                                                       (int)((is_cold ? cold_rows : hot_rows) * ROWB), 0x00020000);
  const uint32_t rows = is_cold ? cold_rows : hot_rows;
  u4 acc = {0, 0, 0, 0};
  u4 a[L], b[L];
  auto load = [&](u4 (&x)[L], uint32_t k) {
    const uint32_t r = mix(group * 0x9E3779B97F4A7C15ull + k) % rows;
    const uint32_t off = r * ROWB + (uint32_t)g * 16;
    const unsigned char *gp = (is_cold ? cold : hot) + off;
#pragma unroll
    for (int t = 0; t < L; ++t) {
      if (AUX >= 100) {                      // global loads: 100 plain, 102 nt
        const u4 *p = reinterpret_cast<const u4 *>(gp + t * G * 16);
        if (AUX == 102 && is_cold) x[t] = __builtin_nontemporal_load(p);   // is_cold is wave-uniform: a real branch
        else x[t] = *p;
      } else {
        x[t] = is_cold ? __builtin_amdgcn_raw_buffer_load_b128(rs, off + t * G * 16, 0, AUX)
                       : __builtin_amdgcn_raw_buffer_load_b128(rs, off + t * G * 16, 0, 0);
      }
    }
  };
  auto fold = [&](const u4 (&x)[L]) {
#pragma unroll
    for (int t = 0; t < L; ++t) acc ^= x[t];
  };
  const unsigned long long t0 = wall_clock64();
  load(a, 0); load(b, 1);
  for (uint32_t k = 0; k + 3 < per_group; k += 2) {
    fold(a); __builtin_amdgcn_sched_barrier(0); load(a, k + 2);
    fold(b); __builtin_amdgcn_sched_barrier(0); load(b, k + 3);
  }
  fold(a); fold(b);
  const unsigned long long t1 = wall_clock64();
  if (lane == 0) atomicAdd(&clk[is_cold ? 1 : 0], t1 - t0);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int AUX>
void run(const char *name, const unsigned char *hot, uint32_t hr, const unsigned char *cold, uint32_t cr, unsigned long long *clk,
         uint32_t *sink, bool last, uint32_t hot_every = 2)
{
  const uint32_t blocks = 3072, per = 1500;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(probe<AUX>, dim3(blocks), dim3(256), 0, 0, hot, hr, cold, cr, per, clk, sink, hot_every);
  CHECK(hipMemset(clk, 0, 16));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<AUX>, dim3(blocks), dim3(256), 0, 0, hot, hr, cold, cr, per, clk, sink, hot_every);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c[2]; CHECK(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
  const double bytes = (double)blocks * 256 / 8 * per * 768.0;
  // mean wall-clock ticks (100 MHz) a hot / a cold wave needed for its gathers
  const double nw = blocks * 4 / 2.0;
  printf("  {\"hot_every\": %u, \"cold_loads\": \"%s\", \"total_GBps\": %.0f, \"kernel_ms\": %.3f, \"hot_wave_us\": %.1f, \"cold_wave_us\": %.1f}%s\n", hot_every, name,
         bytes / (ms * 1e-3) / 1e9, ms, c[0] / nw / 100.0, c[1] / nw / 100.0, last ? "" : ",");
}

int main()
{
  uint32_t *sink; CHECK(hipMalloc(&sink, 4));
  unsigned long long *clk; CHECK(hipMalloc(&clk, 16));
  const uint32_t hr = 3u * 1024 * 1024 / 768, cr = 1000000;
  unsigned char *hot, *cold;
  CHECK(hipMalloc(&hot, (size_t)hr * 768)); CHECK(hipMalloc(&cold, (size_t)cr * 768));
  CHECK(hipMemset(hot, 1, (size_t)hr * 768)); CHECK(hipMemset(cold, 2, (size_t)cr * 768));
  printf("{\"what\": \"half the waves gather 768-byte rows from a 3 MB hot matrix (plain loads), half from a 768 MB cold one with the policy named\", \"results\": [\n");
  run<0>("plain", hot, hr, cold, cr, clk, sink, false);
  run<2>("nt", hot, hr, cold, cr, clk, sink, false);
  run<16>("sc1", hot, hr, cold, cr, clk, sink, false);
  run<17>("sc0 sc1", hot, hr, cold, cr, clk, sink, false);
  run<19>("nt sc0 sc1", hot, hr, cold, cr, clk, sink, false);
  run<0>("plain", hot, hr, cold, cr, clk, sink, false, 0);          // no hot waves at all: what the hint costs on pure misses
  run<2>("nt", hot, hr, cold, cr, clk, sink, false, 0);
  run<0>("plain", hot, hr, cold, cr, clk, sink, false, 4);          // a quarter of the waves hot
  run<2>("nt", hot, hr, cold, cr, clk, sink, false, 4);
  run<100>("global plain", hot, hr, cold, cr, clk, sink, false, 0);
  run<102>("global nt", hot, hr, cold, cr, clk, sink, false, 0);
  run<100>("global plain", hot, hr, cold, cr, clk, sink, false, 2);
  run<102>("global nt", hot, hr, cold, cr, clk, sink, false, 2);
  // round 4: the policy as a property of the ALLOCATION instead of the instruction -- the cold matrix in memory the
  // L2 does not keep (hipDeviceMallocUncached), or fine-grained; plain loads.  A per-nonzero choice between a cached
  // and an uncached copy of a row would be a choice of ADDRESS: no second load form, no divergence.
  for (int kind = 0; kind < 2; ++kind) {
    unsigned char *c2 = nullptr;
    const unsigned flag = kind == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
    const char *nm = kind == 0 ? "plain, cold rows in UNCACHED memory" : "plain, cold rows in FINE-GRAINED memory";
    if (hipExtMallocWithFlags((void **)&c2, (size_t)cr * 768, flag) != hipSuccess) { printf("  {\"cold_loads\": \"%s\", \"error\": \"allocation failed\"},\n", nm); (void)hipGetLastError(); continue; }
    CHECK(hipMemset(c2, 2, (size_t)cr * 768));
    run<0>(nm, hot, hr, c2, cr, clk, sink, false, 2);
    run<0>(nm, hot, hr, c2, cr, clk, sink, false, 4);
    run<0>(nm, hot, hr, c2, cr, clk, sink, false, 0);
    run<100>(nm, hot, hr, c2, cr, clk, sink, false, 2);
    CHECK(hipFree(c2));
  }
  run<0>("plain", hot, hr, cold, cr, clk, sink, true, 2);
  printf("]}\n");
  return 0;
}

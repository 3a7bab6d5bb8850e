// xcd_fabric_probe -- how much of the fabric's L2-miss fill rate can a SUBSET of the eight XCDs
// pull?  The gather kernel of gather_ceiling.hip (random 768-byte rows out of 768 MB, two row-loads
// in flight per wave, 3 waves per SIMD), with the workgroups of the XCDs outside `mask` leaving at
// once (workgroup b runs on XCD b % 8, tools/xcc_probe).  Decides how a tiled pass should lay its
// fabric-bound and its L2-bound work over the XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcd_fabric_probe tools/xcd_fabric_probe.hip && tools/xcd_fabric_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t mix(uint64_t x)
{
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
  return (uint32_t)x;
}
__global__ __launch_bounds__(256) void gather_kernel(const unsigned char *W, uint32_t rows, uint32_t per_group, uint32_t mask,
                                                     uint32_t *sink)
{
  constexpr int G = 8, L = 6;
  constexpr uint32_t ROWB = G * L * 16;
  if (!((mask >> (blockIdx.x & 7)) & 1u)) return;
  const int lane = threadIdx.x & 63, g = lane % G;
  const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const unsigned char *base = W + (size_t)g * 16;
  uint4 acc = {0, 0, 0, 0}, a[L], b[L];
  auto load = [&](uint4 (&x)[L], uint32_t k) {
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
  for (uint32_t k = 0; k + 3 < per_group; k += 2) {
    fold(a); __builtin_amdgcn_sched_barrier(0); load(a, k + 2);
    fold(b); __builtin_amdgcn_sched_barrier(0); load(b, k + 3);
  }
  fold(a); fold(b);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}
int main()
{
  uint32_t *sink; CHECK(hipMalloc(&sink, 4));
  const size_t bytes = (size_t)768 << 20;
  unsigned char *W; CHECK(hipMalloc(&W, bytes)); CHECK(hipMemset(W, 1, bytes));
  const uint32_t rows = (uint32_t)(bytes / 768), blocks = 3072, per = 600;
  const uint32_t masks[] = {0x01, 0x03, 0x0f, 0x55, 0x3f, 0xff};
  printf("{\"what\": \"random 768-byte rows out of 768 MB, gathered by a subset of the XCDs\", \"results\": [\n");
  for (unsigned i = 0; i < sizeof masks / sizeof *masks; ++i) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, W, rows, per, masks[i], sink);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, 0, W, rows, per, masks[i], sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const int nx = __builtin_popcount(masks[i]);
    const double b = 3.0 * (double)per * ((double)blocks * nx / 8 * 256 / 8) * 768;
    printf("  {\"xcd_mask\": \"0x%02x\", \"xcds\": %d, \"row_GBps\": %.0f, \"per_xcd_GBps\": %.0f}%s\n", masks[i], nx,
           b / (ms * 1e-3) / 1e9, b / (ms * 1e-3) / 1e9 / nx, i + 1 < sizeof masks / sizeof *masks ? "," : "");
  }
  printf("]}\n");
  return 0;
}

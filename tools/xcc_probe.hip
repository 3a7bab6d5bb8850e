// xcc_probe -- on which XCD does workgroup b run?  The tiled phi pass hands tile queue b % 8 to
// workgroup b on the assumption that the dispatcher deals workgroups to the eight XCDs in turn,
// for the whole grid and not just its first wave.  This reads HW_REG_XCC_ID in every workgroup of
// a grid many times larger than the machine holds, with workgroups of very unequal length, and
// counts how many ran where b % 8 says.
//   hipcc --offload-arch=gfx950 -O2 -o tools/xcc_probe tools/xcc_probe.hip && tools/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(uint32_t *xcc, uint32_t *sink, uint32_t spin)
{
  const uint32_t id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // XCC_ID[3:0]
  uint32_t v = threadIdx.x, n = spin * (1u + (blockIdx.x * 2654435761u >> 28));  // 1..16 x spin
  for (uint32_t i = 0; i < n; ++i) v = v * 1664525u + 1013904223u;
  if (threadIdx.x == 0) xcc[blockIdx.x] = id;
  if (v == 0x12345u) sink[0] = v;
}
int main(int argc, char **argv)
{
  const uint32_t blocks = argc > 1 ? (uint32_t)atoi(argv[1]) : 200000, spin = argc > 2 ? (uint32_t)atoi(argv[2]) : 2000;
  uint32_t *d = nullptr, *sink = nullptr;
  if (hipMalloc(&d, blocks * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, d, sink, spin);
  std::vector<uint32_t> h(blocks);
  if (hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  uint64_t match = 0, hist[16] = {0};
  for (uint32_t b = 0; b < blocks; ++b) { match += (h[b] & 15u) == (b & 7u); hist[h[b] & 15u]++; }
  printf("{\"blocks\": %u, \"xcc_equals_b_mod_8\": %.6f, \"per_xcc\": [", blocks, (double)match / blocks);
  for (int x = 0; x < 8; ++x) printf("%llu%s", (unsigned long long)hist[x], x < 7 ? ", " : "]}\n");
  return 0;
}

// tools/sweep_probe.hip -- the row sweep on its own (measurement tool, not part of the library): the kernel of
// hpf_kernels.hpp at C2's user-side shape (1M rows, K = 100, ld = 104; G = 16, R = 7) in each way of writing W --
// plain rows, p59 built in LDS (round 3), p59 built in registers (round 4), plain doubles in pieces.  Prints ms per launch and the bytes moved.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DHPF_SWEEP_PIPE=2] -o tools/sweep_probe tools/sweep_probe.hip && tools/sweep_probe [rows [blocks]]
#include "../hgaprec_amd/csrc/hpf_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace hpf;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int G, int R, int MODE>
float run(SweepArgs a, uint32_t blocks, int reps)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((row_sweep_kernel<G, R, MODE>), dim3(blocks), dim3(256), 0, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((row_sweep_kernel<G, R, MODE>), dim3(blocks), dim3(256), 0, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const uint32_t rows = argc > 1 ? (uint32_t)atoi(argv[1]) : 1000000u, K = 100, ld = 104;
  const uint32_t blocks = argc > 2 ? (uint32_t)atoi(argv[2]) : 2048u;     // the library's HPF_SWEEP_BLOCKS default
  constexpr int G = 16, R = 7;
  std::vector<double> S((size_t)rows * ld), cs(ld, 3.0e4);
  unsigned long long x = 88172645463325252ull;
  for (auto &v : S) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) * (1.0 / 9007199254740992.0) * 40.0; }
  double *dS, *dcs, *dpart, *dprior, *drate, *dused; void *dW; uint32_t *dflags;
  CK(hipMalloc(&dS, S.size() * 8)); CK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dcs, ld * 8)); CK(hipMemcpy(dcs, cs.data(), ld * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dpart, (size_t)blocks * 112 * 8)); CK(hipMalloc(&dprior, (size_t)rows * 8)); CK(hipMalloc(&drate, (size_t)rows * 8));
  CK(hipMalloc(&dused, ld * 8)); CK(hipMalloc(&dW, (size_t)rows * 896));
  CK(hipMalloc(&dflags, 16)); CK(hipMemset(dflags, 0, 16));
  CK(hipMemset(dprior, 0, (size_t)rows * 8));
  SweepArgs a;
  a.S = dS; a.W = dW; a.w32 = 0; a.prior_E = dprior; a.prior_rate = drate; a.psi_prior_shape = 0.0;
  a.colsum_oth = dcs; a.colsum_used = dused; a.colsum_part = dpart; a.rows = rows; a.ld = ld; a.K = K;
  a.pk = {8, 13, 6, 768, 3}; a.flags = dflags;
  const PackedRow pkf = {8, 14, 7, 896, 3};
  a.bias_col = -1; a.junk_col = -1; a.bias_rate_add = 0.0; a.s_prior = 0.3; a.r_prior = 0.3; a.hier = 1;
  const int reps = 20;
  printf("{\"rows\": %u, \"ld\": %u, \"blocks\": %u, \"rows_ahead\": %d", rows, ld, blocks, HPF_SWEEP_PIPE);
  {                                   // plain rows want ld = G*R = 112: same S buffer read at that stride over fewer rows
    SweepArgs p = a; p.rows = (uint32_t)((uint64_t)rows * ld / 112); p.ld = 112;
    printf(", \"plain_112_ms\": %.4f", run<G, R, SW_PLAIN>(p, blocks, reps));
  }
  printf(", \"p59_lds_ms\": %.4f", run<G, R, SW_LDS_P59>(a, blocks, reps));
  printf(", \"p59_reg_ms\": %.4f", run<G, R, SW_REG_P59>(a, blocks, reps));
  { SweepArgs f = a; f.pk = pkf; printf(", \"f64_pieces_ms\": %.4f", run<G, R, SW_F64>(f, blocks, reps)); }
  uint32_t fl[4]; CK(hipMemcpy(fl, dflags, 16, hipMemcpyDeviceToHost));
  printf(", \"flags\": %u, \"bytes_read\": %zu, \"bytes_written_p59\": %zu}\n", fl[0], S.size() * 8, (size_t)rows * 768);
  // the two ways of building a p59 row must give the same bytes
  std::vector<unsigned char> w1((size_t)rows * 768), w2((size_t)rows * 768);
  run<G, R, SW_LDS_P59>(a, blocks, 1); CK(hipMemcpy(w1.data(), dW, w1.size(), hipMemcpyDeviceToHost));
  run<G, R, SW_REG_P59>(a, blocks, 1); CK(hipMemcpy(w2.data(), dW, w2.size(), hipMemcpyDeviceToHost));
  size_t diff = 0; for (size_t i = 0; i < w1.size(); ++i) diff += w1[i] != w2[i];
  fprintf(stderr, "p59 rows, LDS build vs register build: %zu differing bytes of %zu\n", diff, w1.size());
  return diff ? 2 : 0;
}

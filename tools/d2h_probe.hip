// d2h_probe.hip -- what moves a device buffer into pinned host memory fastest on this box: one hipMemcpyAsync, the
// same cut over two / four streams (several SDMA engines), or a kernel that stores straight into the mapped host
// buffer.  And the other direction for comparison.  hipcc --offload-arch=gfx950 -O3 -o tools/d2h_probe tools/d2h_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void copy_kernel(const uint4 *src, uint4 *dst, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const size_t bytes = (size_t)1 << 30;
  void *dev, *host;
  CHECK(hipMalloc(&dev, bytes)); CHECK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
  CHECK(hipMemset(dev, 1, bytes));
  hipStream_t st[4]; for (auto &s : st) CHECK(hipStreamCreate(&s));
  printf("{\"bytes\": %zu, \"results\": [\n", bytes);
  for (int dir = 0; dir < 2; ++dir) {
    for (int ns : {1, 2, 4}) {
      double best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipDeviceSynchronize());
        const double t0 = now();
        for (int k = 0; k < ns; ++k) {
          const size_t o = bytes / ns * k, len = bytes / ns;
          if (dir == 0) CHECK(hipMemcpyAsync((char *)host + o, (char *)dev + o, len, hipMemcpyDeviceToHost, st[k]));
          else CHECK(hipMemcpyAsync((char *)dev + o, (char *)host + o, len, hipMemcpyHostToDevice, st[k]));
        }
        CHECK(hipDeviceSynchronize());
        best = std::min(best, now() - t0);
      }
      printf("  {\"direction\": \"%s\", \"how\": \"hipMemcpyAsync on %d stream(s)\", \"GBps\": %.1f},\n", dir ? "h2d" : "d2h", ns, bytes / best / 1e9);
    }
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipDeviceSynchronize());
      const double t0 = now();
      if (dir == 0) hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st[0], (const uint4 *)dev, (uint4 *)host, bytes / 16);
      else hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, st[0], (const uint4 *)host, (uint4 *)dev, bytes / 16);
      CHECK(hipDeviceSynchronize());
      best = std::min(best, now() - t0);
    }
    printf("  {\"direction\": \"%s\", \"how\": \"copy kernel over the mapped host buffer\", \"GBps\": %.1f}%s\n", dir ? "h2d" : "d2h", bytes / best / 1e9, dir ? "" : ",");
  }
  printf("]}\n");
  return 0;
}

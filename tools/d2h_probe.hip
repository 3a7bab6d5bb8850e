// d2h_probe.hip -- what moves a device buffer into pinned host memory fastest on this box: one hipMemcpyAsync, the
// same cut over two / four streams (several SDMA engines), or a kernel that stores straight into the mapped host
// buffer.  And the other direction for comparison.  hipcc --offload-arch=gfx950 -O3 -o tools/d2h_probe tools/d2h_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>
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
  printf("],\n \"host_copy\": [\n");
  // and the host's own copy out of / into the pinned buffer (what the staged route adds), against ordinary memory
  {
    char *a = (char *)malloc(bytes), *b = (char *)malloc(bytes);
    memset(a, 1, bytes); memset(b, 2, bytes);
    auto par = [&](char *dst, const char *src, unsigned T) {
      std::vector<std::thread> th;
      const size_t per = bytes / T;
      const double t0 = now();
      for (unsigned t = 0; t < T; ++t) th.emplace_back([=] { memcpy(dst + t * per, src + t * per, per); });
      for (auto &x : th) x.join();
      return bytes / (now() - t0) / 1e9;
    };
    bool first = true;
    for (unsigned T : {1u, 4u, 16u}) {
      double v[3] = {0, 0, 0};
      for (int rep = 0; rep < 3; ++rep) {
        v[0] = std::max(v[0], par(a, (const char *)host, T));      // pinned -> ordinary (device -> host, second leg)
        v[1] = std::max(v[1], par((char *)host, a, T));            // ordinary -> pinned (host -> device, first leg)
        v[2] = std::max(v[2], par(b, a, T));                       // ordinary -> ordinary
      }
      printf("%s  {\"threads\": %u, \"pinned_to_ordinary_GBps\": %.1f, \"ordinary_to_pinned_GBps\": %.1f, \"ordinary_to_ordinary_GBps\": %.1f}", first ? "" : ",\n", T, v[0], v[1], v[2]);
      first = false;
    }
    printf("\n],\n \"staged_d2h\": [\n");
    // the library's staged route as it is built (hpf_capi.hip d2h): two pinned buffers, chunk c copied out by T
    // threads while chunk c + 1 is on the wire -- per chunk size
    first = true;
    for (size_t chunk : {(size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
      void *stg[2]; hipEvent_t ev[2];
      for (int k = 0; k < 2; ++k) { CHECK(hipHostMalloc(&stg[k], chunk, hipHostMallocDefault)); CHECK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); }
      for (unsigned T : {4u, 16u}) {
        double best = 1e9, t_sync = 0, t_copy = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CHECK(hipDeviceSynchronize());
          double ts = 0, tc = 0;
          const double t0 = now();
          size_t off = 0, prev_off = 0, prev_len = 0; int k = 0; bool have_prev = false;
          while (off < bytes || have_prev) {
            size_t len = 0;
            if (off < bytes) {
              len = std::min(chunk, bytes - off);
              CHECK(hipMemcpyAsync(stg[k], (char *)dev + off, len, hipMemcpyDeviceToHost, st[0]));
              CHECK(hipEventRecord(ev[k], st[0]));
            }
            if (have_prev) {
              const double a0 = now();
              CHECK(hipEventSynchronize(ev[k ^ 1]));
              const double a1 = now();
              std::vector<std::thread> th;
              const size_t per = (prev_len + T - 1) / T;
              for (unsigned t = 0; t < T; ++t) { const size_t o = t * per; if (o < prev_len) th.emplace_back([=] { memcpy(a + prev_off + o, (char *)stg[k ^ 1] + o, std::min(per, prev_len - o)); }); }
              for (auto &x : th) x.join();
              ts += a1 - a0; tc += now() - a1;
            }
            have_prev = len > 0; prev_off = off; prev_len = len; off += len; k ^= 1;
          }
          const double dt = now() - t0;
          if (dt < best) { best = dt; t_sync = ts; t_copy = tc; }
        }
        printf("%s  {\"chunk_MiB\": %zu, \"threads\": %u, \"GBps\": %.1f, \"waiting_for_dma_ms\": %.1f, \"copying_ms\": %.1f, \"total_ms\": %.1f}", first ? "" : ",\n",
               chunk >> 20, T, bytes / best / 1e9, t_sync * 1e3, t_copy * 1e3, best * 1e3);
        first = false;
      }
      for (int k = 0; k < 2; ++k) { CHECK(hipHostFree(stg[k])); CHECK(hipEventDestroy(ev[k])); }
    }
    printf("\n]}\n");
    free(a); free(b);
  }
  return 0;
}

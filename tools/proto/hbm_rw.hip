// HBM throughput of pure writes, pure reads and a 2:1 write:read mix (the forward transform's ratio), streaming 4 GiB (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o tools/proto/hbm_rw tools/proto/hbm_rw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(f4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = f4{v, v + 1, v + 2, v + 3};
}
__global__ __launch_bounds__(256) void rd(const f4* p, size_t n, float* out) {
  f4 s = f4{0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
// reads n words of `a`, writes 2 n words of `b`
__global__ __launch_bounds__(256) void mix21(const f4* a, f4* b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const f4 v = a[i];
    b[2 * i] = v;
    b[2 * i + 1] = v + 1.f;
  }
}
__global__ __launch_bounds__(256) void copy11(const f4* a, f4* b, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
int main() {
  const size_t S = 4ull << 30, n = S / 16;
  f4 *a, *b;
  float* out;
  (void)hipMalloc(&a, S);
  (void)hipMalloc(&b, 2 * S);
  (void)hipMalloc(&out, 64);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int grid : {2048, 8192}) {
    for (int rep = 0; rep < 2; rep++) {
      float ms;
      (void)hipEventRecord(e0); wr<<<grid, 256>>>(a, n, 1.f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d  write only      %.3f ms  %.2f TB/s\n", grid, ms, S / (ms * 1e9));
      (void)hipEventRecord(e0); rd<<<grid, 256>>>(a, n, out); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d  read only       %.3f ms  %.2f TB/s\n", grid, ms, S / (ms * 1e9));
      (void)hipEventRecord(e0); copy11<<<grid, 256>>>(a, b, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d  copy 1:1        %.3f ms  %.2f TB/s (read + written)\n", grid, ms, 2.0 * S / (ms * 1e9));
      (void)hipEventRecord(e0); mix21<<<grid, 256>>>(a, b, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("grid %5d  read 1 : write 2 %.3f ms  %.2f TB/s (read + written)\n", grid, ms, 3.0 * S / (ms * 1e9));
    }
  }
  return 0;
}

// Does a producer -> consumer pair of kernels get more than HBM rate when the working set fits the 256 MB memory-side cache
// (MALL / Infinity Cache)?  write W MB, then read the same W MB; and read the same W MB twice.  Decides whether running the
// forward transform and the multiply-accumulate per GROUP of voices (spectra of one group = 32 voices x 4.2 MB) could take
// the X round trip off HBM.
//   hipcc --offload-arch=gfx950 -O3 -o tools/proto/mall_probe tools/proto/mall_probe.hip && tools/proto/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) f4 gf4;
template <int U>
__global__ __launch_bounds__(256) void rd(const f4* __restrict p, size_t n, float* out) {
  f4 s = f4{0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = blockIdx.x * 256ull + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = ((const gf4*)p)[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) s += v[u];
  }
  for (; i < n; i += stride) s += p[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void wr(f4* __restrict p, size_t n, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  const f4 x = f4{v, v + 1, v + 2, v + 3};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += stride) ((gf4*)p)[i] = x;
}
int main() {
  const size_t maxb = 4ull << 30;
  f4* buf; float* out;
  hipMalloc(&buf, maxb); hipMalloc(&out, 64);
  hipMemset(buf, 1, maxb);
  hipEvent_t e[4]; for (auto& x : e) hipEventCreate(&x);
  const int grid = 256 * 8;
  printf("%8s %12s %12s %12s\n", "MB", "write GB/s", "read-after-write", "read-again");
  for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 4096}) {
    const size_t n = mb * (1ull << 20) / 16;
    double tw = 0, tr = 0, tr2 = 0; const int reps = 30;
    for (int r = -3; r < reps; r++) {
      // a different region of the 4 GB every repetition when the set is small would defeat the purpose: same region, like a reused X buffer
      hipEventRecord(e[0]); hipLaunchKernelGGL(wr, dim3(grid), dim3(256), 0, 0, buf, n, (float)r);
      hipEventRecord(e[1]); hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(256), 0, 0, buf, n, out);
      hipEventRecord(e[2]); hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(256), 0, 0, buf, n, out);
      hipEventRecord(e[3]); hipEventSynchronize(e[3]);
      float a, b, c; hipEventElapsedTime(&a, e[0], e[1]); hipEventElapsedTime(&b, e[1], e[2]); hipEventElapsedTime(&c, e[2], e[3]);
      if (r >= 0) { tw += a; tr += b; tr2 += c; }
    }
    const double gb = mb * 1.048576e-3 * reps;
    printf("%8zu %12.0f %12.0f %12.0f\n", mb, gb / (tw * 1e-3), gb / (tr * 1e-3), gb / (tr2 * 1e-3));
  }
  return 0;
}

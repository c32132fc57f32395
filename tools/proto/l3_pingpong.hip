// Does the Infinity Cache keep a producer kernel's stores for the consumer kernel that follows (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/proto/l3_pingpong tools/proto/l3_pingpong.hip
// 32 x { write S bytes ; read S bytes }: (a) always the same S-byte buffer, (b) 32 different buffers (streams through HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wr(f4* p, size_t n, float v) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = f4{v, v + 1, v + 2, v + 3};
}
__global__ __launch_bounds__(256) void rd(const f4* p, size_t n, float* out) {
  f4 s = f4{0, 0, 0, 0};
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
int main() {
  const int reps = 32;
  float* out;
  (void)hipMalloc(&out, 64);
  for (size_t mb : {32, 64, 128, 160, 192, 256, 512}) {
    const size_t S = mb << 20, n = S / 16;
    f4* big;
    if (hipMalloc(&big, S * reps) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int same = 1; same >= 0; same--) {
      for (int warm = 0; warm < 2; warm++) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; r++) {
          f4* b = big + (same ? 0 : (size_t)r * n);
          wr<<<2048, 256>>>(b, n, (float)r);
          rd<<<2048, 256>>>(b, n, out);
        }
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
      }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("S = %4zu MB  %s  %.3f ms for %d x (write + read)  = %.2f TB/s over written + read bytes\n", mb, same ? "same buffer     " : "different buffers",
             ms, reps, 2.0 * S * reps / (ms * 1e9));
    }
    (void)hipFree(big);
  }
  return 0;
}

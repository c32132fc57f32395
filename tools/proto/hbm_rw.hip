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
// the multiply-accumulate kernel's read pattern: workgroup (tile, job) reads 512-byte pieces (one per frame) at a stride of
// `stride` bytes, 2272 frames per job; 512 threads = 16 pieces per round, `ahead` rounds in flight
template <int AHEAD>
__global__ __launch_bounds__(512, 2) void rd_tiles(const f4* p, size_t stride16, int frames_per_job, float* out) {
  const int tile = blockIdx.x, job = blockIdx.y, row = threadIdx.x >> 5, of = threadIdx.x & 31;
  const f4* base = p + (size_t)job * frames_per_job * stride16 + (size_t)tile * 32 + of;
  f4 s = f4{0, 0, 0, 0};
  for (int f = row; f + 16 * (AHEAD - 1) < frames_per_job; f += 16 * AHEAD) {
    f4 v[AHEAD];
#pragma unroll
    for (int u = 0; u < AHEAD; u++) v[u] = base[(size_t)(f + 16 * u) * stride16];
#pragma unroll
    for (int u = 0; u < AHEAD; u++) s += v[u];
  }
  if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = 1.f;
}
int main(int argc, char** argv) {
  if (argc > 1) {
    const size_t S = 4ull << 30;
    f4* a;
    float* out;
    (void)hipMalloc(&a, S + (1 << 20));
    (void)hipMalloc(&out, 64);
    wr<<<4096, 256>>>(a, S / 16, 0.37f);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++)
      for (size_t stride : {65536, 4096, 512}) {
        // same bytes in every case: 128 tiles x 512 B = one 64 KB "frame" per 128 workgroups; jobs x frames x 64 KB = S
        const int fpj = 2272 - 2272 % 64, jobs = (int)(S / 65536 / fpj);
        // stride 65536: frame-major (as stored now); 4096: 8 tiles interleaved per 4 KB ; 512: tile-major (contiguous per workgroup)
        const size_t stride16 = stride / 16;
        // for the smaller strides a job's region is still frames_per_job x 64 KB: tiles index a different sub-block
        float ms;
        (void)hipEventRecord(e0);
        if (stride == 65536) rd_tiles<4><<<dim3(128, jobs), 512>>>(a, stride16, fpj, out);
        else if (stride == 4096) rd_tiles<4><<<dim3(8, jobs * 16), 512>>>(a, stride16, fpj, out);   // 8 tiles per 4 KB piece, 16 groups = separate regions
        else rd_tiles<4><<<dim3(1, jobs * 128), 512>>>(a, stride16, fpj, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("pieces of 512 B at stride %6zu: %.3f ms  %.2f TB/s\n", stride, ms, (double)jobs * fpj * 65536 / (ms * 1e9));
      }
    return 0;
  }
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

// Micro-benchmark: issue rate of the fp32 multiply-accumulate forms the coarse multiply-accumulate kernel could use (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pkfma_rate tools/proto/pkfma_rate.hip && /tmp/pkfma_rate
// Prints complex multiply-accumulates per cycle-equivalent (per SIMD, from the wall time at the measured shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE, int BLK = 256>
__global__ __launch_bounds__(BLK) void k(float* out, int iters, float seed, long long* cyc) {
  extern __shared__ float dyn_lds[];
  const int l = threadIdx.x;
  if (iters < 0) dyn_lds[l] = seed;
  const long long c0 = clock64();
  if constexpr (MODE == 4 || MODE == 5) {
    // the sweep's shape: 4 x 2 accumulators, 8 partitions, sliding window of 11 frames, 16 spectra -- all in registers
    f2 acc[4][2], h[8][2], xv[11];
    for (int i = 0; i < 8; i++) { acc[i >> 1][i & 1] = f2{seed * i, seed + i}; }
    for (int i = 0; i < 16; i++) h[i >> 1][i & 1] = f2{1.f + 1e-7f * i, 1e-7f * l};
    for (int i = 0; i < 11; i++) xv[i] = f2{seed + l + i, seed - i};
    if constexpr (MODE == 5) {   // noise in [-1, 1): audio spectra, not constants
      auto rnd = [&](unsigned k) { unsigned h = (k * 2654435761u) ^ (l * 40503u) ^ (blockIdx.x * 69069u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return (float)(int)h * 4.6566e-10f; };
      for (int i = 0; i < 8; i++) acc[i >> 1][i & 1] = f2{rnd(i), rnd(100 + i)};
      for (int i = 0; i < 16; i++) h[i >> 1][i & 1] = f2{rnd(200 + i), rnd(300 + i)};
      for (int i = 0; i < 11; i++) xv[i] = f2{rnd(400 + i), rnd(500 + i)};
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int rep = 0; rep < 2; rep++)
#pragma unroll
      for (int jj = 0; jj < 8; jj++) {
        const int j = 7 - jj;
#pragma unroll
        for (int tt = 0; tt < 4; tt++)
#pragma unroll
          for (int c = 0; c < 2; c++) {
            f2 t;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(xv[tt - j + 7]), "v"(h[j][c]), "v"(acc[tt][c]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(acc[tt][c]) : "v"(xv[tt - j + 7]), "v"(h[j][c]), "v"(t));
          }
      }
    }
    f2 s = f2{0.f, 0.f};
    for (int i = 0; i < 8; i++) s += acc[i >> 1][i & 1];
    out[blockIdx.x * blockDim.x + l] = s.x + s.y;
    if (l == 0 && blockIdx.x == 0) *cyc = clock64() - c0;
    return;
  }
  if constexpr (MODE <= 2) {
    f2 acc[16], a[4], b[4];
    for (int i = 0; i < 16; i++) acc[i] = f2{seed * i, seed + i};
    for (int i = 0; i < 4; i++) { a[i] = f2{seed + l + i, seed - i}; b[i] = f2{1.f + 1e-7f * i, 1e-7f * l}; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) {
          if constexpr (MODE == 0) {        // plain packed fma, independent accumulators: 2 per "complex mac"
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + r) & 3]), "v"(b[i & 3]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b[(i + r) & 3]), "v"(a[i & 3]));
          } else if constexpr (MODE == 1) { // the kernel's complex mac: dependent pair with op_sel
            f2 t;
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a[(i + r) & 3]), "v"(b[i & 3]), "v"(acc[i]));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(acc[i]) : "v"(a[(i + r) & 3]), "v"(b[i & 3]), "v"(t));
          } else {                            // four scalar fmas
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a[(i + r) & 3].x), "v"(b[i & 3].x));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(a[(i + r) & 3].x), "v"(b[i & 3].y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a[(i + r) & 3].y), "v"(b[i & 3].y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(a[(i + r) & 3].y), "v"(b[i & 3].x));
          }
        }
    }
    f2 s = f2{0.f, 0.f};
    for (int i = 0; i < 16; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + l] = s.x + s.y;
    if (l == 0 && blockIdx.x == 0) *cyc = clock64() - c0;
  } else {
    // MFMA 4x4x1 (16 blocks of 4x4 outer products): two instructions = 16 bins x 4 rows x 2 complex columns = 128 complex macs
    f4 acc[8];
    float a[4], b[4];
    for (int i = 0; i < 8; i++) acc[i] = f4{seed, seed * i, 0.f, 1.f};
    for (int i = 0; i < 4; i++) { a[i] = seed + l + i; b[i] = 1.f + 1e-7f * (l + i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if constexpr (MODE == 3) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(i + r) & 3], b[i & 3], acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(i + r) & 3], b[i & 3], acc[i], 0, 0, 0);
        }
    }
    f4 s = f4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + l] = s.x + s.y + s.z + s.w;
    if (l == 0 && blockIdx.x == 0) *cyc = clock64() - c0;
  }
}

template <int MODE>
static void run(const char* name, double cmac_per_wave_iter, int wg_per_cu, double instr_per_iter) {
  const int iters = 4096, grid = 256 * wg_per_cu;
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * grid * 256);
  hipMalloc(&cyc, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<grid, 256>>>(out, 64, 0.5f, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<grid, 256>>>(out, iters, 0.5f, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // each SIMD runs wg_per_cu waves (256 threads = 4 waves = one per SIMD)
  const double cmac_per_simd = cmac_per_wave_iter * iters * wg_per_cu;
  long long hc = 0;
  hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  const double n_instr = instr_per_iter * iters;   // per wave
  printf("%-34s waves/SIMD %d  %.3f ms  %.2f cmac lanes/ns/SIMD  | wave 0: %.2f counter ticks per instr, %.2f ns per instr per SIMD, counter %.0f MHz\n", name,
         wg_per_cu, ms, cmac_per_simd / (ms * 1e6), hc / n_instr, ms * 1e6 / (n_instr * wg_per_cu), hc / (ms * 1e3));
  hipFree(out);
}

template <int BLK>
static void run_big(const char* name, size_t lds) {
  const int iters = 4096, grid = 256;
  float* out;
  long long* cyc;
  (void)hipMalloc(&out, sizeof(float) * grid * 1024);
  (void)hipMalloc(&cyc, 8);
  (void)hipFuncSetAttribute((const void*)k<4, BLK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int nblk = 256 * (1024 / BLK);
  k<4, BLK><<<nblk, BLK, lds>>>(out, 64, 0.5f, cyc);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<4, BLK><<<nblk, BLK, lds>>>(out, iters, 0.5f, cyc);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long hc = 0;
  (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-50s %.3f ms  %.2f ns per pk_fma per SIMD (4 waves/SIMD), wave 0: %.2f ticks per instr\n", name, ms, ms * 1e6 / (256.0 * iters * 4), hc / (256.0 * iters));
}
int main() {
  run_big<256>("sweep-shaped, 4 workgroups of 256 per CU", 0);
  run_big<1024>("sweep-shaped, 1 workgroup of 1024 per CU", 0);
  run_big<1024>("sweep-shaped, 1 workgroup of 1024 per CU, 100 KB LDS", 100 * 1024);
  run_big<256>("sweep-shaped, 4 workgroups of 256 per CU", 0);
  for (int w : {1, 2, 4}) {
    run<0>("v_pk_fma_f32 plain", 64.0 * 64, w, 128);            // 64 pairs x 64 lanes
    run<1>("v_pk_fma_f32 complex pair (op_sel)", 64.0 * 64, w, 128);
    run<4>("complex pair, sweep-shaped registers", 128.0 * 64, w, 256);
    run<5>("the same on noise operands", 128.0 * 64, w, 256);
    run<2>("v_fma_f32 x4", 64.0 * 64, w, 256);
    run<3>("v_mfma_f32_4x4x1 (pair = 128 cmac)", 32.0 * 128, w, 64);   // 64 mfma = 32 pairs
  }
  return 0;
}

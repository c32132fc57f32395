// Microbenchmark: issue rate of v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 vs v_fma_f32 on gfx950 (one wave per SIMD .. many).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
  f2 a[8];
  float s[8];
  for (int i = 0; i < 8; i++) { a[i] = f2{1.0f + threadIdx.x * 1e-6f + i, 0.5f}; s[i] = 1.0f + i; }
  f2 m = {1.000001f, 0.999999f};
  float ms = 1.000001f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
        if (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[i]) : "v"(ms));
        if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(ms));
      }
  }
  float acc = 0;
  for (int i = 0; i < 8; i++) acc += a[i].x + a[i].y + s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE>
void run(const char* name, int wavesPerSimd) {
  float* out;
  int blocks = 256 * wavesPerSimd;  // 256 CUs, 256-thread blocks = 4 waves = 1 wave per SIMD
  hipMalloc(&out, blocks * 256 * 4);
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double instr_per_wave = (double)iters * 32;
  double ns_per_instr = ms * 1e6 / instr_per_wave / wavesPerSimd;   // per SIMD
  printf("%-14s waves/SIMD %d: %.3f ms, %.3f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, wavesPerSimd, ms, ns_per_instr, ns_per_instr * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_pk_fma_f32", w);
    run<1>("v_pk_add_f32", w);
    run<2>("v_pk_mul_f32", w);
    run<3>("v_fma_f32", w);
    run<4>("v_add_f32", w);
  }
  return 0;
}

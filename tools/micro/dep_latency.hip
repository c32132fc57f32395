// Microbenchmark: latency of DEPENDENT VALU instructions on gfx950 (one chain per lane), the DPP hand-off, and the shader clock
// under a light load (few waves on the chip, as in the serial biquad kernels).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a = 1.0f + threadIdx.x * 1e-6f, b = 0.999999f, c = 1e-7f, y = a;
  long long t0 = __builtin_readcyclecounter();
  long long m0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      if (MODE == 1) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c)); }
      if (MODE == 2) { asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(y) : "v"(a)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(a) : "v"(y), "v"(c)); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  long long m1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + y;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = m1 - m0; }
}
template <int MODE>
void run(const char* name, int blocks, int threads, int per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, blocks * threads * 4); hipMalloc(&cyc, 16);
  int iters = 20000;
  k<MODE><<<blocks, threads>>>(out, cyc, 100);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<blocks, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  double n = (double)iters * 32 * per_iter;
  printf("%-22s blocks %4d x %3d thr: %.3f ms, %.2f ns per dependent instr; s_memtime ticks/instr %.2f, readcyclecounter/instr %.2f; clock ~ %.2f GHz (memtime @100MHz: %.0f ticks)\n",
         name, blocks, threads, ms, ms * 1e6 / n, h[1] / n, h[0] / n, h[0] / (ms * 1e6), (double)h[1]);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int blocks : {128, 512, 4096}) {
    run<0>("dependent v_fma_f32", blocks, 64, 1);
    run<1>("dependent mul -> add", blocks, 64, 2);
    run<2>("dpp(row_shr) -> add", blocks, 64, 2);
  }
  return 0;
}

// Does the VGPR bank of the three 64-bit sources of v_pk_fma_f32 change its issue rate (gfx950)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/proto/vgpr_bank tools/proto/vgpr_bank.hip
// 4 waves per SIMD, 16 independent accumulators, sources pinned to explicit registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#define PK(d, a, b) "v_pk_fma_f32 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #d ":" #d "+1] op_sel_hi:[0,1,1]\n"
// accumulators v[32..63] (16 pairs); a-operands and b-operands chosen per pattern
#define BODY(A0, A1, B0, B1)                                                                                                  \
  PK(32, A0, B0) PK(34, A0, B1) PK(36, A1, B0) PK(38, A1, B1) PK(40, A0, B0) PK(42, A0, B1) PK(44, A1, B0) PK(46, A1, B1)        \
  PK(48, A0, B0) PK(50, A0, B1) PK(52, A1, B0) PK(54, A1, B1) PK(56, A0, B0) PK(58, A0, B1) PK(60, A1, B0) PK(62, A1, B1)
template <int PAT>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  // initialise v[8..63] to something finite
  asm volatile(
      "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 0.5\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n"
      "v_mov_b32 v14, 1.0\n v_mov_b32 v15, 0.5\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 0.5\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 0.5\n"
      "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 0.5\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 0.5\n" ::
          : "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23");
  for (int it = 0; it < iters; it++) {
#define CLOB                                                                                                                                       \
  "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", \
      "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
    // accumulator pairs start at even registers: v32 (bank 0), v34 (bank 2), ...
    if constexpr (PAT == 0) asm volatile(BODY(8, 12, 16, 20) BODY(8, 12, 16, 20) BODY(8, 12, 16, 20) BODY(8, 12, 16, 20)::: CLOB);       // a, b in banks {0,1}
    if constexpr (PAT == 1) asm volatile(BODY(8, 12, 18, 22) BODY(8, 12, 18, 22) BODY(8, 12, 18, 22) BODY(8, 12, 18, 22)::: CLOB);       // a {0,1}, b {2,3}
    if constexpr (PAT == 2) asm volatile(BODY(10, 14, 18, 22) BODY(10, 14, 18, 22) BODY(10, 14, 18, 22) BODY(10, 14, 18, 22)::: CLOB);   // a, b in {2,3}
    if constexpr (PAT == 3) asm volatile(BODY(8, 14, 16, 22) BODY(8, 14, 16, 22) BODY(8, 14, 16, 22) BODY(8, 14, 16, 22)::: CLOB);       // mixed
  }
  float r;
  asm volatile("v_add_f32 %0, v32, v63" : "=v"(r));
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int PAT>
static void run(const char* name) {
  const int iters = 8192, grid = 256 * 4;
  float* out;
  (void)hipMalloc(&out, sizeof(float) * grid * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<PAT><<<grid, 256>>>(out, 64);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<PAT><<<grid, 256>>>(out, iters);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %.3f ms  %.2f ns per v_pk_fma_f32 per SIMD\n", name, ms, ms * 1e6 / (64.0 * iters * 4));
  (void)hipFree(out);
}
int main(int argc, char** argv) {
  if (argc > 1) {   // sustained: the same kernel back to back for a few seconds, rate per ~0.1 s
    float* out;
    (void)hipMalloc(&out, sizeof(float) * 1024 * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 30; rep++) {
      (void)hipEventRecord(e0);
      for (int q = 0; q < 25; q++) k<3><<<1024, 256>>>(out, 8192);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("t = %2d: %.1f ms  %.2f ns per v_pk_fma_f32 per SIMD\n", rep, ms, ms * 1e6 / (64.0 * 8192 * 4 * 25));
    }
    return 0;
  }
  for (int rep = 0; rep < 2; rep++) {
    run<0>("src0, src1 in banks {0,1}; acc alternating");
    run<1>("src0 {0,1}, src1 {2,3}; acc alternating");
    run<2>("src0, src1 in banks {2,3}; acc alternating");
    run<3>("mixed");
  }
  return 0;
}

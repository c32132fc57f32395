// ga_fft16.hpp -- device-side FFT building blocks shared by ga_kernels.hip (formulation C, block-axis FFT) and ga_coarse.hip
// (formulation D, coarse partitions): packed-f32 complex arithmetic and the register / LDS hybrid radix-16 transform.
// Device code only (inline gfx950 assembly); include from .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace ga {

// ---- explicit GLOBAL address space for pointers that come out of job tables ------------------------------------------
// A pointer loaded from memory is a generic ("flat") pointer to the compiler: every access through it becomes flat_load /
// flat_store, which counts on vmcnt AND lgkmcnt, completes out of order and therefore forces `s_waitcnt vmcnt(0) lgkmcnt(0)`
// before anything that depends on it -- a prefetch issued in front of LDS work is waited for at once.  Every such pointer in
// this library addresses device memory, so say so: the accesses become global_load / global_store.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define GA_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ GA_GLOBAL T* gptr(T* p) { return (GA_GLOBAL T*)p; }
__device__ __forceinline__ v4f ldg4(const float* p) { return *(const GA_GLOBAL v4f*)p; }
__device__ __forceinline__ v2f ldg2(const float* p) { return *(const GA_GLOBAL v2f*)p; }
__device__ __forceinline__ v4f ldg4(const float2* p) { return *(const GA_GLOBAL v4f*)p; }
__device__ __forceinline__ v2f ldg2(const float2* p) { return *(const GA_GLOBAL v2f*)p; }
__device__ __forceinline__ float ldg1(const float* p) { return *(const GA_GLOBAL float*)p; }
__device__ __forceinline__ void stg4(float* p, v4f v) { *(GA_GLOBAL v4f*)p = v; }
__device__ __forceinline__ void stg4(float2* p, v4f v) { *(GA_GLOBAL v4f*)p = v; }
__device__ __forceinline__ void stg2(float2* p, v2f v) { *(GA_GLOBAL v2f*)p = v; }
__device__ __forceinline__ void stg1(float* p, float v) { *(GA_GLOBAL float*)p = v; }

// ---- packed float2 arithmetic --------------------------------------------------------------------------------------
// A complex number lives in an aligned VGPR pair; v_pk_add/mul/fma_f32 work on both halves at the full f32 rate, and their
// op_sel / neg modifiers give the swaps and sign flips of complex arithmetic for free: a +- (-i) b is ONE instruction, a
// complex product two.  (The scalar form the compiler otherwise emits needs twice the issue slots, and this stage is
// issue-bound: SQ_ACTIVE_INST_ANY x waves/SIMD ~ 100 % in the rocprof counters of round 1.)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 add_mi(f2 a, f2 b) {   // a + (-i) b = (a.x + b.y, a.y - b.x)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 sub_mi(f2 a, f2 b) {   // a - (-i) b = (a.x - b.y, a.y + b.x)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ f2 cmulp(f2 a, f2 b) {    // (a.x b.x - a.y b.y, a.x b.y + a.y b.x), second step fused
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
__device__ __forceinline__ f2 cmulp_conj(f2 a, f2 b) {   // conj(a b) = (a.x b.x - a.y b.y, -(a.x b.y + a.y b.x))
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
__device__ __forceinline__ f2 cmulp_swap(f2 a, f2 b) {   // swap(a b) = (a.x b.y + a.y b.x, a.x b.x - a.y b.y)
  f2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}
__device__ __forceinline__ f2 rot_m45(f2 a) {   // a (1 - i) = (a.x + a.y, a.y - a.x)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ f2 rot_p45(f2 a) {   // a (1 + i) = (a.x - a.y, a.y + a.x)
  f2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
__device__ __forceinline__ void pdft4(f2& a0, f2& a1, f2& a2, f2& a3) {   // forward, natural order: 8 instructions
  f2 t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, t3);
  a3 = sub_mi(t1, t3);
}
// dft4 whose third input still has to be multiplied by -i (folded into the first butterflies)
__device__ __forceinline__ void pdft4_a2mi(f2& a0, f2& a1, f2& a2, f2& a3) {
  f2 t0 = add_mi(a0, a2), t1 = sub_mi(a0, a2), t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_mi(t1, t3);
  a3 = sub_mi(t1, t3);
}
__device__ __forceinline__ void pdft8(f2 (&v)[8]) {   // forward 8-point DFT, natural order in / out: 26 instructions
  const float h = 0.70710678118654752440f;
  const f2 hh = {h, h};
  f2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  f2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  pdft4(e0, e1, e2, e3);
  pdft4(o0, o1, o2, o3);
  f2 r1 = rot_m45(o1);            // o1 W8   = h * r1
  f2 r3 = rot_p45(o3);            // o3 W8^3 = -h * r3
  v[0] = e0 + o0; v[4] = e0 - o0;
  v[1] = __builtin_elementwise_fma(r1, hh, e1); v[5] = __builtin_elementwise_fma(-r1, hh, e1);
  v[2] = add_mi(e2, o2); v[6] = sub_mi(e2, o2);
  v[3] = __builtin_elementwise_fma(-r3, hh, e3); v[7] = __builtin_elementwise_fma(r3, hh, e3);
}
__device__ __forceinline__ void pdft16(f2 (&v)[16]) {   // forward 16-point DFT, natural order in / out
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  const f2 hh = {h, h}, nhh = {-h, -h};
  f2 y[4][4];   // y[b][c] = sum_a v[4a + b] W4^(ac)
#pragma unroll
  for (int b = 0; b < 4; b++) {
    f2 a0 = v[b], a1 = v[4 + b], a2 = v[8 + b], a3 = v[12 + b];
    pdft4(a0, a1, a2, a3);
    y[b][0] = a0; y[b][1] = a1; y[b][2] = a2; y[b][3] = a3;
  }
  const f2 w1 = {c1, -s1}, w3 = {s1, -c1}, w9 = {-c1, s1};
  y[1][1] = cmulp(y[1][1], w1);
  y[1][2] = rot_m45(y[1][2]) * hh;          // W16^2 = h (1 - i)
  y[1][3] = cmulp(y[1][3], w3);
  y[2][1] = rot_m45(y[2][1]) * hh;
  // y[2][2] * W16^4 = -i y[2][2]: folded into the butterfly of column 2
  y[2][3] = rot_p45(y[2][3]) * nhh;         // W16^6 = -h (1 + i)
  y[3][1] = cmulp(y[3][1], w3);
  y[3][2] = rot_p45(y[3][2]) * nhh;
  y[3][3] = cmulp(y[3][3], w9);             // W16^9 = -W16^1
#pragma unroll
  for (int c = 0; c < 4; c++) {   // X[c + 4d] = sum_b z[b][c] W4^(bd)
    f2 a0 = y[0][c], a1 = y[1][c], a2 = y[2][c], a3 = y[3][c];
    if (c == 2) pdft4_a2mi(a0, a1, a2, a3); else pdft4(a0, a1, a2, a3);
    v[c] = a0; v[c + 4] = a1; v[c + 8] = a2; v[c + 12] = a3;
  }
}

// ---- radix-16 variant ------------------------------------------------------------------------------------------
// N2 = 16 * 16 * R3 (R3 = 4, 8, 16).  A transform is owned by T = N2/16 threads (16 points each), so a 256-thread workgroup
// runs G = 256/T transforms side by side (4, 2, 1).  Plan: radix 16 on the thread-owned points -> LDS -> radix 16 -> LDS
// (in place) -> radix R3 back into thread-owned registers: two LDS round trips and three barriers per G transforms instead
// of three round trips and three barriers per single transform.  One pad slot per 32 points (`i + (i >> 5)`) makes the
// stride-16 stores of the first pass conflict-free and keeps every other access (32 consecutive, aligned points) inside one
// row; all LDS addresses are `lane base + compile-time constant`, so they fold into the DS instruction offsets.
// Twiddles come from two small per-pass tables whose reads are consecutive in the lane index.
#define TC16_PADDED(n) ((n) + ((n) >> 5))
template <int N>
struct R16Plan {
  static constexpr int T = N / 16;          // threads per transform
  static constexpr int G = 256 / T;         // transforms per workgroup
  static constexpr int R3 = N / 256;        // last radix
  static constexpr int U3 = 16 / R3;        // last-pass butterflies per thread
  static constexpr int T2 = 15 * 16;        // table of W_256^(kk m), [m-1][kk]
  static constexpr int T3 = (R3 - 1) * 256; // table of W_N^(j m),   [m-1][j]
};
// forward FFT of the 16 thread-owned points {t + T m}; `buf` is this transform's N-point LDS buffer.  Contains three
// workgroup barriers; every thread of the workgroup must call it (threads of an idle transform pass live = false).
// PW ("power twiddles", N = 4096 only): `tw3` holds W_N^(j m) for m = 1, 2, 4, 8 only ([4][256], 8 KB instead of 30 KB) and the
// other eleven twiddles of a butterfly are products of those (<= 3 roundings each) -- for kernels that are short of LDS,
// not of VALU issue slots.
template <int N, bool PW = false>
__device__ __forceinline__ void fft16_own(f2 (&own)[16], f2* __restrict buf, const f2* __restrict tw2,
                                          const f2* __restrict tw3, int t) {
  using PL = R16Plan<N>;
  constexpr int T = PL::T;
  pdft16(own);
  const int b1 = 16 * t + (t >> 1);        // pad(16 t + m) = b1 + m
  const int b2 = t + (t >> 5);             // pad(t + c)    = b2 + c + (c >> 5) for c a multiple of 32
#pragma unroll
  for (int m = 0; m < 16; m++) buf[b1 + m] = own[m];
  __syncthreads();
  {
    const int kk = t & 15;
#pragma unroll
    for (int m = 0; m < 16; m++) own[m] = buf[b2 + T * m + (T / 32) * m];
    // twiddles in groups of four with a scheduling fence in between: keeps the compiler from hoisting all 15 table
    // reads above the products (30 more live registers cost a wave per SIMD)
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
      for (int m = 4 * q; m < 4 * q + 4; m++)
        if (m > 0) own[m] = cmulp(own[m], tw2[(m - 1) * 16 + kk]);
      __builtin_amdgcn_sched_barrier(0);
    }
    pdft16(own);
    __syncthreads();   // every read of this pass is done: write in place
    const int b3 = (t >> 4) * 264 + kk;    // pad(256 (t / 16) + kk + 16 m) = b3 + 16 m + (m >> 1)
#pragma unroll
    for (int m = 0; m < 16; m++) buf[b3 + 16 * m + (m >> 1)] = own[m];
  }
  __syncthreads();
  {
    constexpr int R3 = PL::R3, U3 = PL::U3;
#pragma unroll
    for (int u = 0; u < U3; u++) {
      const int j = t + T * u;   // < 256
      f2 v[R3];
#pragma unroll
      for (int m = 0; m < R3; m++) v[m] = buf[b2 + (T + T / 32) * u + 264 * m];
      if constexpr (PW) {
        static_assert(!PW || R3 == 16, "power twiddles: 4096 points only");
        // W^(j m) = W^(j (m & 12)) W^(j (m & 3)): only W^j, W^2j, W^3j stay live; the group base (W^4j, W^8j, W^12j) comes and goes
        const f2 w1 = tw3[j], w2 = tw3[256 + j];
        const f2 w3 = cmulp(w1, w2);
        v[1] = cmulp(v[1], w1); v[2] = cmulp(v[2], w2); v[3] = cmulp(v[3], w3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 1; g < 4; g++) {
          const f2 w4 = tw3[512 + j];
          const f2 b = g == 1 ? w4 : g == 2 ? tw3[768 + j] : cmulp(tw3[768 + j], w4);
          v[4 * g] = cmulp(v[4 * g], b);
          v[4 * g + 1] = cmulp(v[4 * g + 1], cmulp(b, w1));
          v[4 * g + 2] = cmulp(v[4 * g + 2], cmulp(b, w2));
          v[4 * g + 3] = cmulp(v[4 * g + 3], cmulp(b, w3));
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
      for (int q = 0; q < R3 / 4; q++) {
#pragma unroll
        for (int m = 4 * q; m < 4 * q + 4; m++)
          if (m > 0) v[m] = cmulp(v[m], tw3[(m - 1) * 256 + j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      if constexpr (R3 == 16) pdft16(v);
      else if constexpr (R3 == 8) pdft8(v);
      else pdft4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int m = 0; m < R3; m++) own[u + U3 * m] = v[m];   // position j + 256 m = t + T (u + U3 m)
    }
  }
}

}  // namespace ga

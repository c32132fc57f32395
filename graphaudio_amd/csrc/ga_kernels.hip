// ga_kernels.hip -- hand-written gfx950 (CDNA4, MI355X) kernels of the GraphAudio offline render path.
//
// Reference semantics each kernel reproduces are cited as (file:line) of the-byte-bender/GraphAudio.
// Design notes (DESIGN.md has the full account):
//  * wavefront = 64 lanes everywhere; one 256-point real FFT per wavefront, two complex points per lane,
//    radix-2 butterflies exchanged with wave shuffles, double precision like the reference's FftFlat.
//  * the spectral multiply-accumulate runs over ALL blocks of a render chunk at once: for one frequency bin it is
//    a banded-Toeplitz GEMM  Y[rows x time] = X[rows x time'] * T(H)  executed on the f32 matrix cores
//    (v_mfma_f32_16x16x4_f32, exact f32 fma chain), rows = convolver channel-instances sharing one IR channel.
//  * serial-in-time recurrences (biquad, resampler) stage 64x64 tiles through LDS so HBM accesses stay coalesced.
// Compiled with -ffp-contract=off: float32 elementwise arithmetic is unfused like the reference's; fused
// multiply-adds appear only where written explicitly (fma(), MFMA).

#include "ga_kernels.hpp"
#include "ga_fft16.hpp"
#include <type_traits>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace ga {

// gridDim.y is limited to 65535: a level of a fragmented chunk (thousands of voices x tens of segments) can hold more jobs than
// that, so every (x = frames, y = job) launcher walks the job table in windows.  Kernels index `jobs[blockIdx.y]`.
constexpr int kMaxGridY = 32768;
#define GA_LAUNCH_JOBS(kernel, gx_, block_, jobs_, njobs_, ...)                                                      \
  for (int j0_ = 0; j0_ < (njobs_); j0_ += kMaxGridY)                                                                \
    hipLaunchKernelGGL(kernel, dim3((gx_), std::min(kMaxGridY, (njobs_) - j0_)), dim3(block_), 0, s, (jobs_) + j0_, ##__VA_ARGS__)

// phase timestamps for tools/micro/rfft_phase.hip (compiled out of the product)
#ifdef GA_EXP_TIMELINE
__device__ unsigned long long* ga_tl = nullptr;
#define GA_TL(i) do { if (ga_tl && threadIdx.x == 0) ga_tl[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GA_TL(i) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));

// =====================================================================================================
//  256-point real FFT on one wavefront (forward: RealFourierTransform.cs:62-88; inverse: :101-131), double precision.
//  z[n] = x[2n] + i x[2n+1] (n < 128); lane l owns z[l] (slot 0) and z[l+64] (slot 1).  One in-lane radix-2 stage
//  and six cross-lane stages; the cross-lane exchanges are DPP moves (xor 1,2,4,8), ds_swizzle (xor 16) and one
//  ds_bpermute (xor 32) -- no LDS storage, no barriers.  Butterflies are select-free: every lane evaluates
//  d = fma(sigma, mine, partner) followed by one complex multiply whose twiddle is (1, 0) on the "lower" lanes.
//  Each wavefront keeps FFT_ILP independent transforms in flight to hide the exchange latency.
// =====================================================================================================
template <int MASK>
__device__ __forceinline__ int xchg32(int v) {
  if constexpr (MASK == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);           // quad_perm [1,0,3,2]
  else if constexpr (MASK == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
  else if constexpr (MASK == 4) {                                                                // xor 7 then xor 3
    int t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);                                  // row_half_mirror
    return __builtin_amdgcn_mov_dpp(t, 0x1B, 0xF, 0xF, true);                                    // quad_perm [3,2,1,0]
  } else if constexpr (MASK == 8) {                                                              // xor 15 then xor 7
    int t = __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);                                  // row_mirror
    return __builtin_amdgcn_mov_dpp(t, 0x141, 0xF, 0xF, true);
  } else if constexpr (MASK == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                // bit mode: xor 0x10
  else return __shfl_xor(v, MASK, 64);
}
template <int MASK>
__device__ __forceinline__ double xchg(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(xchg32<MASK>(hi), xchg32<MASK>(lo));
}
template <int MASK>
__device__ __forceinline__ float xchg(float v) { return __int_as_float(xchg32<MASK>(__float_as_int(v))); }
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ float shfl_d(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int rev6(int l) { return (int)(__brev((unsigned)l) >> 26); }

// T = double reproduces the reference's double-precision FftFlat transform (spectra equal after the float32 cast with
// overwhelming probability); T = float is the default fast path: the transform error (~1e-7 relative) stays an order of
// magnitude inside the float32 accumulation noise of the partition sum that follows.
template <class T>
struct LaneTwT {   // per-lane constants, loaded once per kernel
  T c[6], s[6];    // stage twiddle for half = 32,16,8,4,2,1: W128^{(l & (half-1)) * 64/half} on upper lanes, (1,0) on lower
  T sg[6];         // -1 on upper lanes (l & half), +1 on lower lanes
  T c1, s1;        // W128^l  (the in-lane stage)
};
using LaneTw = LaneTwT<double>;
template <class T>
__device__ __forceinline__ LaneTwT<T> load_lane_tw_t(const double2* __restrict w128, int lane) {
  LaneTwT<T> t;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    int half = 32 >> i;
    bool up = (lane & half) != 0;
    double2 w = w128[(lane & (half - 1)) * (64 / half)];
    t.c[i] = up ? (T)w.x : (T)1.0;
    t.s[i] = up ? (T)w.y : (T)0.0;
    t.sg[i] = up ? (T)-1.0 : (T)1.0;
  }
  double2 w = w128[lane];
  t.c1 = (T)w.x;
  t.s1 = (T)w.y;
  return t;
}
__device__ __forceinline__ LaneTw load_lane_tw(const double2* __restrict w128, int lane) { return load_lane_tw_t<double>(w128, lane); }
// decimation in frequency: lower <- a + b ; upper <- (a - b) W   (natural order in, bit-reversed out)
template <int HALF, int IDX, class T>
__device__ __forceinline__ void dif_stage(T& ar, T& ai, const LaneTwT<T>& t) {
  T pr = xchg<HALF>(ar), pi = xchg<HALF>(ai);
  T dr = fma(t.sg[IDX], ar, pr), di = fma(t.sg[IDX], ai, pi);   // lower: mine + partner ; upper: partner - mine
  ar = fma(dr, t.c[IDX], -(di * t.s[IDX]));
  ai = fma(dr, t.s[IDX], di * t.c[IDX]);
}
// decimation in time with conjugated twiddles: lower <- a + b conj(W) ; upper <- a - b conj(W)  (bit-reversed in, natural out)
template <int HALF, int IDX, class T>
__device__ __forceinline__ void dit_stage(T& ar, T& ai, const LaneTwT<T>& t) {
  T qr = fma(ar, t.c[IDX], ai * t.s[IDX]);       // mine * conj(tw)   (tw = (1,0) on lower lanes)
  T qi = fma(ai, t.c[IDX], -(ar * t.s[IDX]));
  T pr = xchg<HALF>(qr), pi = xchg<HALF>(qi);
  ar = fma(t.sg[IDX], qr, pr);                   // lower: a + q_partner ; upper: partner - q_mine
  ai = fma(t.sg[IDX], qi, pi);
}

constexpr int FFT_ROWS = 32;        // rows (channel-instances) per workgroup
constexpr int STAGE_LD = 132;       // staging tile is [row][bin], bins fastest, 129 padded to 132 floats
constexpr int FFT_ILP = 4;          // independent transforms in flight per wavefront

// ---- forward ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rfft_fwd_kernel(const ConvRowIO* __restrict rows, int nrows, int nblocks, int hist,
                                                       ConvPlanes pl, Twiddles tw) {
  __shared__ float st_r[FFT_ROWS * STAGE_LD];
  __shared__ float st_i[FFT_ROWS * STAGE_LD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowtile = blockIdx.x;
  const int t = blockIdx.y;
  const LaneTw ltw = load_lane_tw(tw.w128, lane);
  // after the DIF stages slot j of lane l holds Z[k], k = 2 m + j, m = rev6(l).  X[k] needs Z[128 - k]:
  //   j = 0: index 2 ((64 - m) & 63) -> lane rev6((64 - m) & 63) ; j = 1: index 2 (63 - m) + 1 -> lane 63 - l
  const int m = rev6(lane);
  const int k0 = 2 * m, k1 = 2 * m + 1;
  const int src0 = rev6((64 - m) & 63), src1 = 63 - lane;
  const double2 wk0 = tw.w256[k0];
  const double2 wk1 = tw.w256[k1];

  for (int q0 = 0; q0 < FFT_ROWS / 4; q0 += FFT_ILP) {
    double s0r[FFT_ILP], s0i[FFT_ILP], s1r[FFT_ILP], s1i[FFT_ILP];
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) {
      const int row = rowtile * FFT_ROWS + wave * (FFT_ROWS / 4) + q0 + u;
      const float* in = row < nrows ? rows[row].in : nullptr;
      s0r[u] = 0.0;
      s0i[u] = 0.0;
      if (in) {
        const GA_GLOBAL float* p = gptr(in) + (int64_t)t * kBlock + 2 * lane;
        s0r[u] = (double)p[0];     // float -> double, PartitionedConvolver.cs:106
        s0i[u] = (double)p[1];
      }
      // in-lane stage with the zero upper half (PartitionedConvolver.cs:107): (a + 0, (a - 0) * W128^l)
      s1r[u] = fma(s0r[u], ltw.c1, -(s0i[u] * ltw.s1));
      s1i[u] = fma(s0r[u], ltw.s1, s0i[u] * ltw.c1);
    }
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) { dif_stage<32, 0>(s0r[u], s0i[u], ltw); dif_stage<32, 0>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) { dif_stage<16, 1>(s0r[u], s0i[u], ltw); dif_stage<16, 1>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) { dif_stage<8, 2>(s0r[u], s0i[u], ltw); dif_stage<8, 2>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) { dif_stage<4, 3>(s0r[u], s0i[u], ltw); dif_stage<4, 3>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) { dif_stage<2, 4>(s0r[u], s0i[u], ltw); dif_stage<2, 4>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) { dif_stage<1, 5>(s0r[u], s0i[u], ltw); dif_stage<1, 5>(s1r[u], s1i[u], ltw); }
    // real-FFT split: X[k] = E[k] + W256^k * (-i) * D[k],  E = (Z[k] + conj Z[128-k]) / 2,  D = (Z[k] - conj Z[128-k]) / 2
#pragma unroll
    for (int u = 0; u < FFT_ILP; u++) {
      const int rl = wave * (FFT_ROWS / 4) + q0 + u;
      {
        double ax = s0r[u], ay = s0i[u];
        double bx = shfl_d(ax, src0), by = shfl_d(ay, src0);
        double er = 0.5 * (ax + bx), ei = 0.5 * (ay - by);
        double dr = 0.5 * (ax - bx), di = 0.5 * (ay + by);
        double pr = fma(dr, wk0.x, -(di * wk0.y)), pi = fma(dr, wk0.y, di * wk0.x);
        float xr = (float)(er + pi), xi = (float)(ei - pr);   // double -> float, PartitionedConvolver.cs:117-118
        if (k0 == 0) {  // lane 0: DC, plus Nyquist X[128] = Re Z[0] - Im Z[0]; both purely real (RealFourierTransform.cs:76-78)
          xi = 0.f;
          st_r[rl * STAGE_LD + 128] = (float)(ax - ay);
          st_i[rl * STAGE_LD + 128] = 0.f;
        }
        st_r[rl * STAGE_LD + k0] = xr;
        st_i[rl * STAGE_LD + k0] = xi;
      }
      {
        double ax = s1r[u], ay = s1i[u];
        double bx = shfl_d(ax, src1), by = shfl_d(ay, src1);
        double er = 0.5 * (ax + bx), ei = 0.5 * (ay - by);
        double dr = 0.5 * (ax - bx), di = 0.5 * (ay + by);
        double pr = fma(dr, wk1.x, -(di * wk1.y)), pi = fma(dr, wk1.y, di * wk1.x);
        st_r[rl * STAGE_LD + k1] = (float)(er + pi);
        st_i[rl * STAGE_LD + k1] = (float)(ei - pr);
      }
    }
  }
  __syncthreads();
  // coalesced 16-byte stores: 8 lanes cover the 32 consecutive rows (one 128-byte line) of one bin
  const size_t plane_t = (size_t)pl.tx * pl.rp;
  const size_t base = (size_t)(hist + t) * pl.rp + (size_t)rowtile * FFT_ROWS;
  for (int idx = tid; idx < kBins * (FFT_ROWS / 4); idx += 256) {
    int k = idx / (FFT_ROWS / 4), r4 = (idx % (FFT_ROWS / 4)) * 4;
    size_t o = (size_t)k * plane_t + base + r4;
    float4 vr = make_float4(st_r[(r4 + 0) * STAGE_LD + k], st_r[(r4 + 1) * STAGE_LD + k], st_r[(r4 + 2) * STAGE_LD + k], st_r[(r4 + 3) * STAGE_LD + k]);
    float4 vi = make_float4(st_i[(r4 + 0) * STAGE_LD + k], st_i[(r4 + 1) * STAGE_LD + k], st_i[(r4 + 2) * STAGE_LD + k], st_i[(r4 + 3) * STAGE_LD + k]);
    *reinterpret_cast<float4*>(pl.xr + o) = vr;
    *reinterpret_cast<float4*>(pl.xi + o) = vi;
  }
}

void launch_rfft_fwd(hipStream_t s, const ConvRowIO* rows_dev, int nrows, int nblocks, int hist, ConvPlanes pl, Twiddles tw) {
  if (nrows <= 0 || nblocks <= 0) return;
  dim3 grid((nrows + FFT_ROWS - 1) / FFT_ROWS, nblocks);
  hipLaunchKernelGGL(rfft_fwd_kernel, grid, dim3(256), 0, s, rows_dev, nrows, nblocks, hist, pl, tw);
}

// ---- inverse + overlap-add ----------------------------------------------------------------------------
constexpr int OLA_RUN = 16;   // consecutive blocks per workgroup (one extra inverse FFT recovers the incoming tail)

__global__ __launch_bounds__(256) void irfft_ola_kernel(const ConvRowIO* __restrict rows, int nrows, int nblocks, ConvPlanes pl,
                                                        const float* __restrict overlap_in, float* __restrict overlap_out, Twiddles tw) {
  __shared__ float ys_r[FFT_ROWS * STAGE_LD];
  __shared__ float ys_i[FFT_ROWS * STAGE_LD];
  __shared__ float tail[FFT_ROWS][kBlock];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rowtile = blockIdx.x;
  const int ta = blockIdx.y * OLA_RUN;
  const int tb = min(ta + OLA_RUN, nblocks);
  const LaneTw ltw = load_lane_tw(tw.w128, lane);
  const int k0 = rev6(lane) << 1, k1 = k0 | 1;
  const double2 wk0 = tw.w256[k0];
  const double2 wk1 = tw.w256[k1];
  const size_t plane_t = (size_t)pl.ty * pl.rp;

  if (ta == 0) {  // incoming overlap of the chunk's first block = persistent _overlap (PartitionedConvolver.cs:28,148-149)
    for (int idx = tid; idx < FFT_ROWS * kBlock; idx += 256) {
      int r = idx / kBlock, i = idx % kBlock;
      int row = rowtile * FFT_ROWS + r;
      tail[r][i] = row < nrows ? overlap_in[(size_t)row * kBlock + i] : 0.f;
    }
  }
  for (int t = (ta == 0 ? 0 : ta - 1); t < tb; t++) {
    const bool pre = t < ta;   // only recompute block ta-1 to obtain its second half
    __syncthreads();
    const size_t base = (size_t)t * pl.rp + (size_t)rowtile * FFT_ROWS;
    for (int idx = tid; idx < kBins * (FFT_ROWS / 4); idx += 256) {   // 16-byte loads, 8 lanes per 128-byte bin line
      int k = idx / (FFT_ROWS / 4), r4 = (idx % (FFT_ROWS / 4)) * 4;
      size_t o = (size_t)k * plane_t + base + r4;
      float4 vr = *reinterpret_cast<const float4*>(pl.yr + o);
      float4 vi = *reinterpret_cast<const float4*>(pl.yi + o);
      ys_r[(r4 + 0) * STAGE_LD + k] = vr.x; ys_r[(r4 + 1) * STAGE_LD + k] = vr.y;
      ys_r[(r4 + 2) * STAGE_LD + k] = vr.z; ys_r[(r4 + 3) * STAGE_LD + k] = vr.w;
      ys_i[(r4 + 0) * STAGE_LD + k] = vi.x; ys_i[(r4 + 1) * STAGE_LD + k] = vi.y;
      ys_i[(r4 + 2) * STAGE_LD + k] = vi.z; ys_i[(r4 + 3) * STAGE_LD + k] = vi.w;
    }
    __syncthreads();
    for (int q0 = 0; q0 < FFT_ROWS / 4; q0 += FFT_ILP) {
      double s0r[FFT_ILP], s0i[FFT_ILP], s1r[FFT_ILP], s1i[FFT_ILP];
      // Z[k] = (X[k] + conj X[128-k]) + i conj(W256^k) (X[k] - conj X[128-k])   (float -> double, :136)
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) {
        const int rl = wave * (FFT_ROWS / 4) + q0 + u;
        {
          double ar = ys_r[rl * STAGE_LD + k0], ai = ys_i[rl * STAGE_LD + k0];
          double br = ys_r[rl * STAGE_LD + (128 - k0)], bi = -(double)ys_i[rl * STAGE_LD + (128 - k0)];
          if (k0 == 0) { ai = 0.0; bi = 0.0; }   // rdft ignores Im of DC / Nyquist (RealFourierTransform.cs:120-123)
          double er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
          double pr = fma(dr, wk0.x, di * wk0.y), pi = fma(di, wk0.x, -(dr * wk0.y));   // d * conj(w)
          s0r[u] = er - pi;
          s0i[u] = ei + pr;
        }
        {
          double ar = ys_r[rl * STAGE_LD + k1], ai = ys_i[rl * STAGE_LD + k1];
          double br = ys_r[rl * STAGE_LD + (128 - k1)], bi = -(double)ys_i[rl * STAGE_LD + (128 - k1)];
          double er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
          double pr = fma(dr, wk1.x, di * wk1.y), pi = fma(di, wk1.x, -(dr * wk1.y));
          s1r[u] = er - pi;
          s1i[u] = ei + pr;
        }
      }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) { dit_stage<1, 5>(s0r[u], s0i[u], ltw); dit_stage<1, 5>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) { dit_stage<2, 4>(s0r[u], s0i[u], ltw); dit_stage<2, 4>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) { dit_stage<4, 3>(s0r[u], s0i[u], ltw); dit_stage<4, 3>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) { dit_stage<8, 2>(s0r[u], s0i[u], ltw); dit_stage<8, 2>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) { dit_stage<16, 1>(s0r[u], s0i[u], ltw); dit_stage<16, 1>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) { dit_stage<32, 0>(s0r[u], s0i[u], ltw); dit_stage<32, 0>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < FFT_ILP; u++) {
        const int rl = wave * (FFT_ROWS / 4) + q0 + u;
        const int row = rowtile * FFT_ROWS + rl;
        // in-lane stage: z[l] = s0 + s1 conj(W128^l), z[l+64] = s0 - s1 conj(W128^l); scale 2/n * 1/2 (:46,129)
        double qr = fma(s1r[u], ltw.c1, s1i[u] * ltw.s1), qi = fma(s1i[u], ltw.c1, -(s1r[u] * ltw.s1));
        const double scale = 1.0 / 256.0;
        double h0 = (s0r[u] + qr) * scale, h1 = (s0i[u] + qi) * scale;   // time samples 2l, 2l+1
        double g0 = (s0r[u] - qr) * scale, g1 = (s0i[u] - qi) * scale;   // time samples 128+2l, 128+2l+1
        if (row < nrows) {
          if (!pre) {
            float* out = rows[row].out;
            float o0 = (float)h0 + tail[rl][2 * lane];        // (float)y[i] + overlap[i]  (:148)
            float o1 = (float)h1 + tail[rl][2 * lane + 1];
            if (out) {
              *reinterpret_cast<GA_GLOBAL v2f*>(gptr(out) + (int64_t)t * kBlock + 2 * lane) = v2f{o0, o1};
            }
          }
          tail[rl][2 * lane] = (float)g0;                      // overlap[i] = (float)y[i + 128]  (:149)
          tail[rl][2 * lane + 1] = (float)g1;
        }
      }
    }
  }
  if (tb == nblocks) {
    __syncthreads();
    for (int idx = tid; idx < FFT_ROWS * kBlock; idx += 256) {
      int r = idx / kBlock, i = idx % kBlock;
      int row = rowtile * FFT_ROWS + r;
      if (row < nrows) overlap_out[(size_t)row * kBlock + i] = tail[r][i];
    }
  }
}

void launch_irfft_ola(hipStream_t s, const ConvRowIO* rows_dev, int nrows, int nblocks, ConvPlanes pl, const float* overlap_in,
                      float* overlap_out, Twiddles tw) {
  if (nrows <= 0 || nblocks <= 0) return;
  dim3 grid((nrows + FFT_ROWS - 1) / FFT_ROWS, (nblocks + OLA_RUN - 1) / OLA_RUN);
  hipLaunchKernelGGL(irfft_ola_kernel, grid, dim3(256), 0, s, rows_dev, nrows, nblocks, pl, overlap_in, overlap_out, tw);
}

// =====================================================================================================
//  Shared-IR spectral multiply-accumulate on the f32 matrix cores.
//  Reference: PartitionedConvolver.ProcessSpectralConvolution, PartitionedConvolver.cs:154-223
//     acc[k] = sum_{p<P} FDL[(w+p)%P][k] * H[p][k]            (one block, one channel-instance)
//  Here, for one bin k and all blocks t of the chunk and all rows r sharing H:
//     Y[r][t] = sum_u X[r][t0 + u] * G[u - j],  j = t - t0,  G[m] = H[P-1-m] (0 <= m < P), a banded Toeplitz GEMM.
//  Workgroup tile 128 rows x 64 blocks; wave tile 32 rows x 64 blocks = 2 x 4 MFMA 16x16 tiles, re+im accumulators.
//  Taps are processed in segments of <= MAC_PSEG so the reversed taps fit LDS whatever P is.
// =====================================================================================================
constexpr int MAC_ROWS = 128;
constexpr int MAC_TIME = 64;
constexpr int MAC_KC = 16;                   // time steps (u) per LDS chunk
constexpr int MAC_LD = MAC_ROWS + 16;        // padded row stride: lanes 16..31 land 16 banks away from lanes 0..15
constexpr int MAC_PSEG = 1024;
constexpr int MAC_GLEN = MAC_PSEG + 160;

// One 16-step chunk of the banded Toeplitz product for a wave tile (2 M-tiles x 4 N-tiles).
// N-tile n (blocks 16n..16n+15 of the time tile) is inside the band exactly for chunks [n, n + (Ps+14)/16]: the band
// edges coincide with chunk boundaries, so activity is decided per (chunk, N-tile) with no wasted MFMA, and the partial
// edge chunks are covered by the zero padding of the reversed tap table.  A fragments are read once per chunk, B
// fragments are double buffered one N-tile ahead so their LDS latency hides under the previous tile's 24 MFMAs.
__device__ __forceinline__ void mac_chunk(const float* __restrict xsr, const float* __restrict xsi, const float* __restrict g0,
                                          const float* __restrict g1, const float* __restrict g2, int c, int Ps, int wave, int la,
                                          int lk, f32x4 (&acc1)[2][4], f32x4 (&acc2)[2][4], f32x4 (&acc3)[2][4]) {
  float ar[4][2], ai[4][2], as[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ks++)
#pragma unroll
    for (int m = 0; m < 2; m++) {
      int off = (ks * 4 + lk) * MAC_LD + wave * 32 + m * 16 + la;
      ar[ks][m] = xsr[off];
      ai[ks][m] = xsi[off];
    }
  float br[2][4], bi[2][4], bs[2][4];
  const int gbase = c * MAC_KC + lk - la + 64;
#pragma unroll
  for (int ks = 0; ks < 4; ks++) {
    br[0][ks] = g0[gbase + ks * 4];
    bi[0][ks] = g1[gbase + ks * 4];
    bs[0][ks] = g2[gbase + ks * 4];
  }
#pragma unroll
  for (int ks = 0; ks < 4; ks++)
#pragma unroll
    for (int m = 0; m < 2; m++) as[ks][m] = ar[ks][m] + ai[ks][m];
  const int last = (Ps + 14) / 16;
#pragma unroll
  for (int n = 0; n < 4; n++) {
    const int cur = n & 1, nxt = cur ^ 1;
    if (n + 1 < 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ks++) {
        int gi = gbase + ks * 4 - 16 * (n + 1);
        br[nxt][ks] = g0[gi];
        bi[nxt][ks] = g1[gi];
        bs[nxt][ks] = g2[gi];
      }
    }
    if (c >= n && c <= n + last) {   // wave uniform
#pragma unroll
      for (int ks = 0; ks < 4; ks++)
#pragma unroll
        for (int m = 0; m < 2; m++) {
          // (a + ib)(c + id) = (ac - bd) + i(ad + bc)   (PartitionedConvolver.cs:195-204) on the matrix core
          acc1[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[ks][m], br[cur][ks], acc1[m][n], 0, 0, 0);
          acc2[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[ks][m], bi[cur][ks], acc2[m][n], 0, 0, 0);
          acc3[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(as[ks][m], bs[cur][ks], acc3[m][n], 0, 0, 0);
        }
    }
  }
}

__global__ __launch_bounds__(256) void spectral_mac_shared_kernel(ConvPlanes pl, const float* __restrict hr, const float* __restrict hi,
                                                                 int P, int ntt, int nrt, int total) {
  __shared__ __attribute__((aligned(16))) float xs[2][2][MAC_KC * MAC_LD];   // [buffer][re/im][u][row]
  __shared__ float gs[3][MAC_GLEN];                                           // reversed, zero padded taps: re, im, re + im

  // XCD-aware mapping: workgroups b and b+8 share an XCD (round-robin dispatch), give each XCD a contiguous range of
  // (bin, row tile, time tile) so neighbouring time tiles -- which re-read 8/9 of the same X rows -- share one L2.
  const int per = (total + 7) >> 3;
  const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (L >= total) return;
  const int tt = L % ntt;
  const int rt = (L / ntt) % nrt;
  const int k = L / (ntt * nrt);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = tt * MAC_TIME;
  const int row0 = rt * MAC_ROWS;
  const size_t xplane = (size_t)pl.tx * pl.rp;
  const float* __restrict xr = pl.xr + (size_t)k * xplane + row0;
  const float* __restrict xi = pl.xi + (size_t)k * xplane + row0;

  // Three-multiplication complex product (Gauss): with P1 = sum ar*br, P2 = sum ai*bi, P3 = sum (ar+ai)*(br+bi)
  //   re = P1 - P2 ,  im = P3 - P1 - P2      -> 3 MFMAs per complex tile step instead of 4.
  f32x4 acc1[2][4], acc2[2][4], acc3[2][4];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) {
      acc1[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc3[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  // global -> LDS staging is WAVE-PRIVATE: wave w only ever reads rows [32w, 32w+32) of a chunk, so it stages exactly
  // those rows itself (lane -> u = lane/8 and +8, rows 32w + 4*(lane%8) .. +3).  LDS operations of one wave execute in
  // order, so the main loop needs no workgroup barrier and the four waves drift freely over the shared matrix pipes.
  const int su = lane >> 3, sr4 = wave * 32 + (lane & 7) * 4;
  const int la = lane & 15, lk = lane >> 4;

  for (int pa = 0; pa < P; pa += MAC_PSEG) {
    const int Ps = min(MAC_PSEG, P - pa);
    const int u0 = P - (pa + Ps);                  // plane-row offset of this tap segment
    const int nch = (Ps + 63 + MAC_KC - 1) / MAC_KC + 0;
    __syncthreads();                               // previous segment's LDS fully consumed
    // G[m] for m in [-64, Ps+96): gs[m + 64]
    for (int i = tid; i < MAC_GLEN; i += 256) {
      int m = i - 64;
      bool ok = (m >= 0) && (m < Ps);
      int p = pa + Ps - 1 - m;
      float gr_ = ok ? hr[(size_t)k * P + p] : 0.f;
      float gi_ = ok ? hi[(size_t)k * P + p] : 0.f;
      gs[0][i] = gr_;
      gs[1][i] = gi_;
      gs[2][i] = gr_ + gi_;
    }
    // prologue: chunk 0
    float4 pr0, pr1, pi0, pi1;
    {
      size_t o0 = (size_t)(t0 + u0 + su) * pl.rp + sr4;
      size_t o1 = (size_t)(t0 + u0 + su + 8) * pl.rp + sr4;
      pr0 = *reinterpret_cast<const float4*>(xr + o0);
      pr1 = *reinterpret_cast<const float4*>(xr + o1);
      pi0 = *reinterpret_cast<const float4*>(xi + o0);
      pi1 = *reinterpret_cast<const float4*>(xi + o1);
    }
    *reinterpret_cast<float4*>(&xs[0][0][su * MAC_LD + sr4]) = pr0;
    *reinterpret_cast<float4*>(&xs[0][0][(su + 8) * MAC_LD + sr4]) = pr1;
    *reinterpret_cast<float4*>(&xs[0][1][su * MAC_LD + sr4]) = pi0;
    *reinterpret_cast<float4*>(&xs[0][1][(su + 8) * MAC_LD + sr4]) = pi1;
    __syncthreads();

    for (int c = 0; c < nch; c++) {
      const int buf = c & 1;
      const bool more = (c + 1) < nch;
      if (more) {
        size_t o0 = (size_t)(t0 + u0 + (c + 1) * MAC_KC + su) * pl.rp + sr4;
        size_t o1 = o0 + (size_t)8 * pl.rp;
        pr0 = *reinterpret_cast<const float4*>(xr + o0);
        pr1 = *reinterpret_cast<const float4*>(xr + o1);
        pi0 = *reinterpret_cast<const float4*>(xi + o0);
        pi1 = *reinterpret_cast<const float4*>(xi + o1);
      }
      mac_chunk(xs[buf][0], xs[buf][1], gs[0], gs[1], gs[2], c, Ps, wave, la, lk, acc1, acc2, acc3);
      if (more) {
        const int nb = buf ^ 1;
        *reinterpret_cast<float4*>(&xs[nb][0][su * MAC_LD + sr4]) = pr0;
        *reinterpret_cast<float4*>(&xs[nb][0][(su + 8) * MAC_LD + sr4]) = pr1;
        *reinterpret_cast<float4*>(&xs[nb][1][su * MAC_LD + sr4]) = pi0;
        *reinterpret_cast<float4*>(&xs[nb][1][(su + 8) * MAC_LD + sr4]) = pi1;
      }
      __builtin_amdgcn_wave_barrier();   // scheduling fence only: keep the staging writes ahead of the next chunk's reads
    }
  }

  // epilogue: C/D layout of v_mfma_f32_16x16x4_f32: column (time) = lane & 15, row = 4 * (lane >> 4) + reg
  const size_t yplane = (size_t)pl.ty * pl.rp;
  float* __restrict yr = pl.yr + (size_t)k * yplane + row0;
  float* __restrict yi = pl.yi + (size_t)k * yplane + row0;
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 4; n++) {
      int t = t0 + 16 * n + la;
      size_t o = (size_t)t * pl.rp + wave * 32 + m * 16 + lk * 4;
      *reinterpret_cast<f32x4*>(yr + o) = acc1[m][n] - acc2[m][n];
      *reinterpret_cast<f32x4*>(yi + o) = (acc3[m][n] - acc1[m][n]) - acc2[m][n];
    }
}

void launch_spectral_mac_shared(hipStream_t s, ConvPlanes pl, const float* hr, const float* hi, int P, int nblocks, int nrows) {
  if (nrows <= 0 || nblocks <= 0 || P <= 0) return;
  int ntt = (nblocks + MAC_TIME - 1) / MAC_TIME;
  int nrt = (nrows + MAC_ROWS - 1) / MAC_ROWS;
  int total = kBins * ntt * nrt;
  int grid = ((total + 7) / 8) * 8;
  hipLaunchKernelGGL(spectral_mac_shared_kernel, dim3(grid), dim3(256), 0, s, pl, hr, hi, P, ntt, nrt, total);
}

// =====================================================================================================
//  Formulation B kernels (see ga_kernels.hpp): planes [row][bin][block], block index fastest.
// =====================================================================================================
constexpr int FB_RUN = 32;            // blocks per workgroup in the B-layout FFT kernels: 32 blocks = one 128-byte line per bin
// Staging tile [block][bin] in LDS (64 banks of 4 bytes).  A lane of the 128-point transforms owns bins (2m, 2m + 1) with
// m = rev6(lane) and needs their mirrors (128 - 2m, 127 - 2m): with the bins in natural order every one of those accesses
// is a 2-way bank conflict (64 lanes on the even or the odd banks only).  So a column keeps its even bins packed in
// E[0..64] (bin 2e at e) and its odd bins in O[0..63] (bin 2o + 1 at FB_OB + o): all four accesses become 4-byte operations
// whose 64 lanes hit 64 different banks (m, 64 - m, FB_OB + m, FB_OB + 63 - m are permutations of the banks).
// FB_OB == 4 (mod 8) and FB_KP == 18 (mod 64) keep the transposed side conflict-free too: a wave covers 8 consecutive bins
// (banks b..b+3 from E, b+4..b+7 from O) of 8 columns 4 apart (4 * FB_KP == 8 mod 64).
constexpr int FB_OB = 68;
constexpr int FB_KP = 146;
__device__ __forceinline__ int fb_pos(int k) { return (k & 1) * FB_OB + (k >> 1); }

// forward: workgroup = (x-row, run of 32 blocks); wave w transforms blocks w, w+4, ..., w+28 of the run, 4 in flight
template <class T>
__global__ __launch_bounds__(256) void rfft_fwd_b_kernel(const ConvRowIO* __restrict xrows, int nx, int nblocks, int hist,
                                                         ConvPlanesB pl, Twiddles tw, int row0) {
  // staging tile [block of the run][bin] in the even/odd layout described at FB_KP
  __shared__ __attribute__((aligned(16))) float st_r[FB_RUN * FB_KP];
  __shared__ __attribute__((aligned(16))) float st_i[FB_RUN * FB_KP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // consecutive workgroups take consecutive runs of ONE row: their 128-byte pieces of every bin line are adjacent in HBM
  const int xrow = row0 + blockIdx.y;
  const int t0 = blockIdx.x * FB_RUN;
  const LaneTwT<T> ltw = load_lane_tw_t<T>(tw.w128, lane);
  const int m = rev6(lane);
  const int k0 = 2 * m, k1 = 2 * m + 1;
  const int src0 = rev6((64 - m) & 63), src1 = 63 - lane;
  const T wk0x = (T)tw.w256[k0].x, wk0y = (T)tw.w256[k0].y;
  const T wk1x = (T)tw.w256[k1].x, wk1y = (T)tw.w256[k1].y;
  const GA_GLOBAL float* in = gptr(xrows[xrow].in);
  GA_TL(0);

  // all eight input blocks of this wave are requested up front: their HBM latency overlaps the first batch of transforms
  float inr[FB_RUN / 4], ini[FB_RUN / 4];
#pragma unroll
  for (int q = 0; q < FB_RUN / 4; q++) {
    const int t = t0 + wave + 4 * q;
    inr[q] = 0.f;
    ini[q] = 0.f;
    if (in && t < nblocks) {
      const GA_GLOBAL float* p = in + (int64_t)t * kBlock + 2 * lane;
      inr[q] = p[0];
      ini[q] = p[1];
    }
  }
#pragma unroll
  for (int bq = 0; bq < FB_RUN / 16; bq++) {
    T s0r[4], s0i[4], s1r[4], s1i[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      s0r[u] = (T)inr[bq * 4 + u];   // float -> T, PartitionedConvolver.cs:106
      s0i[u] = (T)ini[bq * 4 + u];
      s1r[u] = fma(s0r[u], ltw.c1, -(s0i[u] * ltw.s1));
      s1i[u] = fma(s0r[u], ltw.s1, s0i[u] * ltw.c1);
    }
#ifndef GA_EXP_SKIP_FFT
#pragma unroll
    for (int u = 0; u < 4; u++) { dif_stage<32, 0>(s0r[u], s0i[u], ltw); dif_stage<32, 0>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < 4; u++) { dif_stage<16, 1>(s0r[u], s0i[u], ltw); dif_stage<16, 1>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < 4; u++) { dif_stage<8, 2>(s0r[u], s0i[u], ltw); dif_stage<8, 2>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < 4; u++) { dif_stage<4, 3>(s0r[u], s0i[u], ltw); dif_stage<4, 3>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < 4; u++) { dif_stage<2, 4>(s0r[u], s0i[u], ltw); dif_stage<2, 4>(s1r[u], s1i[u], ltw); }
#pragma unroll
    for (int u = 0; u < 4; u++) { dif_stage<1, 5>(s0r[u], s0i[u], ltw); dif_stage<1, 5>(s1r[u], s1i[u], ltw); }
#endif

#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int tl = wave + 4 * (bq * 4 + u);   // block within the run
      float x0r, x0i;
      {
        T ax = s0r[u], ay = s0i[u];
        T bx = shfl_d(ax, src0), by = shfl_d(ay, src0);
        T er = (T)0.5 * (ax + bx), ei = (T)0.5 * (ay - by);
        T dr = (T)0.5 * (ax - bx), di = (T)0.5 * (ay + by);
        T pr = fma(dr, wk0x, -(di * wk0y)), pi = fma(dr, wk0y, di * wk0x);
        float xr = (float)(er + pi), xi = (float)(ei - pr);
        if (k0 == 0) {
          xi = 0.f;
          st_r[tl * FB_KP + 64] = (float)(ax - ay);   // bin 128 = E[64]
          st_i[tl * FB_KP + 64] = 0.f;
        }
        x0r = xr;
        x0i = xi;
      }
      {
        T ax = s1r[u], ay = s1i[u];
        T bx = shfl_d(ax, src1), by = shfl_d(ay, src1);
        T er = (T)0.5 * (ax + bx), ei = (T)0.5 * (ay - by);
        T dr = (T)0.5 * (ax - bx), di = (T)0.5 * (ay + by);
        T pr = fma(dr, wk1x, -(di * wk1y)), pi = fma(dr, wk1y, di * wk1x);
        st_r[tl * FB_KP + m] = x0r;                           // bin 2m
        st_i[tl * FB_KP + m] = x0i;
        st_r[tl * FB_KP + FB_OB + m] = (float)(er + pi);      // bin 2m + 1
        st_i[tl * FB_KP + FB_OB + m] = (float)(ei - pr);
      }
    }
  }
  GA_TL(1);
  __syncthreads();
  GA_TL(2);
  // store: per bin 32 consecutive blocks = 128 bytes, 8 lanes x 16 B
  const size_t rowbase = (size_t)xrow * kBins * pl.tx + hist + t0;
  for (int idx = tid; idx < kBins * 8; idx += 256) {
    int k = idx >> 3, q = (idx & 7) * 4;
    if (t0 + q >= nblocks) continue;
    size_t o = rowbase + (size_t)k * pl.tx + q;
#ifdef GA_EXP_CONTIG
    o = ((size_t)(xrow * gridDim.x + blockIdx.x) * kBins + k) * 32 + q;
#endif
    const int pk = q * FB_KP + fb_pos(k);
    *reinterpret_cast<float4*>(pl.xr + o) = make_float4(st_r[pk], st_r[pk + FB_KP], st_r[pk + 2 * FB_KP], st_r[pk + 3 * FB_KP]);
    *reinterpret_cast<float4*>(pl.xi + o) = make_float4(st_i[pk], st_i[pk + FB_KP], st_i[pk + 2 * FB_KP], st_i[pk + 3 * FB_KP]);
  }
  GA_TL(3);
}
void launch_rfft_fwd_b(hipStream_t s, const ConvRowIO* xrows_dev, int nx, int nblocks, int hist, ConvPlanesB pl, Twiddles tw, bool fp64) {
  if (nx <= 0 || nblocks <= 0) return;
  for (int r0 = 0; r0 < nx; r0 += 65535) {   // gridDim.y limit
    dim3 grid((nblocks + FB_RUN - 1) / FB_RUN, std::min(65535, nx - r0));
    if (fp64) hipLaunchKernelGGL(rfft_fwd_b_kernel<double>, grid, dim3(256), 0, s, xrows_dev, nx, nblocks, hist, pl, tw, r0);
    else hipLaunchKernelGGL(rfft_fwd_b_kernel<float>, grid, dim3(256), 0, s, xrows_dev, nx, nblocks, hist, pl, tw, r0);
  }
}

// MAC B: workgroup = (set, bin, 256-block time tile); wave = 64 blocks (4 M-tiles) x 16 columns; taps in segments of 256
constexpr int MB_TW = 256;
constexpr int MB_PSEG = 256;
constexpr int MB_HLD = MB_PSEG + 2;            // column stride: 2j + kk distinct banks for lanes (j, kk)
constexpr int MB_XLEN = MB_TW + MB_PSEG + 8;

__global__ __launch_bounds__(256) void spectral_mac_b_kernel(const ConvSetB* __restrict sets, int nblocks, int hist, ConvPlanesB pl,
                                                             int ntt) {
  __shared__ float hs[3][16 * MB_HLD];   // taps re, im, re+im  [column][p]
  __shared__ float xw[3][MB_XLEN];       // input spectra window re, im, re+im : xw[i] = X[t0 - pa - (Ps-1) + i ... ]
  const ConvSetB* __restrict Sp = &sets[blockIdx.z];   // accessed in place: a by-value copy (32 pointers) would go to scratch
  struct { int x, y0, ncol, P; } S = {Sp->x, Sp->y0, Sp->ncol, Sp->P};
  const int k = blockIdx.y;
  const int tt = blockIdx.x;
  const int t0 = tt * MB_TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int la = lane & 15, lk = lane >> 4;
  const int P = S.P;
  const float* __restrict xr = pl.xr + ((size_t)S.x * kBins + k) * pl.tx;
  const float* __restrict xi = pl.xi + ((size_t)S.x * kBins + k) * pl.tx;

  f32x4 acc1[4], acc2[4], acc3[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    acc1[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc3[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int pa = 0; pa < P; pa += MB_PSEG) {
    const int Ps = min(MB_PSEG, P - pa);
    const int Ps4 = (Ps + 3) & ~3;
    __syncthreads();
    // taps of this segment, zero padded to a multiple of 4 (and unused columns zero)
    for (int i = tid; i < 16 * MB_HLD; i += 256) {
      int j = i / MB_HLD, p = i % MB_HLD;
      float vr = 0.f, vi = 0.f;
      if (j < S.ncol && p < Ps) {
        vr = Sp->hr[j][(size_t)k * P + pa + p];
        vi = Sp->hi[j][(size_t)k * P + pa + p];
      }
      hs[0][i] = vr;
      hs[1][i] = vi;
      hs[2][i] = vr + vi;
    }
    // window: plane index of block t is hist + t; tap p needs block t - p.  xw[i] <-> block  t0 - pa - (Ps4 - 1) + i
    const int wlen = MB_TW + Ps4 - 1;
    const int b0 = t0 - pa - (Ps4 - 1);
    for (int i = tid; i < wlen; i += 256) {
      int blk = b0 + i;            // may be < -hist (before any history): zero
      float vr = 0.f, vi = 0.f;
      if (blk >= -hist && blk < nblocks) {
        vr = xr[hist + blk];
        vi = xi[hist + blk];
      }
      xw[0][i] = vr;
      xw[1][i] = vi;
      xw[2][i] = vr + vi;
    }
    __syncthreads();
    // Y[t] += sum_p X[t - p] H[p];  A[i][kk] = X[t0w + 16m + i - (p0 + kk)] -> xw index (Ps4 - 1) + 64w + 16m + i - p0 - kk
    const int abase = (Ps4 - 1) + wave * 64 + la - lk;
    for (int p0 = 0; p0 < Ps4; p0 += 4) {
      const int hoff = la * MB_HLD + p0 + lk;
      float br = hs[0][hoff], bi = hs[1][hoff], bs = hs[2][hoff];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int xo = abase + 16 * m - p0;
        float ar = xw[0][xo], ai = xw[1][xo], as = xw[2][xo];
        acc1[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar, br, acc1[m], 0, 0, 0);
        acc2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai, bi, acc2[m], 0, 0, 0);
        acc3[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(as, bs, acc3[m], 0, 0, 0);
      }
    }
  }
  // D layout: column (IR channel) = lane & 15, row (block) = 4 * (lane >> 4) + reg -> four consecutive blocks per lane
  if (la < S.ncol) {
    float* __restrict yr = pl.yr + ((size_t)(S.y0 + la) * kBins + k) * pl.ty;
    float* __restrict yi = pl.yi + ((size_t)(S.y0 + la) * kBins + k) * pl.ty;
#pragma unroll
    for (int m = 0; m < 4; m++) {
      int t = t0 + wave * 64 + 16 * m + 4 * lk;
      if (t < pl.ty) {
        *reinterpret_cast<f32x4*>(yr + t) = acc1[m] - acc2[m];
        *reinterpret_cast<f32x4*>(yi + t) = (acc3[m] - acc1[m]) - acc2[m];
      }
    }
  }
}
void launch_spectral_mac_b(hipStream_t s, const ConvSetB* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl) {
  if (nsets <= 0 || nblocks <= 0) return;
  int ntt = (nblocks + MB_TW - 1) / MB_TW;
  for (int z0 = 0; z0 < nsets; z0 += 32768) {   // gridDim.z limit
    int nz = std::min(32768, nsets - z0);
    hipLaunchKernelGGL(spectral_mac_b_kernel, dim3(ntt, kBins, nz), dim3(256), 0, s, sets_dev + z0, nblocks, hist, pl, ntt);
  }
}

// =====================================================================================================
//  Reference-order partition sum (formulation R): PartitionedConvolver.ProcessSpectralConvolution
//  (PartitionedConvolver.cs:154-223) evaluated EXACTLY as the reference evaluates it -- for every (row, block t, bin k)
//      acc = 0 ;  for p = 0 .. P-1:  acc.re += (X.re * H.re) - (X.im * H.im) ;  acc.im += (X.re * H.im) + (X.im * H.re)
//  with X = X[t - p][k], H = H[p][k], every operation a separately rounded float32 operation (the reference's AVX
//  Multiply / Subtract / Add and its scalar tail: no fused multiply-add anywhere) and the partitions in ascending order.
//  All (t, k) chains are independent: only the order INSIDE a chain is serial.  Around it run the double-precision
//  256-point transforms of the B layout (rfft_fwd_b_kernel<double> / irfft_ola_b_kernel<double>), so a convolver served by
//  this route reproduces the reference's float32 output bit for bit (up to the ~1e-9 chance per value that two correct
//  double-precision transforms round to different floats).  The planner takes it for convolvers whose output reaches
//  arithmetic that amplifies or quantises last-bit differences (Context::refOrderSensitivity).
//
//  Workgroup = (set, bin, tile of 1024 blocks), 4 waves; lane l of wave w owns the four consecutive blocks
//  t0 + 256 w + 4 l + {0..3}.  The window X[t0 - (Ps-1) .. t0 + 1023] of the bin and the taps H[p] sit in LDS as separate
//  re / im float arrays: per 4 taps a lane reads ONE aligned 16-byte quad of each (consecutive lanes: consecutive quads, no
//  bank conflict) and the taps by broadcast; the window slides through registers (two quads cover four taps x four blocks).  128 VALU operations per 4 LDS reads; taps beyond 1024 in further segments, accumulators carried.
// =====================================================================================================
constexpr int RM_TW = 1024;     // blocks per workgroup
constexpr int RM_PSEG = 1024;   // taps per segment

// Each operation is spelled out as one scalar VALU instruction: left to itself the compiler pairs the products and sums of
// neighbouring chains into v_pk_mul_f32 / v_pk_add_f32 and pays for the pairs with moves -- 58 v_mov per 68 packed operations in
// this loop, while a packed f32 instruction costs the same SIMD cycles as the two scalar ones it replaces on gfx950
// (profiles/r01_micro_pk_f32_issue_rate.txt).
#define GA_RM_OP(op, r, a, b) asm(op " %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
__device__ __forceinline__ void rm_cmac(float& ar, float& ai, float dr, float di, float hr, float hi) {
  // (dr * ir) - (di * ii) ; (dr * ii) + (di * ir) ; acc += ...   PartitionedConvolver.cs:196-206,218-219
  float p0, p1, p2, p3, re, im;
  GA_RM_OP("v_mul_f32", p0, dr, hr);
  GA_RM_OP("v_mul_f32", p1, di, hi);
  GA_RM_OP("v_mul_f32", p2, dr, hi);
  GA_RM_OP("v_mul_f32", p3, di, hr);
  GA_RM_OP("v_sub_f32", re, p0, p1);
  GA_RM_OP("v_add_f32", im, p2, p3);
  GA_RM_OP("v_add_f32", ar, ar, re);
  GA_RM_OP("v_add_f32", ai, ai, im);
}

__global__ __launch_bounds__(256) void refmac_kernel(const ConvSetB* __restrict sets, int nblocks, int hist, ConvPlanesB pl) {
  // w[i] <-> block  b0 + i ,  b0 = t0 - pa - (Ps4 - 1):  output block t0 + T + r needs, for tap pa + pp,  w[c + r - pp],  c = T + Ps4 - 1
  __shared__ __attribute__((aligned(16))) float wr[RM_TW + RM_PSEG + 4];
  __shared__ __attribute__((aligned(16))) float wi[RM_TW + RM_PSEG + 4];
  __shared__ __attribute__((aligned(16))) float hrs[RM_PSEG];
  __shared__ __attribute__((aligned(16))) float his[RM_PSEG];
  const ConvSetB* __restrict Sp = &sets[blockIdx.z];
  const int Sx = Sp->x, Sy0 = Sp->y0, ncol = Sp->ncol, P = Sp->P;
  const int k = blockIdx.y;
  const int t0 = blockIdx.x * RM_TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int T = wave * 256 + 4 * lane;
  const GA_GLOBAL float* xr = gptr(pl.xr + ((size_t)Sx * kBins + k) * pl.tx);
  const GA_GLOBAL float* xi = gptr(pl.xi + ((size_t)Sx * kBins + k) * pl.tx);
  const bool oneSeg = P <= RM_PSEG;
  for (int j = 0; j < ncol; j++) {
    const GA_GLOBAL float* hr = gptr(Sp->hr[j] + (size_t)k * P);
    const GA_GLOBAL float* hi = gptr(Sp->hi[j] + (size_t)k * P);
    float ar[4] = {0.f, 0.f, 0.f, 0.f}, ai[4] = {0.f, 0.f, 0.f, 0.f};   // Array.Clear(_accReal / _accImag), :157-158
    for (int pa = 0; pa < P; pa += RM_PSEG) {
      const int Ps = min(RM_PSEG, P - pa);
      const int Ps4 = (Ps + 3) & ~3;   // (taps beyond Ps are zero: they add (+-)0 to the sums, which leaves every finite sum as it is)
      __syncthreads();
      for (int p = tid; p < Ps4; p += 256) {
        hrs[p] = p < Ps ? hr[pa + p] : 0.f;
        his[p] = p < Ps ? hi[pa + p] : 0.f;
      }
      if (!(oneSeg && j > 0)) {   // (one segment: the window is the same for every column)
        const int b0 = t0 - pa - (Ps4 - 1);
        const int wlen = RM_TW + Ps4 + 3;   // (+ 4: the unused tail of the newest quad)
        for (int i = tid; i < wlen; i += 256) {
          const int blk = b0 + i;   // plane index of block t is hist + t ; outside [-hist, nblocks): zero
          float vr = 0.f, vi = 0.f;
          if (blk >= -hist && blk < nblocks) {
            vr = xr[hist + blk];
            vi = xi[hist + blk];
          }
          wr[i] = vr;
          wi[i] = vi;
        }
      }
      __syncthreads();
      if (t0 + T < nblocks) {
        // quad Q_q = w[c - 4 q - 3 .. c - 4 q] (aligned: c == 3 mod 4).  Tap 4 q + s reads, for output r, w[c + r - 4 q - s]:
        //   s = 0: (Q_q.w, Q_{q-1}.x, .y, .z)   s = 1: (Q_q.z, Q_q.w, Q_{q-1}.x, .y)   s = 2: (Q_q.y, .z, .w, Q_{q-1}.x)   s = 3: Q_q
        int qi = T + Ps4;
        float4 pr = *reinterpret_cast<const float4*>(&wr[qi]);   // Q_{-1}: the three blocks behind the lane's first one
        float4 pi = *reinterpret_cast<const float4*>(&wi[qi]);
        for (int q = 0; q < Ps4; q += 4) {
          qi -= 4;
          const float4 cr = *reinterpret_cast<const float4*>(&wr[qi]);
          const float4 ci = *reinterpret_cast<const float4*>(&wi[qi]);
          const float4 gr = *reinterpret_cast<const float4*>(&hrs[q]);   // (broadcast reads)
          const float4 gi = *reinterpret_cast<const float4*>(&his[q]);
          rm_cmac(ar[0], ai[0], cr.w, ci.w, gr.x, gi.x);
          rm_cmac(ar[1], ai[1], pr.x, pi.x, gr.x, gi.x);
          rm_cmac(ar[2], ai[2], pr.y, pi.y, gr.x, gi.x);
          rm_cmac(ar[3], ai[3], pr.z, pi.z, gr.x, gi.x);
          rm_cmac(ar[0], ai[0], cr.z, ci.z, gr.y, gi.y);
          rm_cmac(ar[1], ai[1], cr.w, ci.w, gr.y, gi.y);
          rm_cmac(ar[2], ai[2], pr.x, pi.x, gr.y, gi.y);
          rm_cmac(ar[3], ai[3], pr.y, pi.y, gr.y, gi.y);
          rm_cmac(ar[0], ai[0], cr.y, ci.y, gr.z, gi.z);
          rm_cmac(ar[1], ai[1], cr.z, ci.z, gr.z, gi.z);
          rm_cmac(ar[2], ai[2], cr.w, ci.w, gr.z, gi.z);
          rm_cmac(ar[3], ai[3], pr.x, pi.x, gr.z, gi.z);
          rm_cmac(ar[0], ai[0], cr.x, ci.x, gr.w, gi.w);
          rm_cmac(ar[1], ai[1], cr.y, ci.y, gr.w, gi.w);
          rm_cmac(ar[2], ai[2], cr.z, ci.z, gr.w, gi.w);
          rm_cmac(ar[3], ai[3], cr.w, ci.w, gr.w, gi.w);
          pr = cr;
          pi = ci;
        }
      }
    }
    const int t = t0 + T;
    if (t < nblocks) {   // (t is a multiple of 4 and so is the planes' pitch: the quad stays inside the row)
      float* __restrict yr = pl.yr + ((size_t)(Sy0 + j) * kBins + k) * pl.ty;
      float* __restrict yi = pl.yi + ((size_t)(Sy0 + j) * kBins + k) * pl.ty;
      *reinterpret_cast<float4*>(yr + t) = make_float4(ar[0], ar[1], ar[2], ar[3]);
      *reinterpret_cast<float4*>(yi + t) = make_float4(ai[0], ai[1], ai[2], ai[3]);
    }
  }
}
void launch_refmac(hipStream_t s, const ConvSetB* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl) {
  if (nsets <= 0 || nblocks <= 0) return;
  const int ntt = (nblocks + RM_TW - 1) / RM_TW;
  for (int z0 = 0; z0 < nsets; z0 += 32768) {   // gridDim.z limit
    const int nz = std::min(32768, nsets - z0);
    hipLaunchKernelGGL(refmac_kernel, dim3(ntt, kBins, nz), dim3(256), 0, s, sets_dev + z0, nblocks, hist, pl);
  }
}

// inverse + overlap-add, B layout: workgroup = (y-row, run of 32 blocks).  All 33 inverse transforms of the run (the extra
// one recovers the tail of the block before the run) are independent: 4 waves x 2 batches of 4, and the 33rd alone on one
// wave.  Heads and tails land in LDS, then
// out[t] = (float)head[t] + tail[t-1] is written as one contiguous, fully coalesced 16 KB range.
template <class T>
__global__ __launch_bounds__(256) void irfft_ola_b_kernel(const ConvRowIO* __restrict yrows, int ny, int nblocks, ConvPlanesB pl,
                                                          const float* const* __restrict overlap_in, float* const* __restrict overlap_out,
                                                          Twiddles tw, int row0) {
  // staging tile [33 columns][bin] (even/odd layout, see FB_KP): column c <-> block ta - 1 + c ; the memory is reused for
  // the head/tail tiles afterwards
  constexpr int NC = FB_RUN + 1;
  __shared__ __attribute__((aligned(16))) float smem[2 * NC * FB_KP];   // 9636 floats >= (33 + 33) * 128 = 8448
  float* sr = smem;
  float* si = smem + NC * FB_KP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = row0 + blockIdx.y;
  const int ta = blockIdx.x * FB_RUN;
  const int tb = min(ta + FB_RUN, nblocks);
  const int nrun = tb - ta;
  const LaneTwT<T> ltw = load_lane_tw_t<T>(tw.w128, lane);
  const int k0 = rev6(lane) << 1, k1 = k0 | 1;
  const T wk0x = (T)tw.w256[k0].x, wk0y = (T)tw.w256[k0].y;
  const T wk1x = (T)tw.w256[k1].x, wk1y = (T)tw.w256[k1].y;
  const float* __restrict yr = pl.yr + (size_t)row * kBins * pl.ty;
  const float* __restrict yi = pl.yi + (size_t)row * kBins * pl.ty;
  GA_TL(0);
  // columns 1..nrun <- blocks ta .. tb-1 (16-byte loads, 8 lanes per 128-byte bin line); column 0 <- block ta - 1
  {
    // all global loads of the tile first (10 x 16 B in flight per thread), then the LDS transposition: one exposed HBM latency
    // per workgroup instead of one per loop iteration
    constexpr int NIT = (kBins * 8 + 255) / 256;   // 5
    float4 vr[NIT], vi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + 256 * it;
      const int k = idx >> 3, q = (idx & 7) * 4;
      vr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      vi[it] = vr[it];
      if (idx < kBins * 8 && ta + q < nblocks) {
#ifdef GA_EXP_CONTIG   // (measurement only: the tile as one contiguous 16.5 KB range per plane)
        vr[it] = *reinterpret_cast<const float4*>(pl.yr + ((size_t)(row * gridDim.x + blockIdx.x) * kBins + k) * 32 + q);
        vi[it] = *reinterpret_cast<const float4*>(pl.yi + ((size_t)(row * gridDim.x + blockIdx.x) * kBins + k) * 32 + q);
#else
        vr[it] = *reinterpret_cast<const float4*>(yr + (size_t)k * pl.ty + ta + q);
        vi[it] = *reinterpret_cast<const float4*>(yi + (size_t)k * pl.ty + ta + q);
#endif
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int idx = tid + 256 * it;
      const int k = idx >> 3, q = (idx & 7) * 4;
      if (idx < kBins * 8) {
        const int pk = (1 + q) * FB_KP + fb_pos(k);
        sr[pk] = vr[it].x; sr[pk + FB_KP] = vr[it].y; sr[pk + 2 * FB_KP] = vr[it].z; sr[pk + 3 * FB_KP] = vr[it].w;
        si[pk] = vi[it].x; si[pk + FB_KP] = vi[it].y; si[pk + 2 * FB_KP] = vi[it].z; si[pk + 3 * FB_KP] = vi[it].w;
      }
    }
  }
  if (tid < kBins) {
    float vr = 0.f, vi = 0.f;
    if (ta > 0) {
      vr = yr[(size_t)tid * pl.ty + ta - 1];
      vi = yi[(size_t)tid * pl.ty + ta - 1];
    }
    sr[fb_pos(tid)] = vr;
    si[fb_pos(tid)] = vi;
  }
  GA_TL(1);
  __syncthreads();
  GA_TL(2);
  // transform columns c = 0..nrun (c = 0 only when ta > 0); results kept in registers until the staging tile is dead.
  // Batches 0 and 1: column (4 * bq + u) * 4 + wave ; batch 2: column 32 alone, on wave (run & 3) -- which wave takes it
  // rotates with the run, so the extra work spreads over the SIMDs.  (As a fifth transform of batch 1 it costs 50 more
  // VGPRs and a wave of occupancy: slower.)
  const int m = rev6(lane);
  float hd[3][4][2], tl[3][4][2];
#pragma unroll
  for (int bq = 0; bq < 3; bq++) {
    constexpr int kFull = 4;
    const int NU = bq < 2 ? kFull : 1;
    const bool act = bq < 2 ? (bq * 16 <= nrun) : (wave == (int)(blockIdx.x & 3) && nrun == FB_RUN);   // wave uniform
#pragma unroll
    for (int u = 0; u < kFull; u++) hd[bq][u][0] = hd[bq][u][1] = tl[bq][u][0] = tl[bq][u][1] = 0.f;
    if (act) {
      T s0r[kFull], s0i[kFull], s1r[kFull], s1i[kFull];
#pragma unroll
      for (int u = 0; u < NU; u++) {
        const int c = bq < 2 ? (bq * 4 + u) * 4 + wave : FB_RUN;
        const float* cr = sr + c * FB_KP;
        const float* ci = si + c * FB_KP;
        {
          T ar = cr[m], ai = ci[m];                               // bin 2m
          T br = cr[64 - m], bi = -(T)ci[64 - m];                 // bin 128 - 2m
          if (m == 0) { ai = (T)0.0; bi = (T)0.0; }
          T er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
          T pr = fma(dr, wk0x, di * wk0y), pi = fma(di, wk0x, -(dr * wk0y));
          s0r[u] = er - pi;
          s0i[u] = ei + pr;
        }
        {
          T ar = cr[FB_OB + m], ai = ci[FB_OB + m];               // bin 2m + 1
          T br = cr[FB_OB + 63 - m], bi = -(T)ci[FB_OB + 63 - m]; // bin 127 - 2m
          T er = ar + br, ei = ai + bi, dr = ar - br, di = ai - bi;
          T pr = fma(dr, wk1x, di * wk1y), pi = fma(di, wk1x, -(dr * wk1y));
          s1r[u] = er - pi;
          s1i[u] = ei + pr;
        }
      }
#ifndef GA_EXP_SKIP_FFT
#pragma unroll
      for (int u = 0; u < NU; u++) { dit_stage<1, 5>(s0r[u], s0i[u], ltw); dit_stage<1, 5>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < NU; u++) { dit_stage<2, 4>(s0r[u], s0i[u], ltw); dit_stage<2, 4>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < NU; u++) { dit_stage<4, 3>(s0r[u], s0i[u], ltw); dit_stage<4, 3>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < NU; u++) { dit_stage<8, 2>(s0r[u], s0i[u], ltw); dit_stage<8, 2>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < NU; u++) { dit_stage<16, 1>(s0r[u], s0i[u], ltw); dit_stage<16, 1>(s1r[u], s1i[u], ltw); }
#pragma unroll
      for (int u = 0; u < NU; u++) { dit_stage<32, 0>(s0r[u], s0i[u], ltw); dit_stage<32, 0>(s1r[u], s1i[u], ltw); }
#endif
#pragma unroll
      for (int u = 0; u < NU; u++) {
        T qr = fma(s1r[u], ltw.c1, s1i[u] * ltw.s1), qi = fma(s1i[u], ltw.c1, -(s1r[u] * ltw.s1));
        const T scale = (T)(1.0 / 256.0);
        hd[bq][u][0] = (float)((s0r[u] + qr) * scale);   // time samples 2l, 2l+1
        hd[bq][u][1] = (float)((s0i[u] + qi) * scale);
        tl[bq][u][0] = (float)((s0r[u] - qr) * scale);   // time samples 128+2l, 128+2l+1
        tl[bq][u][1] = (float)((s0i[u] - qi) * scale);
      }
    }
  }
  GA_TL(3);
  __syncthreads();   // every wave is done reading the staging tile: reuse it as head[33][128] | tail[33][128]
  GA_TL(4);
  float* head = smem;
  float* tail = smem + 33 * kBlock;
#pragma unroll
  for (int bq = 0; bq < 3; bq++) {
    const int NU = bq < 2 ? 4 : 1;
    const bool mine = bq < 2 || wave == (int)(blockIdx.x & 3);
#pragma unroll
    for (int u = 0; u < NU; u++) {
      const int c = bq < 2 ? (bq * 4 + u) * 4 + wave : FB_RUN;
      if (mine && c <= nrun && (c > 0 || ta > 0)) {
        *reinterpret_cast<float2*>(&head[c * kBlock + 2 * lane]) = make_float2(hd[bq][u][0], hd[bq][u][1]);
        *reinterpret_cast<float2*>(&tail[c * kBlock + 2 * lane]) = make_float2(tl[bq][u][0], tl[bq][u][1]);
      }
    }
  }
  if (ta == 0 && tid < kBlock) tail[tid] = ldg1(overlap_in[row] + tid);   // incoming overlap of the chunk's first block
  __syncthreads();
  GA_TL(5);
  float* out = yrows[row].out;
  if (out) {
    float* o4 = out + (int64_t)ta * kBlock;
    for (int idx = tid; idx < nrun * (kBlock / 4); idx += 256) {
      const int c = idx / (kBlock / 4) + 1, s4 = (idx % (kBlock / 4)) * 4;
      const float4 h = *reinterpret_cast<const float4*>(&head[c * kBlock + s4]);
      const float4 t = *reinterpret_cast<const float4*>(&tail[(c - 1) * kBlock + s4]);
      stg4(o4 + 4 * idx, v4f{h.x + t.x, h.y + t.y, h.z + t.z, h.w + t.w});   // (float)y[i] + overlap[i]  (:148)
    }
  }
  if (tb == nblocks && tid < kBlock) stg1(overlap_out[row] + tid, tail[nrun * kBlock + tid]);   // overlap[i] = (float)y[i+128]  (:149)
  GA_TL(6);
}
void launch_irfft_ola_b(hipStream_t s, const ConvRowIO* yrows_dev, int ny, int nblocks, ConvPlanesB pl,
                        const float* const* overlap_in_dev, float* const* overlap_out_dev, Twiddles tw, bool fp64) {
  if (ny <= 0 || nblocks <= 0) return;
  for (int r0 = 0; r0 < ny; r0 += 65535) {   // gridDim.y limit
    dim3 grid((nblocks + FB_RUN - 1) / FB_RUN, std::min(65535, ny - r0));
    if (fp64) hipLaunchKernelGGL(irfft_ola_b_kernel<double>, grid, dim3(256), 0, s, yrows_dev, ny, nblocks, pl, overlap_in_dev, overlap_out_dev, tw, r0);
    else hipLaunchKernelGGL(irfft_ola_b_kernel<float>, grid, dim3(256), 0, s, yrows_dev, ny, nblocks, pl, overlap_in_dev, overlap_out_dev, tw, r0);
  }
}

__global__ __launch_bounds__(128) void stale_copy_kernel(const StaleJob* __restrict jobs) {
  const StaleJob j = jobs[blockIdx.x];
  const int i = threadIdx.x;
  gptr(j.dst)[i] = j.src ? ldg1(j.src + i) * (j.curve ? ldg1(j.curve + i) : j.scale) : 0.f;
}
void launch_stale_copy(hipStream_t s, const StaleJob* jobs_dev, int njobs) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(stale_copy_kernel, dim3(njobs), dim3(128), 0, s, jobs_dev);
}

// one wavefront per (job, bin): 4 bins per workgroup (a workgroup per bin moved 2 KB and was launch-rate bound)
__global__ __launch_bounds__(256) void hist_copy_b_kernel(const HistJobB* __restrict jobs) {
  const HistJobB j = jobs[blockIdx.y];
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= kBins) return;
  GA_GLOBAL float* __restrict d = gptr(j.dst) + (size_t)k * j.dst_stride;
  if (j.src) {
    const GA_GLOBAL float* __restrict sp = gptr(j.src) + (size_t)k * j.src_stride;
    for (int i = lane; i < j.n; i += 64) d[i] = sp[i];
  } else {
    for (int i = lane; i < j.n; i += 64) d[i] = 0.f;
  }
}
void launch_hist_copy_b(hipStream_t s, const HistJobB* jobs_dev, int njobs, int max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  for (int j0 = 0; j0 < njobs; j0 += 32768) {
    int nj = std::min(32768, njobs - j0);
    hipLaunchKernelGGL(hist_copy_b_kernel, dim3((kBins + 3) / 4, nj), dim3(256), 0, s, jobs_dev + j0);
  }
}

// =====================================================================================================
//  Formulation C: FFT convolution along the block axis (see ga_kernels.hpp).  One workgroup of 256 threads owns one
//  N2-point complex sequence in LDS; Stockham autosort passes of radix 8 / 4 (natural order in and out), float32.
// =====================================================================================================
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {   // explicit fma: 4 instructions, one rounding less per component
  return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)

__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {   // forward, natural order
  float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_mi(csub(a1, a3));
  a0 = cadd(t0, t2);
  a2 = csub(t0, t2);
  a1 = cadd(t1, t3);
  a3 = csub(t1, t3);
}
__device__ __forceinline__ void dft8(float2 (&v)[8]) {   // forward 8-point DFT, natural order in / out
  const float h = 0.70710678118654752440f;
  float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  float2 o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  // odd part times W8^q
  float2 w1 = make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));      // o1 * (1 - i)/sqrt2
  float2 w2 = cmul_mi(o2);                                            // o2 * (-i)
  float2 w3 = make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));     // o3 * (-1 - i)/sqrt2
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, w1); v[5] = csub(e1, w1);
  v[2] = cadd(e2, w2); v[6] = csub(e2, w2);
  v[3] = cadd(e3, w3); v[7] = csub(e3, w3);
}

// LDS index padding: one extra float2 every 16 keeps the strided writes of the early passes off a single bank pair
__device__ __forceinline__ int PADI(int i) { return i + (i >> 4); }
#define TC_PADDED(n) ((n) + ((n) >> 4))

// one Stockham pass of radix R over the N-point sequence in `buf` (in place: read, barrier, write, barrier)
template <int N, int R>
__device__ __forceinline__ void stockham_pass(float2* __restrict buf, const float2* __restrict tw, int Ns, int tid) {
  constexpr int NB = N / R;                 // butterflies
  constexpr int PER = (NB + 255) / 256;     // per thread
  float2 v[PER][R];
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int j = tid + u * 256;
    if (NB % 256 == 0 || j < NB) {
      const int kk = j % Ns;
#pragma unroll
      for (int m = 0; m < R; m++) {
        float2 x = buf[PADI(j + m * NB)];
        if (m > 0) x = cmul(x, tw[(kk * m) * (N / (Ns * R))]);
        v[u][m] = x;
      }
      if constexpr (R == 8) {
        dft8(v[u]);
      } else {
        dft4(v[u][0], v[u][1], v[u][2], v[u][3]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int j = tid + u * 256;
    if (NB % 256 == 0 || j < NB) {
      const int kk = j % Ns;
      const int j0 = (j / Ns) * Ns * R + kk;
#pragma unroll
      for (int m = 0; m < R; m++) buf[PADI(j0 + m * Ns)] = v[u][m];
    }
  }
  __syncthreads();
}
template <int N>
__device__ __forceinline__ void fft_lds(float2* buf, const float2* tw, int tid) {   // forward N-point FFT in place
  if (N == 1024) {
    stockham_pass<N, 8>(buf, tw, 1, tid);
    stockham_pass<N, 8>(buf, tw, 8, tid);
    stockham_pass<N, 4>(buf, tw, 64, tid);
    stockham_pass<N, 4>(buf, tw, 256, tid);
  } else if (N == 2048) {
    stockham_pass<N, 8>(buf, tw, 1, tid);
    stockham_pass<N, 8>(buf, tw, 8, tid);
    stockham_pass<N, 8>(buf, tw, 64, tid);
    stockham_pass<N, 4>(buf, tw, 512, tid);
  } else {
    stockham_pass<N, 8>(buf, tw, 1, tid);
    stockham_pass<N, 8>(buf, tw, 8, tid);
    stockham_pass<N, 8>(buf, tw, 64, tid);
    stockham_pass<N, 8>(buf, tw, 512, tid);
  }
}

template <int N2>
__global__ __launch_bounds__(256) void tap_spectra_kernel(float2* __restrict hs, const float* __restrict hr, const float* __restrict hi,
                                                          int P, const float2* __restrict tw) {
  __shared__ float2 buf[TC_PADDED(N2)];
  const int k = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
  const size_t src = ((size_t)c * kBins + k) * P;
  for (int i = tid; i < N2; i += 256) buf[PADI(i)] = i < P ? make_float2(hr[src + i], hi[src + i]) : make_float2(0.f, 0.f);
  __syncthreads();
  fft_lds<N2>(buf, tw, tid);
  float2* dst = hs + ((size_t)c * kBins + k) * N2;
  // stored with the 1/N2 of the inverse transform folded in (a power of two: exact)
  const float sc = 1.0f / N2;
  for (int i = tid; i < N2; i += 256) dst[i] = make_float2(buf[PADI(i)].x * sc, buf[PADI(i)].y * sc);
}
void launch_tap_spectra(hipStream_t s, float2* hs, const float* hr, const float* hi, int nch, int P, int N2, const float2* tw) {
  dim3 g(kBins, nch), b(256);
  if (N2 == 1024) hipLaunchKernelGGL(tap_spectra_kernel<1024>, g, b, 0, s, hs, hr, hi, P, tw);
  else if (N2 == 2048) hipLaunchKernelGGL(tap_spectra_kernel<2048>, g, b, 0, s, hs, hr, hi, P, tw);
  else if (N2 == 4096) hipLaunchKernelGGL(tap_spectra_kernel<4096>, g, b, 0, s, hs, hr, hi, P, tw);
  else launch_fail("no taps-spectrum kernel for this block-axis FFT length");
}

// ---- register/LDS hybrid FFT for the tconv kernel -------------------------------------------------------------------
// Every thread owns the PT = N/256 points {tid + 256 m}.  With the radix plans below these are exactly the inputs of the
// FIRST Stockham pass (butterfly j = tid + 256u reads j + (N/R1) m) and the outputs of the LAST one (j + (N/RL) m), so
// a forward FFT, a pointwise product and an inverse FFT chain through registers; only the middle passes go through LDS
// (ping-pong buffers, one barrier per pass).   N = 1024: radices 4,8,8,4   N = 2048: 8,8,8,4   N = 4096: 8,8,8,8
template <int R>
__device__ __forceinline__ void dftR(float2 (&v)[R]) {
  if constexpr (R == 8) dft8(v);
  else dft4(v[0], v[1], v[2], v[3]);
}
template <int N, int R>
__device__ __forceinline__ void first_pass(const float2 (&own)[N / 256], float2* __restrict dst, int tid) {
  constexpr int NBT = (N / R) / 256;   // butterflies per thread
#pragma unroll
  for (int u = 0; u < NBT; u++) {
    const int j = tid + 256 * u;
    float2 v[R];
#pragma unroll
    for (int m = 0; m < R; m++) v[m] = own[u + NBT * m];
    dftR<R>(v);
#pragma unroll
    for (int m = 0; m < R; m++) dst[PADI(j * R + m)] = v[m];   // Ns = 1: j0 = j * R
  }
}
template <int N, int R, int Ns>
__device__ __forceinline__ void mid_pass(const float2* __restrict src, float2* __restrict dst, const float2* __restrict tw, int tid) {
  constexpr int NB = N / R;
  constexpr int PER = (NB + 255) / 256;
#pragma unroll
  for (int u = 0; u < PER; u++) {
    const int j = tid + 256 * u;
    if (NB % 256 == 0 || j < NB) {
      const int kk = j % Ns;
      float2 v[R];
#pragma unroll
      for (int m = 0; m < R; m++) {
        float2 x = src[PADI(j + m * NB)];
        if (m > 0) x = cmul(x, tw[PADI((kk * m) * (N / (Ns * R)))]);
        v[m] = x;
      }
      dftR<R>(v);
      const int j0 = (j / Ns) * Ns * R + kk;
#pragma unroll
      for (int m = 0; m < R; m++) dst[PADI(j0 + m * Ns)] = v[m];
    }
  }
}
template <int N, int R>
__device__ __forceinline__ void last_pass(const float2* __restrict src, float2 (&own)[N / 256], const float2* __restrict tw, int tid) {
  constexpr int NB = N / R, Ns = N / R;   // last pass: Ns * R == N
  constexpr int NBT = NB / 256;
#pragma unroll
  for (int u = 0; u < NBT; u++) {
    const int j = tid + 256 * u;           // j < Ns: kk = j, j0 = j
    float2 v[R];
#pragma unroll
    for (int m = 0; m < R; m++) {
      float2 x = src[PADI(j + m * NB)];
      if (m > 0) x = cmul(x, tw[PADI(j * m)]);
      v[m] = x;
    }
    dftR<R>(v);
#pragma unroll
    for (int m = 0; m < R; m++) own[u + (Ns / 256) * m] = v[m];   // position j + m * Ns = tid + 256 (u + (Ns/256) m)
  }
}
// forward FFT of the thread-owned points; `a` is written first.  One barrier per pass.
template <int N>
__device__ __forceinline__ void fft_own(float2 (&own)[N / 256], float2* a, float2* b, const float2* tw, int tid) {
  if constexpr (N == 1024) {
    first_pass<N, 4>(own, a, tid);
    __syncthreads();
    mid_pass<N, 8, 4>(a, b, tw, tid);
    __syncthreads();
    mid_pass<N, 8, 32>(b, a, tw, tid);
    __syncthreads();
    last_pass<N, 4>(a, own, tw, tid);
  } else if constexpr (N == 2048) {
    first_pass<N, 8>(own, a, tid);
    __syncthreads();
    mid_pass<N, 8, 8>(a, b, tw, tid);
    __syncthreads();
    mid_pass<N, 8, 64>(b, a, tw, tid);
    __syncthreads();
    last_pass<N, 4>(a, own, tw, tid);
  } else {
    first_pass<N, 8>(own, a, tid);
    __syncthreads();
    mid_pass<N, 8, 8>(a, b, tw, tid);
    __syncthreads();
    mid_pass<N, 8, 64>(b, a, tw, tid);
    __syncthreads();
    last_pass<N, 8>(a, own, tw, tid);
  }
}

// Persistent workgroups: grid = (TC_GROUPS, bins).  A workgroup owns one bin and walks over its share of the
// (set, overlap-save segment) items: window of N2 block-spectra of the x-row -> FFT -> per column: * taps spectrum, inverse
// FFT (conjugation trick), store the N2 - P + 1 valid blocks of the y-row.  The twiddle table is brought into LDS once
// per workgroup, the bin's taps spectra stay in L2 (every concurrently running workgroup of the bin reads the same ones),
// and the NEXT item's window is prefetched into registers while the current item is transformed.
constexpr int TC_GROUPS = 17;   // 17 x 129 workgroups = 2.9 rounds of the 768 resident slots (3 per CU)
template <int N2>
__global__ __launch_bounds__(256) void tconv_kernel(const ConvSetC* __restrict sets, int nsets, int nseg, int nblocks, int hist,
                                                    ConvPlanesB pl, const float2* __restrict twg) {
  constexpr int PT = N2 / 256;
  extern __shared__ float2 lds[];
  float2* bufA = lds;
  float2* bufB = lds + TC_PADDED(N2);
  float2* tw = lds + 2 * TC_PADDED(N2);      // twiddle table exp(-2 pi i j / N2) in LDS
  const int k = blockIdx.y, tid = threadIdx.x;
  const int total = nsets * nseg;
  for (int i = tid; i < N2; i += 256) tw[PADI(i)] = twg[i];

  auto load_window = [&](int item, float2 (&w)[PT]) {
    const ConvSetC* __restrict S = &sets[item / nseg];
    const int seg = item % nseg;
    const int P = S->P;
    const int b0 = seg * (N2 - (P - 1)) - (P - 1);     // block of window element 0
    const float* __restrict xr = pl.xr + ((size_t)S->x * kBins + k) * pl.tx;
    const float* __restrict xi = pl.xi + ((size_t)S->x * kBins + k) * pl.tx;
#pragma unroll
    for (int m = 0; m < PT; m++) {
      const int blk = b0 + tid + 256 * m;
      w[m] = make_float2(0.f, 0.f);
      if (blk >= -hist && blk < nblocks) w[m] = make_float2(xr[hist + blk], xi[hist + blk]);
    }
  };

  int item = blockIdx.x;
  __syncthreads();                             // twiddles visible
  int flip = 0;
  while (item < total) {
    const ConvSetC* __restrict S = &sets[item / nseg];
    const int seg = item % nseg;
    const int P = S->P, ncol = S->ncol;
    const int L = N2 - (P - 1);
    const int t0 = seg * L;                    // first output block of this segment
    float2 xf[PT];
    load_window(item, xf);
    const int next = item + gridDim.x;
    if (t0 < nblocks) {
      // buffers alternate between consecutive transforms so that a fast thread never overwrites what a slower one still reads
      if (flip) fft_own<N2>(xf, bufB, bufA, tw, tid); else fft_own<N2>(xf, bufA, bufB, tw, tid);
      flip ^= 1;
      const float scale = 1.0f;   // 1/N2 is folded into the taps spectra (tap_spectra_kernel)
      const int nvalid = min(L, nblocks - t0);
      for (int j = 0; j < ncol; j++) {
        const float2* __restrict hs = S->hs[j] + (size_t)k * N2;
        float2 y[PT];
#pragma unroll
        for (int m = 0; m < PT; m++) {
          float2 p = cmul(xf[m], hs[tid + 256 * m]);
          y[m] = make_float2(p.x, -p.y);       // conj: ifft(v) = conj(fft(conj(v))) / N
        }
        if (flip) fft_own<N2>(y, bufB, bufA, tw, tid); else fft_own<N2>(y, bufA, bufB, tw, tid);
        flip ^= 1;
        float* __restrict yr = pl.yr + ((size_t)(S->y0 + j) * kBins + k) * pl.ty + t0;
        float* __restrict yi = pl.yi + ((size_t)(S->y0 + j) * kBins + k) * pl.ty + t0;
#pragma unroll
        for (int m = 0; m < PT; m++) {
          const int i = tid + 256 * m - (P - 1);   // output block index within the segment
          if (i >= 0 && i < nvalid) {
            yr[i] = y[m].x * scale;
            yi[i] = -y[m].y * scale;
          }
        }
      }
    }
    item = next;
  }
}
void launch_tconv(hipStream_t s, const ConvSetC* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl, int N2, const float2* tw,
                  int nseg) {
  if (nsets <= 0 || nblocks <= 0) return;
  size_t lds = ((size_t)3 * TC_PADDED(N2)) * sizeof(float2);
  // N2 = 4096 needs 104 KB of dynamic LDS (gfx950 has 160 KB per CU); per device, so set on every launch of this legacy path
  if (N2 == 4096) (void)hipFuncSetAttribute((const void*)tconv_kernel<4096>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
  // every set of one launch has the same P, hence the same segment length L = N2 - P + 1
  const long long total = (long long)nsets * nseg;
  const int gx = (int)std::min<long long>(total, TC_GROUPS);
  dim3 grid(gx, kBins), block(256);
  if (N2 == 1024) hipLaunchKernelGGL(tconv_kernel<1024>, grid, block, lds, s, sets_dev, nsets, nseg, nblocks, hist, pl, tw);
  else if (N2 == 2048) hipLaunchKernelGGL(tconv_kernel<2048>, grid, block, lds, s, sets_dev, nsets, nseg, nblocks, hist, pl, tw);
  else if (N2 == 4096) hipLaunchKernelGGL(tconv_kernel<4096>, grid, block, lds, s, sets_dev, nsets, nseg, nblocks, hist, pl, tw);
  else launch_fail("no block-axis FFT kernel for this length");
}

// (packed-f32 complex arithmetic and the radix-16 register / LDS transform live in ga_fft16.hpp)
// Persistent workgroups (one per resident slot).  The G transforms of a workgroup serve G different (set, segment) items.
template <int N2>
__global__ __launch_bounds__(256, 3) void tconv16_kernel(const ConvSetC* __restrict sets, int nsets, int nseg, int nblocks, int hist,
                                                      ConvPlanesB pl, const float2* __restrict twg, int tbase) {
  using PL = R16Plan<N2>;
  constexpr int T = PL::T, G = PL::G;
  extern __shared__ f2 lds16[];
  f2* lds = lds16;
  f2* tw2 = lds;
  f2* tw3 = lds + PL::T2;
  const int tid = threadIdx.x;
  // T >= 64: a wavefront belongs to ONE transform, so everything indexed by g is wave-uniform -- say so (scalar loads of
  // the set record, SGPR base addresses for the global accesses)
  const int g = __builtin_amdgcn_readfirstlane(tid / T), t = tid % T;
  f2* buf = lds + PL::T2 + PL::T3 + g * TC16_PADDED(N2);
  for (int i = tid; i < PL::T2 + PL::T3; i += 256) lds[i] = f2{twg[i].x, twg[i].y};
  const int total = nsets * nseg;
  __syncthreads();
  // work unit = (bin, step of G items); the units are dealt to the resident workgroups in contiguous, equal ranges (bin-major:
  // a workgroup stays on one bin, whose taps spectra stay in its L2)
  const int steps = (total + G - 1) / G;
  const long long units = (long long)kBins * steps;
  const long long per = (units + gridDim.x - 1) / gridDim.x;
  const long long w0 = (long long)blockIdx.x * per, w1 = min(units, w0 + per);
  const unsigned ut = (unsigned)t;
  // window of unit w -> registers.  Called for unit w + 1 as soon as the last product of unit w has consumed the spectrum,
  // so the HBM latency of the next window hides behind the last inverse transform.
  auto load_window = [&](long long w, f2 (&xw)[16]) {
    const int k = (int)(w / steps);
    const int item = (int)(w % steps) * G + g;
    const bool live = item < total;
    const ConvSetC* __restrict S = &sets[live ? item / nseg : 0];
    const int seg = live ? item % nseg : 0;
    const int P = S->P;
    const int t0 = tbase + seg * (N2 - (P - 1));
    // window element e = t + T m is block t0 - (P - 1) + e; plane index hist + block >= 0 because hist >= P - 1
    const int first = hist + t0 - (P - 1);
    const int lim = (live && t0 < nblocks) ? nblocks - (t0 - (P - 1)) : 0;   // elements at or beyond `lim` are zero padding
    const float* __restrict xr = pl.xr + ((size_t)S->x * kBins + k) * pl.tx + first;
    const float* __restrict xi = pl.xi + ((size_t)S->x * kBins + k) * pl.tx + first;
#pragma unroll
    for (int m = 0; m < 16; m++) {
      xw[m] = f2{0.f, 0.f};
      if ((int)(ut + T * m) < lim) xw[m] = f2{xr[ut + T * m], xi[ut + T * m]};
    }
  };
  f2 xf[16];
  if (w0 < w1) load_window(w0, xf);
  for (long long w = w0; w < w1; w++) {
    const int k = (int)(w / steps);
    const int base = (int)(w % steps) * G;
    const int item = base + g;
    const bool live = item < total;
    const ConvSetC* __restrict S = &sets[live ? item / nseg : 0];
    const int seg = live ? item % nseg : 0;
    const int P = S->P;
    const int ncol = live ? S->ncol : 0;
    const int L = N2 - (P - 1);
    const int t0 = tbase + seg * L;
    const bool work = live && t0 < nblocks;
    int maxcol = 0;   // the same for every thread of the workgroup: barriers inside fft16_own
#pragma unroll
    for (int gg = 0; gg < G; gg++) {
      const int it = base + gg;
      if (it < total && tbase + (it % nseg) * L < nblocks) maxcol = max(maxcol, sets[it / nseg].ncol);
    }
    if (maxcol == 0) {
      if (w + 1 < w1) load_window(w + 1, xf);
      continue;
    }
    fft16_own<N2>(xf, buf, tw2, tw3, t);
    __syncthreads();   // the last-pass reads are done before the next transform stores into the buffer
    const int nvalid = min(L, nblocks - t0);
    for (int j = 0; j < maxcol; j++) {
      const bool act = work && j < ncol;
      f2 y[16];
      if (act) {
        const GA_GLOBAL f2* __restrict hs = reinterpret_cast<const GA_GLOBAL f2*>(gptr(S->hs[j]) + (size_t)k * N2);
#pragma unroll
        for (int m = 0; m < 16; m++) y[m] = cmulp_swap(xf[m], hs[ut + T * m]);   // ifft(v) = swap(fft(swap(v))) / N, 1/N inside hs
      } else {
#pragma unroll
        for (int m = 0; m < 16; m++) y[m] = f2{0.f, 0.f};
      }
      if (j == maxcol - 1 && w + 1 < w1) load_window(w + 1, xf);
      fft16_own<N2>(y, buf, tw2, tw3, t);
      if (act) {
        // element e = t + T m of the result is output block t0 + e - (P - 1); valid for P - 1 <= e < P - 1 + nvalid
        float* __restrict yr = pl.yr + ((size_t)(S->y0 + j) * kBins + k) * pl.ty + t0 - (P - 1);
        float* __restrict yi = pl.yi + ((size_t)(S->y0 + j) * kBins + k) * pl.ty + t0 - (P - 1);
        const int e0 = P - 1, e1 = P - 1 + nvalid;
#pragma unroll
        for (int m = 0; m < 16; m++) {
          const int e = (int)(ut + T * m);
          if (e >= e0 && e < e1) {
            yr[ut + T * m] = y[m].y;   // swapped back
            yi[ut + T * m] = y[m].x;
          }
        }
      }
      __syncthreads();
    }
  }
}
template <int N2>
static void launch_tconv16_n(hipStream_t s, const ConvSetC* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl, const float2* tw16,
                             int nseg, int tbase) {
  using PL = R16Plan<N2>;
  const size_t lds = (size_t)(PL::T2 + PL::T3 + PL::G * TC16_PADDED(N2)) * sizeof(float2);
  // resident workgroups per device (a process may drive several devices: the attribute and the occupancy are per device)
  static int groupsOf[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) launch_fail("device ordinal out of range");
  if (!groupsOf[dev]) {
    if (hipFuncSetAttribute((const void*)tconv16_kernel<N2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096) != hipSuccess)
      launch_fail("cannot raise the dynamic LDS limit of the block-axis FFT kernel");
    int occ = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, tconv16_kernel<N2>, 256, lds) != hipSuccess || occ < 1)
      launch_fail("block-axis FFT kernel does not fit a compute unit");
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    groupsOf[dev] = occ * cus;   // exactly one resident round
  }
  const int groups = groupsOf[dev];
  const long long steps = ((long long)nsets * nseg + PL::G - 1) / PL::G;
  dim3 grid((unsigned)std::min<long long>(steps * kBins, groups)), block(256);
  hipLaunchKernelGGL(tconv16_kernel<N2>, grid, block, lds, s, sets_dev, nsets, nseg, nblocks, hist, pl, tw16, tbase);
}
void launch_tconv16(hipStream_t s, const ConvSetC* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl, int N2, const float2* tw16,
                    int nseg, int tbase) {
  if (nsets <= 0 || nblocks <= 0 || nseg <= 0) return;
  if (N2 == 1024) launch_tconv16_n<1024>(s, sets_dev, nsets, nblocks, hist, pl, tw16, nseg, tbase);
  else if (N2 == 2048) launch_tconv16_n<2048>(s, sets_dev, nsets, nblocks, hist, pl, tw16, nseg, tbase);
  else if (N2 == 4096) launch_tconv16_n<4096>(s, sets_dev, nsets, nblocks, hist, pl, tw16, nseg, tbase);
  else launch_fail("no block-axis FFT kernel for this length");
}

// ---- plane utilities ----------------------------------------------------------------------------------
__global__ void plane_copy_kernel(float* __restrict dst, int dst_t, int dst_t0, const float* __restrict src, int src_t, int src_t0,
                                  int n, int rp) {
  const int k = blockIdx.y;
  const size_t per = (size_t)n * rp / 4;
  float4* d = reinterpret_cast<float4*>(dst + ((size_t)k * dst_t + dst_t0) * rp);
  const float4* s = src ? reinterpret_cast<const float4*>(src + ((size_t)k * src_t + src_t0) * rp) : nullptr;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x)
    d[i] = s ? s[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_plane_copy(hipStream_t s, float* dst, int dst_t, int dst_t0, const float* src, int src_t, int src_t0, int n, int rp) {
  if (n <= 0) return;
  size_t per = (size_t)n * rp / 4;
  int gx = (int)((per + 255) / 256);
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL(plane_copy_kernel, dim3(gx, kBins), dim3(256), 0, s, dst, dst_t, dst_t0, src, src_t, src_t0, n, rp);
}

__global__ void extract_ir_kernel(float* __restrict hr, float* __restrict hi, const float* __restrict xr, const float* __restrict xi,
                                  int tx, int rp, int P, int nch) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over nch * kBins * P
  int total = nch * kBins * P;
  if (idx >= total) return;
  int p = idx % P, k = (idx / P) % kBins, c = idx / (P * kBins);
  size_t o = ((size_t)k * tx + p) * rp + c;
  hr[idx] = xr[o];
  hi[idx] = xi[o];
}
void launch_extract_ir(hipStream_t s, float* hr, float* hi, const float* xr, const float* xi, int tx, int rp, int P, int nch) {
  int total = nch * kBins * P;
  if (total <= 0) return;
  hipLaunchKernelGGL(extract_ir_kernel, dim3((total + 255) / 256), dim3(256), 0, s, hr, hi, xr, xi, tx, rp, P, nch);
}

__global__ void pair_sum_kernel(float* __restrict out, const float* __restrict a, const float* __restrict b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = a[i] + b[i];
}
void launch_pair_sum(hipStream_t s, float* out, const float* a, const float* b, int64_t n) {
  if (n <= 0) return;
  int g = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(pair_sum_kernel, dim3(g), dim3(256), 0, s, out, a, b, n);
}

// =====================================================================================================
//  Mix (AudioNodeInput.Pull / MixBuffer, AudioNodeInput.cs:100-244): sequential float32 sum in connection order,
//  starting from the cleared buffer (0 + t0 + t1 + ...), one thread per 1 or 4 frames.
// =====================================================================================================
template <int VEC, bool SCALED>
__global__ __launch_bounds__(256) void mix_kernel(const MixJob* __restrict jobs, const float* const* __restrict terms, const float* __restrict gains,
                                                  const float* const* __restrict curves) {
  const MixJob job = jobs[blockIdx.y];
  const int64_t nv = (job.n + VEC - 1) / VEC;
  const float* const* __restrict tp = terms + job.term0;
  const float* __restrict gp = SCALED ? gains + job.term0 : nullptr;   // SCALED: term j contributes fl(x * gain[j]) -- a folded constant GainNode
  // ... or fl(x * curve_j[f]) where the folded GainNode's gain follows a timeline (curves[j] != null): GainNode.cs:52-57
  const float* const* __restrict cp = (SCALED && curves) ? curves + job.term0 : nullptr;
  if (SCALED && job.nterms == 1) {   // a gain on its own: out = in * g, as gain_kernel writes it (no 0 + in front)
    const float g = gp[0];
    const float* __restrict cv = cp ? cp[0] : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
      const int64_t f = job.f0 + i * VEC;
      if (VEC == 4) {
        const v4f v = ldg4(tp[0] + f);
        const v4f gv = cv ? ldg4(cv + f) : v4f{g, g, g, g};
        const v4f r = v4f{v.x * gv.x, v.y * gv.y, v.z * gv.z, v.w * gv.w};
        stg4(job.out + f, r);
        if (job.out2) stg4(job.out2 + f, r);
      } else {
        const float r = ldg1(tp[0] + f) * (cv ? ldg1(cv + f) : g);
        gptr(job.out)[f] = r;
        if (job.out2) gptr(job.out2)[f] = r;
      }
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i * VEC;
    if (VEC == 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int j = 0;
      for (; j + 4 <= job.nterms; j += 4) {   // four loads in flight, adds strictly in term order
        v4f v0 = ldg4(tp[j] + f);
        v4f v1 = ldg4(tp[j + 1] + f);
        v4f v2 = ldg4(tp[j + 2] + f);
        v4f v3 = ldg4(tp[j + 3] + f);
        if (SCALED) {   // (products rounded on their own: -ffp-contract=off, no fma with the add below)
          const float g0 = gp[j], g1 = gp[j + 1], g2 = gp[j + 2], g3 = gp[j + 3];
          v4f m0 = v4f{g0, g0, g0, g0}, m1 = v4f{g1, g1, g1, g1}, m2 = v4f{g2, g2, g2, g2}, m3 = v4f{g3, g3, g3, g3};
          if (cp) {
            if (cp[j]) m0 = ldg4(cp[j] + f);
            if (cp[j + 1]) m1 = ldg4(cp[j + 1] + f);
            if (cp[j + 2]) m2 = ldg4(cp[j + 2] + f);
            if (cp[j + 3]) m3 = ldg4(cp[j + 3] + f);
          }
          v0 = v4f{v0.x * m0.x, v0.y * m0.y, v0.z * m0.z, v0.w * m0.w};
          v1 = v4f{v1.x * m1.x, v1.y * m1.y, v1.z * m1.z, v1.w * m1.w};
          v2 = v4f{v2.x * m2.x, v2.y * m2.y, v2.z * m2.z, v2.w * m2.w};
          v3 = v4f{v3.x * m3.x, v3.y * m3.y, v3.z * m3.z, v3.w * m3.w};
        }
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
        acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
        acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
      }
      for (; j < job.nterms; j++) {
        v4f v = ldg4(tp[j] + f);
        if (SCALED) {
          const float g = gp[j];
          const v4f m = (cp && cp[j]) ? ldg4(cp[j] + f) : v4f{g, g, g, g};
          v = v4f{v.x * m.x, v.y * m.y, v.z * m.z, v.w * m.w};
        }
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      stg4(job.out + f, v4f{acc.x, acc.y, acc.z, acc.w});
      if (job.out2) stg4(job.out2 + f, v4f{acc.x, acc.y, acc.z, acc.w});
    } else {
      float acc = 0.f;
      for (int j = 0; j < job.nterms; j++) {
        float v = ldg1(tp[j] + f);
        if (SCALED) v = v * ((cp && cp[j]) ? ldg1(cp[j] + f) : gp[j]);
        acc += v;
      }
      gptr(job.out)[f] = acc;
      if (job.out2) gptr(job.out2)[f] = acc;
    }
  }
}
void launch_mix(hipStream_t s, const MixJob* jobs_dev, int njobs, const float* const* terms_dev, int64_t max_n, bool vec4, const float* gains_dev,
                const float* const* curves_dev) {
  if (njobs <= 0 || max_n <= 0) return;
  int64_t nv = vec4 ? (max_n + 3) / 4 : max_n;
  int gx = (int)std::min<int64_t>((nv + 255) / 256, 2048);
  if (gains_dev) {
    if (vec4)
      GA_LAUNCH_JOBS((mix_kernel<4, true>), gx, 256, jobs_dev, njobs, terms_dev, gains_dev, curves_dev);
    else
      GA_LAUNCH_JOBS((mix_kernel<1, true>), gx, 256, jobs_dev, njobs, terms_dev, gains_dev, curves_dev);
  } else if (vec4) {
    GA_LAUNCH_JOBS((mix_kernel<4, false>), gx, 256, jobs_dev, njobs, terms_dev, gains_dev, curves_dev);
  } else {
    GA_LAUNCH_JOBS((mix_kernel<1, false>), gx, 256, jobs_dev, njobs, terms_dev, gains_dev, curves_dev);
  }
}

// ---- a bus of MANY terms (hundreds to thousands of voices into one input): few jobs, each a long strictly ordered sum ----------
// With one job per channel the launch above is a few hundred waves whose every step waits for a scalar load (the term's pointer,
// gain, curve) and then for a vector load: at 4 terms per step a 4096-term bus is 1024 such round trips per wave (measured: 0.95 ms
// for 2 x 4096 terms x 120,000 frames, 0.26 of the HBM rate).  Here a wave takes 64 frames, one per lane, reads the descriptors of
// 64 terms with ONE coalesced load each (pointer / gain / curve: lane l holds term t0 + l, handed out with v_readlane) and keeps
// the loads of 32 terms in flight; the additions stay in term order (AudioNodeInput.cs:118-132).
template <bool SCALED>
__global__ __launch_bounds__(64) void mix_wide_kernel(const MixJob* __restrict jobs, const float* const* __restrict terms, const float* __restrict gains,
                                                     const float* const* __restrict curves) {
  const MixJob job = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  const float* const* __restrict tp = terms + job.term0;
  const float* __restrict gp = SCALED ? gains + job.term0 : nullptr;
  const float* const* __restrict cp = (SCALED && curves) ? curves + job.term0 : nullptr;
  constexpr int NB = 32;
  for (int64_t i0 = (int64_t)blockIdx.x * 64; i0 < job.n; i0 += (int64_t)gridDim.x * 64) {
    const int64_t i = i0 + lane;
    const bool live = i < job.n;
    const int64_t f = job.f0 + (live ? i : 0);
    float acc = 0.f;
    for (int t0 = 0; t0 < job.nterms; t0 += 64) {
      const int cnt = min(64, job.nterms - t0);
      const bool mine = lane < cnt;
      const float* myp = mine ? tp[t0 + lane] : nullptr;
      const float myg = (SCALED && mine) ? gp[t0 + lane] : 1.f;
      const float* myc = (cp && mine) ? cp[t0 + lane] : nullptr;
      auto ptr_of = [](const float* p, int l) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
        return (const float*)(((unsigned long long)hi << 32) | lo);
      };
#pragma unroll
      for (int u0 = 0; u0 < 64; u0 += NB) {
        if (u0 >= cnt) break;
        if (u0 + NB <= cnt) {
          float v[NB], m[NB];
#pragma unroll
          for (int u = 0; u < NB; u++) v[u] = ldg1(ptr_of(myp, u0 + u) + f);
          if (SCALED) {
#pragma unroll
            for (int u = 0; u < NB; u++) {
              m[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myg), u0 + u));
              if (cp) {
                const float* c = ptr_of(myc, u0 + u);
                if (c) m[u] = ldg1(c + f);
              }
            }
#pragma unroll
            for (int u = 0; u < NB; u++) v[u] = v[u] * m[u];   // (rounded on its own: -ffp-contract=off)
          }
#pragma unroll
          for (int u = 0; u < NB; u++) acc += v[u];
        } else {
          for (int u = u0; u < cnt; u++) {   // the last, partial batch of the bus
            float v = ldg1(tp[t0 + u] + f);
            if (SCALED) v = v * ((cp && cp[t0 + u]) ? ldg1(cp[t0 + u] + f) : gp[t0 + u]);
            acc += v;
          }
        }
      }
    }
    if (live) {
      gptr(job.out)[f] = acc;
      if (job.out2) gptr(job.out2)[f] = acc;
    }
  }
}
void launch_mix_wide(hipStream_t s, const MixJob* jobs_dev, int njobs, const float* const* terms_dev, int64_t max_n, const float* gains_dev,
                     const float* const* curves_dev) {
  if (njobs <= 0 || max_n <= 0) return;
  const int gx = (int)std::min<int64_t>((max_n + 63) / 64, 8192);
  if (gains_dev)
    GA_LAUNCH_JOBS((mix_wide_kernel<true>), gx, 64, jobs_dev, njobs, terms_dev, gains_dev, curves_dev);
  else
    GA_LAUNCH_JOBS((mix_wide_kernel<false>), gx, 64, jobs_dev, njobs, terms_dev, gains_dev, curves_dev);
}

template <bool SCALED>
__global__ __launch_bounds__(256) void downmix_kernel(const DownmixJob* __restrict jobs, const float* const* __restrict terms, const float* __restrict gains) {
  const DownmixJob job = jobs[blockIdx.y];
  const float* const* __restrict tp = terms + job.term0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i;
    float sum = 0.f;
    for (int ch = 0; ch < job.nch; ch++) {   // AudioNodeInput.cs:221-226
      float v = ldg1(tp[ch] + f);
      if (SCALED) v = v * gains[job.term0 + ch];
      sum += v;
    }
    gptr(job.out)[f] = sum * job.scale;
  }
}
void launch_downmix(hipStream_t s, const DownmixJob* jobs_dev, int njobs, const float* const* terms_dev, int64_t max_n, const float* gains_dev) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 2048);
  if (gains_dev) GA_LAUNCH_JOBS(downmix_kernel<true>, gx, 256, jobs_dev, njobs, terms_dev, gains_dev);
  else GA_LAUNCH_JOBS(downmix_kernel<false>, gx, 256, jobs_dev, njobs, terms_dev, gains_dev);
}

// ---- GainNode (GainNode.cs:48-58) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void gain_kernel(const GainJob* __restrict jobs) {
  const GainJob job = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i;
    float g = job.curve ? gptr(job.curve)[f] : job.gain;
    if (job.mod) {   // Math.Clamp(intrinsicValue + modulation, min, max), AudioParam.cs:129
      g = g + gptr(job.mod)[f];
      g = g < job.vmin ? job.vmin : (g > job.vmax ? job.vmax : g);
    }
    gptr(job.out)[f] = gptr(job.in)[f] * g;
  }
}
void launch_gain(hipStream_t s, const GainJob* jobs_dev, int njobs, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 1024);
  GA_LAUNCH_JOBS(gain_kernel, gx, 256, jobs_dev, njobs);
}

// =====================================================================================================
//  BiQuadFilterNode with constant coefficients (BiQuadFilterNode.cs:136-141): one lane per (node, channel),
//  64 jobs per wavefront.  The wave stages a [64 jobs][64 frames] tile through LDS so every HBM access is a
//  coalesced 256-byte row even though each lane walks its own slab serially.
// =====================================================================================================
constexpr int BQ_TILE = 64;
__device__ __forceinline__ const float* bcast_ptr(const float* p, int srcLane) {
  unsigned long long v = (unsigned long long)p;
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, srcLane), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), srcLane);
  return (const float*)(((unsigned long long)hi << 32) | lo);
}
// JPW = cascades (jobs) per wavefront.  The recurrence is latency/issue bound, not lane bound: with few jobs it is far
// better to spread them thinly (4..16 lanes busy per wave, every SIMD of the chip working) than to fill 64 lanes of a
// handful of waves.  All 64 lanes still cooperate on the coalesced 256-byte row loads / stores of the JPW x 64 tile.
template <int NSEC, int JPW>
__global__ __launch_bounds__(64) void biquad_kernel(const BiquadJob* __restrict jobs, int njobs, const BiquadSection* __restrict secs) {
  // tile length: long tiles amortise the load / barrier / store overhead of a tile when few chains share a wavefront
  constexpr int TL = JPW <= 8 ? 256 : BQ_TILE, TM = TL / 64;
  __shared__ float tile[JPW][TL + 1];
  __shared__ float prevw[JPW][2];   // NSEC == 1: W1, W2 at the start of the tile (the FIR half needs w[-1], w[-2])
  const int lane = threadIdx.x;
  const int j0 = blockIdx.x * JPW;
  const int myj = j0 + lane;
  const bool have = lane < JPW && myj < njobs;
  // every compute lane keeps ITS job (cascade coefficients and states) in registers; row pointers are broadcast with v_readlane
  BiquadJob me{};
  if (have) me = jobs[myj];
  const float* inb = have ? me.in + me.f0 : nullptr;
  float* outb = (have && me.out) ? me.out + me.f0 : nullptr;
  float b0[NSEC], b1[NSEC], b2[NSEC], a1[NSEC], a2[NSEC], w1[NSEC], w2[NSEC];
  float* st[NSEC];
#pragma unroll
  for (int q = 0; q < NSEC; q++) {
    b0[q] = b1[q] = b2[q] = a1[q] = a2[q] = w1[q] = w2[q] = 0.f;
    st[q] = nullptr;
    if (have) {
      const BiquadSection sc = secs[me.sec0 + q];
      b0[q] = sc.b0; b1[q] = sc.b1; b2[q] = sc.b2; a1[q] = sc.a1; a2[q] = sc.a2;
      st[q] = me.state ? me.state + 2 * q : sc.state;
      w1[q] = ldg1(st[q]);
      w2[q] = ldg1(st[q] + 1);
    }
  }
  const int64_t n = have ? me.n : 0;
  int64_t nmax = n;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, (int64_t)__shfl_xor((long long)nmax, m, 64));
  const int jcount = min(JPW, njobs - j0);

  float pre[JPW][TM];   // prefetched tile: pre[r][u] = frame (base + 64 u + lane) of job j0 + r
  auto fetch = [&](int64_t base) {
#pragma unroll
    for (int r = 0; r < JPW; r++) {
      const float* p = bcast_ptr(inb, r);
      int64_t nr = __builtin_amdgcn_readlane((int)n, r);   // n < 2^31 frames per segment
#pragma unroll
      for (int u = 0; u < TM; u++) {
        int64_t fi = base + 64 * u + lane;
        pre[r][u] = (r < jcount && fi < nr) ? ldg1(p + fi) : 0.f;
      }
    }
  };
  fetch(0);
  for (int64_t base = 0; base < nmax; base += TL) {
#pragma unroll
    for (int r = 0; r < JPW; r++)
#pragma unroll
      for (int u = 0; u < TM; u++) tile[r][64 * u + lane] = pre[r][u];
    __syncthreads();
    if (base + TL < nmax) fetch(base + TL);   // next tile's loads fly during the serial recurrence below
    const int cnt = (int)max<int64_t>(0, min<int64_t>(TL, n - base));
    if constexpr (NSEC == 1) {
      // A single section splits into a RECURSIVE half  w[n] = x[n] - a1 w[n-1] - a2 w[n-2]  (serial, one lane per chain) and a
      // FIR half  y[n] = b0 w[n] + b1 w[n-1] + b2 w[n-2]  (no recurrence: all 64 lanes, one sample each).  The same products
      // and sums in the same order as BiQuadFilterNode.cs:137-138 -> bit-exact, with 4 instead of 9 flops on the serial chain.
      if (lane < JPW) {
        prevw[lane][0] = w1[0];
        prevw[lane][1] = w2[0];
        float ww1 = w1[0], ww2 = w2[0];
        const float ca1 = a1[0], ca2 = a2[0];
        // the chain  a1 w1 -> (x - .) -> (. - a2 w2)  is three dependent VALU results per sample (~20 cycles each on gfx950): keep
        // everything else -- the select of a partial tile included -- off it
        if (cnt == TL) {
#pragma unroll
          for (int i0 = 0; i0 < TL; i0 += 16) {
            float xv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) xv[i] = tile[lane][i0 + i];
#pragma unroll
            for (int i = 0; i < 16; i++) {
              const float w = xv[i] - ca1 * ww1 - ca2 * ww2;
              ww2 = ww1;
              ww1 = w;
              xv[i] = w;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) tile[lane][i0 + i] = xv[i];
          }
        } else {
          for (int i = 0; i < cnt; i++) {
            const float w = tile[lane][i] - ca1 * ww1 - ca2 * ww2;
            ww2 = ww1;
            ww1 = w;
            tile[lane][i] = w;
          }
        }
        w1[0] = ww1;
        w2[0] = ww2;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < JPW; r++) {
        if (r < jcount) {
          const float cb0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b0[0]), r));
          const float cb1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b1[0]), r));
          const float cb2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b2[0]), r));
          float yv[TM];
#pragma unroll
          for (int u = 0; u < TM; u++) {
            const int x = 64 * u + lane;
            const float w0 = tile[r][x];
            const float wm1 = x >= 1 ? tile[r][x - 1] : prevw[r][0];
            const float wm2 = x >= 2 ? tile[r][x - 2] : prevw[r][1 - x];   // x = 1: w[-1] ; x = 0: w[-2]
            yv[u] = cb0 * w0 + cb1 * wm1 + cb2 * wm2;
          }
          // (one wavefront: its LDS reads above are served before the writes below)
#pragma unroll
          for (int u = 0; u < TM; u++) tile[r][64 * u + lane] = yv[u];
        }
      }
    } else if (lane < JPW) {
      if (cnt == TL) {
        // register batches of 16 keep LDS latency off the W1/W2 dependency chains
#pragma unroll
        for (int i0 = 0; i0 < TL; i0 += 16) {
          float xv[16];
#pragma unroll
          for (int i = 0; i < 16; i++) xv[i] = tile[lane][i0 + i];
#pragma unroll
          for (int i = 0; i < 16; i++) {
            float x = xv[i];
#pragma unroll
            for (int q = 0; q < NSEC; q++) {
              float w = x - a1[q] * w1[q] - a2[q] * w2[q];            // BiQuadFilterNode.cs:137
              float y = b0[q] * w + b1[q] * w1[q] + b2[q] * w2[q];    // :138
              w2[q] = w1[q];
              w1[q] = w;
              x = y;
            }
            xv[i] = x;
          }
#pragma unroll
          for (int i = 0; i < 16; i++) tile[lane][i0 + i] = xv[i];
        }
      } else {
        for (int i = 0; i < cnt; i++) {   // the job's last, partial tile
          float x = tile[lane][i];
#pragma unroll
          for (int q = 0; q < NSEC; q++) {
            float w = x - a1[q] * w1[q] - a2[q] * w2[q];
            float y = b0[q] * w + b1[q] * w1[q] + b2[q] * w2[q];
            w2[q] = w1[q];
            w1[q] = w;
            x = y;
          }
          tile[lane][i] = x;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < JPW; r++) {
      if (r < jcount) {
        float* q = (float*)bcast_ptr(outb, r);
        if (!q) continue;   // (a state-only job: pass A of a cascade split along time)
        int64_t nr = __builtin_amdgcn_readlane((int)n, r);
#pragma unroll
        for (int u = 0; u < TM; u++) {
          int64_t fi = base + 64 * u + lane;
          if (fi < nr) stg1(q + fi, tile[r][64 * u + lane]);
        }
      }
    }
    __syncthreads();
  }
  if (have) {
#pragma unroll
    for (int q = 0; q < NSEC; q++)
      for (int t = 0; t < me.twins; t++) {   // (twin channels: BiquadJob::twins)
        stg1(st[q] + 2 * t, w1[q]);
        stg1(st[q] + 2 * t + 1, w2[q]);
      }
  }
}
template <int JPW>
static void launch_biquad_jpw(hipStream_t s, const BiquadJob* jobs_dev, int njobs, const BiquadSection* secs_dev, int nsec) {
  dim3 g((njobs + JPW - 1) / JPW), b(64);
  switch (nsec) {
    case 1: hipLaunchKernelGGL((biquad_kernel<1, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    case 2: hipLaunchKernelGGL((biquad_kernel<2, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    case 3: hipLaunchKernelGGL((biquad_kernel<3, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    case 4: hipLaunchKernelGGL((biquad_kernel<4, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    case 5: hipLaunchKernelGGL((biquad_kernel<5, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    case 6: hipLaunchKernelGGL((biquad_kernel<6, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    case 7: hipLaunchKernelGGL((biquad_kernel<7, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
    default: hipLaunchKernelGGL((biquad_kernel<8, JPW>), g, b, 0, s, jobs_dev, njobs, secs_dev); break;
  }
}
// ---- cascades of NSEC >= 2 sections: the sections of one cascade sit on NEIGHBOURING LANES -----------------------------------
// Lane q of a group runs section q on sample k - SK q at step k and takes its input from lane q - 1 with one DPP row shift: a
// software pipeline across lanes.  Per step the wave issues ONE section's arithmetic instead of NSEC sections back to back on
// a single lane; every section still sees exactly the reference's sample-by-sample float arithmetic
// (BiQuadFilterNode.cs:137-138: w = (x - a1 w1) - a2 w2 ; y = (b0 w + b1 w1) + b2 w2), so the result stays bit-exact.
// 16 / NSEC cascades per 16-lane row.
//
// A wave that is alone on its SIMD pays ~5 cycles per instruction, whatever the instruction (vector, scalar, s_nop, LDS; measured with
// tools/micro/bq_pipe_probe.hip: 221 instructions per 16 steps = 1131 cycles), and ~4 more when it needs the result of the
// instruction right before it: the walk costs its instruction COUNT.  Its steady state is therefore ONE inline-assembly
// statement with fixed registers (ga_biquad_pipe_asm.inc, written by tools/gen_biquad_pipe_asm.py, where the schedule is
// described): 8 vector instructions per step --
//   the four products of w[k-1] as two packed multiplies ({a1, b1} w and {a2, b2} w: P1, R1 of this step, P2, R2 of the next);
//   the recursion  t = x - P1 ; w = t - P2 ;
//   the output half (m = b0 w ; s = m + R1 ; y = s + R2) one and two steps behind, in the gaps of the recursion;
//   the hand-over from the lane to the left and the select of the cascade's input for section 0 as one v_cndmask_b32_dpp
// -- plus one instruction per step for the LDS traffic and the loop; no instruction follows its producer.  SK = 4 steps between
// neighbouring sections keep the hand-over off the dependency chain.  The pipeline is filled once per JOB and drained once
// (masked batches, plain C++), not per tile: step k hands out sample k - D, D = SK (NSEC - 1), the tile is stored D samples late.
#ifdef GA_BQ_PROBE   // tools/micro/bq_pipe_probe.hip: where a wave's cycles go (s_memtime around the phases of every tile)
__device__ unsigned long long ga_bq_probe[8];
}   // namespace ga
// (a library built with -DGA_BQ_PROBE hands the sums out and clears them: tools/bq_probe_product.py)
extern "C" __attribute__((visibility("default"))) void ga_bq_probe_read(unsigned long long* out) {
  unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ga::ga_bq_probe), sizeof zero);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(ga::ga_bq_probe), zero, sizeof zero);
}
namespace ga {
#define GA_BQ_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define GA_BQ_ACC(i, v) probe_acc[i] += (v)
#else
#define GA_BQ_T(v)
#define GA_BQ_ACC(i, v)
#endif
template <int NSEC>
__global__ __launch_bounds__(64) void biquad_pipe_kernel(const BiquadJob* __restrict jobs, int njobs, const BiquadSection* __restrict secs,
                                                         int jpw) {
  constexpr int GPR = 16 / NSEC;        // cascades (groups of NSEC lanes) per row
  constexpr int MAXJ = 4 * GPR < 16 ? 4 * GPR : 16;   // (16: the tile's rows live in registers twice, prefetched and outgoing)
  constexpr int PT = 256, TM = PT / 64;   // steps per tile
  constexpr int SK = 4;                   // steps between a section and the next
  constexpr int D = SK * (NSEC - 1);      // step k hands out sample k - D
  constexpr int LD = PT + 4;              // 16-byte aligned rows: a batch of 16 steps reads / writes its samples as four b128
  __shared__ __attribute__((aligned(16))) float tile[MAXJ][LD];
  __shared__ __attribute__((aligned(16))) float dump[64][16];   // where the lanes that are not a last section "write their outputs"
  const int lane = threadIdx.x;
  const int row = lane >> 4, lr = lane & 15;
  const int grp = lr / NSEC, q = lr % NSEC;
  const int slot = row * GPR + grp;                       // cascade slot of this lane inside the wave
  const int j0 = blockIdx.x * jpw;
  const int myj = j0 + slot;
  const bool have = grp < GPR && slot < jpw && myj < njobs;
  BiquadJob me{};
  if (have) me = jobs[myj];
  const float* inb = have ? me.in + me.f0 : nullptr;
  float* outb = have ? me.out + me.f0 : nullptr;
  float b0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;   // w3: the w before w2 (for the assembly walk to pick up mid-stream; not state)
  v2f ab1 = {0.f, 0.f}, ab2 = {0.f, 0.f};   // {a1, b1}, {a2, b2}
  float* st = nullptr;
  if (have) {
    const BiquadSection sc = secs[me.sec0 + q];
    b0 = sc.b0;
    ab1 = v2f{sc.a1, sc.b1};
    ab2 = v2f{sc.a2, sc.b2};
    st = me.state ? me.state + 2 * q : sc.state;
    w1 = ldg1(st);
    w2 = ldg1(st + 1);
  }
  const int n = have ? (int)me.n : 0;
  int nmax = n, nmin = have ? n : 0x7fffffff;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    nmax = max(nmax, __shfl_xor(nmax, m, 64));
    nmin = min(nmin, __shfl_xor(nmin, m, 64));
  }
  nmax = __builtin_amdgcn_readfirstlane(nmax);
  nmin = __builtin_amdgcn_readfirstlane(nmin);
  const int jcount = min(jpw, njobs - j0);
  auto lane_of = [](int r) { return (r / GPR) * 16 + (r % GPR) * NSEC; };   // lane of section 0 of slot r

  // A tile row in flight (prefetched / outgoing) is four registers per lane, in one of two layouts:
  //   whole tiles   lane l holds frames 4 l .. 4 l + 3 of the row: ONE 16-byte global access and one b128 LDS access per row
  //   ragged tiles  register u holds frame 64 u + l, every access bounds-checked (a job's first and last tiles)
  v4f pre[MAXJ];
  bool pre_whole = false;
  auto fetch = [&](int base) {
    pre_whole = base + PT <= nmin;
    if (pre_whole) {
#pragma unroll
      for (int r = 0; r < MAXJ; r++)
        if (r < jcount) pre[r] = ldg4(bcast_ptr(inb, lane_of(r)) + base + 4 * lane);
      return;
    }
#pragma unroll
    for (int r = 0; r < MAXJ; r++) {
      pre[r] = v4f{0.f, 0.f, 0.f, 0.f};
      if (r < jcount) {
        const float* p = bcast_ptr(inb, lane_of(r));
        const int nr = __builtin_amdgcn_readlane(n, lane_of(r));
#pragma unroll
        for (int u = 0; u < TM; u++) {
          const int fi = base + 64 * u + lane;
          if (fi < nr) pre[r][u] = ldg1(p + fi);
        }
      }
    }
  };
  const int srow = have ? slot : 0;
  const bool q0 = q == 0;
  const bool qlast = have && q == NSEC - 1;
  const int jofs = SK * q;
  float yh[SK];   // this lane's outputs of the last SK steps, yh[0] the latest
#pragma unroll
  for (int i = 0; i < SK; i++) yh[i] = 0.f;
  auto dpp_left = [](float v) {   // lane l takes lane l - 1's value (row_shr:1); a row's lane 0 is always a section 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, false));
  };
  // nb batches of 16 steps in which every section of every cascade of the wave has a sample (tile positions p0 ...)
  const unsigned long long q0mask = __builtin_amdgcn_ballot_w64(q0);
  auto steady_run = [&](int p0, int nb) {
    typedef __attribute__((address_space(3))) float* lds_ptr;   // (a 32-bit LDS offset: what ds_read / ds_write address)
    const unsigned ain = (unsigned)(size_t)(lds_ptr)&tile[srow][p0];
    const unsigned aout = qlast ? ain : (unsigned)(size_t)(lds_ptr)&dump[lane][0];
    const unsigned ainc = qlast ? 64u : 0u;
    asm volatile(
#include "ga_biquad_pipe_asm.inc"
        : [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [y0] "+v"(yh[0]), [y1] "+v"(yh[1]), [y2] "+v"(yh[2]), [y3] "+v"(yh[3]), [nb] "+s"(nb)
        : [ab1] "v"(ab1), [ab2] "v"(ab2), [b0] "v"(b0), [ain] "v"(ain), [aout] "v"(aout), [ainc] "v"(ainc), [q0] "s"(q0mask)
        :
#include "ga_biquad_pipe_asm_clobbers.inc"
    );
  };
  // 16 steps of the fill / the drain of the pipeline, or past the end of a shorter cascade: lanes without a sample hold still
  auto masked = [&](int k0, int p0) {
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
      const int j = k0 + i - jofs;
      const bool active = have && j >= 0 && j < n;
      const float up = dpp_left(yh[SK - 1]);
      const float x = q0 ? tile[srow][p0 + i] : up;
      const float w = x - ab1.x * w1 - ab2.x * w2;           // BiQuadFilterNode.cs:137
      const float yy = b0 * w + ab1.y * w1 + ab2.y * w2;     // :138
      w3 = active ? w2 : w3;
      w2 = active ? w1 : w2;
      w1 = active ? w : w1;
#pragma unroll
      for (int h = SK - 1; h > 0; h--) yh[h] = yh[h - 1];
      yh[0] = yy;
      if (qlast && active) tile[srow][p0 + i] = yy;
    }
  };
  const int nsteps = nmax + D;
  // (the coefficient and state loads complete HERE: left pending, their first use inside the loop would wait for vmcnt(0), that is
  // for the next tile's prefetch as well, once per tile -- the whole memory latency on the recurrence's clock)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  // Order of the memory operations of a tile: the samples of tile i + 1 are requested before tile i is walked and waited for
  // after it; the results of tile i leave AFTER that wait -- issued before it, every tile would wait for its stores to be
  // acknowledged (vmcnt counts loads and stores in one queue), a few microseconds on the recurrence's clock.
  v4f outv[MAXJ];
  bool out_whole = false;
  auto store_tile = [&](int base) {   // position p of the tile holds sample base + p - D
    if (out_whole) {
#pragma unroll
      for (int r = 0; r < MAXJ; r++)
        if (r < jcount) stg4((float*)bcast_ptr(outb, lane_of(r)) + (base - D) + 4 * lane, outv[r]);
      return;
    }
#pragma unroll
    for (int r = 0; r < MAXJ; r++) {
      if (r < jcount) {
        float* o = (float*)bcast_ptr(outb, lane_of(r));
        const int nr = __builtin_amdgcn_readlane(n, lane_of(r));
#pragma unroll
        for (int u = 0; u < TM; u++) {
          const int fi = base + 64 * u + lane - D;
          if (fi >= 0 && fi < nr) stg1(o + fi, outv[r][u]);
        }
      }
    }
  };
#ifdef GA_BQ_PROBE
  unsigned long long probe_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  GA_BQ_T(tk0);
  fetch(0);
  for (int base = 0; base < nsteps; base += PT) {
    GA_BQ_T(ta);
    // (one explicit wait for the prefetch on every path: the compiler cannot see that a row without a cascade was never requested
    // and would otherwise wait for "its" load -- that is for the stores below -- when fetch() next overwrites the registers)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    if (pre_whole) {
#pragma unroll
      for (int r = 0; r < MAXJ; r++)
        if (r < jcount) *(v4f*)&tile[r][4 * lane] = pre[r];
    } else {
#pragma unroll
      for (int r = 0; r < MAXJ; r++)
        if (r < jcount) {
#pragma unroll
          for (int u = 0; u < TM; u++) tile[r][64 * u + lane] = pre[r][u];
        }
    }
    if (base > 0) store_tile(base - PT);
    __syncthreads();
    if (base + PT < nmax) fetch(base + PT);   // next tile's loads fly during the recurrence below
    GA_BQ_T(tb);
    // the tile's batches: [0, ps) masked (fill) | [ps, pe) steady, one assembly run | [pe, end) masked (drain, shorter cascades)
    const int pend = min(PT, (nsteps - base + 15) & ~15);
    const int ps = min(pend, max(0, (D - base + 15) & ~15));
    const int pe = max(ps, min(pend, (nmin - base) & ~15));
#pragma unroll 1
    for (int p0 = 0; p0 < ps; p0 += 16) { masked(base + p0, p0); GA_BQ_ACC(5, 1); }
    if (pe > ps) { steady_run(ps, (pe - ps) >> 4); GA_BQ_ACC(4, (pe - ps) >> 4); }
#pragma unroll 1
    for (int p0 = pe; p0 < pend; p0 += 16) { masked(base + p0, p0); GA_BQ_ACC(5, 1); }
    GA_BQ_T(tc);
    __syncthreads();
    out_whole = base >= D && base + PT - D <= nmin;
    if (out_whole) {
#pragma unroll
      for (int r = 0; r < MAXJ; r++)
        if (r < jcount) outv[r] = *(const v4f*)&tile[r][4 * lane];
    } else {
#pragma unroll
      for (int r = 0; r < MAXJ; r++)
        if (r < jcount) {
#pragma unroll
          for (int u = 0; u < TM; u++) outv[r][u] = tile[r][64 * u + lane];
        }
    }
    __syncthreads();
    GA_BQ_T(td);
    GA_BQ_ACC(0, tb - ta);
    GA_BQ_ACC(1, tc - tb);
    GA_BQ_ACC(2, td - tc);
  }
  store_tile((nsteps - 1) / PT * PT);
#ifdef GA_BQ_PROBE
  {
    GA_BQ_T(tk1);
    probe_acc[3] = tk1 - tk0;
    probe_acc[6] = 1;   // waves
    probe_acc[7] = (unsigned long long)jcount;
    if (lane == 0)
      for (int i = 0; i < 8; i++) atomicAdd(&ga_bq_probe[i], probe_acc[i]);
  }
#endif
  if (have) {
    for (int t = 0; t < me.twins; t++) {   // (twin channels: BiquadJob::twins)
      stg1(st + 2 * t, w1);
      stg1(st + 2 * t + 1, w2);
    }
  }
}
template <int NSEC>
static void launch_biquad_pipe(hipStream_t s, const BiquadJob* jobs_dev, int njobs, const BiquadSection* secs_dev) {
  constexpr int MAXJ = 4 * (16 / NSEC) < 16 ? 4 * (16 / NSEC) : 16;
  // The walk is bound by one wave's instruction issue, not by lanes, and a cascade more in a wave costs ~200 cycles of staging per
  // tile of 256 steps (12,200): up to 512 waves (one per two SIMDs) one cascade each, then fuller waves -- measured on MI355X
  // (tools/micro/bq_pipe_probe.hip, 4096 cascades of 5 sections x 120,000 frames): 512 waves of 8: 3.1-3.2 ms, 342 of 12: 3.1-3.3.
  int jpw = std::min(MAXJ, std::max(1, (njobs + 511) / 512));
  if (const char* e = expenv("GA_BQ_JPW")) jpw = std::min(MAXJ, std::max(1, atoi(e)));
  hipLaunchKernelGGL(biquad_pipe_kernel<NSEC>, dim3((njobs + jpw - 1) / jpw), dim3(64), 0, s, jobs_dev, njobs, secs_dev, jpw);
}
// ---- ONE section per job, one job per lane, nothing staged --------------------------------------------------------------------
// The pieces of cascades that are split along time (BiquadScanJob: tens of thousands of short walks) and any other single
// section.  The lane-per-cascade kernel above stages [jobs][frames] tiles through LDS so that every global access is a coalesced
// row -- ~80 instructions per row and tile of 64 frames, 85 per step of a 64-lane wave, and a wave pays ~5 cycles per instruction
// (config 2: 0.41 ms per pass over 0.49 GB).  Here a lane reads ITS piece 16 bytes at a time (four consecutive loads of a lane
// cover one 64-byte line, which the vector cache keeps), walks the four samples in registers -- the two packed products of w[n-1]
// ({a1, b1} w, {a2, b2} w) serve the recursion and the output half alike, 7 instructions per sample, 3 when only the state is
// wanted -- and writes 16 bytes.  The arithmetic is BiQuadFilterNode.cs:137-138 operation by operation (as biquad_pipe_kernel).
template <bool STATE_ONLY>
__global__ __launch_bounds__(64) void biquad1_kernel(const BiquadJob* __restrict jobs, int njobs, const BiquadSection* __restrict secs) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= njobs) return;
  const BiquadJob me = jobs[j];
  const BiquadSection sc = secs[me.sec0];
  float* st = me.state ? me.state : sc.state;
  const float b0 = sc.b0;
  const v2f ab1 = {sc.a1, sc.b1}, ab2 = {sc.a2, sc.b2};
  float w1 = ldg1(st), w2 = ldg1(st + 1);
  const float* __restrict in = me.in + me.f0;
  float* __restrict out = (!STATE_ONLY && me.out) ? me.out + me.f0 : nullptr;
  const int n = (int)me.n;
  v2f Bp = {ab2.x * w2, ab2.y * w2};   // {a2, b2} w[n-2]
  auto step = [&](float x) {
    const v2f A = {ab1.x * w1, ab1.y * w1};   // {a1, b1} w[n-1]
    const float t = x - A.x;
    const float w = t - Bp.x;                 // w = (x - a1 w1) - a2 w2
    float y = 0.f;
    if (!STATE_ONLY) {
      const float m = b0 * w;
      const float sum = m + A.y;
      y = sum + Bp.y;                         // y = (b0 w + b1 w1) + b2 w2
    }
    Bp = v2f{ab2.x * w1, ab2.y * w1};
    w2 = w1;
    w1 = w;
    return y;
  };
  // 32 frames = 128 bytes = one cache line of the lane's stream per round: the eight loads of a round go out back to back (the line is
  // used up while it is resident -- with 16 bytes per round the 64 lines of every wave of a CU evict each other between two uses and each
  // is fetched from L2 several times), the next round's are in flight during this round's 32 steps
  int i = 0;
  if (n >= 32) {
    v4f cur[8], nxt[8];
#pragma unroll
    for (int u = 0; u < 8; u++) cur[u] = ldg4(in + 4 * u);
    for (; i + 32 <= n; i += 32) {
      const bool more = i + 64 <= n;
#pragma unroll
      for (int u = 0; u < 8; u++) nxt[u] = more ? ldg4(in + i + 32 + 4 * u) : cur[u];
      v4f y[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        y[u].x = step(cur[u].x);
        y[u].y = step(cur[u].y);
        y[u].z = step(cur[u].z);
        y[u].w = step(cur[u].w);
      }
      if (!STATE_ONLY && out) {
#pragma unroll
        for (int u = 0; u < 8; u++) stg4(out + i + 4 * u, y[u]);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) cur[u] = nxt[u];
    }
  }
  for (; i + 4 <= n; i += 4) {
    const v4f c = ldg4(in + i);
    const float y0 = step(c.x), y1 = step(c.y), y2 = step(c.z), y3 = step(c.w);
    if (!STATE_ONLY && out) stg4(out + i, v4f{y0, y1, y2, y3});
  }
  for (; i < n; i++) {
    const float y = step(ldg1(in + i));
    if (!STATE_ONLY && out) stg1(out + i, y);
  }
  for (int t = 0; t < me.twins; t++) {   // (twin channels: BiquadJob::twins)
    stg1(st + 2 * t, w1);
    stg1(st + 2 * t + 1, w2);
  }
}
void launch_biquad_lanes(hipStream_t s, const BiquadJob* jobs_dev, int njobs, const BiquadSection* secs_dev, int nsec, bool state_only) {
  if (njobs <= 0) return;
  static const bool staged1 = expenv("GA_BQ_STAGED1") != nullptr;   // (measurement: the staged kernel for single sections too)
  if (nsec == 1 && !staged1) {
    if (state_only)
      hipLaunchKernelGGL(biquad1_kernel<true>, dim3((njobs + 63) / 64), dim3(64), 0, s, jobs_dev, njobs, secs_dev);
    else
      hipLaunchKernelGGL(biquad1_kernel<false>, dim3((njobs + 63) / 64), dim3(64), 0, s, jobs_dev, njobs, secs_dev);
    return;
  }
  if (njobs <= 16 * 1024) launch_biquad_jpw<32>(s, jobs_dev, njobs, secs_dev, nsec);   // (more, emptier waves while the chip has room)
  else launch_biquad_jpw<64>(s, jobs_dev, njobs, secs_dev, nsec);
}
// the pieces of every cascade as jobs of the lane-per-cascade kernel: thread = (cascade, piece)
__global__ __launch_bounds__(256) void biquad_split_expand_kernel(const BiquadScanJob* __restrict casc, int ncasc, int G, int64_t K,
                                                                 BiquadJob* __restrict passA, BiquadJob* __restrict passB) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= ncasc * G) return;
  const int j = v / G, l = v % G;
  const BiquadScanJob C = casc[j];
  BiquadJob b;
  b.in = C.in;
  b.out = C.out;
  b.sec0 = C.sec0;
  b.nsec = C.nsec;
  b.f0 = C.f0 + (int64_t)l * K;
  b.n = min(K, C.n - (int64_t)l * K);
  b.state = l + 1 < G ? C.scratch + (size_t)l * C.nsec * 2 : nullptr;   // the last piece runs on the cascade's own state
  b.twins = l + 1 < G ? 1 : C.twins;
  b.pad_ = 0;
  passB[(size_t)j * G + l] = b;
  if (l + 1 < G) {
    b.out = nullptr;
    b.twins = 1;
    passA[(size_t)j * (G - 1) + l] = b;
  }
}
void launch_biquad_split_expand(hipStream_t s, const BiquadScanJob* casc_dev, int ncasc, int G, int64_t K, BiquadJob* passA, BiquadJob* passB) {
  if (ncasc <= 0) return;
  hipLaunchKernelGGL(biquad_split_expand_kernel, dim3((ncasc * G + 255) / 256), dim3(256), 0, s, casc_dev, ncasc, G, K, passA, passB);
}
// s_{l+1} = A^K s_l + z_l (float64 accumulation); one thread per cascade
__global__ __launch_bounds__(64) void biquad_scan_kernel(const BiquadScanJob* __restrict jobs, int njobs, int G, const BiquadSection* __restrict secs,
                                                        const uint8_t* __restrict tables) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= njobs) return;
  const BiquadScanJob J = jobs[j];
  const int D = 2 * J.nsec;
  const float* Mg = (const float*)(tables + J.m_off);
  double cur[2 * kMaxBiquadSections], nxt[2 * kMaxBiquadSections];
  for (int q = 0; q < J.nsec; q++) {
    const float* st = secs[J.sec0 + q].state;
    cur[2 * q] = ldg1(st);
    cur[2 * q + 1] = ldg1(st + 1);
  }
  for (int l = 0; l + 1 < G; l++) {
    float* sc = J.scratch + (size_t)l * D;
    for (int r = 0; r < D; r++) {
      double a = (double)ldg1(sc + r);   // z_l
      for (int c = 0; c < D; c++) a += (double)ldg1(Mg + r * D + c) * cur[c];
      nxt[r] = a;
    }
    for (int r = 0; r < D; r++) {
      stg1(sc + r, (float)cur[r]);       // s_l: where pass B's piece l starts
      cur[r] = (double)(float)nxt[r];    // (the state is float in the filter)
    }
  }
  for (int q = 0; q < J.nsec; q++) {     // s_{G-1}: the last piece starts from (and ends in) the cascade's own state
    float* st = secs[J.sec0 + q].state;
    stg1(st, (float)cur[2 * q]);
    stg1(st + 1, (float)cur[2 * q + 1]);
  }
}
// The same scan with the cascade length known at compile time: state, matrix and the pieces' z_l live in registers (with a run-time
// length they are indexed arrays in scratch memory, and every step waits for its own loads: 0.19 ms for config 2's 255 steps), and
// z_{l+1} is requested before s_l is written over z_l.  Same operations in the same order: the same bits.
template <int NS>
__global__ __launch_bounds__(64) void biquad_scan_fixed_kernel(const BiquadScanJob* __restrict jobs, int njobs, int G, const BiquadSection* __restrict secs,
                                                              const uint8_t* __restrict tables) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= njobs) return;
  const BiquadScanJob J = jobs[j];
  constexpr int D = 2 * NS;
  const float* Mg = (const float*)(tables + J.m_off);
  float M[D * D];
#pragma unroll
  for (int e = 0; e < D * D; e++) M[e] = ldg1(Mg + e);
  double cur[D];
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const float* st = secs[J.sec0 + q].state;
    cur[2 * q] = ldg1(st);
    cur[2 * q + 1] = ldg1(st + 1);
  }
  float zn[D];
  if (G > 1) {
#pragma unroll
    for (int r = 0; r < D; r++) zn[r] = ldg1(J.scratch + r);
  }
  for (int l = 0; l + 1 < G; l++) {
    float* sc = J.scratch + (size_t)l * D;
    float z[D];
#pragma unroll
    for (int r = 0; r < D; r++) z[r] = zn[r];
    if (l + 2 < G) {
#pragma unroll
      for (int r = 0; r < D; r++) zn[r] = ldg1(sc + D + r);
    }
    double nxt[D];
#pragma unroll
    for (int r = 0; r < D; r++) {
      double a = (double)z[r];   // z_l
#pragma unroll
      for (int c = 0; c < D; c++) a += (double)M[r * D + c] * cur[c];
      nxt[r] = a;
    }
#pragma unroll
    for (int r = 0; r < D; r++) {
      stg1(sc + r, (float)cur[r]);       // s_l: where pass B's piece l starts
      cur[r] = (double)(float)nxt[r];    // (the state is float in the filter)
    }
  }
#pragma unroll
  for (int q = 0; q < NS; q++) {
    float* st = secs[J.sec0 + q].state;
    stg1(st, (float)cur[2 * q]);
    stg1(st + 1, (float)cur[2 * q + 1]);
  }
}
// (the launch's jobs all have `nsec` sections: Exec::bqScans is kept by cascade length)
void launch_biquad_scan(hipStream_t s, const BiquadScanJob* jobs_dev, int njobs, int G, const BiquadSection* secs_dev, const uint8_t* tables, int nsec) {
  if (njobs <= 0) return;
  const dim3 g((njobs + 63) / 64), b(64);
  switch (nsec) {
    case 1: hipLaunchKernelGGL(biquad_scan_fixed_kernel<1>, g, b, 0, s, jobs_dev, njobs, G, secs_dev, tables); return;
    case 2: hipLaunchKernelGGL(biquad_scan_fixed_kernel<2>, g, b, 0, s, jobs_dev, njobs, G, secs_dev, tables); return;
    case 3: hipLaunchKernelGGL(biquad_scan_fixed_kernel<3>, g, b, 0, s, jobs_dev, njobs, G, secs_dev, tables); return;
    case 4: hipLaunchKernelGGL(biquad_scan_fixed_kernel<4>, g, b, 0, s, jobs_dev, njobs, G, secs_dev, tables); return;
    default: break;
  }
  hipLaunchKernelGGL(biquad_scan_kernel, g, b, 0, s, jobs_dev, njobs, G, secs_dev, tables);
}
void launch_biquad(hipStream_t s, const BiquadJob* jobs_dev, int njobs, const BiquadSection* secs_dev, int nsec) {
  if (njobs <= 0) return;
  static const bool pipe = !expenv("GA_BQ_NOPIPE");
  if (pipe && nsec >= 2) {
    switch (nsec) {
      case 2: launch_biquad_pipe<2>(s, jobs_dev, njobs, secs_dev); return;
      case 3: launch_biquad_pipe<3>(s, jobs_dev, njobs, secs_dev); return;
      case 4: launch_biquad_pipe<4>(s, jobs_dev, njobs, secs_dev); return;
      case 5: launch_biquad_pipe<5>(s, jobs_dev, njobs, secs_dev); return;
      case 6: launch_biquad_pipe<6>(s, jobs_dev, njobs, secs_dev); return;
      case 7: launch_biquad_pipe<7>(s, jobs_dev, njobs, secs_dev); return;
      default: launch_biquad_pipe<8>(s, jobs_dev, njobs, secs_dev); return;
    }
  }
  // measured on MI355X (config 4, 8,192 cascades of 5 sections): ~512 waves on the chip (one per two SIMDs) is the sweet
  // spot -- 4 / 8 / 16 / 32 jobs per wave took 81 / 65 / 47 / 54 ms.  A wave issues one VALU instruction per ~4 cycles
  // however many lanes are busy, so fewer, fuller waves only pay once that many waves exist.
  int per = (njobs + 511) / 512;
  if (const char* e = expenv("GA_BQ_JPW")) per = atoi(e);   // tuning override
  if (per <= 4) launch_biquad_jpw<4>(s, jobs_dev, njobs, secs_dev, nsec);
  else if (per <= 8) launch_biquad_jpw<8>(s, jobs_dev, njobs, secs_dev, nsec);
  else if (per <= 16) launch_biquad_jpw<16>(s, jobs_dev, njobs, secs_dev, nsec);
  else if (per <= 32) launch_biquad_jpw<32>(s, jobs_dev, njobs, secs_dev, nsec);
  else launch_biquad_jpw<64>(s, jobs_dev, njobs, secs_dev, nsec);
}

// ---- BiQuadFilterNode with automated parameters --------------------------------------------------------
__device__ void biquad_update_coefficients(int type, float frequency, float q, float gain, float sample_rate, float& b0, float& b1,
                                           float& b2, float& a1, float& a2) {   // BiQuadFilterNode.cs:149-258
  const float PI = 3.14159274f;
  float w0 = 2.f * PI * frequency / sample_rate;
  // The reference's MathF.Cos / MathF.Sin are the C library's, which evaluate in double and round once; the device library's
  // single-precision versions are 1-2 ulp off that, which a resonant section amplifies ~fs/(pi*f/Q) times.  Evaluating in
  // double here lands on the same float except when the true value sits within ~1e-9 of a rounding boundary.
  float cosW0 = (float)cos((double)w0);
  float sinW0 = (float)sin((double)w0);
  float alpha = sinW0 / (2.f * q);
  float a0, A1, A2, B0, B1, B2;
  switch (type) {
    case 0: B0 = (1.f - cosW0) / 2.f; B1 = 1.f - cosW0; B2 = (1.f - cosW0) / 2.f; a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha; break;
    case 1: B0 = (1.f + cosW0) / 2.f; B1 = -(1.f + cosW0); B2 = (1.f + cosW0) / 2.f; a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha; break;
    case 2: B0 = alpha; B1 = 0.f; B2 = -alpha; a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha; break;
    case 3: B0 = 1.f; B1 = -2.f * cosW0; B2 = 1.f; a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha; break;
    case 4: B0 = 1.f - alpha; B1 = -2.f * cosW0; B2 = 1.f + alpha; a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha; break;
    case 5: {
      float A = (float)pow(10.0, (double)(gain / 40.f));
      B0 = 1.f + alpha * A; B1 = -2.f * cosW0; B2 = 1.f - alpha * A; a0 = 1.f + alpha / A; A1 = -2.f * cosW0; A2 = 1.f - alpha / A;
      break;
    }
    case 6: {
      float A = (float)pow(10.0, (double)(gain / 40.f));
      float beta = sqrtf(A) / q;
      B0 = A * ((A + 1.f) - (A - 1.f) * cosW0 + beta * sinW0);
      B1 = 2.f * A * ((A - 1.f) - (A + 1.f) * cosW0);
      B2 = A * ((A + 1.f) - (A - 1.f) * cosW0 - beta * sinW0);
      a0 = (A + 1.f) + (A - 1.f) * cosW0 + beta * sinW0;
      A1 = -2.f * ((A - 1.f) + (A + 1.f) * cosW0);
      A2 = (A + 1.f) + (A - 1.f) * cosW0 - beta * sinW0;
      break;
    }
    case 7: {
      float A = (float)pow(10.0, (double)(gain / 40.f));
      float beta = sqrtf(A) / q;
      B0 = A * ((A + 1.f) + (A - 1.f) * cosW0 + beta * sinW0);
      B1 = -2.f * A * ((A - 1.f) + (A + 1.f) * cosW0);
      B2 = A * ((A + 1.f) + (A - 1.f) * cosW0 - beta * sinW0);
      a0 = (A + 1.f) - (A - 1.f) * cosW0 + beta * sinW0;
      A1 = 2.f * ((A - 1.f) - (A + 1.f) * cosW0);
      A2 = (A + 1.f) - (A - 1.f) * cosW0 - beta * sinW0;
      break;
    }
    default: B0 = 1.f; B1 = 0.f; B2 = 0.f; a0 = 1.f; A1 = 0.f; A2 = 0.f; break;
  }
  b0 = B0 / a0;
  b1 = B1 / a0;
  b2 = B2 / a0;
  a1 = A1 / a0;
  a2 = A2 / a0;
}
__global__ __launch_bounds__(64) void biquad_dynamic_kernel(const BiquadDynJob* __restrict jobs, int njobs) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= njobs) return;
  const BiquadDynJob* __restrict job = &jobs[j];
  GA_GLOBAL BiquadDynState* st = gptr(job->state);
  float b0 = st->b0, b1 = st->b1, b2 = st->b2, a1 = st->a1, a2 = st->a2;
  // (bit 8 of filter_type: the Type setter ran on the host while the state lived here -- _coefficientsDirty, BiQuadFilterNode.cs:24-36)
  bool dirty = st->dirty != 0 || (job->filter_type & 0x100) != 0;
  const int C = job->channels;
  const int type = job->filter_type & 0xFF;
  const float nyq = job->nyquist, sr = job->sample_rate;
  for (int64_t b = 0; b < job->nblocks; b++) {
    const int64_t f0 = (job->b0 + b) * kBlock;
    const float gainDb = job->gcurve ? ldg1(job->gcurve + f0) : job->gval;
    float lb0 = b0, lb1 = b1, lb2 = b2, la1 = a1, la2 = a2;   // lastB0 ... (:110)
    float usedFreq = 1000.f, usedQ = 1.0f;                     // _lastFrequency / _lastQ never change (:13-14,111-112)
    for (int ch = 0; ch < C; ch++) {
      const GA_GLOBAL float* in = gptr(job->in[ch]);
      GA_GLOBAL float* out = gptr(job->out[ch]);
      float w1 = st->w[2 * ch], w2 = st->w[2 * ch + 1];
      for (int i = 0; i < kBlock; i++) {
        float f = job->fcurve ? ldg1(job->fcurve + f0 + i) : job->fval;
        f = f < 1.f ? 1.f : (f > nyq ? nyq : f);
        float q = job->qcurve ? ldg1(job->qcurve + f0 + i) : job->qval;
        q = q > 0.001f ? q : 0.001f;
        if (dirty || fabsf(f - usedFreq) > 0.001f || fabsf(q - usedQ) > 0.0001f) {   // usedGain == gainDb always (:113,126)
          biquad_update_coefficients(type, f, q, gainDb, sr, b0, b1, b2, a1, a2);
          usedFreq = f;
          usedQ = q;
          dirty = false;
          lb0 = b0; lb1 = b1; lb2 = b2; la1 = a1; la2 = a2;
        }
        float x = in ? in[f0 + i] : 0.f;
        float w = x - la1 * w1 - la2 * w2;
        float y = lb0 * w + lb1 * w1 + lb2 * w2;
        w2 = w1;
        w1 = w;
        out[f0 + i] = y;
      }
      st->w[2 * ch] = w1;
      st->w[2 * ch + 1] = w2;
    }
  }
  st->b0 = b0; st->b1 = b1; st->b2 = b2; st->a1 = a1; st->a2 = a2;
  st->dirty = dirty ? 1 : 0;
}
void launch_biquad_dynamic(hipStream_t s, const BiquadDynJob* jobs_dev, int njobs) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(biquad_dynamic_kernel, dim3((njobs + 63) / 64), dim3(64), 0, s, jobs_dev, njobs);
}

// =====================================================================================================
//  AudioParam timeline (AudioParam.ComputeValueAtTime and helpers, AudioParam.cs:169-247), double precision,
//  no contraction: sampleTime = blockTime + i * deltaTime (:116-120); k-rate samples at block start (:146).
// =====================================================================================================
__global__ __launch_bounds__(128) void param_curve_kernel(const ParamJob* __restrict jobs, const ParamEvent* __restrict events,
                                                          const double* __restrict block_times, double delta_time) {
  const ParamJob job = jobs[blockIdx.y];
  const int i = threadIdx.x;
  for (int64_t b = blockIdx.x; b < job.nblocks; b += gridDim.x) {
    const int64_t blk = job.b0 + b;
    const double bt = block_times[blk];
    double st = job.arate ? bt + i * delta_time : bt;
    gptr(job.out)[blk * kBlock + i] = param_value_at(events + job.ev0, job.nev, job.value, st);
  }
}
void launch_param_curve(hipStream_t s, const ParamJob* jobs_dev, int njobs, const ParamEvent* events_dev,
                        const double* block_times_dev, double delta_time, int64_t max_blocks) {
  if (njobs <= 0 || max_blocks <= 0) return;
  int gx = (int)std::min<int64_t>(max_blocks, 4096);
  GA_LAUNCH_JOBS(param_curve_kernel, gx, 128, jobs_dev, njobs, events_dev, block_times_dev, delta_time);
}

// ---- looping rate-1 source (AudioBufferSourceNode.cs:186-235) --------------------------------------------
__global__ __launch_bounds__(256) void loop_source_kernel(const LoopJob* __restrict jobs) {
  const LoopJob job = jobs[blockIdx.y];
  const int64_t len = job.loop_end - job.loop_start;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t u = job.pos0 + i;
    if (u >= job.loop_end) u = job.loop_start + (len > 0 ? (u - job.loop_end) % len : 0);
    gptr(job.out)[job.f0 + i] = gptr(job.buf)[u];
  }
}
void launch_loop_source(hipStream_t s, const LoopJob* jobs_dev, int njobs, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 1024);
  GA_LAUNCH_JOBS(loop_source_kernel, gx, 256, jobs_dev, njobs);
}

// =====================================================================================================
//  CubicResampler (CubicResampler.cs:26-63).  The double-precision position recurrence is replayed on the host
//  once per distinct rate (the "trajectory"); every (job, block) pair is then independent: one lane walks the 128
//  outputs of its block with exactly the reference's per-sample arithmetic.  64 consecutive blocks of one job per
//  wavefront; outputs go through LDS so the HBM stores are coalesced.
// =====================================================================================================
__global__ __launch_bounds__(64) void resample_kernel(const ResampleJob* __restrict jobs, const ResampleBlock* __restrict traj) {
  __shared__ float tile[64][kBlock + 1];
  const ResampleJob job = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  const int64_t bl0 = (int64_t)blockIdx.x * 64;
  if (bl0 >= job.nblocks) return;
  const int64_t b = bl0 + lane;
  const bool have = b < job.nblocks;
  if (have) {
    const ResampleBlock rb = traj[job.traj0 + b];
    const GA_GLOBAL float* __restrict in = gptr(job.buf) + job.start_pos;
    int64_t ip = rb.consumed;           // next input index (relative)
    double Pos = rb.pos;
    int ready = rb.ready;
    float S0 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f;
    // the window holds the last `ready` consumed samples (Shift, CubicResampler.cs:91-97)
    if (ready >= 1) S3 = in[ip - 1];
    if (ready >= 2) S2 = in[ip - 2];
    if (ready >= 3) S1 = in[ip - 3];
    if (ready >= 4) S0 = in[ip - 4];
    while (ready < 4 && ip < job.avail) {   // priming (:31-35)
      S0 = S1; S1 = S2; S2 = S3; S3 = in[ip++];
      ready++;
    }
    int outp = 0;
    if (ready == 4) {
      for (; outp < rb.produced; outp++) {
        int consume = (int)Pos;
        for (int i = 0; i < consume; i++) { S0 = S1; S1 = S2; S2 = S3; S3 = in[ip++]; }
        Pos -= consume;
        float t = (float)Pos;
        tile[lane][outp] = S1 + t * (0.5f * (S2 - S0) + t * ((S0 - 2.5f * S1 + 2.f * S2 - 0.5f * S3) + t * (0.5f * (S3 - S0) + 1.5f * (S1 - S2))));
        Pos += job.rate;
      }
    }
    for (; outp < kBlock; outp++) tile[lane][outp] = 0.f;   // outputSpan.Slice(outIdx).Clear()
  }
  __syncthreads();
  const int nb = (int)min<int64_t>(64, job.nblocks - bl0);
  GA_GLOBAL float* __restrict out = gptr(job.out) + (job.b0 + bl0) * kBlock;
  for (int r = 0; r < nb; r++) {
    out[(int64_t)r * kBlock + lane] = tile[r][lane];
    out[(int64_t)r * kBlock + 64 + lane] = tile[r][64 + lane];
  }
}
__global__ __launch_bounds__(256) void resample_fast_kernel(const ResampleFastJob* __restrict jobs) {
  const ResampleFastJob job = jobs[blockIdx.y];
  const int64_t total = job.nblocks * kBlock;
  const GA_GLOBAL float* __restrict in = gptr(job.buf) + job.start_pos;
  const GA_GLOBAL v2f* __restrict sm = (const GA_GLOBAL v2f*)job.samples;   // (ip as the bits of .x, t = .y: one 8-byte load)
  GA_GLOBAL float* __restrict out = gptr(job.out) + job.b0 * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const v2f e = sm[i];
    const GA_GLOBAL float* __restrict w = in + (int64_t)__float_as_uint(e.x) - 4;
    const float S0 = w[0], S1 = w[1], S2 = w[2], S3 = w[3];
    const float t = e.y;
    // CubicResampler.cs:52-57, the reference's expression tree
    out[i] = S1 + t * (0.5f * (S2 - S0) + t * ((S0 - 2.5f * S1 + 2.f * S2 - 0.5f * S3) + t * (0.5f * (S3 - S0) + 1.5f * (S1 - S2))));
  }
}
void launch_resample_fast(hipStream_t s, const ResampleFastJob* jobs_dev, int njobs, int64_t max_blocks) {
  if (njobs <= 0 || max_blocks <= 0) return;
  // (8 rounds per workgroup when there are thousands of jobs: 4096 voices x 469 one-round workgroups were bound by the dispatch rate)
  int rounds = 1;
  if (const char* e = expenv("GA_RS_ROUNDS")) rounds = std::max(1, atoi(e));
  else if ((int64_t)njobs * ((max_blocks * kBlock + 255) / 256) > 200000) rounds = 8;
  const int gx = (int)std::min<int64_t>((max_blocks * kBlock + 256 * rounds - 1) / (256 * rounds), 512);
  GA_LAUNCH_JOBS(resample_fast_kernel, gx, 256, jobs_dev, njobs);
}
void launch_resample(hipStream_t s, const ResampleJob* jobs_dev, int njobs, const ResampleBlock* traj_dev, int64_t max_blocks) {
  if (njobs <= 0 || max_blocks <= 0) return;
  int gx = (int)((max_blocks + 63) / 64);
  GA_LAUNCH_JOBS(resample_kernel, gx, 64, jobs_dev, njobs, traj_dev);
}


// =====================================================================================================
//  General source replay (see GsrBlock): one lane per block, fed samples gathered by index with the loop wrap rule
//  `pos >= loopEnd -> loopStart` (AudioBufferSourceNode.cs:262-265,297-314).  A rare path (looping + resampling,
//  moving playbackRate): gathers are uncoalesced, stores go through LDS like resample_kernel.
// =====================================================================================================
__global__ __launch_bounds__(64) void gsr_kernel(const GsrJob* __restrict jobs, const uint8_t* __restrict base) {
  __shared__ float tile[64][kBlock + 1];
  const GsrJob job = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  const int64_t bl0 = (int64_t)blockIdx.x * 64;
  if (bl0 >= job.nblocks) return;
  const int64_t b = bl0 + lane;
  if (b < job.nblocks) {
    const GsrBlock d = ((const GsrBlock*)(base + job.desc_off))[b];   // (`base` is a kernel argument: global)
    const GA_GLOBAL float* __restrict in = gptr(job.buf);
    int64_t ip = d.next;
    auto feed = [&]() {
      float v = in[ip++];
      if (job.loop && ip >= job.loop_end) ip = job.loop_start;
      return v;
    };
    int outp = 0;
    if (d.copy) {
      for (; outp < d.produced; outp++) tile[lane][outp] = feed();
    } else {
      float S0 = d.w[0] >= 0 ? in[d.w[0]] : 0.f, S1 = d.w[1] >= 0 ? in[d.w[1]] : 0.f;
      float S2 = d.w[2] >= 0 ? in[d.w[2]] : 0.f, S3 = d.w[3] >= 0 ? in[d.w[3]] : 0.f;
      double Pos = d.pos;
      int ready = d.ready;
      if (d.produced > 0) {
        while (ready < 4) {   // priming (CubicResampler.cs:31-35); the host has checked that the inputs exist
          S0 = S1; S1 = S2; S2 = S3; S3 = feed();
          ready++;
        }
        for (; outp < d.produced; outp++) {
          int consume = (int)Pos;
          for (int i = 0; i < consume; i++) { S0 = S1; S1 = S2; S2 = S3; S3 = feed(); }
          Pos -= consume;
          float t = (float)Pos;
          tile[lane][outp] = S1 + t * (0.5f * (S2 - S0) + t * ((S0 - 2.5f * S1 + 2.f * S2 - 0.5f * S3) + t * (0.5f * (S3 - S0) + 1.5f * (S1 - S2))));
          Pos += d.rate;
        }
      }
    }
    for (; outp < kBlock; outp++) tile[lane][outp] = 0.f;
  }
  __syncthreads();
  const int nb = (int)min<int64_t>(64, job.nblocks - bl0);
  GA_GLOBAL float* __restrict out = gptr(job.out) + (job.b0 + bl0) * kBlock;
  for (int r = 0; r < nb; r++) {
    out[(int64_t)r * kBlock + lane] = tile[r][lane];
    out[(int64_t)r * kBlock + 64 + lane] = tile[r][64 + lane];
  }
}
void launch_gsr(hipStream_t s, const GsrJob* jobs_dev, int njobs, const uint8_t* plan_base_dev, int64_t max_blocks) {
  if (njobs <= 0 || max_blocks <= 0) return;
  int gx = (int)((max_blocks + 63) / 64);
  GA_LAUNCH_JOBS(gsr_kernel, gx, 64, jobs_dev, njobs, plan_base_dev);
}


// =====================================================================================================
//  AudioStreamNodeBase.Process (see StreamPiece): one lane per block, pieces of a block in order; stores go through LDS like
//  resample_kernel.  A rare path (streams are few): gathers are per lane.
// =====================================================================================================
__global__ __launch_bounds__(64) void stream_kernel(const StreamJob* __restrict jobs, const uint8_t* __restrict base) {
  __shared__ float tile[64][kBlock + 1];
  const StreamJob job = jobs[blockIdx.y];
  const int lane = threadIdx.x;
  const int64_t bl0 = (int64_t)blockIdx.x * 64;
  if (bl0 >= job.nblocks) return;
  const StreamSeg* __restrict segs = (const StreamSeg*)(base + job.segs_off);
  const GA_GLOBAL float* win_in = gptr(job.win_in);
  auto sample = [&](int seg, int64_t idx) -> float {
    if (seg == -1) return 0.f;
    if (seg == -2) return win_in[idx];
    const StreamSeg sg = segs[seg];
    return ldg1(sg.base + (int64_t)job.ch * sg.stride + idx);
  };
  const int64_t b = bl0 + lane;
  if (b < job.nblocks) {
    const StreamBlock blk = ((const StreamBlock*)(base + job.blocks_off))[job.b0 + b];
    for (int i = 0; i < kBlock; i++) tile[lane][i] = 0.f;   // frames no piece covers are cleared
    for (int q = 0; q < blk.npieces; q++) {
      const StreamPiece d = ((const StreamPiece*)(base + job.pieces_off))[blk.piece0 + q];
      const StreamSeg sg = segs[d.seg];
      const GA_GLOBAL float* __restrict in = gptr(sg.base) + (int64_t)job.ch * sg.stride;
      int64_t ip = d.next;
      int outp = d.out0;
      const int oend = d.out0 + d.produced;
      if (d.copy) {
        for (; outp < oend; outp++) tile[lane][outp] = in[ip++];
      } else {
        float S0 = sample(d.wseg[0], d.w[0]), S1 = sample(d.wseg[1], d.w[1]);
        float S2 = sample(d.wseg[2], d.w[2]), S3 = sample(d.wseg[3], d.w[3]);
        double Pos = d.pos;
        int ready = d.ready;
        if (d.produced > 0) {
          while (ready < 4) {   // priming (CubicResampler.cs:31-35); the host has checked that the inputs exist
            S0 = S1; S1 = S2; S2 = S3; S3 = in[ip++];
            ready++;
          }
          for (; outp < oend; outp++) {
            int consume = (int)Pos;
            for (int i = 0; i < consume; i++) { S0 = S1; S1 = S2; S2 = S3; S3 = in[ip++]; }
            Pos -= consume;
            float t = (float)Pos;
            tile[lane][outp] = S1 + t * (0.5f * (S2 - S0) + t * ((S0 - 2.5f * S1 + 2.f * S2 - 0.5f * S3) + t * (0.5f * (S3 - S0) + 1.5f * (S1 - S2))));
            Pos += d.rate;
          }
        }
      }
    }
  }
  __syncthreads();
  const int nb = (int)min<int64_t>(64, job.nblocks - bl0);
  GA_GLOBAL float* __restrict out = gptr(job.out) + (job.b0 + bl0) * kBlock;
  for (int r = 0; r < nb; r++) {
    out[(int64_t)r * kBlock + lane] = tile[r][lane];
    out[(int64_t)r * kBlock + 64 + lane] = tile[r][64 + lane];
  }
  if (job.win_out && blockIdx.x == 0 && lane < 4) gptr(job.win_out)[lane] = sample(job.wend_seg[lane], job.wend[lane]);
}
void launch_stream(hipStream_t s, const StreamJob* jobs_dev, int njobs, const uint8_t* plan_base_dev, int64_t max_blocks) {
  if (njobs <= 0 || max_blocks <= 0) return;
  int gx = (int)((max_blocks + 63) / 64);
  GA_LAUNCH_JOBS(stream_kernel, gx, 64, jobs_dev, njobs, plan_base_dev);
}

// =====================================================================================================
//  ConstantSourceNode / OscillatorNode / StereoPannerNode (see ga_kernels.hpp)
// =====================================================================================================
__global__ __launch_bounds__(256) void const_source_kernel(const ConstJob* __restrict jobs) {
  const ConstJob job = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i;
    float v = 0.f;
    if (f >= job.lo && f < job.hi) v = job.curve ? gptr(job.curve)[f] : job.value;
    gptr(job.out)[f] = v;
  }
}
void launch_const_source(hipStream_t s, const ConstJob* jobs_dev, int njobs, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 1024);
  GA_LAUNCH_JOBS(const_source_kernel, gx, 256, jobs_dev, njobs);
}

__device__ __forceinline__ float osc_sample(double ph, int type) {   // GenerateSample, OscillatorNode.cs:171-195
  const double PI = 3.14159265358979323846;
  switch (type) {
    case 0: return (float)sin(ph);
    case 1: return ph < PI ? 1.0f : -1.0f;
    case 2: return (float)(2.0 * (ph / (2.0 * PI)) - 1.0);
    case 3: {
      double t = ph / (2.0 * PI);
      return (float)(4.0 * fabs(t - floor(t + 0.5)) - 1.0);
    }
    default: return 0.f;
  }
}
__global__ __launch_bounds__(64) void oscillator_kernel(const OscJob* __restrict jobs) {
  extern __shared__ double osc_lds[];
  // [64] block-start phases | [64][129] floats: samples out | (curve jobs only) [64][128] doubles: phase increments
  double* start_ph = osc_lds;
  float (*tile)[kBlock + 1] = reinterpret_cast<float (*)[kBlock + 1]>(osc_lds + 64);
  double* incs = osc_lds + 64 + (64 * (kBlock + 1) * sizeof(float) + 7) / 8;
  const OscJob job = jobs[blockIdx.x];
  const int lane = threadIdx.x;
  const double PI2 = 2.0 * 3.14159265358979323846;
  const double sr = (double)job.sample_rate;
  const bool curve = job.curve != nullptr;
  const double inc_const = (PI2 * (double)job.value) / sr;   // `(2.0 * Math.PI * freqValues[i]) / Context.SampleRate` (:140)
  double ph = *gptr(job.phase);
  const int64_t nblk = job.n / kBlock;
  for (int64_t g0 = 0; g0 < nblk; g0 += 64) {
    const int nb = (int)min<int64_t>(64, nblk - g0);
    const int64_t fg = job.f0 + g0 * kBlock;   // first frame of this group of blocks
    if (curve) {   // the increments of the whole group, computed 64-wide (the division stays out of the serial chain)
      for (int r = 0; r < nb; r++) {
        const int64_t f = fg + (int64_t)r * kBlock;
        incs[r * kBlock + lane] = (PI2 * (double)gptr(job.curve)[f + lane]) / sr;
        incs[r * kBlock + 64 + lane] = (PI2 * (double)gptr(job.curve)[f + 64 + lane]) / sr;
      }
      __syncthreads();
    }
    if (lane == 0) {   // the serial recurrence (:139-143), only the phase: `_phase += inc; if (_phase >= 2 pi) _phase -= 2 pi`
      for (int r = 0; r < nb; r++) {
        start_ph[r] = ph;
        const int64_t f = fg + (int64_t)r * kBlock;
        if (f + kBlock <= job.lo || f >= job.hi) continue;
        const int i0 = (int)max<int64_t>(job.lo - f, 0), i1 = (int)min<int64_t>(job.hi - f, kBlock);
        if (curve) {
          for (int i = i0; i < i1; i++) {
            ph += incs[r * kBlock + i];
            if (ph >= PI2) ph -= PI2;
          }
        } else {
          for (int i = i0; i < i1; i++) {
            ph += inc_const;
            if (ph >= PI2) ph -= PI2;
          }
        }
      }
    }
    __syncthreads();
    if (lane < nb) {
      double p = start_ph[lane];
      const int64_t f = fg + (int64_t)lane * kBlock;
      for (int i = 0; i < kBlock; i++) {
        float v = 0.f;
        if (f + i >= job.lo && f + i < job.hi) {
          v = osc_sample(p, job.type);
          p += curve ? incs[lane * kBlock + i] : inc_const;
          if (p >= PI2) p -= PI2;
        }
        tile[lane][i] = v;
      }
    }
    __syncthreads();
    for (int r = 0; r < nb; r++) {
      GA_GLOBAL float* o = gptr(job.out) + fg + (int64_t)r * kBlock;
      o[lane] = tile[r][lane];
      o[64 + lane] = tile[r][64 + lane];
    }
    __syncthreads();
    ph = __shfl(ph, 0);   // every lane carries the running phase (only lane 0 advanced it)
  }
  if (lane == 0) *gptr(job.phase) = ph;
}
void launch_oscillator(hipStream_t s, const OscJob* jobs_dev, int njobs, bool any_curve) {
  if (njobs <= 0) return;
  size_t lds = 64 * sizeof(double) + ((64 * (kBlock + 1) * sizeof(float) + 7) / 8) * 8 + (any_curve ? 64 * kBlock * sizeof(double) : 0);
  if (lds > 48 * 1024)   // (a per-device attribute: set whenever it is needed, a process may drive several devices)
    (void)hipFuncSetAttribute((const void*)oscillator_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  hipLaunchKernelGGL(oscillator_kernel, dim3(njobs), dim3(64), lds, s, jobs_dev);
}

__global__ __launch_bounds__(256) void stereo_panner_kernel(const PanJob* __restrict jobs) {
  const PanJob job = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i;
    if (!job.stereo) {   // ProcessMono, :103-105
      const float x = gptr(job.in_l)[f];
      gptr(job.out_l)[f] = x * job.gain_l;
      gptr(job.out_r)[f] = x * job.gain_r;
    } else {             // ProcessStereo, :135-145
      const float inl = gptr(job.in_l)[f], inr = gptr(job.in_r)[f];
      if (job.pan <= 0.0f) {
        gptr(job.out_l)[f] = inl + inr * job.gain_l;
        gptr(job.out_r)[f] = inr * job.gain_r;
      } else {
        gptr(job.out_l)[f] = inl * job.gain_l;
        gptr(job.out_r)[f] = inr + inl * job.gain_r;
      }
    }
  }
}
void launch_stereo_panner(hipStream_t s, const PanJob* jobs_dev, int njobs, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 1024);
  GA_LAUNCH_JOBS(stereo_panner_kernel, gx, 256, jobs_dev, njobs);
}


__global__ __launch_bounds__(256) void delay_kernel(const DelayJob* __restrict jobs) {
  const DelayJob job = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i;
    const float dt = job.curve ? gptr(job.curve)[f] : job.value;
    int d = (int)(dt * (float)job.sample_rate);   // float * int -> float, truncated (:68, :88)
    d = min(max(d, 0), job.max_delay);
    gptr(job.out)[f] = d > 0 ? gptr(job.line)[f - d] : 0.f;
  }
}
void launch_delay(hipStream_t s, const DelayJob* jobs_dev, int njobs, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 1024);
  GA_LAUNCH_JOBS(delay_kernel, gx, 256, jobs_dev, njobs);
}


__device__ __forceinline__ void pan_gains(float pan, int stereo, float& gl, float& gr) {   // :94-98 / :129-133
  const float PIf = 3.14159265358979323846f;
  const float x = stereo ? (pan <= 0.0f ? pan + 1.0f : pan) : (pan + 1.0f) * 0.5f;
  const float a = x * PIf / 2.0f;
  gl = (float)cos((double)a);   // rounded once from double, as the C library's cosf / sinf behind MathF are
  gr = (float)sin((double)a);
}
__global__ __launch_bounds__(64) void stereo_panner_dynamic_kernel(const PanDynJob* __restrict jobs) {
  __shared__ float chg_pan[64];
  __shared__ int chg_has[64];
  __shared__ PanState carry[65];
  const PanDynJob job = jobs[blockIdx.x];
  const int lane = threadIdx.x;
  const GA_GLOBAL float* psrc = (const GA_GLOBAL float*)job.state;   // {last_pan, gain_l, gain_r, pad}
  PanState st = job.init ? job.init_state : PanState{psrc[0], psrc[1], psrc[2], psrc[3]};
  const int64_t nblk = job.n / kBlock;
  auto panAt = [&](int64_t f) { return fminf(fmaxf(job.curve ? gptr(job.curve)[f] : job.value, -1.0f), 1.0f); };   // Math.Clamp(panValues[i], -1, 1)
  for (int64_t g0 = 0; g0 < nblk; g0 += 64) {
    const int nb = (int)min<int64_t>(64, nblk - g0);
    const int64_t fb = job.f0 + (g0 + lane) * kBlock;
    // last index of this block at which pan differs from the previous sample's pan (the previous sample of the block's first
    // frame is the last frame of the block before; for the job's very first frame it is the carried _lastPan)
    int has = 0;
    float cp = 0.f;
    if (lane < nb) {
      float prev = (g0 + lane == 0) ? st.last_pan : panAt(fb - 1);
      for (int i = 0; i < kBlock; i++) {
        const float p = panAt(fb + i);
        if (p != prev) { has = 1; cp = p; }   // NaN != NaN: a NaN _lastPan (initial state) forces the first computation
        prev = p;
      }
    }
    chg_has[lane] = has;
    chg_pan[lane] = cp;
    __syncthreads();
    if (lane == 0) {   // state at the start of every block of the group
      PanState cur = st;
      for (int r = 0; r < nb; r++) {
        carry[r] = cur;
        if (chg_has[r]) {
          cur.last_pan = chg_pan[r];
          pan_gains(cur.last_pan, job.stereo, cur.gain_l, cur.gain_r);
        }   // no change in the block: every pan of the block equals the carried _lastPan
      }
      carry[nb] = cur;
    }
    __syncthreads();
    if (lane < nb) {
      PanState s0 = carry[lane];
      float gl = s0.gain_l, gr = s0.gain_r, lp = s0.last_pan;
      for (int i = 0; i < kBlock; i++) {
        const int64_t f = fb + i;
        const float pan = panAt(f);
        if (pan != lp) {
          pan_gains(pan, job.stereo, gl, gr);
          lp = pan;
        }
        if (!job.stereo) {
          const float x = gptr(job.in_l)[f];
          gptr(job.out_l)[f] = x * gl;
          gptr(job.out_r)[f] = x * gr;
        } else {
          const float inl = gptr(job.in_l)[f], inr = gptr(job.in_r)[f];
          if (pan <= 0.0f) {
            gptr(job.out_l)[f] = inl + inr * gl;
            gptr(job.out_r)[f] = inr * gr;
          } else {
            gptr(job.out_l)[f] = inl * gl;
            gptr(job.out_r)[f] = inr + inl * gr;
          }
        }
      }
    }
    __syncthreads();
    st = carry[nb];
    __syncthreads();
  }
  if (lane == 0) {
    GA_GLOBAL float* pdst = (GA_GLOBAL float*)job.state;
    pdst[0] = st.last_pan; pdst[1] = st.gain_l; pdst[2] = st.gain_r; pdst[3] = st.pad_;
  }
}
void launch_stereo_panner_dynamic(hipStream_t s, const PanDynJob* jobs_dev, int njobs) {
  if (njobs <= 0) return;
  hipLaunchKernelGGL(stereo_panner_dynamic_kernel, dim3(njobs), dim3(64), 0, s, jobs_dev);
}


__global__ __launch_bounds__(256) void interleave_kernel(float* __restrict dst, InterleaveSrc src, int channels, int used, int64_t f0, int64_t n) {
  // one thread per output element: consecutive threads write consecutive floats (coalesced); the reads of one channel are
  // strided by `channels` across the wave and served from L2
  const int64_t total = n * channels;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = i / channels;
    const int ch = (int)(i - f * channels);
    dst[(f0 + f) * channels + ch] = ch < used ? src.ch[ch][f0 + f] : 0.f;
  }
}
void launch_interleave(hipStream_t s, float* dst, InterleaveSrc src, int channels, int used, int64_t f0, int64_t n) {
  if (n <= 0 || channels <= 0) return;
  int gx = (int)std::min<int64_t>((n * channels + 255) / 256, 4096);
  hipLaunchKernelGGL(interleave_kernel, dim3(gx), dim3(256), 0, s, dst, src, channels, used, f0, n);
}


__global__ __launch_bounds__(256) void param_mod_kernel(const ParamModJob* __restrict jobs) {
  const ParamModJob job = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = job.f0 + i;
    const int64_t fs = job.krate ? f - (f % kBlock) : f;   // k-rate: `_input.Buffer.GetChannelSpan(0)[0]` and the block-start value
    const float intr = job.intrinsic ? gptr(job.intrinsic)[fs] : job.value;
    gptr(job.out)[f] = fminf(fmaxf(intr + gptr(job.mod)[fs], job.vmin), job.vmax);   // Math.Clamp(intrinsicValue + modulation, min, max)
  }
}
void launch_param_mod(hipStream_t s, const ParamModJob* jobs_dev, int njobs, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return;
  int gx = (int)std::min<int64_t>((max_n + 255) / 256, 1024);
  GA_LAUNCH_JOBS(param_mod_kernel, gx, 256, jobs_dev, njobs);
}

}  // namespace ga

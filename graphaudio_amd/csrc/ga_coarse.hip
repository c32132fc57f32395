// ga_coarse.hip -- formulation D of the convolver: COARSE partitions with the consumer's sum fused in the frequency domain.
//
// The reference evaluates  y = x * h  as a uniformly partitioned convolution with 128-sample partitions because it has to
// answer every 128 frames (PartitionedConvolver.cs:104-223).  An offline render knows a whole chunk of input, so the same
// linear convolution can use partitions of CB = 8192 samples: P' = ceil(taps / 8192) partitions (8 for the 65,536-tap
// impulse response of BASELINE.json configs[2]) instead of 512, overlap-save with 16,384-point real transforms:
//
//   X[u]   = RFFT( x[(u-1) CB .. (u+1) CB) )                         coarse_fwd_kernel   (one transform per window)
//   Y[t]   = sum over terms (x, h) of  sum_{p<P'} X_x[t-p] . H_h[p]   coarse_mac_kernel   (sliding window along t in LDS)
//   y[t]   = IRFFT( sum of the Y rows of an output )[CB .. 2 CB)      coarse_inv_kernel
//
// "Terms" is where the destination mix goes (AudioNodeInput.cs:118-132,195-198): convolver outputs that are only consumed by
// one summing input are accumulated as spectra, so a 1024-voice bus costs one inverse transform per output channel and
// coarse block instead of one per voice -- the per-voice spectra Y and the per-voice output slabs never exist in HBM.
// Every voice's forward transform and its product with its OWN impulse-response spectra are evaluated per voice; nothing is
// pre-summed across voices in front of the convolution.  State between chunks is the last P' x CB INPUT samples per row
// (time domain): overlap-save has no output-side state, so nodes keep nothing that depends on who consumes them.
//
// HBM traffic per 10 s step of config 3 (1024 voices): input 2.0 GB (+0.3 history) in, X 4.4 GB out, 4.4 GB in, Y 0.25 GB
// out + in, bus out -- ~11 GB against ~35 GB for formulation C.
//
// Real transform of N = 16,384 points through TWO complex transforms of 4,096 points (the radix-16 register / LDS
// transform of ga_fft16.hpp) on z_a[m] = x[4m] + i x[4m+2], z_b[m] = x[4m+1] + i x[4m+3], and one combine pass
// (tools/proto/coarse_math.py is the numpy statement of the index arithmetic).  Spectra are "packed": 8192 complex values per
// frame, bin 0 holding the two real bins (X[0], X[8192]).  All power-of-two scale factors (the 1/2 of the even/odd
// splits, 1/4096 of the inverse) are folded into the impulse-response spectra: exact.
#include "ga_kernels.hpp"
#include "ga_fft16.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace ga {

constexpr int CM = 4096;                       // complex transform length
constexpr int CPAD = TC16_PADDED(CM);          // LDS slots of one transform buffer (one pad slot per 32)
__device__ __forceinline__ int cpad(int j) { return j + (j >> 5); }
__device__ __forceinline__ int cslot(int s) { return (s >> 12) * CPAD + cpad(s & (CM - 1)); }   // packed bin / z index -> LDS slot

__device__ __forceinline__ f2 cj(f2 a) { return f2{a.x, -a.y}; }
__device__ __forceinline__ f2 mul_mi(f2 a) { return f2{a.y, -a.x}; }    // a * (-i)
__device__ __forceinline__ f2 mul_pi(f2 a) { return f2{-a.y, a.x}; }    // a * (+i)
__device__ __forceinline__ f2 cmulc(f2 a, f2 b) { return cmulp(a, b); }

// acc + a b  (two packed fmas)
__device__ __forceinline__ f2 cfmap(f2 a, f2 b, f2 acc) {
  f2 t, r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(b), "v"(acc));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}

// =====================================================================================================================
//  forward: one workgroup of 512 threads = the two complex transforms (g = tid / 256) of one window, side by side; it walks a
//  run of consecutive windows of one row so that a window's second half is the next window's first half (each input sample
//  is fetched once per run) and the next half is in flight while the current window is transformed.
// =====================================================================================================================
struct HalfRegs {
  f2 v[8];
};
// the 8 (re, im) points this thread owns in half-window `j` of the row: x[4 (t + 256 q) + g], x[.. + g + 2]
__device__ __forceinline__ void coarse_issue_half(const CoarseXRow& R, int j, int g, int t, v4f (&raw)[8]) {
  const float* src = nullptr;
  int64_t lim = 0;   // samples of this half that exist (the rest is zero)
  if (j < 0) {
    const int64_t off = (int64_t)R.hist_len + (int64_t)j * kCoarseBlock;
    if (R.hist && off >= 0) {
      src = R.hist + off;
      lim = kCoarseBlock;
    }
  } else if (R.in) {
    const int64_t off = (int64_t)j * kCoarseBlock;
    src = R.in + off;
    lim = std::min<int64_t>(kCoarseBlock, R.nvalid - off);
  }
  const bool aligned = ((uintptr_t)src & 15) == 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int o = 4 * (t + 256 * q);
    raw[q] = v4f{0.f, 0.f, 0.f, 0.f};
    if (src && o < lim) {   // lim is a multiple of 4 (chunks are whole 128-frame blocks, histories whole coarse blocks)
      if (aligned) {
        raw[q] = ldg4(src + o);
      } else {
        raw[q] = v4f{ldg1(src + o), ldg1(src + o + 1), ldg1(src + o + 2), ldg1(src + o + 3)};
      }
    }
  }
  (void)g;
}
__device__ __forceinline__ void coarse_pick_half(const v4f (&raw)[8], int g, float scale, HalfRegs& h) {
#pragma unroll
  for (int q = 0; q < 8; q++) h.v[q] = g ? f2{raw[q].y * scale, raw[q].w * scale} : f2{raw[q].x * scale, raw[q].z * scale};
}

__global__ __launch_bounds__(512) void coarse_fwd_kernel(const CoarseXRow* __restrict rows, int run, float2* __restrict X,
                                                         const float2* __restrict twg, const float2* __restrict twab, int exp) {
  using PL = R16Plan<CM>;
  extern __shared__ f2 clds[];
  f2* tw2 = clds;
  f2* tw3 = clds + PL::T2;
  f2* zb0 = clds + PL::T2 + PL::T3;
  const int tid = threadIdx.x;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 8), t = tid & 255;
  f2* buf = zb0 + g * CPAD;
  for (int i = tid; i < PL::T2 + PL::T3; i += 512) clds[i] = f2{twg[i].x, twg[i].y};
  const CoarseXRow R = rows[blockIdx.y];
  const int w0 = blockIdx.x * run;
  const int w1 = min(R.n_frames, w0 + run);
  if (w0 >= w1) return;   // (uniform)
  // combine-pass twiddles of this thread's bins k = tid + 512 q: a = W_8192^k, b = W_16384^k
  f2 ta[4], tb[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    ta[q] = f2{twab[tid + 512 * q].x, twab[tid + 512 * q].y};
    tb[q] = f2{twab[2049 + tid + 512 * q].x, twab[2049 + tid + 512 * q].y};
  }
  const bool zero2 = (R.flags & 1) != 0;   // impulse-response partitions: [h_p | 0]
  HalfRegs cur, nxt;
  {
    v4f raw[8];
    coarse_issue_half(R, R.u0 + w0 - 1, g, t, raw);
    coarse_pick_half(raw, g, R.scale, cur);
    if (zero2) {
#pragma unroll
      for (int q = 0; q < 8; q++) nxt.v[q] = f2{0.f, 0.f};
    } else {
      coarse_issue_half(R, R.u0 + w0, g, t, raw);
      coarse_pick_half(raw, g, R.scale, nxt);
    }
  }
  __syncthreads();
  for (int w = w0; w < w1; w++) {
    const int u = R.u0 + w;
    v4f raw[8];
    const bool more = w + 1 < w1;
    if (more && !(exp & 8)) coarse_issue_half(R, zero2 ? u : u + 1, g, t, raw);   // in flight behind this window's transforms
    f2 own[16];
#pragma unroll
    for (int m = 0; m < 8; m++) {
      own[m] = cur.v[m];
      own[8 + m] = nxt.v[m];
    }
    if (!(exp & 4)) fft16_own<CM>(own, buf, tw2, tw3, t);
    __syncthreads();   // the last pass has read the buffer: store Z_g in natural order
#pragma unroll
    for (int m = 0; m < 16; m++) buf[cpad(t + 256 * m)] = own[m];
    __syncthreads();
    // the prefetched half is consumed BEFORE the combine pass issues its stores: vmcnt counts loads and stores in issue
    // order, so waiting for these loads later would wait for this window's stores to complete as well
    if (more) {
      if (zero2) {
        coarse_pick_half(raw, g, R.scale, cur);
      } else {
        cur = nxt;
        coarse_pick_half(raw, g, R.scale, nxt);
      }
    }
    float2* __restrict F = X + (size_t)(R.frame0 + w) * kCoarseBins;
    auto quad = [&](int k, f2 a, f2 b) {
      const int km = (CM - k) & (CM - 1);
      const f2 zak = zb0[cpad(k)], zam = cj(zb0[cpad(km)]);
      const f2 zbk = zb0[CPAD + cpad(k)], zbm = cj(zb0[CPAD + cpad(km)]);
      // (all values carry a factor 2 -- folded into the impulse-response scale)
      const f2 fea = zak + zam, foa = mul_mi(zak - zam);
      const f2 feb = zbk + zbm, fob = mul_mi(zbk - zbm);
      const f2 afa = cmulc(a, foa), afb = cmulc(a, fob);
      const f2 Ek = fea + afa, Emk = cj(fea - afa);
      const f2 Ok = feb + afb, Omk = cj(feb - afb);
      const f2 S = cmulc(b, Ok);
      const f2 T = cmulc(mul_mi(cj(b)), Omk);
      const f2 Xk = Ek + S, Xnk = cj(Ek - S);
      const f2 Xmk = Emk + T, Xpk = cj(Emk - T);
      if (exp & 1) {
        if (Xk.x == 12345.f && Xnk.x == 54321.f && Xmk.x == 999.f && Xpk.y == 777.f) F[0] = make_float2(0.f, 0.f);
      } else if (k == 0) {
        F[0] = make_float2(Xk.x, Xnk.x);            // packed: (X[0], X[8192])
        F[CM] = make_float2(Xmk.x, Xmk.y);
      } else {
        F[k] = make_float2(Xk.x, Xk.y);
        F[2 * CM - k] = make_float2(Xnk.x, Xnk.y);
        if (k != CM / 2) {
          F[CM - k] = make_float2(Xmk.x, Xmk.y);
          F[CM + k] = make_float2(Xpk.x, Xpk.y);
        }
      }
    };
    if (!(exp & 2)) {
#pragma unroll
    for (int q = 0; q < 4; q++) quad(tid + 512 * q, ta[q], tb[q]);
    }
    if (tid == 0 && !(exp & 2)) quad(CM / 2, f2{twab[CM / 2].x, twab[CM / 2].y}, f2{twab[2049 + CM / 2].x, twab[2049 + CM / 2].y});
    __syncthreads();   // the combine pass has read both buffers before the next transform writes them
  }
}

void launch_coarse_fwd(hipStream_t s, const CoarseXRow* rows_dev, int nrows, int max_frames, int run, float2* X, const float2* tw16,
                       const float2* twab) {
  if (nrows <= 0 || max_frames <= 0) return;
  using PL = R16Plan<CM>;
  const size_t lds = (size_t)(PL::T2 + PL::T3 + 2 * CPAD) * sizeof(float2);
  if (hipFuncSetAttribute((const void*)coarse_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    launch_fail("cannot raise the dynamic LDS limit of the coarse forward transform");
  run = std::max(run, 1);
  for (int r0 = 0; r0 < nrows; r0 += 32768) {
    dim3 grid((max_frames + run - 1) / run, std::min(32768, nrows - r0));
    static const int exp = getenv("GA_COARSE_EXP") ? atoi(getenv("GA_COARSE_EXP")) : 0;   // timing experiments only
    hipLaunchKernelGGL(coarse_fwd_kernel, grid, dim3(512), lds, s, rows_dev + r0, run, X, tw16, twab, exp & 15);
  }
}

// =====================================================================================================================
//  multiply-accumulate: workgroup = (job, tile of 64 bins); lane = bin, wave w = a quarter of the job's coarse blocks.
//  The X frames of one term (all windows the job needs, 512 bytes each) are staged in LDS -- double buffered, the next term's
//  frames are fetched while this one is accumulated -- and every wave slides over them:
//      acc[t][c] += X[t - p] . H_c[p]          p < P', c < CW columns, t in the wave's range
//  with the accumulators of ALL terms of the job in registers: the sum over the job's voices costs no memory traffic.
// =====================================================================================================================
template <int CW, int TW>
__global__ __launch_bounds__(256, 2) void coarse_mac_kernel(const CoarseJob* __restrict jobs, const CoarseTerm* __restrict terms,
                                                            const float2* __restrict X, float2* __restrict Y, int y_frames, int NFA, int exp) {
  extern __shared__ f2 mlds[];
  const CoarseJob J = jobs[blockIdx.y];
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = J.P, nT = J.n_t;
  const int NF = nT + P - 1;                 // frames of a term that are needed
  // NFA = frames an LDS buffer holds (launch-wide: the unrolled sweep of the last active wave may read past NF, never stored)
  f2* xs0 = mlds;
  f2* xs1 = mlds + (size_t)NFA * 64;
  f2* hs0 = mlds + (size_t)2 * NFA * 64;
  f2* hs1 = hs0 + (size_t)P * CW * 64;       // only when the job's terms have different impulse responses
  const int twr = (nT + 3) / 4;              // coarse blocks per wave (<= TW)
  const int t0w = wv * twr;
  const bool special = tile == 0;            // bin 0 of tile 0 is the packed pair of real bins
  const bool lane0 = special && lane == 0;
  const size_t binoff = (size_t)tile * 64;

  // staging of one term: NF x 32 float4 of X (+ P x CW x 32 float4 of H)
  constexpr int XR = (4 * TW + 16 + 7) / 8;  // float4 per thread that cover NF <= 4 TW + 15 frames
  float4 xr[XR];
  auto issue_x = [&](const CoarseTerm& T) {
#pragma unroll
    for (int r = 0; r < XR; r++) {
      const int idx = tid + 256 * r;
      const int fr = idx >> 5, of = idx & 31;
      xr[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (fr < NF) xr[r] = *reinterpret_cast<const float4*>(X + (size_t)(T.frame0 + J.t0 + fr) * kCoarseBins + binoff + 2 * of);
    }
  };
  auto commit_x = [&](f2* xs) {
#pragma unroll
    for (int r = 0; r < XR; r++) {
      const int idx = tid + 256 * r;
      const int fr = idx >> 5, of = idx & 31;
      if (fr < NF) *reinterpret_cast<float4*>(xs + fr * 64 + 2 * of) = xr[r];
    }
  };
  auto load_h = [&](const CoarseTerm& T, f2* hs) {   // straight to LDS (small: P x CW x 512 bytes)
    for (int idx = tid; idx < P * CW * 32; idx += 256) {
      const int pc = idx >> 5, of = idx & 31;
      const int p = pc / CW, c = pc % CW;
      *reinterpret_cast<v4f*>(hs + pc * 64 + 2 * of) = ldg4(T.h[c] + (size_t)p * kCoarseBins + binoff + 2 * of);
    }
  };

  f2 acc[TW][CW];
#pragma unroll
  for (int tt = 0; tt < TW; tt++)
#pragma unroll
    for (int c = 0; c < CW; c++) acc[tt][c] = f2{0.f, 0.f};

  const CoarseTerm* __restrict T = terms + J.term0;
  issue_x(T[0]);
  load_h(T[0], hs0);
  commit_x(xs0);
  __syncthreads();
  for (int i = 0; i < J.n_terms; i++) {
    const bool more = i + 1 < J.n_terms;
    f2* xs = (i & 1) ? xs1 : xs0;
    f2* hs = (J.shared_h || !(i & 1)) ? hs0 : hs1;
    if (more && !(exp & 2)) issue_x(T[i + 1]);
    if (t0w < nT && !(exp & 1)) {
      for (int p = 0; p < P; p++) {
        f2 h[CW];
#pragma unroll
        for (int c = 0; c < CW; c++) h[c] = hs[(p * CW + c) * 64 + lane];
        const f2* __restrict xb = xs + (t0w + (P - 1) - p) * 64 + lane;
        if (!special) {
#pragma unroll
          for (int tt = 0; tt < TW; tt++) {
            const f2 x = xb[tt * 64];
#pragma unroll
            for (int c = 0; c < CW; c++) acc[tt][c] = cfmap(x, h[c], acc[tt][c]);
          }
        } else {
#pragma unroll
          for (int tt = 0; tt < TW; tt++) {
            const f2 x = xb[tt * 64];
#pragma unroll
            for (int c = 0; c < CW; c++) {
              const f2 gen = cfmap(x, h[c], acc[tt][c]);
              const f2 pk = __builtin_elementwise_fma(x, h[c], acc[tt][c]);   // two real bins side by side
              acc[tt][c] = lane0 ? pk : gen;
            }
          }
        }
      }
    }
    if (more) {
      commit_x((i & 1) ? xs0 : xs1);
      if (!J.shared_h) load_h(T[i + 1], (i & 1) ? hs0 : hs1);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tt = 0; tt < TW; tt++) {
    const int t = t0w + tt;
    if (tt < twr && t < nT && !(exp & 4)) {
#pragma unroll
      for (int c = 0; c < CW; c++)
        Y[((size_t)(J.yrow0 + c) * y_frames + J.t0 + t) * kCoarseBins + binoff + lane] = make_float2(acc[tt][c].x, acc[tt][c].y);
    }
  }
}

template <int CW, int TW>
static void launch_coarse_mac_t(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                                int y_frames, int max_t, int maxP, bool any_private) {
  static const int exp = getenv("GA_COARSE_EXP") ? atoi(getenv("GA_COARSE_EXP")) : 0;   // timing experiments only
  // frames the sweep of the last active wave touches: t0w + TW + P - 1 with t0w = (waves - 1) * ceil(n_t / 4)
  int NFA = 0;
  for (int nt = 1; nt <= max_t; nt++) {
    const int twr = (nt + 3) / 4, wl = (nt + twr - 1) / twr - 1;
    NFA = std::max(NFA, wl * twr + TW + maxP - 1);
  }
  const size_t lds = ((size_t)2 * NFA * 64 + (size_t)(any_private ? 2 : 1) * maxP * CW * 64) * sizeof(float2);
  if (lds > 160 * 1024) launch_fail("coarse multiply-accumulate: staging does not fit the LDS");
  if (hipFuncSetAttribute((const void*)coarse_mac_kernel<CW, TW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 65536)) !=
      hipSuccess)
    launch_fail("cannot raise the dynamic LDS limit of the coarse multiply-accumulate");
  if (getenv("GA_COARSE_EXP")) {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, coarse_mac_kernel<CW, TW>, 256, lds);
    fprintf(stderr, "[coarse_mac<%d,%d>] lds %zu B, NFA %d, occupancy %d workgroups/CU, %d jobs\n", CW, TW, lds, NFA, occ, njobs);
  }
  for (int j0 = 0; j0 < njobs; j0 += 32768)
    hipLaunchKernelGGL((coarse_mac_kernel<CW, TW>), dim3(kCoarseBins / 64, std::min(32768, njobs - j0)), dim3(256), lds, s, jobs_dev + j0,
                       terms_dev, X, Y, y_frames, NFA, exp >> 4);
}
// all jobs of one launch have the same column count `cw` (1, 2 or 4) and at most `max_t` coarse blocks (<= kCoarseJobBlocks(cw))
void launch_coarse_mac(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                       int y_frames, int cw, int max_t, int maxP, bool any_private) {
  if (njobs <= 0) return;
  if (max_t > kCoarseJobBlocks(cw)) launch_fail("coarse multiply-accumulate: too many coarse blocks in a job");
  const bool small = max_t <= 16;
  if (cw == 1) {
    if (small) launch_coarse_mac_t<1, 4>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
    else launch_coarse_mac_t<1, 16>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
  } else if (cw == 2) {
    if (small) launch_coarse_mac_t<2, 4>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
    else launch_coarse_mac_t<2, 16>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
  } else if (cw == 4) {
    if (small) launch_coarse_mac_t<4, 4>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
    else launch_coarse_mac_t<4, 8>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);   // 128 accumulator registers
  } else {
    launch_fail("coarse multiply-accumulate: unsupported column count");
  }
}

// =====================================================================================================================
//  inverse: workgroup = (output row, coarse block): sum the Y rows of the output (the frequency-domain mix), undo the
//  combine pass, two inverse complex transforms (forward transform of the swapped values), keep samples [CB, 2 CB).
// =====================================================================================================================
__global__ __launch_bounds__(512) void coarse_inv_kernel(const CoarseOut* __restrict outs, const int* __restrict ylist,
                                                         const float2* __restrict Y, int y_frames, const float2* __restrict twg,
                                                         const float2* __restrict twab) {
  using PL = R16Plan<CM>;
  extern __shared__ f2 clds[];
  f2* tw2 = clds;
  f2* tw3 = clds + PL::T2;
  f2* zb0 = clds + PL::T2 + PL::T3;
  const int tid = threadIdx.x;
  const int g = __builtin_amdgcn_readfirstlane(tid >> 8), t = tid & 255;
  f2* buf = zb0 + g * CPAD;
  const CoarseOut O = outs[blockIdx.y];
  const int tb = blockIdx.x;
  const int64_t nout = std::min<int64_t>(kCoarseBlock, O.nvalid - (int64_t)tb * kCoarseBlock);
  if (nout <= 0 || !O.out) return;   // (uniform)
  for (int i = tid; i < PL::T2 + PL::T3; i += 512) clds[i] = f2{twg[i].x, twg[i].y};
  // ---- frequency-domain mix: bins (2 i, 2 i + 1), i = tid + 512 r ----
  {
    float4 sum[8];
#pragma unroll
    for (int r = 0; r < 8; r++) sum[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < O.ny; j++) {
      const float4* __restrict src = reinterpret_cast<const float4*>(Y + ((size_t)ylist[O.y0 + j] * y_frames + tb) * kCoarseBins);
      float4 v[8];
#pragma unroll
      for (int r = 0; r < 8; r++) v[r] = src[tid + 512 * r];
#pragma unroll
      for (int r = 0; r < 8; r++) {
        sum[r].x += v[r].x; sum[r].y += v[r].y; sum[r].z += v[r].z; sum[r].w += v[r].w;
      }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      const int b0 = 2 * (tid + 512 * r);
      zb0[cslot(b0)] = f2{sum[r].x, sum[r].y};
      zb0[cslot(b0 + 1)] = f2{sum[r].z, sum[r].w};
    }
  }
  __syncthreads();
  // ---- undo the combine pass, in place: slots {k, 4096-k, 4096+k, 8192-k} -> za[k], za[4096-k], zb[k], zb[4096-k] ----
  auto quad = [&](int k, f2 a, f2 b) {
    f2 xk, xn, xm, xp;
    if (k == 0) {
      const f2 p0 = zb0[cslot(0)];
      xk = f2{p0.x, 0.f};
      xn = f2{p0.y, 0.f};
      xm = zb0[cslot(CM)];
      xp = xm;
    } else {
      xk = zb0[cslot(k)];
      xn = zb0[cslot(2 * CM - k)];
      xm = zb0[cslot(CM - k)];
      xp = zb0[cslot(CM + k)];
    }
    const f2 Ek = xk + cj(xn), Ok = cmulc(xk - cj(xn), cj(b));
    const f2 Emk = xm + cj(xp), Omk = cmulc(xm - cj(xp), mul_pi(b));
    const f2 ca = cj(a);
    const f2 fea = Ek + cj(Emk), foa = cmulc(Ek - cj(Emk), ca);
    const f2 feb = Ok + cj(Omk), fob = cmulc(Ok - cj(Omk), ca);
    const f2 zak = fea + mul_pi(foa), zam = cj(fea) + mul_pi(cj(foa));
    const f2 zbk = feb + mul_pi(fob), zbm = cj(feb) + mul_pi(cj(fob));
    zb0[cslot(k)] = zak;
    zb0[cslot(CM + k)] = zbk;
    if (k != 0 && k != CM / 2) {
      zb0[cslot(CM - k)] = zam;
      zb0[cslot(2 * CM - k)] = zbm;
    }
  };
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int k = tid + 512 * q;
    quad(k, f2{twab[k].x, twab[k].y}, f2{twab[2049 + k].x, twab[2049 + k].y});
  }
  if (tid == 0) quad(CM / 2, f2{twab[CM / 2].x, twab[CM / 2].y}, f2{twab[2049 + CM / 2].x, twab[2049 + CM / 2].y});
  __syncthreads();
  // ---- inverse complex transforms: ifft(v) = swap(fft(swap(v))) (1 / 4096 folded into the impulse-response spectra) ----
  f2 own[16];
#pragma unroll
  for (int m = 0; m < 16; m++) {
    const f2 z = buf[cpad(t + 256 * m)];
    own[m] = f2{z.y, z.x};
  }
  __syncthreads();
  fft16_own<CM>(own, buf, tw2, tw3, t);
  __syncthreads();
  // ---- samples [CB, 2 CB): z_g[j], j = t + 256 m >= 2048, holds x[4 j + g] (re) and x[4 j + g + 2] (im) ----
  float* st = reinterpret_cast<float*>(zb0);
#pragma unroll
  for (int m = 8; m < 16; m++) {
    const int o = 4 * (t + 256 * (m - 8)) + g;
    st[o] = own[m].y;       // swapped back
    st[o + 2] = own[m].x;
  }
  __syncthreads();
  float* __restrict dst = O.out + (int64_t)tb * kCoarseBlock;
  if ((((uintptr_t)dst) & 15) == 0) {
    for (int i = tid; i < kCoarseBlock / 4; i += 512)
      if (4 * i < nout) stg4(dst + 4 * i, *reinterpret_cast<const v4f*>(st + 4 * i));
  } else {
    for (int i = tid; i < kCoarseBlock; i += 512)
      if (i < nout) stg1(dst + i, st[i]);
  }
}

void launch_coarse_inv(hipStream_t s, const CoarseOut* outs_dev, int nouts, int n_t, const int* ylist_dev, const float2* Y, int y_frames,
                       const float2* tw16, const float2* twab) {
  if (nouts <= 0 || n_t <= 0) return;
  using PL = R16Plan<CM>;
  const size_t lds = (size_t)(PL::T2 + PL::T3 + 2 * CPAD) * sizeof(float2);
  if (hipFuncSetAttribute((const void*)coarse_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    launch_fail("cannot raise the dynamic LDS limit of the coarse inverse transform");
  for (int r0 = 0; r0 < nouts; r0 += 32768)
    hipLaunchKernelGGL(coarse_inv_kernel, dim3(n_t, std::min(32768, nouts - r0)), dim3(512), lds, s, outs_dev + r0, ylist_dev, Y, y_frames,
                       tw16, twab);
}

// =====================================================================================================================
//  history: the last `hist_len` samples of [old history | this chunk's input] become the next chunk's history
// =====================================================================================================================
__global__ __launch_bounds__(256) void coarse_hist_kernel(const CoarseHistJob* __restrict jobs) {
  const CoarseHistJob J = jobs[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < J.hist_len; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sidx = i + J.n;   // position in the concatenation [old (hist_len) | in (n)]
    float v = 0.f;
    if (sidx < J.hist_len) v = J.old_hist ? ldg1(J.old_hist + sidx) : 0.f;
    else if (J.in) v = ldg1(J.in + (sidx - J.hist_len));
    stg1(J.new_hist + i, v);
  }
}
void launch_coarse_hist(hipStream_t s, const CoarseHistJob* jobs_dev, int njobs, int64_t max_len) {
  if (njobs <= 0 || max_len <= 0) return;
  const int gx = (int)std::min<int64_t>((max_len + 255) / 256, 256);
  for (int j0 = 0; j0 < njobs; j0 += 32768)
    hipLaunchKernelGGL(coarse_hist_kernel, dim3(gx, std::min(32768, njobs - j0)), dim3(256), 0, s, jobs_dev + j0);
}

}  // namespace ga

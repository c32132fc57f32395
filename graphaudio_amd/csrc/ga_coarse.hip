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
// Where the algebra allows it the sum moves further forward: terms that share one impulse response are summed as spectra before ONE
// multiply (coarse_sum_kernel), and a whole fused group on one impulse response is summed in the TIME domain in front of one set
// of transforms (coarse_premix_kernel, option coarse_premix).  Terms with impulse responses of their own are transformed and
// multiplied one by one (coarse_mac_kernel).  State between chunks is the last P' x CB INPUT samples per row (time domain):
// overlap-save has no output-side state, so nodes keep nothing that depends on who consumes them.
//
// HBM traffic per 10 s step of config 3 (1024 voices): input 2.0 GB (+0.3 history) in, X 4.4 GB out, 4.4 GB in, Y 0.25 GB
// out + in, bus out -- ~11 GB against ~35 GB for formulation C.
//
// Real transform of N = 16,384 points through TWO complex transforms of 4,096 points (the radix-16 register / LDS
// transform of ga_fft16.hpp) on z_a[m] = x[4m] + i x[4m+1], z_b[m] = x[4m+2] + i x[4m+3], and one combine pass
// (tools/proto/coarse_math.py is the numpy statement of the index arithmetic).  Spectra are "packed": 8192 complex values per
// frame, bin 0 holding the two real bins (X[0], X[8192]), stored with the two halves of the spectrum interleaved (position
// 2 j = bin j, 2 j + 1 = bin 4096 + j: what the combine pass produces per thread is then two aligned 16-byte words).  All power-of-two scale factors (the 1/2 of the even/odd
// splits, 1/4096 of the inverse) are folded into the impulse-response spectra: exact.
#include "ga_kernels.hpp"
#include "ga_fft16.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <string>
#include <type_traits>
#include <utility>
#include <cstdio>
#include <cstdlib>

namespace ga {

// The dynamic-LDS limit of a kernel is raised once per (kernel, device), not on every launch: to the CU's 160 KB where the
// runtime takes that, else to what the launch at hand needs (then again per launch).
struct LdsLimit {
  std::atomic<uint64_t> done{0};
  void raise(const void* kern, size_t need, const char* what) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_relaxed) & bit) return;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess) {
      done.fetch_or(bit);
      return;
    }
    (void)hipGetLastError();
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(need, 65536)) != hipSuccess) launch_fail(what);
  }
};

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
#if defined(GA_EXPERIMENTS) && defined(GA_MAC_SCALAR_FMA)   // (measurement: four v_fma_f32 instead of two v_pk_fma_f32 -- same values)
  float re, im, re2, im2;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(re) : "v"(a.x), "v"(b.x), "v"(acc.x));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(im) : "v"(a.x), "v"(b.y), "v"(acc.y));
  asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(re2) : "v"(a.y), "v"(b.y), "v"(re));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(im2) : "v"(a.y), "v"(b.x), "v"(im));
  return f2{re2, im2};
#endif
  f2 t, r;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(t) : "v"(a), "v"(b), "v"(acc));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}

// 8-byte LDS reads that STAY `ds_read_b64`.  The compiler merges two reads off one base register into `ds_read2st64_b64`, which
// the LDS serves as two accesses of four 16-lane groups each -- 8 LDS cycles for 1 KB, where two `ds_read_b64` take 2 + 2
// (MI355X_MICROARCH.md, LDS table: 128 vs 256 B/clk/CU).  The sliding sweeps below read 13 values per 64 packed fmas and wave;
// merged, that is ~70 % of the LDS pipe's cycles next to a VALU that wants every issue slot.  Reads issued here are invisible
// to the compiler's wait-count pass: lds_wait_all() must follow before the first use.
template <int OFF>
__device__ __forceinline__ f2 lds_rd_b64(unsigned addr) {
  f2 r;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int STRIDE, int... Q>
__device__ __forceinline__ void lds_rd_seq(f2* dst, unsigned addr, std::integer_sequence<int, Q...>) {
  ((dst[Q] = lds_rd_b64<Q * STRIDE>(addr)), ...);
}
template <int N>
__device__ __forceinline__ void lds_pin(f2* v) {   // values of asm reads: usable only after the wait (volatile asms keep their order)
#pragma unroll
  for (int i = 0; i < N; i++) asm volatile("" : "+v"(v[i]));
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const f2* p) {
  return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) f2*)p;
}

// =====================================================================================================================
//  forward: one workgroup of 256 threads per window, THREE workgroups per CU (44 KB of LDS and <= 168 VGPRs each: while one waits
//  for its loads / stores or at a barrier the others transform).  Thread t owns the points j = t + 256 m (m < 16) of BOTH complex transforms
//      z_a[j] = x[4j] + i x[4j+1],   z_b[j] = x[4j+2] + i x[4j+3]
//  so it fetches x[4j .. 4j+3] as ONE 16-byte word and no sample is fetched twice; the transforms run one after the other
//  through the same LDS buffer.  The combine pass needs Z[k] and Z[4096 - k]: a thread keeps Z[t + 256 m] for m < 8 in
//  registers and gets the mirrored values (held by thread 256 - t as its m >= 8 half) through the buffer.  The workgroup walks a
//  run of consecutive windows of one row: a window's second half is the next window's first half, and the half after that is
//  in flight while the current window is transformed.
// =====================================================================================================================
// the 8 words x[4j .. 4j+3] = (z_a[j], z_b[j]), j = t + 256 q, this thread owns in half-window `hw` of the row
__device__ __forceinline__ void coarse_issue_half(const CoarseXRow& R, int hw, int t, v4f (&raw)[8]) {
  const float* src = nullptr;
  int64_t lim = 0;   // samples of this half that exist (the rest is zero)
  if (hw < 0) {
    const int64_t off = (int64_t)R.hist_len + (int64_t)hw * kCoarseBlock;
    if (R.hist && off >= 0) {
      src = R.hist + off;
      lim = kCoarseBlock;
    }
  } else if (R.in) {
    const int64_t off = (int64_t)hw * kCoarseBlock;
    src = R.in + off;
    lim = std::min<int64_t>(kCoarseBlock, R.nvalid - off);
  }
  const bool aligned = ((uintptr_t)src & 15) == 0;
#pragma unroll
  for (int q = 0; q < 8; q++) {
    const int o = 4 * (t + 256 * q);
    raw[q] = v4f{0.f, 0.f, 0.f, 0.f};
    if (src && o < lim) {   // lim is a multiple of 4 (chunks are whole 128-frame blocks, histories whole coarse blocks)
      if (aligned) raw[q] = ldg4(src + o);
      else raw[q] = v4f{ldg1(src + o), ldg1(src + o + 1), ldg1(src + o + 2), ldg1(src + o + 3)};
    }
  }
}

constexpr int kHandOverWgs = 8;    // workgroups that carry one row of a hand-over across PCIe
constexpr int kFwdPark = 4;         // values per thread parked in LDS across transform b (8 KB: what three workgroups per CU leave)
constexpr int kFwdTw3 = 4 * 256;   // power twiddles of the last pass: W_4096^(j m), m = 1, 2, 4, 8 (ga_fft16.hpp, PW)
// EXPERIMENT: the phase-removal timing variants of tools/coarse_exp.sh (run-time flags `exp_`); the product kernel has none
template <bool EXPERIMENT>
__global__ __launch_bounds__(256, 3) void coarse_fwd_kernel(const CoarseXRow* __restrict rows, int run, float2* __restrict X,
                                                            const float2* __restrict twg, const float2* __restrict twab, int exp_,
                                                            const CoarseHandOver* __restrict handover, int n_handover) {
  if ((int)blockIdx.y < n_handover) {
    // the previous chunk's bus on its way to the caller's page-locked rows (Context::pendingHandOver): a few long-lived workgroups
    // per row write over PCIe while the rest of the launch transforms (chunks without a pre-mix launch to ride in)
    const int nwg = min((int)gridDim.x, kHandOverWgs);
    if ((int)blockIdx.x >= nwg) return;
    const CoarseHandOver H = handover[blockIdx.y];
    const GA_GLOBAL v4f* src = (const GA_GLOBAL v4f*)H.src;
    GA_GLOBAL v4f* dst = (GA_GLOBAL v4f*)H.dst;
    const int64_t nw = H.n / 4, step = (int64_t)nwg * 512;
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < nw; i += step) {
      const int64_t i2 = i + 256;
      const v4f a = src[i], b = i2 < nw ? src[i2] : a;
      dst[i] = a;
      if (i2 < nw) dst[i2] = b;
    }
    return;
  }
  const int exp = EXPERIMENT ? exp_ : 0;
  using PL = R16Plan<CM>;
  extern __shared__ f2 clds[];
  f2* tw2 = clds;
  f2* tw3 = clds + PL::T2;
  f2* buf = clds + PL::T2 + kFwdTw3;
  f2* park = buf + CPAD;   // kFwdPark mirrored values of transform a per thread wait here while transform b runs (registers)
  const int t_ = threadIdx.x;
  for (int i = t_; i < PL::T2 + kFwdTw3; i += 256) clds[i] = f2{twg[i].x, twg[i].y};
  const CoarseXRow R = rows[blockIdx.y - n_handover];
  const int w0 = blockIdx.x * run;
  const int w1 = min(R.n_frames, w0 + run);
  if (w0 >= w1) return;   // (uniform)
  // combine-pass twiddles of this thread's bins k = t + 256 m: b = W_16384^k = W_16384^t W_64^m (and W_8192^k = b^2); only
  // W_16384^t lives in registers, the eight W_64^m are literals
  // (fetched per window, next to the prefetch: held across the loop it is spilled to scratch and reloaded from there instead)
  const bool zero2 = (R.flags & 1) != 0;   // impulse-response partitions: [h_p | 0]
  const float scale = R.scale;
  // Every input sample is fetched ONCE per run, as part of a 16-byte word that holds one point of each transform: a window's
  // second half (held as 8 words) is the next window's first half, and the half after that is requested between the two
  // transforms of the current window (not earlier: its 32 registers would be live during transform a as well)
  const int wave64 = __builtin_amdgcn_readfirstlane(t_ & ~63);
  v4f first[8], second[8];
  coarse_issue_half(R, R.u0 + w0 - 1, t_, first);
  if (zero2) {
#pragma unroll
    for (int q = 0; q < 8; q++) second[q] = v4f{0.f, 0.f, 0.f, 0.f};
  } else {
    coarse_issue_half(R, R.u0 + w0, t_, second);
  }
  __syncthreads();
  for (int w = w0; w < w1; w++) {
    // the thread index is made opaque per window: otherwise every loop-invariant address of the unrolled body (LDS slots,
    // store offsets of the 32 output bins, prefetch offsets) is hoisted out of the loop and held in ~100 registers
    // -- and it is rebuilt from the lane and wave numbers, or the allocator spills threadIdx.x itself and reloads it per window
    int t = wave64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(t));
    const int u = R.u0 + w;
    const bool more = w + 1 < w1;
    f2 own[16];
    // the samples of the window's second half that belong to the next chunk's history leave from here (every input half is
    // the second half of exactly one window of the row)
    if (R.carry && u >= 0 && (int64_t)(u + 1) * kCoarseBlock > R.carry_from) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int64_t sidx = (int64_t)u * kCoarseBlock + 4 * (t + 256 * q);
        if (sidx >= R.carry_from && sidx < R.nvalid) stg4(R.carry + (sidx - R.carry_from), second[q]);
      }
    }
    // ---- transform a ----
#pragma unroll
    for (int m = 0; m < 8; m++) {
      own[m] = f2{first[m].x * scale, first[m].y * scale};
      own[8 + m] = f2{second[m].x * scale, second[m].y * scale};
    }
    // each transform sits in its own basic block (a branch on an opaque, always-true scalar): as straight-line code the
    // scheduler drags values of the neighbouring phases over the transform and the allocator spills them to scratch
    int go = 1;
    asm volatile("" : "+s"(go));
    if (go && !(exp & 4)) fft16_own<CM, true>(own, buf, tw2, tw3, t);
    f2 za[8], pa[8];
    const f2 za8 = own[8];   // Z_a[2048] (thread 0)
    __syncthreads();          // the last pass has read the buffer
    // LDS slots of the exchange: cpad(t + 256 m) = cpad(t) + 264 m and cpad(2048 - t - 256 m) = cpad(2048 - t) - 264 m (256 m is
    // a multiple of 32), written as one base each plus immediates -- the compiler does not see it through the shifts
    const int xw = cpad(t), xr = cpad(2048 - t) - 264 * 7;
#pragma unroll
    for (int m = 0; m < 8; m++) {
      za[m] = own[m];
      buf[xw + 264 * m] = own[8 + m];   // Z[j], j >= 2048, at slot j - 2048
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = t + 256 * m;             // mirror 4096 - k >= 2049 sits at slot 2048 - k  (k = 0: Z[4096] = Z[0], own)
      pa[m] = k == 0 ? za[0] : buf[xr + 264 * (7 - m)];
      if (m >= 8 - kFwdPark) park[t + 256 * (m - (8 - kFwdPark))] = pa[m];
    }
    __syncthreads();          // mirrors fetched before transform b writes the buffer
    // ---- transform b ----
#pragma unroll
    for (int m = 0; m < 8; m++) {
      own[m] = f2{first[m].z * scale, first[m].w * scale};
      own[8 + m] = f2{second[m].z * scale, second[m].w * scale};
    }
    asm volatile("" : "+s"(go));
    if (go && !(exp & 4)) fft16_own<CM, true>(own, buf, tw2, tw3, t);
    f2 pb[8];
    const f2 zb8 = own[8];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) buf[xw + 264 * m] = own[8 + m];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const int k = t + 256 * m;
      pb[m] = k == 0 ? own[0] : buf[xr + 264 * (7 - m)];
    }
    // the window moves on: `second` becomes the first half, the half after it is requested now (not earlier: its 32 registers
    // would be live during the transforms) and arrives behind the combine pass and the next window's first transform
    const float2 tb0g = twab[2049 + t];
    const f2 tb0 = f2{tb0g.x, tb0g.y};
    if (more) {
      if (zero2) {
        if (!(exp & 8)) coarse_issue_half(R, u, t, first);
      } else {
#pragma unroll
        for (int q = 0; q < 8; q++) first[q] = second[q];
        if (!(exp & 8)) coarse_issue_half(R, u + 1, t, second);
      }
    }
    // ---- combine: W = A + a B (8192-point complex spectrum of x[2n] + i x[2n+1]), then the real-input split.
    //      All values carry a factor 2 (the 1/2 of the split is folded into the impulse-response scale).
    float2* __restrict F = X + (size_t)(R.frame0 + w) * kCoarseBins;
    auto quad = [&](int k, f2 Ak, f2 Am, f2 Bk, f2 Bm, f2 b) {
      const f2 a = cmulc(b, b);
      const f2 aB = cmulc(a, Bk), caBm = cmulc(cj(a), Bm);
      const f2 Wk = Ak + aB, Wp = Ak - aB;          // W[k], W[4096 + k]
      const f2 Wm = Am - caBm, Wn = Am + caBm;      // W[4096 - k], W[8192 - k]
      const f2 fe = Wk + cj(Wn), bfo = cmulc(b, mul_mi(Wk - cj(Wn)));
      const f2 Xk = fe + bfo, Xn = cj(fe - bfo);
      const f2 fe2 = Wm + cj(Wp), wfo2 = cmulc(mul_mi(cj(b)), mul_mi(Wm - cj(Wp)));
      const f2 Xm = fe2 + wfo2, Xp = cj(fe2 - wfo2);
      // frame layout: position 2 j = bin j, position 2 j + 1 = bin 4096 + j (j < 4096): the two pairs of a quad are two
      // aligned 16-byte words, consecutive in k across the lanes (1 KB per wave and store instruction)
      if (exp & 1) {
        if (Xk.x == 12345.f && Xn.x == 54321.f && Xm.x == 999.f && Xp.y == 777.f) F[0] = make_float2(0.f, 0.f);
      } else if (k == 0) {
        stg4(F, v4f{Xk.x, Xn.x, Xm.x, Xm.y});                  // bin 0 packed (X[0], X[8192]) ; bin 4096
      } else if (k == CM / 2) {
        stg4(F + 2 * k, v4f{Xk.x, Xk.y, Xn.x, Xn.y});          // bins 2048, 6144
      } else {
        stg4(F + 2 * k, v4f{Xk.x, Xk.y, Xp.x, Xp.y});          // bins k, 4096 + k
        stg4(F + 2 * (CM - k), v4f{Xm.x, Xm.y, Xn.x, Xn.y});   // bins 4096 - k, 8192 - k
      }
    };
    if (!(exp & 2)) {
      constexpr float w64[8][2] = {{1.f, 0.f},
                                   {0.99518472667219688624f, -0.09801714032956060199f},
                                   {0.98078528040323044913f, -0.19509032201612826785f},
                                   {0.95694033573220886494f, -0.29028467725446236764f},
                                   {0.92387953251128675613f, -0.38268343236508977173f},
                                   {0.88192126434835502971f, -0.47139673682599764856f},
                                   {0.83146961230254523708f, -0.55557023301960222474f},
                                   {0.77301045336273696081f, -0.63439328416364549822f}};
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const f2 b = m == 0 ? tb0 : cmulc(tb0, f2{w64[m][0], w64[m][1]});
        quad(t + 256 * m, za[m], m >= 8 - kFwdPark ? park[t + 256 * (m - (8 - kFwdPark))] : pa[m], own[m], pb[m], b);
        __builtin_amdgcn_sched_barrier(0);   // one quad at a time: the unrolled pass would otherwise keep all eight in flight
      }
      if (t == 0) quad(CM / 2, za8, za8, zb8, zb8, f2{twab[2049 + CM / 2].x, twab[2049 + CM / 2].y});
    }
    __syncthreads();   // mirrors of transform b fetched before the next window's transform a writes the buffer
  }
}

const char* launch_coarse_fwd(hipStream_t s, const CoarseXRow* rows_dev, int nrows, int max_frames, int run, float2* X, const float2* tw16,
                              const float2* twab, const CoarseHandOver* handover_dev, int n_handover) {
  if (nrows <= 0 || max_frames <= 0) return "";
  using PL = R16Plan<CM>;
  const size_t lds = (size_t)(PL::T2 + kFwdTw3 + CPAD + kFwdPark * 256) * sizeof(float2);
  static const int exp = (expenv("GA_COARSE_EXP") ? atoi(expenv("GA_COARSE_EXP")) : 0) & 15;   // timing experiments only
  auto kern = exp ? coarse_fwd_kernel<true> : coarse_fwd_kernel<false>;
  static LdsLimit lim[2];
  lim[exp ? 1 : 0].raise((const void*)kern, lds, "cannot raise the dynamic LDS limit of the coarse forward transform");
  run = std::max(run, 1);
  for (int r0 = 0; r0 < nrows; r0 += 32768) {
    const int nh = r0 == 0 ? n_handover : 0;   // (the hand-over rows come first in the grid: their workgroups start at once)
    dim3 grid((max_frames + run - 1) / run, std::min(32768, nrows - r0) + nh);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, rows_dev + r0, run, X, tw16, twab, exp, handover_dev, nh);
  }
  return "coarse_fwd_kernel";
}

// =====================================================================================================================
//  multiply-accumulate: workgroup = (job, tile of 64 bins); lane = bin, wave w = a quarter of the job's coarse blocks.
//  The X frames of one term (all windows the job needs, 512 bytes each) are staged in LDS -- double buffered, the next term's
//  frames are fetched while this one is accumulated -- and every wave slides over them:
//      acc[t][c] += X[t - p] . H_c[p]          p < P', c < CW columns, t in the wave's range
//  with the accumulators of ALL terms of the job in registers: the sum over the job's voices costs no memory traffic.
// =====================================================================================================================
#ifndef GA_MAC_WAVES
#define GA_MAC_WAVES 8
#endif
#ifndef GA_MAC_TW
#define GA_MAC_TW 9      // coarse blocks per wave of a 1- or 2-column job (8 waves x 9 = kCoarseJobBlocks(2))
#endif
#ifndef GA_MAC_PB2
#define GA_MAC_PB2 4      // partition block used for 2-column jobs whose partition count is a multiple of 4 (2: 1.54 ms, 4: 1.49 ms at 1024 private stereo IRs)
#endif
constexpr int kMacWaves16 = 12;                               // the 16-column instance: 12 waves (three per SIMD: 168 registers each) ...
constexpr int kMacTW16 = (kCoarseJobBlocks(16) + kMacWaves16 - 1) / kMacWaves16;   // ... x 3 blocks x 16 columns = 48 complex accumulators per lane
constexpr int kMacWaves = GA_MAC_WAVES;      // waves per workgroup: each takes 1/kMacWaves of the job's coarse blocks
// Workgroups per CU the register budget is set for.  Jobs of 1 or 2 columns span up to 64 coarse blocks: 2 x 71 frames x 512 B of
// double-buffered staging (72.7 KB) + the spectra leave room for ONE workgroup per CU whatever the partition count (measured:
// hipOccupancyMaxActiveBlocksPerMultiprocessor = 1 at 89 KB), so those instances take the 256 registers two waves per SIMD may
// have (the operand prefetch of the sweep needs 130); 4-column jobs span 32 blocks (72 KB at 8 partitions): two per CU, 128.
constexpr int kMacWavesPerSimd(int cw) { return kMacWaves * (cw >= 4 ? 2 : 1) / 4; }
// (launches whose terms all share one impulse response take coarse_sum_kernel below instead)
template <int CW, int TW, int PB, int WV>
__global__ __launch_bounds__(64 * WV, WV == 8 ? kMacWavesPerSimd(CW) : WV / 4) void coarse_mac_kernel(const CoarseJob* __restrict jobs, const CoarseTerm* __restrict terms,
                                                                    const float2* __restrict X, float2* __restrict Y, int y_frames, int NFA,
                                                                    int exp) {
  extern __shared__ f2 mlds[];   // (ALL of the kernel's LDS is this one array: a second object beside a direct-to-LDS target costs a vmcnt(0) per read)
  const CoarseJob J = jobs[blockIdx.y];
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = J.P, nT = J.n_t;
  const int NF = nT + P - 1;                 // frames of a term that are needed: frame fr = window J.t0 - (P - 1) + fr
  const int fr_lo = max(0, J.u_lo - (J.t0 - (P - 1))), fr_hi = min(NF - 1, J.u_hi - (J.t0 - (P - 1)));   // those that exist (the others are zero)
  // NFA = frames an LDS buffer holds (launch-wide: the unrolled sweep of the last active wave may read past NF, never stored)
  f2* xs0 = mlds;
  f2* xs1 = mlds + (size_t)NFA * 64;
  f2* hs0 = mlds + (size_t)2 * NFA * 64;
  f2* hs1 = hs0 + (size_t)P * CW * 64;       // only when the job's terms have different impulse responses
  const int twr = (nT + WV - 1) / WV;   // coarse blocks per wave (<= TW)
  const int t0w = wv * twr;
  const bool special = tile == 0;            // bin 0 of tile 0 is the packed pair of real bins
  const bool lane0 = special && lane == 0;
  const size_t binoff = (size_t)tile * 64;

  // staging of one term: NF x 32 float4 of X (+ P x CW x 32 float4 of H), global memory -> LDS directly (global_load_lds_dwordx4:
  // no staging registers, no ds_write pass).  One wave instruction moves two consecutive 512-byte rows: the LDS image is
  // linear in the lane (destination = wave-uniform base + 16 lane), the source address is per lane.
  typedef __attribute__((address_space(3))) void* lds_t;
  constexpr int XR = ((WV * TW + kCoarseMaxP) * 32 + (64 * WV) - 1) / (64 * WV);  // instructions per wave that cover NF <= WV TW + P - 1 frames
  auto issue_x = [&](const CoarseTerm& T, f2* xs) {
#pragma unroll
    for (int r = 0; r < XR; r++) {
      const int fr0 = ((64 * WV) / 32) * r + 2 * wv;   // (uniform)
      const int fr = fr0 + (lane >> 5), of = lane & 31;
      if (fr >= fr_lo && fr <= fr_hi)
        __builtin_amdgcn_global_load_lds(gptr(X + (size_t)(T.frame0 + J.t0 + fr) * kCoarseBins + binoff + 2 * of), (lds_t)(xs + fr0 * 64), 16, 0, 0);
    }
  };
  auto issue_h = [&](const CoarseTerm& T, f2* hs) {   // P x CW rows of 512 bytes
    for (int pc0 = 2 * wv; pc0 < P * CW; pc0 += 2 * WV) {
      const int pc = pc0 + (lane >> 5), of = lane & 31;
      // the two rows' base addresses are wave-uniform (scalar loads of the term's descriptor: a per-lane descriptor load is a
      // vector-memory instruction whose result the address needs at once -- a vmcnt(0) in the middle of the staging)
      const int pA = pc0 / CW, cA = pc0 % CW, pB = (pc0 + 1) / CW, cB = (pc0 + 1) % CW;
      const float2* rowA = T.h[cA] + (size_t)pA * kCoarseBins;
      const float2* rowB = pc0 + 1 < P * CW ? T.h[cB] + (size_t)pB * kCoarseBins : rowA;
      if (pc < P * CW)
        __builtin_amdgcn_global_load_lds(gptr(((lane >> 5) ? rowB : rowA) + binoff + 2 * of), (lds_t)(hs + pc0 * 64), 16, 0, 0);
    }
  };

  f2 acc[TW][CW];
#pragma unroll
  for (int tt = 0; tt < TW; tt++)
#pragma unroll
    for (int c = 0; c < CW; c++) acc[tt][c] = f2{0.f, 0.f};

  const CoarseTerm* __restrict T = terms + J.term0;
  // windows that do not exist are zero rows in both buffers (no term ever writes them)
  for (int idx = tid; idx < NF * 32; idx += (64 * WV)) {
    const int fr = idx >> 5;
    if (fr < fr_lo || fr > fr_hi) {
      *reinterpret_cast<v4f*>(xs0 + fr * 64 + 2 * (idx & 31)) = v4f{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<v4f*>(xs1 + fr * 64 + 2 * (idx & 31)) = v4f{0.f, 0.f, 0.f, 0.f};
    }
  }
  issue_x(T[0], xs0);
  issue_h(T[0], hs0);
  __syncthreads();   // (waits for the workgroup's direct-to-LDS loads: the barrier's fence includes vmcnt(0))
  for (int i = 0; i < J.n_terms; i++) {
    const bool more = i + 1 < J.n_terms;
    f2* xs = (i & 1) ? xs1 : xs0;
    f2* hs = (J.shared_h || !(i & 1)) ? hs0 : hs1;
    // the next term lands in the other buffers while this one is accumulated (their last readers passed the barrier below)
    if (more && !(exp & 2)) {
      if (!J.shared_h) issue_h(T[i + 1], (i & 1) ? hs0 : hs1);   // (first: anything its addressing waits for must not wait for the X loads)
      issue_x(T[i + 1], (i & 1) ? xs0 : xs1);
    }
    if (t0w < nT && !(exp & 1)) {
      // blocks of PB partitions (P is a multiple of PB): their spectra sit in registers, and the frames b - (PB - 1) .. b +
      // TW - 1 of the sweep (b = frame of (tt = 0, j = 0); (tt, j) uses frame b + tt - j) are requested up front: one LDS
      // latency per block of partitions, then the block is pure VALU with j outermost, so that consecutive fmas belong to
      // different accumulators.  `SP` (tile 0: lane 0 holds two real bins) is a compile-time copy of the loop: as a run-time
      // flag inside the unrolled body the compiler evaluates both products for every tile.
      auto sweep = [&](auto sp) {
        constexpr bool SP = decltype(sp)::value;
        constexpr int PBX = PB, NX = TW + PBX - 1, NH = PBX * CW;
        constexpr bool EARLY_H = CW <= 4;   // (16 columns: the spectra are read where they are used, or 32 more registers spill)
        auto fma1 = [&](int tt, int c, f2 x, f2 h) {
          if constexpr (!SP) {
            acc[tt][c] = cfmap(x, h, acc[tt][c]);
          } else {
            const f2 gen = cfmap(x, h, acc[tt][c]);
            const f2 pk = __builtin_elementwise_fma(x, h, acc[tt][c]);   // two real bins side by side
            acc[tt][c] = lane0 ? pk : gen;
          }
        };
        if constexpr (EARLY_H) {
          // One LDS latency per TERM, not per block of partitions: the operands of block b + 1 (NH spectra values, NX frames: rows
          // 512 bytes apart, plain ds_read_b64 -- see lds_rd_b64) are requested before the fmas of block b are issued and waited
          // for after them.  With the one workgroup per CU the double-buffered staging leaves room for (two waves per SIMD) a
          // wave that waits for the LDS at every block idles the VALU a third of the time.
          f2 hA[NH], xA[NX], hB[NH], xB[NX];
          auto fetch = [&](int pb, f2* h, f2* x) {
            lds_rd_seq<512>(h, lds_addr(hs + pb * CW * 64 + lane), std::make_integer_sequence<int, NH>{});
            lds_rd_seq<512>(x, lds_addr(xs + (t0w + (P - 1) - pb - (PBX - 1)) * 64 + lane), std::make_integer_sequence<int, NX>{});   // (pb + PBX - 1 <= P - 1: inside the buffer)
          };
          auto arrive = [&](f2* h, f2* x) {
            lds_wait_all();
            lds_pin<NH>(h);
            lds_pin<NX>(x);
          };
          auto fmas = [&](const f2* h, const f2* x) {
#pragma unroll
            for (int j = 0; j < PBX; j++)
#pragma unroll
              for (int tt = 0; tt < TW; tt++)
#pragma unroll
                for (int c = 0; c < CW; c++) fma1(tt, c, x[tt - j + (PBX - 1)], h[j * CW + c]);
          };
          // partition blocks whose windows u = J.t0 + t0w + tt - p all lie outside [u_lo, u_hi] multiply zero rows: skipped.  (The
          // blocks behind the chunk's end of a group that carries its tail, the first blocks of a signal without history.)
          const int tw0 = J.t0 + t0w;
          const int pbLo = max(0, (tw0 - (PBX - 1) - J.u_hi + PBX - 1) / PBX * PBX), pbHi = min(P, tw0 + TW - J.u_lo);
          if (pbLo < pbHi) {
            fetch(pbLo, hA, xA);
            for (int pb = pbLo; pb < pbHi; pb += 2 * PBX) {
              arrive(hA, xA);
              const bool second = pb + PBX < pbHi;
              if (second) fetch(pb + PBX, hB, xB);
              fmas(hA, xA);
              if (!second) break;
              arrive(hB, xB);
              if (pb + 2 * PBX < pbHi) fetch(pb + 2 * PBX, hA, xA);
              fmas(hB, xB);
            }
          }
        } else {
          for (int pb = 0; pb < P; pb += PBX) {
            const f2* __restrict xb = xs + (t0w + (P - 1) - pb - (PBX - 1)) * 64 + lane;
            f2 xv[NX];
#pragma unroll
            for (int q = 0; q < NX; q++) xv[q] = xb[q * 64];
#pragma unroll
            for (int j = 0; j < PBX; j++)
#pragma unroll
              for (int c = 0; c < CW; c++) {
                const f2 h = hs[((pb + j) * CW + c) * 64 + lane];
#pragma unroll
                for (int tt = 0; tt < TW; tt++) fma1(tt, c, xv[tt - j + (PBX - 1)], h);
              }
          }
        }
      };
      // (the 16-column instance leaves the packed pair of real bins to coarse_mac_bin0_kernel: a second copy of a sweep with 48
      // accumulators does not fit the registers)
      if (special && CW != 16) sweep(std::true_type{});
      else sweep(std::false_type{});
    }
    if (!(exp & 8)) __syncthreads();
  }
#pragma unroll
  for (int tt = 0; tt < TW; tt++) {
    const int t = t0w + tt;
    if (tt < twr && t < nT && !(exp & 4)) {
#pragma unroll
      for (int c = 0; c < CW; c++)
        stg2(Y + ((size_t)(J.yrow0 + c) * y_frames + J.t0 + t) * kCoarseBins + binoff + lane, v2f{acc[tt][c].x, acc[tt][c].y});
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
//  One impulse response for all terms of a job (many voices through one room: ConvolverNodes that hold the same buffer and
//  feed the same sum): the spectral multiply commutes with the sum over the terms,
//        sum_v sum_p X_v[t - p] H[p]  =  sum_p ( sum_v X_v[t - p] ) H[p],
//  so the job first adds up its terms' X frames -- a pure streaming reduction, every thread owns a few 16-byte words of the
//  (frames x 64 bins) tile and adds them straight from global memory into registers, four terms in flight, no LDS, no barrier
//  -- and multiplies ONCE at the end (sum tile -> LDS, the usual sliding sweep).  The multiply-accumulate work per job drops
//  from (terms x P x blocks) to (P x blocks) products per bin and the kernel runs at the rate it can read X.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef GA_SUM_AHEAD
#define GA_SUM_AHEAD 4
#endif
constexpr int kSumWaves = 8, kSumThreads = 64 * kSumWaves, kSumAhead = GA_SUM_AHEAD;   // terms in flight per thread
template <int CW>
__global__ __launch_bounds__(kSumThreads, 4) void coarse_sum_kernel(const CoarseJob* __restrict jobs, const CoarseTerm* __restrict terms,
                                                                    const float2* __restrict X, float2* __restrict Y, int y_frames, int NFA, int exp) {
  constexpr int TW = kCoarseSumJobBlocks(CW) / kSumWaves;                                   // coarse blocks per wave (9, 9, 5)
  constexpr int XR = ((kCoarseSumJobBlocks(CW) + kCoarseMaxP) * 32 + kSumThreads - 1) / kSumThreads;   // 16-byte words per thread (6, 6, 4)
  extern __shared__ f2 mlds[];   // (all of the kernel's LDS is this one array)
  const CoarseJob J = jobs[blockIdx.y];
  const int tile = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = J.P, nT = J.n_t, NF = nT + P - 1, nterms = J.n_terms;   // frame fr = window J.t0 - (P - 1) + fr
  const int fr_lo = max(0, J.u_lo - (J.t0 - (P - 1))), fr_hi = min(NF - 1, J.u_hi - (J.t0 - (P - 1)));   // the windows that exist
  const int nvf = fr_hi - fr_lo + 1;                                         // (<= 0: nothing to add up)
  const int nw = nvf > 0 ? (nvf * 32 + kSumThreads - 1) / kSumThreads : 0;   // 16-byte words per thread that cover them (uniform)
  const int twr = (nT + kSumWaves - 1) / kSumWaves, t0w = wv * twr;
  const bool special = tile == 0, lane0 = special && lane == 0;
  const size_t binoff = (size_t)tile * 64;
  f2* S = mlds;                           // NFA frames x 64 bins: the summed spectra
  f2* hs = mlds + (size_t)NFA * 64;       // P x CW rows of 64 bins
  typedef __attribute__((address_space(3))) void* lds_t;
  const CoarseTerm* __restrict T = terms + J.term0;
  // the impulse response goes straight to LDS and is first needed after the reduction
  for (int pc0 = 2 * wv; pc0 < P * CW; pc0 += 2 * kSumWaves) {
    const int pc = pc0 + (lane >> 5), of = lane & 31;
    if (pc < P * CW)
      __builtin_amdgcn_global_load_lds(gptr(T[0].h[pc % CW] + (size_t)(pc / CW) * kCoarseBins + binoff + 2 * of), (lds_t)(hs + pc0 * 64), 16, 0, 0);
  }
  static_assert(kCoarseJobTerms <= 64, "one lane per term");
  const int f0v = lane < nterms ? T[lane].frame0 : 0;   // the terms' first frames, read with v_readlane below
  // word r of this thread: existing frame fr_lo + (tid + kSumThreads r) / 32 (clamped: the surplus words of the last round are
  // loaded and never stored), bins 2 of, 2 of + 1
  int64_t woff[XR];
#pragma unroll
  for (int r = 0; r < XR; r++) {
    const int idx = tid + kSumThreads * r;
    const int fr = fr_lo + min(idx >> 5, max(nvf - 1, 0)), of = idx & 31;
    woff[r] = (int64_t)(J.t0 + fr) * kCoarseBins + (int64_t)binoff + 2 * of;
  }
  v4f acc[XR];
#pragma unroll
  for (int r = 0; r < XR; r++) acc[r] = v4f{0.f, 0.f, 0.f, 0.f};
  // the reduction, compiled once per word count (a run-time count inside the unrolled body costs registers and spills)
  auto stream = [&](auto nwc) {
    constexpr int NW = decltype(nwc)::value;
    constexpr int AH = NW >= 6 ? 3 : kSumAhead;   // (6 words x 4 terms in flight would not fit the 128 registers of 4 waves per SIMD)
    int i = 0;
    for (; i + AH <= nterms; i += AH) {
      v4f ld[AH][NW];
#pragma unroll
      for (int u = 0; u < AH; u++) {
        const float2* __restrict Xt = X + (int64_t)__builtin_amdgcn_readlane(f0v, i + u) * kCoarseBins;
#pragma unroll
        for (int r = 0; r < NW; r++) ld[u][r] = ldg4(Xt + woff[r]);
      }
#pragma unroll
      for (int u = 0; u < AH; u++)
#pragma unroll
        for (int r = 0; r < NW; r++) acc[r] += ld[u][r];
    }
    for (; i < nterms; i++) {
      const float2* __restrict Xt = X + (int64_t)__builtin_amdgcn_readlane(f0v, i) * kCoarseBins;
#pragma unroll
      for (int r = 0; r < NW; r++) acc[r] += ldg4(Xt + woff[r]);
    }
  };
  if (!(exp & 2)) {
    static_assert(XR <= 6, "one case per word count below");
    switch (nw) {   // (uniform)
      case 1: stream(std::integral_constant<int, 1>{}); break;
      case 2: stream(std::integral_constant<int, 2>{}); break;
      case 3: stream(std::integral_constant<int, XR >= 3 ? 3 : XR>{}); break;
      case 4: stream(std::integral_constant<int, XR >= 4 ? 4 : XR>{}); break;
      case 5: stream(std::integral_constant<int, XR >= 5 ? 5 : XR>{}); break;
      case 6: stream(std::integral_constant<int, XR >= 6 ? 6 : XR>{}); break;
      default: break;
    }
  }
  // the sum tile: zero rows for the windows that do not exist, the sums for the others
  for (int idx = tid; idx < NF * 32; idx += kSumThreads) {
    const int fr = idx >> 5;
    if (fr < fr_lo || fr > fr_hi) *reinterpret_cast<v4f*>(S + fr * 64 + 2 * (idx & 31)) = v4f{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int r = 0; r < XR; r++) {
    const int idx = tid + kSumThreads * r;
    const int fv = idx >> 5, of = idx & 31;
    if (fv < nvf) *reinterpret_cast<v4f*>(S + (fr_lo + fv) * 64 + 2 * of) = acc[r];
  }
  __syncthreads();   // (its fence also waits for the impulse response's direct-to-LDS loads)
  // ---- one sweep: y[tt][c] = sum_p S[t0w + tt - p] H_c[p]; tile 0, lane 0 (two real bins) multiplies element-wise ----
  f2 y[TW][CW], yS[TW][CW];
#pragma unroll
  for (int tt = 0; tt < TW; tt++)
#pragma unroll
    for (int c = 0; c < CW; c++) y[tt][c] = yS[tt][c] = f2{0.f, 0.f};
  if (t0w < nT && !(exp & 1)) {
    auto sweep = [&](auto sp) {
      constexpr bool SP = decltype(sp)::value;
      const f2* __restrict xb = S + (t0w + (P - 1)) * 64 + lane;   // frame of (tt = 0, p = 0)
      f2 xv[TW];
#pragma unroll
      for (int q = 1; q < TW; q++) xv[q] = xb[q * 64];             // frames of tt = 1 .. TW - 1 at p = 0
      for (int p = 0; p < P; p++) {
        // window slides one frame back per partition: xv[tt] = frame (tt - p)
#pragma unroll
        for (int q = TW - 1; q > 0; q--) xv[q] = p == 0 ? xv[q] : xv[q - 1];
        xv[0] = xb[-p * 64];
        f2 h[CW];
#pragma unroll
        for (int c = 0; c < CW; c++) h[c] = hs[(p * CW + c) * 64 + lane];
#pragma unroll
        for (int tt = 0; tt < TW; tt++)
#pragma unroll
          for (int c = 0; c < CW; c++) {
            y[tt][c] = cfmap(xv[tt], h[c], y[tt][c]);
            if constexpr (SP) yS[tt][c] = __builtin_elementwise_fma(xv[tt], h[c], yS[tt][c]);
          }
      }
    };
    if (special) sweep(std::true_type{});
    else sweep(std::false_type{});
  }
#pragma unroll
  for (int tt = 0; tt < TW; tt++) {
    const int t = t0w + tt;
    if (tt < twr && t < nT && !(exp & 4)) {
#pragma unroll
      for (int c = 0; c < CW; c++) {
        const f2 a = lane0 ? yS[tt][c] : y[tt][c];
        stg2(Y + ((size_t)(J.yrow0 + c) * y_frames + J.t0 + t) * kCoarseBins + binoff + lane, v2f{a.x, a.y});
      }
    }
  }
}

template <int CW>
static const char* launch_coarse_sum(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                              int y_frames, int max_t, int maxP) {
  static const int exp = expenv("GA_COARSE_EXP") ? atoi(expenv("GA_COARSE_EXP")) : 0;   // timing experiments only
  constexpr int TW = kCoarseSumJobBlocks(CW) / kSumWaves;
  int NFA = 0;   // frames the sweep of the last active wave touches
  for (int nt = 1; nt <= max_t; nt++) {
    const int twr = (nt + kSumWaves - 1) / kSumWaves, wl = (nt + twr - 1) / twr - 1;
    NFA = std::max(NFA, wl * twr + TW + maxP - 1);
  }
  const size_t lds = ((size_t)NFA * 64 + (size_t)maxP * CW * 64) * sizeof(float2);
  if (lds > 160 * 1024) launch_fail("coarse multiply-accumulate: staging does not fit the LDS");
  static LdsLimit lim;
  lim.raise((const void*)coarse_sum_kernel<CW>, lds, "cannot raise the dynamic LDS limit of the coarse multiply-accumulate");
  if (expenv("GA_COARSE_EXP")) {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, coarse_sum_kernel<CW>, kSumThreads, lds);
    fprintf(stderr, "[coarse_sum<%d>] lds %zu B, NFA %d, occupancy %d workgroups/CU, %d jobs\n", CW, lds, NFA, occ, njobs);
  }
  for (int j0 = 0; j0 < njobs; j0 += 32768)
    hipLaunchKernelGGL((coarse_sum_kernel<CW>), dim3(kCoarseBins / 64, std::min(32768, njobs - j0)), dim3(kSumThreads), lds, s, jobs_dev + j0, terms_dev, X,
                       Y, y_frames, NFA, exp >> 4);
  return CW == 1 ? "coarse_sum_kernel<1>" : (CW == 2 ? "coarse_sum_kernel<2>" : "coarse_sum_kernel<4>");
}

// bin 0 of the packed spectra holds the two REAL bins (X[0], X[8192]): their products are element-wise.  The 16-column instance of
// the general kernel computes a complex product there like everywhere else; this kernel writes the right value over it:
// thread = (output block, column) of a job, Y[c][t][0] = sum over terms, partitions of X[t - p][0] (.) H_c[p][0].
__global__ __launch_bounds__(256) void coarse_mac_bin0_kernel(const CoarseJob* __restrict jobs, const CoarseTerm* __restrict terms,
                                                             const float2* __restrict X, float2* __restrict Y, int y_frames) {
  const CoarseJob J = jobs[blockIdx.x];
  const CoarseTerm* __restrict T = terms + J.term0;
  const int P = J.P;   // (<= kCoarseMaxP)
  const int idx = blockIdx.y * 256 + threadIdx.x;
  if (idx >= J.n_t * 16) return;
  const int t = idx >> 4, c = idx & 15;
  // partitions whose window exists: u = J.t0 + t - p in [u_lo, u_hi]
  const int p_lo = max(0, J.t0 + t - J.u_hi), p_hi = min(P - 1, J.t0 + t - J.u_lo);
  f2 acc = f2{0.f, 0.f};
  for (int i = 0; i < J.n_terms; i++) {   // (all loads of a term are independent: the partitions' loop is unrolled by four)
    const float2* __restrict h = T[i].h[c];
    const float2* __restrict x = X + (size_t)(T[i].frame0 + J.t0 + t + (P - 1)) * kCoarseBins;   // window u = J.t0 + t - p sits p frames back
    int p = p_lo;
    for (; p + 3 <= p_hi; p += 4) {
      v2f xv[4], hv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        xv[q] = ldg2(x - (size_t)(p + q) * kCoarseBins);
        hv[q] = ldg2(h + (size_t)(p + q) * kCoarseBins);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) acc = __builtin_elementwise_fma(f2{xv[q].x, xv[q].y}, f2{hv[q].x, hv[q].y}, acc);
    }
    for (; p <= p_hi; p++) {
      const v2f xv = ldg2(x - (size_t)p * kCoarseBins), hv = ldg2(h + (size_t)p * kCoarseBins);
      acc = __builtin_elementwise_fma(f2{xv.x, xv.y}, f2{hv.x, hv.y}, acc);
    }
  }
  stg2(Y + ((size_t)(J.yrow0 + c) * y_frames + J.t0 + t) * kCoarseBins, v2f{acc.x, acc.y});
}

// =====================================================================================================================
//  16 columns on the MATRIX CORES.  For one bin the partition sum of a job is a small complex GEMM,
//        Y[c][t] += sum over terms, p < 4 of  H_c[p] . X[t - p]          c < 16 columns, t < 32 blocks,
//  i.e. per (bin, term) a [16 x 4] . [4 x 32] product: two v_mfma_f32_16x16x4_f32 tiles, four real MFMAs each (re += ar br,
//  re += (-ai) bi, im += ar bi, im += ai br).  A = the term's spectra (lane l holds H_{c = l & 15}[p = l >> 4]), B = its frames in
//  Toeplitz order (lane l holds X[t0 + (l & 15) - (l >> 4)]), D: lane l, register r = Y[c = 4 (l >> 4) + r][t = l & 15].  The
//  operand reuse that the register-tiled VALU kernel has to buy with accumulators (3 blocks x 16 columns per lane, one LDS read
//  per 6 packed fmas) is done by the hardware here: 3 LDS reads per 8 MFMAs (8192 real multiply-adds).
//  Workgroup = (job, 64 bins), 8 waves, wave w owns bins 8 w .. 8 w + 7 and keeps their 2 x 2 x 4 accumulators over the job's
//  terms.  A term's frames and spectra are staged global -> registers -> LDS with a row pitch of 65 complex values (a fragment
//  read walks ROWS at a fixed bin: at the natural pitch of 64 every lane would hit the same bank), double buffered.  The results
//  leave through the LDS as whole 512-byte rows.  Bin 0 of tile 0 (the packed pair of real bins) is put right by
//  coarse_mac_bin0_kernel afterwards.
// =====================================================================================================================
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kMfThreads = 512, kMfFrames = kCoarseJobBlocks(16) + 3, kMfRows = kMfFrames + 64, kMfPitch = 65;   // (pitch of the result tile)
constexpr size_t kMfLdsBytes = std::max<size_t>((size_t)2 * kMfRows * 64, (size_t)256 * kMfPitch) * sizeof(float2);
// LDS image of a term: 35 frame rows + 64 spectra rows of 64 bins, 512 bytes apart, written by global_load_lds (no staging registers,
// no ds_write pass after the MFMAs).  A fragment read walks ROWS at a fixed bin -- at this pitch every lane on one bank -- and a
// direct-to-LDS load cannot pad rows (its image is linear in the lane), so the swizzle is applied on the GLOBAL side: slot s of row r
// receives the row's 16-byte word s ^ (r & 31).  32 consecutive rows at one bin then fall on 16 different bank pairs: 2-way, on 3
// reads per 8 MFMAs.
__device__ __forceinline__ int mf_slot(int row, int bin) { return row * 64 + ((((bin >> 1) ^ row) & 31) << 1) + (bin & 1); }
__global__ __launch_bounds__(kMfThreads, 2) void coarse_mfma16_kernel(const CoarseJob* __restrict jobs, const CoarseTerm* __restrict terms,
                                                                     const float2* __restrict X, float2* __restrict Y, int y_frames) {
  extern __shared__ f2 mlds[];
  typedef __attribute__((address_space(3))) void* lds_t;
  const CoarseJob J = jobs[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = J.P, nT = J.n_t;
  const size_t binoff = (size_t)blockIdx.x * 64;
  const CoarseTerm* __restrict T = terms + J.term0;
  // rows 0 .. 34 = frames (row f = window J.t0 + f - 3), rows 35 .. 98 = spectra (p = r >> 4, c = r & 15); rows that do not exist
  // (windows outside the signal, partitions >= P) are zero in both buffers and never loaded
  auto live = [&](int row) {
    if (row < kMfFrames) {
      const int u = J.t0 + row - 3;
      return row < nT + 3 && u >= J.u_lo && u <= J.u_hi;
    }
    return row < kMfRows && ((row - kMfFrames) >> 4) < P;
  };
  f2* st0 = mlds;
  f2* st1 = mlds + (size_t)kMfRows * 64;
  for (int idx = tid; idx < kMfRows * 32; idx += kMfThreads) {
    const int row = idx >> 5;
    if (!live(row)) {
      *reinterpret_cast<v4f*>(st0 + row * 64 + 2 * (idx & 31)) = v4f{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<v4f*>(st1 + row * 64 + 2 * (idx & 31)) = v4f{0.f, 0.f, 0.f, 0.f};
    }
  }
  auto issue = [&](int i, f2* stage) {   // one wave instruction = two consecutive rows (1 KB, linear in the lane)
    const int frame0 = T[i].frame0;                 // (scalar loads: the term index is uniform)
    const float2* __restrict h0 = T[i].h[0];
#pragma unroll
    for (int k = 0; k < (kMfRows + 15) / 16; k++) {
      const int r0 = 16 * k + 2 * wv, row = r0 + (lane >> 5);
      const int sw = ((lane & 31) ^ row) & 31;      // the word of the row this lane fetches into slot (lane & 31)
      const float2* src = row < kMfFrames ? X + (size_t)(frame0 + J.t0 + row - 3 + (P - 1)) * kCoarseBins
                                          : h0 + ((size_t)((row - kMfFrames) & 15) * P + ((row - kMfFrames) >> 4)) * kCoarseBins;
      if (live(row)) __builtin_amdgcn_global_load_lds(gptr(src + binoff + 2 * sw), (lds_t)(stage + r0 * 64), 16, 0, 0);
    }
  };
  f4v accr[8][2], acci[8][2];
#pragma unroll
  for (int b = 0; b < 8; b++)
#pragma unroll
    for (int q = 0; q < 2; q++) accr[b][q] = acci[b][q] = f4v{0.f, 0.f, 0.f, 0.f};
  issue(0, st0);
  __syncthreads();   // (waits for the workgroup's direct-to-LDS loads: the barrier's fence includes vmcnt(0))
  const int arow = kMfFrames + lane;                              // A fragment: spectra row (p = lane >> 4, c = lane & 15)
  const int brow = (lane & 15) - (lane >> 4) + 3;                 // B fragment of tile 0: frame of t = lane & 15, p = lane >> 4
  for (int i = 0; i < J.n_terms; i++) {
    const f2* __restrict st = (i & 1) ? st1 : st0;
    if (i + 1 < J.n_terms) issue(i + 1, (i & 1) ? st0 : st1);   // lands during the MFMAs below (its last readers passed the barrier)
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const int bin = 8 * wv + b;
      const f2 a = st[mf_slot(arow, bin)];
      const f2 x0 = st[mf_slot(brow, bin)], x1 = st[mf_slot(brow + 16, bin)];
      const float nai = -a.y;
      accr[b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, accr[b][0], 0, 0, 0);
      acci[b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.y, acci[b][0], 0, 0, 0);
      accr[b][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, accr[b][1], 0, 0, 0);
      acci[b][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.y, acci[b][1], 0, 0, 0);
      accr[b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(nai, x0.y, accr[b][0], 0, 0, 0);
      acci[b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.x, acci[b][0], 0, 0, 0);
      accr[b][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(nai, x1.y, accr[b][1], 0, 0, 0);
      acci[b][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.x, acci[b][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // results: per tile of 16 blocks through the LDS ([row = c 16 + t][bin], pitch 65), then whole 512-byte rows to Y
#pragma unroll
  for (int q = 0; q < 2; q++) {
#pragma unroll
    for (int b = 0; b < 8; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) mlds[((4 * (lane >> 4) + r) * 16 + (lane & 15)) * kMfPitch + 8 * wv + b] = f2{accr[b][q][r], acci[b][q][r]};
    __syncthreads();
    for (int idx = tid; idx < 256 * 32; idx += kMfThreads) {
      const int row = idx >> 5, of = idx & 31, c = row >> 4, t = 16 * q + (row & 15);
      if (t < nT) {
        const f2 v0 = mlds[row * kMfPitch + 2 * of], v1 = mlds[row * kMfPitch + 2 * of + 1];
        stg4(Y + ((size_t)(J.yrow0 + c) * y_frames + J.t0 + t) * kCoarseBins + binoff + 2 * of, v4f{v0.x, v0.y, v1.x, v1.y});
      }
    }
    __syncthreads();
  }
}

template <int CW, int TW, int PB, int WV = kMacWaves>
static const char* launch_coarse_mac_t(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                                int y_frames, int max_t, int maxP, bool any_private) {
  static const int exp = expenv("GA_COARSE_EXP") ? atoi(expenv("GA_COARSE_EXP")) : 0;   // timing experiments only
  // frames the sweep of the last active wave touches: t0w + TW + P - 1 with t0w = (active waves - 1) * ceil(n_t / waves)
  int NFA = 0;
  for (int nt = 1; nt <= max_t; nt++) {
    const int twr = (nt + WV - 1) / WV, wl = (nt + twr - 1) / twr - 1;
    NFA = std::max(NFA, wl * twr + TW + maxP - 1);
  }
  const size_t lds = ((size_t)2 * NFA * 64 + (size_t)(any_private ? 2 : 1) * maxP * CW * 64) * sizeof(float2);
  if (lds > 160 * 1024) launch_fail("coarse multiply-accumulate: staging does not fit the LDS");
  static LdsLimit lim;
  lim.raise((const void*)coarse_mac_kernel<CW, TW, PB, WV>, lds, "cannot raise the dynamic LDS limit of the coarse multiply-accumulate");
  if (expenv("GA_COARSE_EXP")) {
    int occ = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, coarse_mac_kernel<CW, TW, PB, WV>, 64 * WV, lds);
    fprintf(stderr, "[coarse_mac<%d,%d,%d>] lds %zu B, NFA %d, occupancy %d workgroups/CU, %d jobs\n", CW, TW, PB, lds, NFA, occ, njobs);
  }
  for (int j0 = 0; j0 < njobs; j0 += 32768)
    hipLaunchKernelGGL((coarse_mac_kernel<CW, TW, PB, WV>), dim3(kCoarseBins / 64, std::min(32768, njobs - j0)), dim3(64 * WV), lds, s,
                       jobs_dev + j0, terms_dev, X, Y, y_frames, NFA, exp >> 4);
  static const std::string name = "coarse_mac_kernel<" + std::to_string(CW) + "," + std::to_string(TW) + "," + std::to_string(PB) + "," + std::to_string(WV) + ">";
  return name.c_str();
}
// all jobs of one launch have the same column count `cw` (1, 2 or 4), at most `max_t` coarse blocks (<= kCoarseJobBlocks(cw))
// and partition counts that are multiples of `pb` (1, 2 or 4: the register block of the sweep)
template <int CW, int TWL>   // TWL = coarse blocks per wave: accumulators TWL x CW complex values per lane
static const char* launch_coarse_mac_tw(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                                 int y_frames, int max_t, int maxP, bool any_private, int pb) {
  if (pb >= 4 && (CW == 1 || (CW == 2 && GA_MAC_PB2 == 4))) return launch_coarse_mac_t<CW, TWL, 4>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
  if (pb >= 2) return launch_coarse_mac_t<CW, TWL, 2>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
  return launch_coarse_mac_t<CW, TWL, 1>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
}
template <int CW>
static const char* launch_coarse_mac_cw(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                                 int y_frames, int max_t, int maxP, bool any_private, int pb) {
  if (!any_private) return launch_coarse_sum<CW>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP);   // (the planner cut these jobs for it)
  if (max_t <= 2 * kMacWaves) return launch_coarse_mac_t<CW, 2, 1>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private);
  if constexpr (CW <= 2) {
    // 9 blocks per wave only where a job needs them (a 10 s chunk + its carried tail = 67 blocks): the sweep computes all TWL blocks
    if (max_t <= 8 * kMacWaves) return launch_coarse_mac_tw<CW, 8>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private, pb);
    return launch_coarse_mac_tw<CW, GA_MAC_TW>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private, pb);
  } else {
    return launch_coarse_mac_tw<CW, GA_MAC_TW4>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private, pb);
  }
}
const char* launch_coarse_mac(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                       int y_frames, int cw, int max_t, int maxP, bool any_private, int pb, bool matrix_cores) {
  if (njobs <= 0) return "";
  if (max_t > (any_private ? kCoarseJobBlocks(cw) : kCoarseSumJobBlocks(cw))) launch_fail("coarse multiply-accumulate: too many coarse blocks in a job");
  if (const char* e = expenv("GA_COARSE_PB")) pb = std::min(pb, std::max(1, atoi(e)));   // measurements only
  if (pb != 1 && pb != 2 && pb != 4 && pb != 8 && pb != 16) launch_fail("coarse multiply-accumulate: unsupported partition block");
  if (cw == 1) return launch_coarse_mac_cw<1>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private, pb);
  if (cw == 2) return launch_coarse_mac_cw<2>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private, pb);
  if (cw == 4) return launch_coarse_mac_cw<4>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, any_private, pb);
  // all 16 columns of a source in one job (BASELINE.json configs[4]): 12 waves x 3 blocks x 16 columns, one workgroup per CU --
  // the source's X frames are staged ONCE for its 16 columns (4-column jobs: four times) and its impulse-response spectra once per
  // 36 blocks
  if (cw == 16 && any_private) {
    const char* name;
    if (matrix_cores && maxP <= 4 && max_t <= kCoarseJobBlocks(16)) {
      static LdsLimit lim;
      lim.raise((const void*)coarse_mfma16_kernel, kMfLdsBytes, "cannot raise the dynamic LDS limit of the matrix-core multiply-accumulate");
      for (int j0 = 0; j0 < njobs; j0 += 32768)
        hipLaunchKernelGGL(coarse_mfma16_kernel, dim3(kCoarseBins / 64, std::min(32768, njobs - j0)), dim3(kMfThreads), kMfLdsBytes, s, jobs_dev + j0,
                           terms_dev, X, Y, y_frames);
      name = "coarse_mfma16_kernel";
    } else {
      name = launch_coarse_mac_t<16, kMacTW16, 1, kMacWaves16>(s, jobs_dev, njobs, terms_dev, X, Y, y_frames, max_t, maxP, true);
    }
    hipLaunchKernelGGL(coarse_mac_bin0_kernel, dim3(njobs, (kCoarseJobBlocks(16) * 16 + 255) / 256), dim3(256), 0, s, jobs_dev, terms_dev, X, Y, y_frames);
    return name;
  }
  launch_fail("coarse multiply-accumulate: unsupported column count");
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
  const int64_t s0 = (int64_t)tb * kCoarseBlock;   // first sample of this coarse block
  const int64_t send = O.nvalid + (O.tail_out ? O.tail_len : 0);
  if (s0 >= send || !O.out) return;   // (uniform)
  float* st = reinterpret_cast<float*>(zb0);
  const bool have_y = tb < O.n_y;
  // where the block's samples go: the chunk's output up to nvalid, the carried tail behind it; the previous chunk's tail is added
  auto route = [&](bool zero) {
    __syncthreads();
    for (int i = tid; i < kCoarseBlock / 4; i += 512) {
      const int64_t si = s0 + 4 * i;
      if (si >= send) continue;
      v4f v = zero ? v4f{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const v4f*>(st + 4 * i);
      if (O.tail_in && si < O.tail_len) v += ldg4(O.tail_in + si);
      float* __restrict dst = si < O.nvalid ? O.out + si : O.tail_out + (si - O.nvalid);   // (nvalid, tail_len: multiples of 4)
      if ((((uintptr_t)dst) & 15) == 0) stg4(dst, v);
      else { stg1(dst, v.x); stg1(dst + 1, v.y); stg1(dst + 2, v.z); stg1(dst + 3, v.w); }
    }
  };
  if (!have_y) {   // behind the last transformed block: only the older tail moves on
    route(true);
    return;
  }
  for (int i = tid; i < PL::T2 + PL::T3; i += 512) clds[i] = f2{twg[i].x, twg[i].y};
  // ---- frequency-domain mix: word i = tid + 512 r of a frame holds bins (i, 4096 + i)  (layout: coarse_fwd_kernel) ----
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
      const int b0 = tid + 512 * r;
      zb0[cslot(b0)] = f2{sum[r].x, sum[r].y};
      zb0[cslot(CM + b0)] = f2{sum[r].z, sum[r].w};
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
#pragma unroll
  for (int m = 8; m < 16; m++) {
    const int o = 4 * (t + 256 * (m - 8)) + g;
    st[o] = own[m].y;       // swapped back
    st[o + 2] = own[m].x;
  }
  route(false);
}

// n_blocks: coarse blocks of the longest output, carried tail included (outputs return early behind their own end)
const char* launch_coarse_inv(hipStream_t s, const CoarseOut* outs_dev, int nouts, int n_blocks, const int* ylist_dev, const float2* Y, int y_frames,
                              const float2* tw16, const float2* twab) {
  const int n_t = n_blocks;
  if (nouts <= 0 || n_t <= 0) return "";
  using PL = R16Plan<CM>;
  const size_t lds = (size_t)(PL::T2 + PL::T3 + 2 * CPAD) * sizeof(float2);
  static LdsLimit lim;
  lim.raise((const void*)coarse_inv_kernel, lds, "cannot raise the dynamic LDS limit of the coarse inverse transform");
  for (int r0 = 0; r0 < nouts; r0 += 32768)
    hipLaunchKernelGGL(coarse_inv_kernel, dim3(n_t, std::min(32768, nouts - r0)), dim3(512), lds, s, outs_dev + r0, ylist_dev, Y, y_frames,
                       tw16, twab);
  return "coarse_inv_kernel";
}

// =====================================================================================================================
//  history: the last `hist_len` samples of [old history | this chunk's input] become the next chunk's history
// =====================================================================================================================
__global__ __launch_bounds__(256) void coarse_hist_kernel(const CoarseHistJob* __restrict jobs) {
  const CoarseHistJob J = jobs[blockIdx.y];
  // hist_len and n are multiples of 4 (whole coarse blocks / whole 128-frame blocks): 16-byte words never straddle the seam
  const bool wide = (((uintptr_t)J.old_hist | (uintptr_t)J.in | (uintptr_t)J.new_hist) & 15) == 0;
  if (wide) {
    for (int64_t i = 4 * ((int64_t)blockIdx.x * blockDim.x + threadIdx.x); i < J.hist_len; i += 4 * (int64_t)gridDim.x * blockDim.x) {
      const int64_t sidx = i + J.n;   // position in the concatenation [old (hist_len) | in (n)]
      v4f v = v4f{0.f, 0.f, 0.f, 0.f};
      if (sidx < J.hist_len) {
        if (J.old_hist) v = ldg4(J.old_hist + sidx);
      } else if (J.in) {
        v = ldg4(J.in + (sidx - J.hist_len));
      }
      stg4(J.new_hist + i, v);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < J.hist_len; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t sidx = i + J.n;
    float v = 0.f;
    if (sidx < J.hist_len) v = J.old_hist ? ldg1(J.old_hist + sidx) : 0.f;
    else if (J.in) v = ldg1(J.in + (sidx - J.hist_len));
    stg1(J.new_hist + i, v);
  }
}
// =====================================================================================================================
//  pre-mix: out[f] = sum over the terms of in_t[f].  Workgroup = 256 consecutive frames (one 16-byte word = 4 frames per lane) x
//  all terms of the job; its four waves take a quarter of the terms each (four times the waves to keep words in flight, also
//  for short chunks), stream them from global memory kPremixAhead words at a time per lane, and wave 0 adds the four partial
//  sums.  Term descriptors come 64 at a time, one per lane, and are handed out with v_readlane: the addresses are scalar base +
//  32-bit lane offset.  All sums are Kahan-compensated -- four VALU operations per word in a kernel that waits for HBM -- so
//  the result is the correctly rounded sum to within an ulp whatever the number of terms.  While a wave holds a member's
//  samples it writes the member's input history of the next chunk (`carry`).
//  Job flags (planner): bit 0 = every `in` is 16-byte aligned, bit 1 = some term has a carry, bit 2 = hand-over of a finished bus
//  to page-locked host rows (one term, plain copy; Context::pendingHandOver).
// =====================================================================================================================
#ifndef GA_PREMIX_AHEAD
#define GA_PREMIX_AHEAD 8
#endif
#ifndef GA_PREMIX_WAVES
#define GA_PREMIX_WAVES 4
#endif
#ifndef GA_PREMIX_NT
#define GA_PREMIX_NT 0      // 1: non-temporal loads of the members' samples ; 2: and non-temporal stores of their histories
#endif
constexpr int kPremixAhead = GA_PREMIX_AHEAD, kPremixWaves = GA_PREMIX_WAVES;
#ifndef GA_PREMIX_WORDS
#define GA_PREMIX_WORDS 1   // 16-byte words per lane and term: a wave reads GA_PREMIX_WORDS KB in a row from a member before it moves on
#endif
#ifndef GA_PREMIX_COPY_WGS
#define GA_PREMIX_COPY_WGS 8
#endif
constexpr int kPremixCopyWgs = GA_PREMIX_COPY_WGS;   // workgroups that carry one row of a hand-over across PCIe
constexpr int kPremixWords = GA_PREMIX_WORDS, kPremixTile = 256 * kPremixWords;   // frames per workgroup
struct Kahan4 {
  v4f s, c;
  __device__ __forceinline__ void add(v4f x) {
#ifdef GA_PREMIX_PLAIN   // (measurement: plain float sum)
    s += x;
    return;
#endif
    const v4f y = x - c, t = s + y;
    c = (t - s) - y;
    s = t;
  }
};
template <bool ALIGNED, bool CARRY>
__device__ __forceinline__ void premix_stream(const PremixJob& J, const PremixTerm* __restrict terms, int t0, int t1, int lane,
                                              const uint32_t (&off)[kPremixWords], const bool (&keep)[kPremixWords],
                                              const uint32_t (&coff)[kPremixWords], Kahan4 (&acc)[kPremixWords]) {
  typedef const GA_GLOBAL char* gcp;
  typedef GA_GLOBAL char* gp;
  auto fetch = [&](gcp in, uint32_t o) -> v4f {       // `in` is wave-uniform; words behind the end re-read word 0 and are dropped
    if (ALIGNED) return GA_PREMIX_NT ? __builtin_nontemporal_load((const GA_GLOBAL v4f*)(in + o)) : *(const GA_GLOBAL v4f*)(in + o);
    const GA_GLOBAL float* p = (const GA_GLOBAL float*)(in + o);
    return v4f{p[0], p[1], p[2], p[3]};
  };
  auto put = [&](gp car, uint32_t o, v4f x) {
    if (GA_PREMIX_NT >= 2) __builtin_nontemporal_store(x, (GA_GLOBAL v4f*)(car + o));
    else *(GA_GLOBAL v4f*)(car + o) = x;
  };
  for (int base = t0; base < t1; base += 64) {
    const int nb = min(64, t1 - base);
    uint64_t pin = 0, pcar = 0;
    if (lane < nb) {
      const GA_GLOBAL PremixTerm* T = (const GA_GLOBAL PremixTerm*)terms + (J.term0 + base + lane);
      pin = (uint64_t)(uintptr_t)T->in;
      if (CARRY) pcar = (uint64_t)(uintptr_t)T->carry;
    }
    const int pin_lo = (int)(uint32_t)pin, pin_hi = (int)(uint32_t)(pin >> 32), pc_lo = (int)(uint32_t)pcar, pc_hi = (int)(uint32_t)(pcar >> 32);
    auto in_of = [&](int j) {
      return (gcp)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(pin_hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane(pin_lo, j));
    };
    auto carry_of = [&](int j) {
      return (gp)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(pc_hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane(pc_lo, j));
    };
    int j = 0;
    for (; j + kPremixAhead <= nb; j += kPremixAhead) {
      v4f x[kPremixAhead][kPremixWords];
#pragma unroll
      for (int u = 0; u < kPremixAhead; u++) {
        gcp in = in_of(j + u);
#pragma unroll
        for (int w = 0; w < kPremixWords; w++) x[u][w] = fetch(in, off[w]);
      }
#pragma unroll
      for (int u = 0; u < kPremixAhead; u++) {
        if (CARRY) {
          gp car = carry_of(j + u);
#pragma unroll
          for (int w = 0; w < kPremixWords; w++)
            if (car && keep[w]) put(car, coff[w], x[u][w]);
        }
#pragma unroll
        for (int w = 0; w < kPremixWords; w++) acc[w].add(x[u][w]);
      }
    }
    for (; j < nb; j++) {
      gcp in = in_of(j);
      gp car = CARRY ? carry_of(j) : nullptr;
#pragma unroll
      for (int w = 0; w < kPremixWords; w++) {
        const v4f x = fetch(in, off[w]);
        if (CARRY && car && keep[w]) put(car, coff[w], x);
        acc[w].add(x);
      }
    }
  }
}
__global__ __launch_bounds__(64 * kPremixWaves) void coarse_premix_kernel(const PremixJob* __restrict jobs, const PremixTerm* __restrict terms) {
  __shared__ v4f part[kPremixWaves - 1][2][kPremixWords][64];
  const PremixJob J = jobs[blockIdx.y];
  if (J.flags & 4) {
    // hand-over job: out is page-locked HOST memory.  A few workgroups walk the whole row -- they sit on a handful of wave slots
    // for as long as PCIe needs (~ 70 us for a 10 s stereo bus) while every other workgroup of the launch streams HBM.  (As
    // ordinary jobs -- one short-lived workgroup per 256 frames -- the copies filled the whole chip until PCIe had drained them.)
    if (blockIdx.x >= (unsigned)kPremixCopyWgs) return;
    const GA_GLOBAL v4f* src = (const GA_GLOBAL v4f*)terms[J.term0].in;
    GA_GLOBAL v4f* dst = (GA_GLOBAL v4f*)J.out;
    const int64_t nw = J.n / 4, step = (int64_t)kPremixCopyWgs * blockDim.x * 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x * 2 + threadIdx.x; i < nw; i += step) {
      const int64_t i2 = i + blockDim.x;
      const v4f a = src[i], b = i2 < nw ? src[i2] : a;
      dst[i] = a;
      if (i2 < nw) dst[i2] = b;
    }
    return;
  }
  const int64_t tile0 = (int64_t)blockIdx.x * kPremixTile;
  if (tile0 >= J.n) return;   // (uniform)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int64_t f[kPremixWords];
  bool live[kPremixWords], keep[kPremixWords];
  uint32_t off[kPremixWords], coff[kPremixWords];   // byte offsets (chunks and histories are far below 4 GB)
  Kahan4 acc[kPremixWords];
#pragma unroll
  for (int w = 0; w < kPremixWords; w++) {
    f[w] = tile0 + (int64_t)(w * 64 + lane) * 4;
    live[w] = f[w] < J.n;
    keep[w] = live[w] && f[w] >= J.carry_from;
    off[w] = live[w] ? (uint32_t)(f[w] * 4) : 0u;
    coff[w] = (uint32_t)((f[w] - J.carry_from) * 4);
    acc[w] = Kahan4{v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
  }
  // this wave's share of the terms: a multiple of kPremixAhead each
  const int per = ((J.nterms + kPremixWaves - 1) / kPremixWaves + kPremixAhead - 1) / kPremixAhead * kPremixAhead;
  const int t0 = min(J.nterms, wv * per), t1 = min(J.nterms, t0 + per);
  const bool carry_here = (J.flags & 2) && tile0 + kPremixTile > J.carry_from;   // (uniform)
  if (J.flags & 1) {
    if (carry_here) premix_stream<true, true>(J, terms, t0, t1, lane, off, keep, coff, acc);
    else premix_stream<true, false>(J, terms, t0, t1, lane, off, keep, coff, acc);
  } else {
    if (carry_here) premix_stream<false, true>(J, terms, t0, t1, lane, off, keep, coff, acc);
    else premix_stream<false, false>(J, terms, t0, t1, lane, off, keep, coff, acc);
  }
  if (wv > 0) {
#pragma unroll
    for (int w = 0; w < kPremixWords; w++) {
      part[wv - 1][0][w][lane] = acc[w].s;
      part[wv - 1][1][w][lane] = acc[w].c;
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int w = 0; w < kPremixWords; w++) {
#pragma unroll
      for (int q = 0; q < kPremixWaves - 1; q++) {   // (a partial sum is s - c: add both parts)
        acc[w].add(part[q][0][w][lane]);
        acc[w].add(-part[q][1][w][lane]);
      }
      if (live[w]) stg4(J.out + f[w], acc[w].s);
    }
  }
}
const char* launch_coarse_premix(hipStream_t s, const PremixJob* jobs_dev, int njobs, const PremixTerm* terms_dev, int64_t max_n) {
  if (njobs <= 0 || max_n <= 0) return "";
  const int64_t gx = (max_n + kPremixTile - 1) / kPremixTile;
  if (gx > 0x7fffffff || max_n >= ((int64_t)1 << 29)) launch_fail("coarse pre-mix: chunk too long");
  for (int j0 = 0; j0 < njobs; j0 += 32768)
    hipLaunchKernelGGL(coarse_premix_kernel, dim3((unsigned)gx, std::min(32768, njobs - j0)), dim3(64 * kPremixWaves), 0, s, jobs_dev + j0, terms_dev);
  return "coarse_premix_kernel";
}

void launch_coarse_hist(hipStream_t s, const CoarseHistJob* jobs_dev, int njobs, int64_t max_len) {
  if (njobs <= 0 || max_len <= 0) return;
  const int gx = (int)std::min<int64_t>((max_len / 4 + 255) / 256, 64);
  for (int j0 = 0; j0 < njobs; j0 += 32768)
    hipLaunchKernelGGL(coarse_hist_kernel, dim3(gx, std::min(32768, njobs - j0)), dim3(256), 0, s, jobs_dev + j0);
}

__global__ __launch_bounds__(256) void table_upload_kernel(v4f* __restrict dst, const v4f* __restrict src, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) ((GA_GLOBAL v4f*)dst)[i] = ((const GA_GLOBAL v4f*)src)[i];
}
void launch_table_upload(hipStream_t s, void* dst, const void* src_pinned, size_t bytes) {
  const size_t n16 = (bytes + 15) / 16;
  if (!n16) return;
  const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 64);
  hipLaunchKernelGGL(table_upload_kernel, dim3(grid), dim3(256), 0, s, (v4f*)dst, (const v4f*)src_pinned, n16);
}

}  // namespace ga

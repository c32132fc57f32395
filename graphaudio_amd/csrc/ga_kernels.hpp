// ga_kernels.hpp -- launch wrappers of the hand-written gfx950 kernels (ga_kernels.hip).
// Host code (ga_engine.cpp) builds small job tables in pinned memory, uploads them once per chunk and
// calls these wrappers; every wrapper only enqueues work on `stream` (no allocation, no sync).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace ga {

// A launcher that cannot serve its arguments (a planner bug, e.g. an FFT length without a kernel) reports it as an error
// code through the C ABI (GA_ERR_INVALID_OPERATION): the library never aborts the host process.  Defined in ga_engine.cpp.
[[noreturn]] void launch_fail(const char* what);

// Measurement-only environment switches (kernel variant A/B tests, phase-removal experiments of tools/) exist only in builds made
// with -DGA_EXPERIMENTS (tools/build_variant.sh): the shipped library reads no environment variable that changes what it computes.
#ifdef GA_EXPERIMENTS
inline const char* expenv(const char* name) { return getenv(name); }
#else
inline const char* expenv(const char*) { return nullptr; }
#endif

constexpr int kBlock = 128;  // AudioBuffer.FramesPerBlock (AudioBuffer.cs:10)
constexpr int kBins = 129;   // complexCount for fftSize 256 (PartitionedConvolver.cs:40-41)

// ---- convolver pipeline -----------------------------------------------------------------------------
// Spectra planes are split re/im float32 like the reference's _delayReal/_delayImag (PartitionedConvolver.cs:23-24)
// but laid out [bin k][block t][row r] with the row (channel-instance) index fastest, so that one MFMA operand
// fragment is one coalesced LDS/HBM line of rows.
struct ConvPlanes {
  float* xr;       // [kBins][tx][rp]  forward spectra incl. (P-1) history rows at the front
  float* xi;
  float* yr;       // [kBins][ty][rp]  accumulated spectra
  float* yi;
  int tx;          // allocated block extent of x planes
  int ty;          // allocated block extent of y planes
  int rp;          // padded row count (multiple of 128)
};

// one row (convolver channel-instance) of a group: where its time-domain input / output for this chunk live.
struct ConvRowIO {
  const float* in;   // chunk-frame indexed input (in[f], f in [0, nblocks*128)); nullptr = silence
  float* out;        // chunk-frame indexed output slab; nullptr = discard (padding row)
};

// twiddle tables: w128[j] = exp(-2 pi i j / 128) j<64 ; w256[k] = exp(-2 pi i k / 256) k<=128 (double2 = re, im)
struct Twiddles {
  const double2* w128;
  const double2* w256;
};

// forward rfft256 of every (row, block): X planes rows [hist + t] for t in [0, nblocks)
void launch_rfft_fwd(hipStream_t s, const ConvRowIO* rows_dev, int nrows, int nblocks, int hist, ConvPlanes pl, Twiddles tw);
// shared-IR spectral multiply-accumulate (PartitionedConvolver.cs:154-223 over all blocks of the chunk at once)
//   Y[k][t][r] = sum_{p<P} X[k][t + (P-1) - p][r] * H[k][p]
void launch_spectral_mac_shared(hipStream_t s, ConvPlanes pl, const float* hr, const float* hi, int P, int nblocks, int nrows);
// inverse rfft256 + overlap-add (PartitionedConvolver.cs:130-151); the overlap state [rows][128] persists across
// chunks and is double buffered (read overlap_in, write overlap_out) because workgroups run in any order
void launch_irfft_ola(hipStream_t s, const ConvRowIO* rows_dev, int nrows, int nblocks, ConvPlanes pl, const float* overlap_in,
                      float* overlap_out, Twiddles tw);
// history: copy n block-rows of every bin plane  dst[k][dst_t0 + i][:] = src[k][src_t0 + i][:]
void launch_plane_copy(hipStream_t s, float* dst, int dst_t, int dst_t0, const float* src, int src_t, int src_t0, int n, int rp);
// IR spectra extraction  h[c][k][p] = x[k][p][c]  for c < nch (planes produced by launch_rfft_fwd with hist = 0)
void launch_extract_ir(hipStream_t s, float* hr, float* hi, const float* xr, const float* xi, int tx, int rp, int P, int nch);
// out[f] = a[f] + b[f]   (true-stereo pair sum, ConvolverNode.cs:157-164)
void launch_pair_sum(hipStream_t s, float* out, const float* a, const float* b, int64_t n);

// ---- convolver pipeline, formulation B: one input against up to 16 IR channels ("private IR" nodes) ----------------
// Used when few nodes share an impulse response (unique IR per voice, multi-channel HRTF-style IRs): rows of the
// shared-IR GEMM would be empty.  Here one input spectrum sequence X[k][t] (an "x-row") is multiplied with the spectra of
// up to 16 (slot -> IR channel) columns:  Y[col][k][t] = sum_p X[k][t - p] * H[col][k][p]   -- M = time, N = columns, K = taps.
// Planes are [row][bin][block] with the block index fastest.
struct ConvPlanesB {
  float* xr;   // [nx][kBins][tx]   tx = hist + chunk blocks (padded)
  float* xi;
  float* yr;   // [ny][kBins][ty]
  float* yi;
  int tx, ty;
};
struct ConvSetB {          // one MAC problem: x-row `x` against `ncol` columns
  int x;                   // x-row index
  int y0;                  // first y-row index (columns are consecutive y rows)
  int ncol;                // 1..16
  int P;
  const float* hr[16];     // per column: spectra [kBins][P] (re)
  const float* hi[16];
};
struct HistJobB {          // history save / restore of one x-row: kBins runs of `n` floats
  float* dst;
  const float* src;        // nullptr = fill with zeros
  int dst_stride, src_stride, n, pad_;
};
void launch_rfft_fwd_b(hipStream_t s, const ConvRowIO* xrows_dev, int nx, int nblocks, int hist, ConvPlanesB pl, Twiddles tw, bool fp64);
void launch_spectral_mac_b(hipStream_t s, const ConvSetB* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl);
// the same partition sum in the reference's own order and arithmetic (formulation R: every product and sum a separately rounded
// float32 operation, partitions ascending -- PartitionedConvolver.cs:154-223); bit-exact with the reference, ~50x the time
void launch_refmac(hipStream_t s, const ConvSetB* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl);
void launch_irfft_ola_b(hipStream_t s, const ConvRowIO* yrows_dev, int ny, int nblocks, ConvPlanesB pl,
                        const float* const* overlap_in_dev, float* const* overlap_out_dev, Twiddles tw, bool fp64);
void launch_hist_copy_b(hipStream_t s, const HistJobB* jobs_dev, int njobs, int max_n);
// feedback cycles: dst[0..128) = src[0..128) * scale (src == nullptr: zeros) -- the block a producer put out, kept for the consumers
// that pull it while it is being processed in the NEXT block (Nodes/AudioNode.cs:153-156)
struct StaleJob {
  float* dst;
  const float* src;     // nullptr: zeros
  float scale;
  int pad_;
  const float* curve = nullptr;   // non-null: dst = src x curve (a GainNode that was folded into its consumer with its gain curve)
};
void launch_stale_copy(hipStream_t s, const StaleJob* jobs_dev, int njobs);

// ---- convolver pipeline, formulation C: the partition sum as an FFT convolution ALONG THE BLOCK AXIS -----------------
// For one (row, bin) the reference's  acc[t] = sum_{p<P} X[t-p] H[p]  (PartitionedConvolver.cs:154-223) is a length-P
// linear convolution over the block index t.  With all blocks of a chunk known it is evaluated by overlap-save with
// complex FFTs of N2 = 4 * 2^ceil(log2 P) points (float32): forward FFT of the spectra window, product with the
// precomputed N2-point spectrum of the taps, inverse FFT, keep the last N2 - P + 1 outputs.  ~20x fewer flops than the
// direct sum, so the stage becomes HBM-bound.  Same [row][bin][block] planes, sets and histories as formulation B.
struct ConvSetC {
  int x;                   // x-row index
  int y0;                  // first y-row (columns consecutive)
  int ncol;                // 1..16
  int P;
  const float2* hs[16];    // per column: [kBins][N2] spectrum of the taps along the partition axis
};
// spectra of the taps: hs[c][k][0..N2) = FFT_N2( H[c][k][0..P) zero padded )
void launch_tap_spectra(hipStream_t s, float2* hs, const float* hr, const float* hi, int nch, int P, int N2, const float2* tw);
void launch_tconv(hipStream_t s, const ConvSetC* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl, int N2, const float2* tw,
                  int nseg);
// radix-16 variant; tw16 = [15][16] W_256^(kk m) followed by [N2/256 - 1][256] W_N2^(j m) (Context::twiddles16)
// the launch covers the nseg segments that start at block `tbase` (a chunk may be covered by launches of different N2)
void launch_tconv16(hipStream_t s, const ConvSetC* sets_dev, int nsets, int nblocks, int hist, ConvPlanesB pl, int N2, const float2* tw16,
                    int nseg, int tbase);

// ---- convolver pipeline, formulation D: coarse partitions, consumer sum fused in the frequency domain (ga_coarse.hip) ----
constexpr int kCoarseBlock = 8192;   // samples per coarse partition / output block
constexpr int kCoarseMaxP = 16;      // partitions a job can slide over (impulse responses up to 131,072 taps)
constexpr int kCoarseJobTerms = 32;   // terms (signals) whose products one multiply-accumulate job sums in registers
// ... and one reduction job (coarse_sum_kernel: the terms share their impulse response): 8 waves x 9 (x 5) -- a 10 s chunk with its
// carried tail is 59 + 8 output blocks
constexpr int kCoarseSumJobBlocks(int columns) { return columns <= 2 ? 72 : 40; }   // (1, 2, 4 columns)
#ifndef GA_MAC_TW4
#define GA_MAC_TW4 4   // coarse blocks per wave of the 4-column instance of the general multiply-accumulate kernel
#endif
constexpr int kCoarseJobBlocks(int columns) { return columns <= 2 ? 72 : (columns == 16 ? 32 : 8 * GA_MAC_TW4); }   // coarse blocks one multiply-accumulate job covers: 8 waves x 9 (x 4) -- a 10 s chunk (59 blocks) + the 8 blocks of its carried tail in ONE job
// floating-point operations of one 16,384-point real transform as the kernels evaluate it: two complex radix-16 transforms of 4096
// points (3 passes x 256 radix-16 butterflies of ~ 200 flops incl. twiddles, each) + the combine pass (~ 30 flops per bin quad pair)
constexpr double kCoarseTransformFlops = 2.0 * 3.0 * 256.0 * 200.0 + 2048.0 * 60.0;
constexpr int kCoarseBins = 8192;    // packed complex bins of one 16,384-point real spectrum (bin 0 = (X[0], X[8192]))
struct CoarseXRow {        // one transformed signal: a convolver input channel, or an impulse-response channel
  const float* hist;       // the hist_len samples in front of the chunk (nullptr = zeros)
  const float* in;         // chunk-frame indexed input (nullptr = zeros)
  int64_t nvalid;          // samples of `in` (whole 128-frame blocks); zeros beyond
  int frame0;              // index of the row's first frame in the X buffer
  int n_frames;            // windows to transform
  int u0;                  // window index of frame 0: window u covers samples [(u - 1) CB, (u + 1) CB) of the chunk
  int hist_len;            // multiple of CB
  int flags;               // bit 0: second half of every window is zero (impulse-response partitions [h_p | 0])
  float scale;             // applied to every sample (impulse responses: normalisation and transform scale factors)
  // the next chunk's history, written by the forward kernel itself while it holds the samples (no separate copy pass):
  // carry[s - carry_from] = in[s] for carry_from <= s < nvalid.  nullptr: the row's history is copied by coarse_hist_kernel
  // (chunks shorter than the history, silent inputs, unaligned views).
  float* carry;
  int64_t carry_from;      // multiple of 4
};
struct CoarseHandOver {    // a finished bus row on its way to page-locked host memory, carried by a few workgroups of a long launch
  const float* src;        // device staging row
  float* dst;              // device-visible address of the caller's row
  int64_t n;               // frames (a multiple of 4; both pointers 16-byte aligned)
};
struct CoarseTerm {        // one (signal, impulse response) product feeding a job's accumulators
  int frame0;              // frame of window u = -(P - 1) of the signal
  int pad_;
  const float2* h[16];     // per column: packed spectra [P][kCoarseBins] of the column's impulse-response channel
};
struct CoarseJob {         // accumulators Y[yrow0 + c][t] = sum over terms sum_p X[t - p] H_c[p],  c < columns of the launch
  int term0, n_terms;
  int P;                   // coarse partitions (all terms of a job)
  int t0, n_t;             // coarse blocks [t0, t0 + n_t) of the chunk, n_t <= kCoarseJobBlocks(columns)
  int yrow0;
  int shared_h;            // every term uses the same spectra (loaded once)
  int u_lo, u_hi;          // windows of the terms that exist: X[u] = 0 for u outside [u_lo, u_hi] (no input history in front of the
                           // chunk when the outputs carry their tails; nothing transformed behind the chunk's last window)
  int pad_;
};
struct CoarseOut {         // one time-domain output: the sum of `ny` Y rows
  float* out;              // chunk-frame indexed
  int64_t nvalid;          // chunk frames
  int y0, ny;              // ylist[y0 .. y0 + ny)
  // Tail carried from chunk to chunk (what the input so far contributes to samples behind the chunk's end):
  //   v[i] = y[i] (0 for coarse blocks >= n_y) + tail_in[i] (i < tail_len) ;  out[i] = v[i] for i < nvalid ;
  //   tail_out[i - nvalid] = v[i] for nvalid <= i < nvalid + tail_len.   tail_in / tail_out may be nullptr.
  const float* tail_in;
  float* tail_out;
  int64_t tail_len;        // multiple of 4
  int n_y;                 // coarse blocks the Y rows hold for this output
  int pad_;
};
struct CoarseHistJob {
  const float* old_hist;   // nullptr = zeros
  const float* in;         // nullptr = zeros
  float* new_hist;
  int64_t hist_len, n;
};
// Time-domain pre-mix of a fused group whose members all convolve with ONE impulse response (option "coarse_premix"):
//   sum_v (x_v * h) = (sum_v x_v) * h     -- the distributive law once more, in front of the transforms --
// so the group is ONE signal to transform: out[f] = sum over the job's terms of in_t[f] (compensated float32 summation: the
// result is the correctly rounded sum to within an ulp, whatever the number of terms).  While a thread holds a member's
// samples it also writes the member's own input history of the next chunk (carry), exactly as the forward kernel does for
// rows it transforms itself: members keep their private state, so a group can re-form or dissolve between chunks.
struct PremixTerm {
  const float* in;         // chunk-frame indexed (or a history row); never nullptr (silent members are left out of the list)
  float* carry;            // carry[f - carry_from] = in[f] for carry_from <= f < n (16-byte aligned); nullptr = none
};
struct PremixJob {
  float* out;              // n floats, 16-byte aligned
  int term0, nterms;       // (no terms: zeros)
  int64_t n;               // multiple of 4
  int64_t carry_from;      // multiple of 4
  int flags;               // bit 0: every `in` is 16-byte aligned ; bit 1: some term has a carry ; bit 2: hand-over copy to host rows
  int pad_;
};
const char* launch_coarse_premix(hipStream_t s, const PremixJob* jobs_dev, int njobs, const PremixTerm* terms_dev, int64_t max_n);
// (the coarse launchers return the name of the kernel instance they ran: ga_stats.stage_kernel)
// forward: tw16 = Context::twiddles16pw() ; inverse: tw16 = Context::twiddles16(4096) ; twab = [2][2049]: W_8192^k, W_16384^k
const char* launch_coarse_fwd(hipStream_t s, const CoarseXRow* rows_dev, int nrows, int max_frames, int run, float2* X, const float2* tw16,
                       const float2* twab, const CoarseHandOver* handover_dev = nullptr, int n_handover = 0);
// matrix_cores (16-column jobs with at most 4 partitions whose columns' spectra are h[0] + c x P x kCoarseBins): the per-bin complex
// GEMM  Y[c][t] += sum_p H_c[p] X[t - p]  on v_mfma_f32_16x16x4_f32 (coarse_mfma16_kernel)
const char* launch_coarse_mac(hipStream_t s, const CoarseJob* jobs_dev, int njobs, const CoarseTerm* terms_dev, const float2* X, float2* Y,
                       int y_frames, int cw, int max_t, int maxP, bool any_private, int pb, bool matrix_cores = false);
const char* launch_coarse_inv(hipStream_t s, const CoarseOut* outs_dev, int nouts, int n_t, const int* ylist_dev, const float2* Y, int y_frames,
                       const float2* tw16, const float2* twab);
void launch_coarse_hist(hipStream_t s, const CoarseHistJob* jobs_dev, int njobs, int64_t max_len);
// A chunk's job tables, page-locked host memory -> device memory, by a kernel on the chunk's own stream: a copy by the DMA engine
// in front of the first kernel costs two engine hand-overs (~ 20 us each way in the traces), a kernel in front of a kernel one
// launch gap.  `bytes` is a multiple of 16; src is the device-visible address of hipHostMalloc'ed memory.
void launch_table_upload(hipStream_t s, void* dst, const void* src_pinned, size_t bytes);

// ---- graph plumbing kernels ------------------------------------------------------------------------
// out[f0 + i] = ((0 + t0[f0+i]) + t1[f0+i]) + ...  in term order (AudioNodeInput.cs:118-132,182-244)
struct MixJob {
  float* out;
  int term0;      // index of first term in the term table
  int nterms;
  int64_t f0;     // first chunk frame
  int64_t n;      // frames
  float* out2 = nullptr;   // a second row that receives the same sums (a channel whose term list equals this one's: mono material in a
                           // stereo input, AudioNodeInput.cs:182-244 adds the same values to both channels in the same order)
};
// gains_dev (may be null): one factor per entry of the term table -- term j contributes fl(term[j][f] * gain[j]) (a folded constant GainNode)
void launch_mix(hipStream_t s, const MixJob* jobs_dev, int njobs, const float* const* terms_dev, int64_t max_n, bool vec4, const float* gains_dev = nullptr,
                const float* const* curves_dev = nullptr);   // curves: per term, null or a chunk-frame indexed gain curve (a folded automated GainNode)

// the same sums for jobs of many terms (a bus of hundreds of voices; any alignment): one frame per lane, 32 terms' loads in flight
void launch_mix_wide(hipStream_t s, const MixJob* jobs_dev, int njobs, const float* const* terms_dev, int64_t max_n, const float* gains_dev = nullptr,
                     const float* const* curves_dev = nullptr);
constexpr int kMixWideMinTerms = 256;   // (the planner hands jobs of at least this many terms to launch_mix_wide while they are few)

// down-mix N -> 1: out[f] = (sum_ch in[ch][f]) * scale   (AudioNodeInput.cs:214-228); 'ins' index the term table
struct DownmixJob {
  float* out;
  int term0;
  int nch;
  float scale;
  int64_t f0;
  int64_t n;
};
void launch_downmix(hipStream_t s, const DownmixJob* jobs_dev, int njobs, const float* const* terms_dev, int64_t max_n, const float* gains_dev = nullptr);

// out = in * gain  (GainNode.cs:48-58); curve != nullptr -> per-sample a-rate values, else constant
struct GainJob {
  const float* in;
  float* out;
  const float* curve;
  const float* mod;     // audio-rate modulation (AudioParam.cs:123-135): value = clamp(intrinsic + mod, min, max); nullptr = none
  float gain;
  float vmin, vmax;
  int64_t f0;
  int64_t n;
};
void launch_gain(hipStream_t s, const GainJob* jobs_dev, int njobs, int64_t max_n);

// float32 Direct-Form-II biquad cascade with constant coefficients (BiQuadFilterNode.cs:136-141).  A job is a chain of
// 1..8 BiQuadFilterNodes connected output -> single input (the planner fuses them): every section is evaluated with
// exactly the per-node arithmetic, its float32 output feeding the next section, so results equal node-by-node processing.
constexpr int kMaxBiquadSections = 8;
struct BiquadSection {
  float b0, b1, b2, a1, a2;
  float pad_;
  float* state;   // {W1, W2} of this (node, channel)
};
struct BiquadJob {
  const float* in;
  float* out;     // nullptr: the job only advances its state (pass A of a cascade split along time)
  int sec0;       // first section in the section table
  int nsec;
  int64_t f0;
  int64_t n;
  float* state;   // nullptr: every section's own `state` ; else {W1, W2} of section q at state + 2 q (a piece of a split cascade)
  int twins = 1;  // the job stands for this many channels of its node(s) that carry the SAME signal from the same state (mono
  int pad_ = 0;   // material in a stereo node): the end state is written to the state slots of all of them (state + 2 t, t < twins)
};
void launch_biquad(hipStream_t s, const BiquadJob* jobs_dev, int njobs, const BiquadSection* secs_dev, int nsec);

// ---- constant-coefficient cascades split ALONG TIME (option "biquad_time_split") ---------------------------------------
// A cascade is a linear time-invariant system  s[n+1] = A s[n] + B x[n]  with the 2 x nsec direct-form-II values (W1, W2 of every
// section) as its state, so a long segment need not be one serial walk: cut into G pieces of K frames,
//   pass A   every piece but the last is run from the ZERO state, outputs dropped: its final state z_l          (G - 1 pieces in parallel)
//   scan     s_0 = the cascade's state ; s_{l+1} = A^K s_l + z_l                                             (per cascade, G - 1 steps)
//   pass B   every piece is run from its true initial state s_l and writes its outputs                        (G pieces in parallel)
// Each pass is the per-sample arithmetic of BiQuadFilterNode.cs:137-138 (the lane-per-cascade kernel, biquad_kernel<NSEC, 64>);
// A^K comes from the host (float64, repeated squaring).  What differs from the one-walk evaluation is the rounding of the
// pieces' initial states (not bit-exact; tests/test_gpu_biquad_split.py measures it): twice the arithmetic, G times the lanes.
struct BiquadScanJob {   // one cascade (node chain x channel) cut into G pieces of K frames (the last one shorter)
  const float* in;
  float* out;
  uint64_t m_off;     // offset of A^K (row-major [2 nsec][2 nsec], float) in the chunk's table
  float* scratch;     // [G - 1][nsec][2]: in: z_l of pass A ; out: s_l, where pass B's piece l starts
  int sec0, nsec;     // the cascade's own sections (their `state` is the persistent state: in s_0, out s_{G-1})
  int64_t f0, n;
  int twins = 1;      // as BiquadJob::twins (handed to the last piece)
  int pad_ = 0;
};
// the pieces of `ncasc` cascades as BiquadJobs, written on the device (the host tables stay one record per cascade):
// pass A: ncasc x (G - 1) state-only jobs ; pass B: ncasc x G jobs
void launch_biquad_split_expand(hipStream_t s, const BiquadScanJob* casc_dev, int ncasc, int G, int64_t K, BiquadJob* passA, BiquadJob* passB);
// lane-per-cascade kernel for any number of jobs (a job with out == nullptr only advances its state)
// (state_only: every job of the launch has out == nullptr -- pass A of a split: the output half is not evaluated)
void launch_biquad_lanes(hipStream_t s, const BiquadJob* jobs_dev, int njobs, const BiquadSection* secs_dev, int nsec, bool state_only = false);
void launch_biquad_scan(hipStream_t s, const BiquadScanJob* casc_dev, int ncasc, int G, const BiquadSection* secs_dev, const uint8_t* tables, int nsec = 0);   // nsec: the cascade length of every job (0 = mixed)

// BiQuadFilterNode with automated parameters (BiQuadFilterNode.cs:87-147): one lane per NODE walks block by block and
// channel by channel exactly like the reference, refreshing the coefficients whenever the per-sample frequency / Q move
// by more than 1e-3 Hz / 1e-4 (usedFreq / usedQ restart at 1000 / 1.0 every block, :111-112,126).
struct BiquadDynState {   // persistent per node
  float b0, b1, b2, a1, a2;
  int dirty;
  float w[64];            // {W1, W2} per channel (32 channels)
};
struct BiquadDynJob {
  const float* in[32];    // per channel, chunk-frame indexed; nullptr = zeros
  float* out[32];
  const float* fcurve;    // per-sample frequency curve or nullptr
  const float* qcurve;
  const float* gcurve;    // k-rate gain (dB) curve or nullptr
  float fval, qval, gval;
  int channels;
  int filter_type;
  float nyquist;
  float sample_rate;
  BiquadDynState* state;
  int64_t b0;             // first block (chunk relative)
  int64_t nblocks;
};
void launch_biquad_dynamic(hipStream_t s, const BiquadDynJob* jobs_dev, int njobs);

// AudioParam timeline evaluation (AudioParam.cs:114-247)
struct ParamEvent {
  int type;  // 0 SetValue, 1 LinearRamp, 2 ExponentialRamp, 3 SetTarget
  float value;
  float target;
  float pad_;
  double time;
  double time_constant;
};
struct ParamJob {
  float* out;         // chunk-frame indexed curve
  int ev0;            // first event in the event table
  int nev;
  float value;        // AudioParam._value
  int arate;          // 1 = per sample, 0 = per block
  int64_t b0;         // first block (chunk relative)
  int64_t nblocks;
};
// AudioParam.ComputeValueAtTime and helpers (AudioParam.cs:169-247), double precision, no contraction.  Host + device:
// the device evaluates a-rate / k-rate curves; the host evaluates the k-rate playbackRate that steers source replay.
__host__ __device__ inline float param_interp_linear(float v0, double t0, float v1, double t1, double t) {
  double u = (t - t0) / (t1 - t0);
  u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
  float d = v1 - v0;
  return (float)((double)v0 + (double)d * u);
}
__host__ __device__ inline float param_interp_exp(float v0, double t0, float v1, double t1, double t) {
  if (v0 <= 0 || v1 <= 0) return param_interp_linear(v0, t0, v1, t1, t);
  double u = (t - t0) / (t1 - t0);
  u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
  float ratio = v1 / v0;
  return (float)((double)v0 * pow((double)ratio, u));
}
__host__ __device__ inline float param_set_target(const ParamEvent& e, float baseline, double time) {
  double elapsed = time - e.time;
  if (elapsed <= 0) return baseline;
  double tc = e.time_constant > 0.001 ? e.time_constant : 0.001;
  float d = baseline - e.target;
  return (float)((double)e.target + (double)d * exp(-elapsed / tc));
}
__host__ __device__ inline float param_value_at(const ParamEvent* ev, int count, float value, double time) {
  if (count == 0) return value;
  float boundary = value;
  for (int i = 0; i < count; i++) {
    const ParamEvent& e = ev[i];
    if (time < e.time) {
      if (i == 0) return boundary;
      const ParamEvent& prev = ev[i - 1];
      if (e.type == 1) return param_interp_linear(prev.value, prev.time, e.value, e.time, time);
      if (e.type == 2) return param_interp_exp(prev.value, prev.time, e.value, e.time, time);
      if (prev.type == 3) return param_set_target(prev, boundary, time);
      return prev.value;
    }
    if (e.type != 3) boundary = e.value;
  }
  const ParamEvent& last = ev[count - 1];
  if (last.type == 3) return param_set_target(last, boundary, time);
  return last.value;
}
void launch_param_curve(hipStream_t s, const ParamJob* jobs_dev, int njobs, const ParamEvent* events_dev,
                        const double* block_times_dev, double delta_time, int64_t max_blocks);

// Looping rate-1 playback (AudioBufferSourceNode.cs:186-235): out[f] = buf[map(pos0 + f - f0)]
struct LoopJob {
  const float* buf;
  float* out;
  int64_t pos0;       // _playbackPosition at the first frame of the job
  int64_t loop_start;
  int64_t loop_end;
  int64_t f0;
  int64_t n;
};
void launch_loop_source(hipStream_t s, const LoopJob* jobs_dev, int njobs, int64_t max_n);

// CubicResampler playback (CubicResampler.cs:26-63) driven by a host-computed block trajectory
struct ResampleBlock {  // state at the start of one 128-frame block
  int64_t consumed;     // input samples consumed before this block (relative to the start position)
  double pos;           // CubicResampler.Pos
  int ready;            // CubicResampler.Ready
  int produced;         // outputs produced in this block (128 unless the input ran out)
};
struct ResampleJob {
  const float* buf;      // channel data
  float* out;            // chunk-frame indexed
  int64_t start_pos;     // buffer index of the first consumed sample
  int64_t avail;         // number of input samples available from start_pos
  int traj0;             // index of the trajectory entry for the job's first block
  double rate;
  int64_t b0;            // first block (chunk relative)
  int64_t nblocks;
};
void launch_resample(hipStream_t s, const ResampleJob* jobs_dev, int njobs, const ResampleBlock* traj_dev, int64_t max_blocks);
// The same arithmetic with ONE LANE PER OUTPUT SAMPLE.  The only serial part of CubicResampler.Process is the position recurrence
// (Pos += rate in double, consume = (int)Pos: CubicResampler.cs:40-60); the host replays it once per distinct rate anyway (Resampler)
// and leaves, per output sample, where the window ends and the interpolation fraction -- every voice of that rate reads the same
// table.  A lane then gathers its four taps and evaluates the reference's float32 polynomial: bit-exact, coalesced, bound by HBM
// instead of by 128 dependent steps per lane (config 4 at 4096 voices: 4.3 -> ~1 ms per 2.5 s).
struct ResampleSample {
  uint32_t ip;   // input samples consumed when this output is produced (relative to the job's start position): taps ip-4 .. ip-1
  float t;       // (float)Pos
};
struct ResampleFastJob {
  const float* buf;                // channel data
  float* out;                      // chunk-frame indexed
  const ResampleSample* samples;   // the trajectory's samples from the job's first block on (128 per block)
  int64_t start_pos;
  int64_t b0;
  int64_t nblocks;
};
void launch_resample_fast(hipStream_t s, const ResampleFastJob* jobs_dev, int njobs, int64_t max_blocks);

// General source replay: the host replays AudioBufferSourceNode.Process block by block (AudioBufferSourceNode.cs:165-358:
// k-rate playbackRate per block, loop wrap, the copy path when the effective rate is exactly 1, the resampler's window
// carried as buffer INDICES) and hands the device one descriptor per block; the device does the per-sample arithmetic.
// Used for looping playback with resampling and for a playbackRate that changes while the source plays.
struct GsrBlock {        // state at the start of one processed block
  int64_t pp;            // _playbackPosition (host bookkeeping)
  int64_t next;          // buffer index of the first sample fed in this block
  int64_t w[4];          // buffer indices held in S0..S3 (-1: never fed -> 0.0f)
  double pos;            // CubicResampler.Pos
  double rate;           // effective rate of this block
  int ready;             // CubicResampler.Ready
  int produced;          // outputs produced; the rest of the block is cleared
  int copy;              // 1: effective rate == 1.0 -> plain copy path (:186-222), resampler untouched
  int pad_;
};
struct GsrJob {
  const float* buf;      // channel data
  float* out;            // chunk-frame indexed
  uint64_t desc_off;     // byte offset of the first GsrBlock in the plan buffer
  int64_t b0;            // first block (chunk relative)
  int64_t nblocks;
  int64_t loop_start, loop_end;
  int loop;
  int pad_;
};
void launch_gsr(hipStream_t s, const GsrJob* jobs_dev, int njobs, const uint8_t* plan_base_dev, int64_t max_blocks);


// ---- AudioStreamNodeBase.Process (GraphAudio.IO/AudioStreamSourceNodeBase.cs:132-301): a queue of buffers played back to back through
// the CubicResampler.  The host replays Process block by block ON INDICES (which buffer and which samples every block feeds, the
// resampler's position, the window as four sample references) and uploads, per block, the pieces the block is made of --
// one per `while (framesRendered < framesToRender)` iteration of the reference, each fed from ONE buffer; the device does the
// per-sample arithmetic.  A window reference is (segment, index): segment >= 0 = a buffer of the chunk's segment table, -1 =
// never fed (0.0f), -2 = the value the window slot held at the start of the chunk (device state, per channel).
struct StreamSeg {
  const float* base;     // channel ch of the buffer at base + ch * stride
  int64_t stride;
};
struct StreamPiece {
  int64_t next;          // index (in the piece's buffer) of the first sample fed
  int64_t w[4];          // window S0..S3 at the start of the piece
  int wseg[4];
  int seg;               // the buffer this piece feeds from
  int out0, produced;    // output frames [out0, out0 + produced) of the block
  int copy;              // 1: effective rate == 1.0 -> plain copy (:222-240), the resampler is not touched
  int ready;             // CubicResampler.Ready
  int pad_;
  double pos, rate;      // CubicResampler.Pos, effective rate
};
struct StreamBlock {
  int piece0, npieces;   // pieces of this block (index into the node's piece table); frames not covered are cleared
};
struct StreamJob {
  float* out;            // chunk-frame indexed
  const float* win_in;   // [4] window values of this channel at the start of the chunk
  float* win_out;        // [4] window values at the end of the chunk (null: not written by this job)
  uint64_t blocks_off, pieces_off, segs_off;   // byte offsets of the node's tables in the plan buffer
  int64_t b0, nblocks;   // chunk-relative blocks of this job
  int64_t wend[4];       // window at the end of the chunk
  int wend_seg[4];
  int ch, pad_;
};
void launch_stream(hipStream_t s, const StreamJob* jobs_dev, int njobs, const uint8_t* plan_base_dev, int64_t max_blocks);

// ---- remaining pure-Core nodes ------------------------------------------------
// ConstantSourceNode.Process (ConstantSourceNode.cs:76-141): out[f] = offset[f] for frames inside [lo, hi) of the chunk,
// 0 outside (sample-accurate start / stop)
struct ConstJob {
  const float* curve;   // a-rate offset curve (chunk-frame indexed) or null -> `value`
  float* out;
  float value;
  int pad_;
  int64_t f0, n;        // frames of this job
  int64_t lo, hi;       // playing window in chunk frames
};
void launch_const_source(hipStream_t s, const ConstJob* jobs_dev, int njobs, int64_t max_n);

// OscillatorNode.Process (OscillatorNode.cs:91-158): phase accumulated in double with a conditional 2 pi wrap -- a serial
// recurrence.  One wavefront per oscillator: lane 0 walks the phase over 64 blocks at a time and leaves the phase at every
// block start in LDS, then 64 lanes generate one block each (the sin / saw / triangle evaluation runs 64-wide).
struct OscJob {
  const float* curve;   // a-rate frequency curve or null -> `value`
  float* out;
  double* phase;        // device: OscillatorNode._phase, read at the start and written back at the end
  float value;
  int type;             // 0 sine, 1 square, 2 sawtooth, 3 triangle (GA_OSC_*)
  int sample_rate;
  int pad_;
  int64_t f0, n;        // frames of this job (multiples of 128)
  int64_t lo, hi;       // playing window in chunk frames
};
void launch_oscillator(hipStream_t s, const OscJob* jobs_dev, int njobs, bool any_curve);

// StereoPannerNode.Process with a constant pan (StereoPannerNode.cs:76-153): the gains in force are tracked on the host
// (they only change when pan changes, :92/:127), the device applies the mono or the stereo law
struct PanJob {
  const float* in_l;
  const float* in_r;    // null on the mono path
  float* out_l;
  float* out_r;
  float gain_l, gain_r, pan;
  int stereo;           // 0: mono law (:103-105) ; 1: stereo law (:135-145)
  int64_t f0, n;
};
void launch_stereo_panner(hipStream_t s, const PanJob* jobs_dev, int njobs, int64_t max_n);

// StereoPannerNode with an automated pan: the gains are recomputed whenever pan differs from the previous SAMPLE's pan
// (`if (pan != lastPan)`, :92/:127) with the law of the path in force at that moment, and persist across blocks and paths.
// One wavefront per job: every lane finds the last change inside its block, the carried gains are resolved across the 64
// blocks of a group, then every lane walks its block.
struct PanState {
  float last_pan, gain_l, gain_r, pad_;
};
struct PanDynJob {
  const float* in_l;
  const float* in_r;     // null on the mono path
  float* out_l;
  float* out_r;
  const float* curve;    // a-rate pan curve (chunk-frame indexed); null: the constant `value` (a modulation input that fell silent)
  PanState* state;       // device-resident (_lastPan, _lastGainL, _lastGainR); read at the start unless `init`, written at the end
  PanState init_state;   // host-tracked state handed over when the node turns dynamic
  int init;
  int stereo;
  float value;           // Pan.Value, used where `curve` is null
  int pad_;
  int64_t f0, n;         // multiples of 128
};
void launch_stereo_panner_dynamic(hipStream_t s, const PanDynJob* jobs_dev, int njobs);

// DelayNode.Process (DelayNode.cs:43-100): out[f] = line[f - d(f)], d(f) = clamp((int)(delayTime[f] * sampleRate), 0, max);
// d = 0 reads 0 (CircularBuffer.Read, :136-144).  `line` is indexed like the chunk (line[f] = input sample of frame f) and
// preceded by the history of the previous chunks, so every read is a plain gather.
struct DelayJob {
  const float* line;    // line[f0 - max_delay .. f0 + n) must be readable
  const float* curve;   // a-rate delayTime curve or null -> `value`
  float* out;
  float value;
  int sample_rate;
  int max_delay;
  int pad_;
  int64_t f0, n;
};
void launch_delay(hipStream_t s, const DelayJob* jobs_dev, int njobs, int64_t max_n);

// AudioParam.ComputeARate / ComputeKRate with a non-silent modulation input (AudioParam.cs:123-135,148-160):
//   a-rate: out[f] = clamp(intrinsic[f] + mod[f], min, max) ; k-rate: the block's first sample of both, repeated over the block
struct ParamModJob {
  const float* intrinsic;   // timeline curve (chunk-frame indexed) or null -> `value`
  const float* mod;         // channel 0 of the mixed modulation input
  float* out;
  float value, vmin, vmax;
  int krate;
  int64_t f0, n;
};
void launch_param_mod(hipStream_t s, const ParamModJob* jobs_dev, int njobs, int64_t max_n);

// ProcessBlockInterleaved (AudioContextBase.cs:125-155): dst[(f0 + i) * channels + ch] = ch < used ? src[ch][f0 + i] : 0
struct InterleaveSrc {
  const float* ch[32];
};
void launch_interleave(hipStream_t s, float* dst, InterleaveSrc src, int channels, int used, int64_t f0, int64_t n);

}  // namespace ga

// ga_engine.hpp -- host side of libgraphaudio_hip.so: graph snapshot, control-plane simulation, chunk executor.
//
// How the reference's per-block pull model (AudioContextBase.ProcessBlock -> AudioNode.ProcessInternal ->
// AudioNodeInput.Pull, AudioContextBase.cs:52-81, Nodes/AudioNode.cs:152-183, AudioNodeInput.cs:100-138) becomes a
// device plan:
//   1. CONTROL PLANE (host, no sample data).  Everything that decides *which* arithmetic happens -- channel counts,
//      IsSilent flags, start/stop/ended scheduling, queued commands, the lagged ComputeOutputChannelCount -- depends
//      only on the schedule, never on sample values.  simulate_chunk() replays the reference's depth-first traversal
//      on that state alone and cuts the chunk into SEGMENTS of consecutive blocks with identical control state.
//   2. DATA PLANE (device).  For every segment the nodes are executed level by level as batched kernels over all
//      blocks of the segment at once; convolvers run once per chunk over all blocks (time-batched spectral MAC).
#pragma once
#include <cstring>
#include <initializer_list>
#include <new>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <tuple>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/graphaudio_hip.h"
#include "ga_kernels.hpp"

namespace ga {

struct Err {
  int code;
  std::string msg;
};
[[noreturn]] inline void fail(int code, const std::string& m) { throw Err{code, m}; }
#define GA_HIP(expr)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (expr);                                                                                 \
    if (e_ != hipSuccess)                                                                                   \
      ::ga::fail(e_ == hipErrorOutOfMemory ? GA_ERR_OUT_OF_MEMORY : GA_ERR_DEVICE,                          \
                 std::string(#expr) + ": " + hipGetErrorString(e_));                                        \
  } while (0)

// ---- AudioParam host state (AudioParam.cs:11-392) ----
struct ParamS {
  float def, minv, maxv;
  bool arate;
  float value;
  std::vector<ParamEvent> events;
  std::vector<std::pair<int, int>> modulation;  // connected outputs (node, output) -- unsupported on device yet
  // per-chunk
  float* curve = nullptr;  // chunk-frame indexed device curve when events are present
};

struct Conn {
  int node;
  int out;
  bool operator==(const Conn& o) const { return node == o.node && out == o.out; }
};
struct InRef {
  int node;
  int input;  // >= 0 input index; < 0: param (-1 - param)
  bool operator==(const InRef& o) const { return node == o.node && input == o.input; }
};

// AudioNodeInput (AudioNodeInput.cs:11-98)
struct InputS {
  int channelCount = 2;
  int mode = GA_COUNT_MODE_MAX;
  int interp = GA_INTERP_SPEAKERS;
  std::vector<Conn> connected;
  bool dirty = true;
  int bufCh = 0;  // channel count of the input's AudioBuffer; 0 = none yet
  bool silent = true;
};
struct OutputS {
  std::vector<InRef> connectedInputs;
  int bufCh = 0;  // AudioNodeOutput.Buffer channel count; 0 = null (owner not processed yet)
  bool silent = true;
  // flagged non-silent, but every sample is an exact zero: a ConvolverNode marks its output non-silent from its first block
  // (ConvolverNode.cs:153) while nothing has reached its input yet; nodes without state of their own pass the property on.  Only a
  // DelayNode asks: its output flag is raised by the first non-ZERO sample (DelayNode.cs:72,92), which such an input never delivers.
  bool zero = false;
};

struct PlayBuf {
  int channels = 0;
  int64_t length = 0;
  int64_t stride = 0;  // floats between channels on the device
  int sampleRate = 0;
  std::vector<std::vector<float>> host;
  float* dev = nullptr;   // first sample (inside the allocation `devBase`: Context::dallocSkewed)
  void* devBase = nullptr;
  size_t devBytes = 0;
  bool released = false;   // ga_buffer_release: the host no longer holds the buffer; storage goes when no node refers to it
};

struct IrSpectra {  // P zero-padded 256-point spectra per IR channel (PartitionedConvolver.cs:65-91)
  int P = 0;
  int nch = 0;
  float* hr = nullptr;  // [nch][129][P]
  float* hi = nullptr;
  // formulation C: N2-point spectra of the taps along the partition axis, [nch][129][N2], for N2 = 1024, 2048, 4096
  // (built on first use: the segment plan of a chunk may mix FFT lengths)
  float2* hspecN[3] = {nullptr, nullptr, nullptr};
  static int n2Index(int N2) { return N2 == 1024 ? 0 : N2 == 2048 ? 1 : 2; }
  size_t hBytes = 0, hspecBytes = 0;
  // formulation D: the scaled taps [nch][tapsStride] (kept for the coarse spectra, built on first use) and the packed
  // 16,384-point spectra of the coarse partitions [nch][coarseP][kCoarseBins]
  float* taps = nullptr;
  int64_t tapsStride = 0;
  size_t tapsBytes = 0;
  int coarseP = 0;
  float2* coarse = nullptr;
  size_t coarseBytes = 0;
  int64_t* devBytesRef = nullptr;   // the owning context's byte counter
  IrSpectra() = default;
  IrSpectra(const IrSpectra&) = delete;
  IrSpectra& operator=(const IrSpectra&) = delete;
  ~IrSpectra() {   // shared by the convolver nodes that use it and the context's cache; the last owner frees the device memory
    if (hr) (void)hipFree(hr);
    if (hi) (void)hipFree(hi);
    for (float2* h : hspecN)
      if (h) (void)hipFree(h);
    if (taps) (void)hipFree(taps);
    if (coarse) (void)hipFree(coarse);
    if (devBytesRef) *devBytesRef -= (int64_t)(2 * hBytes + hspecBytes + tapsBytes + coarseBytes);
  }
};

struct ConvGroup;   // rows sharing one IR channel's spectra
struct ConvRowRef {
  ConvGroup* group = nullptr;
  int idx = -1;
};

// source phase timeline inside one chunk
enum SrcPhase { SRC_IDLE = 0, SRC_PLAY = 1, SRC_END = 2, SRC_GONE = 3 };
struct SrcSpan {
  int64_t b0;      // chunk-relative first block
  int phase;
  int64_t pos;     // _playbackPosition at b0 (SRC_PLAY)
  int64_t blkIdx;  // index of b0 in the resampler trajectory (resampled playback)
};

struct NodeS {
  int id = 0, type = 0;
  bool disposed = false;
  std::vector<InputS> inputs;
  std::vector<OutputS> outputs;
  std::vector<ParamS> params;
  int64_t lastProcessedBlock = -1;
  bool isProcessing = false;
  bool reachable = false;
  bool prevReachable = false;   // reachable in the topology the PREVIOUS chunk ran on (= it was evaluated there; chunkTopology)
  int level = 0, depth = 0;

  // AudioBufferSourceNode (AudioBufferSourceNode.cs:15-30)
  int bufId = -1;
  bool hasStarted = false, hasStopped = false, endedRaised = false;
  double startTime = std::nan(""), stopTime = std::nan("");
  double offset = 0, duration = std::numeric_limits<double>::infinity();
  int64_t playbackPosition = 0;
  bool loop = false;
  double loopStart = 0, loopEnd = 0;
  int64_t rsBlocks = 0;          // resampler: blocks played so far (index into the trajectory)
  int64_t rsStartPos = 0;        // resampler: buffer index where consumption started
  std::vector<SrcSpan> spans;    // per-chunk plan
  // general replay mode (GsrBlock): entered -- for good -- when the source loops while resampling or its rate moves
  bool gsr = false;
  double rsRate = 0.0;           // trajectory mode: the effective rate the trajectory belongs to
  int rsBufId = -1;              // buffer the resampler window refers to
  int rsChannels = 0;            // `_resamplers.Length` (AudioBufferSourceNode.cs:238-245)
  int64_t gsrW[4] = {-1, -1, -1, -1};
  double gsrPos = 0.0;
  int gsrReady = 0;
  std::vector<GsrBlock> gsrBlocks;  // per chunk: one per processed block from the first played block, + the end state
  uint64_t gsrDevOff = 0;
  bool gsrUploaded = false;
  // ConstantSourceNode / OscillatorNode share hasStarted/hasStopped/startTime/stopTime/endedRaised with the buffer source
  int oscType = 0;                 // OscillatorNode._type
  double* oscPhase = nullptr;      // device: OscillatorNode._phase (one double)
  bool oscPhaseReset = false;      // Start() sets _phase = 0 (OscillatorNode.cs:62)
  int64_t schedLo = 0, schedHi = 0;  // per chunk: frames [lo, hi) of the chunk in which the scheduled source plays
  // StereoPannerNode (StereoPannerNode.cs:12-14)
  float panLast = std::nanf(""), panGL = 0.5f, panGR = 0.5f;
  PanState* panDev = nullptr;      // device copy of the three, authoritative while pan is automated (panOnDevice)
  bool panOnDevice = false;
  uint64_t bqDynSeq = ~0ull, panDynSeq = ~0ull;   // (chunk in which the node ran its per-sample kernel: Context::chunkSeq)
  bool bqDynChunk_unused = false;         // the same for a biquad whose parameter modulation falls silent inside a chunk (coefficient state on the device)
  bool panDynChunk_unused = false;        // (control plane) evaluated by the dynamic kernel in an earlier segment of THIS chunk: the gains in force live on the
                                   // device until the chunk ends, so the rest of the chunk stays on that kernel (a modulation input that falls silent)
  // DelayNode (DelayNode.cs:13-15)
  int maxDelaySamples = 0;
  int delayCh = 0;                 // channels of `_outputBuffer` (re-rented, i.e. silent again, when the count changes)
  int delayRings = 0;              // CircularBuffers allocated so far (2 at construction, grown on demand)
  bool delayAudible = false;       // the output buffer's non-silent flag (sticky, :96-97)
  float* delayHist = nullptr;      // device [rings][maxDelaySamples]: the samples written just before the current chunk
  int delayHistRings = 0;
  float* delayLine = nullptr;      // device scratch [rings][maxDelaySamples + delayCap]: history followed by the chunk's input
  int64_t delayCap = 0;
  int delayLineRings = 0;
  std::vector<int64_t> delayR;     // the same count as seen by the READER of a DelayNode at which a loop is cut (planned a convolver depth earlier)
  std::vector<int64_t> delayW;     // per ring: frames appended in the current chunk (a ring only advances while it is processed)
  bool delayLoaded = false;        // per chunk: history copied in front of the line
  // control-plane model of what the rings hold, each ring on its OWN time line (a ring beyond the input's channel count is not
  // written and keeps its content and position until the channel count grows again, DelayNode.cs:62-94)
  struct DelayRingModel {
    int64_t pos = 0;                                      // frames appended so far
    bool open = false;                                    // the last run is still growing
    std::vector<std::pair<int64_t, int64_t>> runs;        // [from, to) frame ranges that came from non-silent input blocks
  };
  std::vector<DelayRingModel> delayModel;
  int64_t delayPrevEval = -1;                             // absolute block of the previous evaluation
  int delayPrevCh = 0;
  // BiQuadFilterNode (BiQuadFilterNode.cs:12-19)
  int filterType = GA_FILTER_LOWPASS;
  float b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0;
  bool coefDirty = true;
  bool coefMemoValid = false;      // Context::updateBiquadCoefficients: its last arguments and result
  int coefMemoType = -1;
  float coefMemoIn[3] = {0, 0, 0}, coefMemoOut[5] = {0, 0, 0, 0, 0};
  BiquadDynState* bqDyn = nullptr;  // device: coefficients + dirty flag + {W1, W2} per channel
  float* bqState = nullptr;         // = bqDyn->w
  int bqTwinN = 32;                 // channels 0 .. bqTwinN-1 hold the SAME filter state (all zero at first; kept equal while the channels are
                                    // fed the same signal -- mono material in a stereo node -- and evaluated once: planBiquad, BiquadJob::twins)
  bool coefOnDevice = false;        // the device copy of the coefficients is newer than b0..a2 above (automated run)
  // ConvolverNode (ConvolverNode.cs:12-16,87,95)
  int irBuf = -1;
  std::shared_ptr<IrSpectra> ir;
  int effectiveOutCh = 0;
  bool isTrueStereo = false, normalize = true, enableTrueStereo = true;
  std::vector<ConvRowRef> convRows;
  // convolution path: 0 = not assigned yet, 1 = shared-IR groups (formulation A), 2 = private IR (formulation B)
  int convPath = 0;
  int bInCh = 0, bSlots = 0;       // formulation B: input channels / (input channel, IR channel) slots
  float* bHistR = nullptr;         // [bInCh][129][P-1] spectra history of every input channel
  float* bHistI = nullptr;
  float* bOverlap = nullptr;       // [bSlots][2][128] double-buffered overlap
  int bOvCur = 0;
  bool bShared = true;             // all input channels have carried identical signals so far (one x-row serves all)
  // where the last P-1 input spectra live: -1 = bHistR/bHistI (or all zero), 0/1 = still in that pair of x planes, at
  // rows [bHistRow, bHistRow + bHistNx), time offset bHistOff, row pitch bHistTxb (the next chunk reads them from there)
  int bHistPlane = -1;
  int bHistRow = 0, bHistNx = 1, bHistOff = 0, bHistTxb = 0;
  bool bHistZero = true;
  // formulation D (convPath 4): the state of a node is the last coarseP x 8192 INPUT samples of every input channel (time
  // domain, double buffered: a chunk reads one copy and writes the other); overlap-save keeps nothing on the output side
  // ... unless the samples in front of the next chunk ARE device memory that stays: a source played without resampling hands its
  // PlayableAudioBuffer itself to the convolver (zero-copy views), so the history of the next chunk is a span of that buffer --
  // nothing is copied (dHistExt[ch] = {first sample of the span, buffer id}; the buffer is kept alive by Context::collectGarbage)
  std::vector<std::pair<const float*, int>> dHistExt;
  float* dHist[2] = {nullptr, nullptr};   // [bInCh][dHistLen]  (inside the allocations dHistBase: Context::dallocSkewed)
  void* dHistBase[2] = {nullptr, nullptr};
  size_t dHistBytes = 0;
  int64_t dHistLen = 0;
  int dHistCur = 0;
  bool dHistZero = true;
  bool everFed = false;   // (control plane) a non-silent, not-known-zero block has reached this node's input: its state / tail may be non-zero
  // ... and the LEADER of a fused group (or a convolver on its own) carries, per output channel, what the input so far adds to
  // the samples behind the last chunk's end (ga_plan_conv.cpp, planCoarseStage): valid for the next chunk only, and only while the
  // group's signature is the same; the input histories above stay the authoritative state
  float* dTail[2] = {nullptr, nullptr};   // [dTailCh][dTailLen]
  int64_t dTailLen = 0;
  int dTailCh = 0, dTailCur = 0;
  uint64_t dTailSig = 0, dTailSeq = ~0ull - 1;
  // AudioStreamNodeBase (GraphAudio.IO/AudioStreamSourceNodeBase.cs:21-28): queue state lives on the host (indices only), the
  // resampler window of every channel lives on the device between chunks (stWin, double buffered), Pos / Ready on the host
  std::deque<int> stQueued, stProcessed;   // buffer ids
  int stCurrent = -1;
  int64_t stPos = 0;
  int stLastRate = 0;
  int stState = GA_STREAM_STOPPED;
  int stChannels = -1;                      // _resamplers.Length ; -1 = null
  double stRsPos = 0.0;
  int stRsReady = 0;
  bool stWinValid = false;                  // false: every window slot is 0 (cleared / never fed)
  float* stWin[2] = {nullptr, nullptr};     // [32 channels][4]
  int stWinCur = 0;
  struct StreamBlockInfo { int outCh; bool silent; };
  std::vector<StreamBlockInfo> stInfo;      // per chunk: channel count / silence of every block
  std::vector<StreamBlock> stBlocks;        // per chunk tables for stream_kernel
  std::vector<StreamPiece> stPieces;
  std::vector<StreamSeg> stSegs;
  int64_t stWend[4] = {0, 0, 0, 0};
  int stWendSeg[4] = {-1, -1, -1, -1};
  bool stFed = false;                       // the chunk's pieces moved a resampler window
  uint64_t stBlocksOff = 0, stPiecesOff = 0, stSegsOff = 0;
  bool stUploaded = false;
  // per chunk: convolver outputs that one summing input consumes are summed as spectra (Context::planCoarseFusion);
  // the leader's output slabs carry the sum, the other members contribute no time-domain signal of their own
  int dLeader = -1;
  int dGroupSize = 0;   // (on a leader) members of its fused group in this chunk, itself included
  // formulation R (the reference's own order and arithmetic, launch_refmac): refSens = the output reaches arithmetic that amplifies
  // or quantises last-bit differences (Context::refOrderSensitivity, per chunk); refOrder = this chunk evaluates the node that way
  // (nodes on the B / C state layout: spectra history + overlap, which R shares)
  bool refSens = false, refOrder = false;
  // feedback cycles: a node that some consumer pulls while it is being processed keeps a copy of the block it put out last
  // (shape of its output views: one row of 128 frames per channel, per output for a ChannelSplitterNode)
  bool staleProducer = false;
  bool delaySplit = false;     // (per chunk) a DelayNode at which a feedback loop is cut: reader in front of everything, writer at its level
  float* staleBuf = nullptr;   // the copy consumers read in this chunk: [staleRows][128]
  float* staleNext = nullptr;  // ... and the one this chunk's output is written to (the two swap when the chunk is planned)
  int staleRows = 0;
  uint64_t staleSeq = 0;       // Context::chunkSeq of the chunk that last wrote staleBuf (chunkStaleCommit): a node that stops being pulled
                               // from inside its own evaluation and later is again must not find the block it kept back then
};

// A vector with inline room for N elements (heap only beyond): the per-node, per-segment records of the control-plane
// simulation and of the planner are almost always one input with one term and a channel or two, and with tens of thousands of
// nodes their heap traffic (five to ten allocations per node and segment) was most of the host time of a chunk.
// The subset of std::vector's interface the engine uses; elements may be non-trivial (InSeg holds a SmallVec itself).
template <class T, int N>
class SmallVec {
 public:
  SmallVec() {}
  explicit SmallVec(size_t n) { resize(n); }
  SmallVec(size_t n, const T& v) { assign(n, v); }
  SmallVec(std::initializer_list<T> il) { for (const T& v : il) push_back(v); }
  SmallVec(const SmallVec& o) { for (const T& v : o) push_back(v); }
  SmallVec(SmallVec&& o) noexcept { take(std::move(o)); }
  ~SmallVec() { clear(); release(); }
  SmallVec& operator=(const SmallVec& o) {
    if (this != &o) { clear(); reserve(o.n_); for (const T& v : o) push_back(v); }
    return *this;
  }
  SmallVec& operator=(SmallVec&& o) noexcept {
    if (this != &o) { clear(); release(); take(std::move(o)); }
    return *this;
  }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  T* data() { return p_; }
  const T* data() const { return p_; }
  T& operator[](size_t i) { return p_[i]; }
  const T& operator[](size_t i) const { return p_[i]; }
  T* begin() { return p_; }
  T* end() { return p_ + n_; }
  const T* begin() const { return p_; }
  const T* end() const { return p_ + n_; }
  T& front() { return p_[0]; }
  const T& front() const { return p_[0]; }
  T& back() { return p_[n_ - 1]; }
  const T& back() const { return p_[n_ - 1]; }
  void clear() {
    for (size_t i = 0; i < n_; i++) p_[i].~T();
    n_ = 0;
  }
  void reserve(size_t c) {
    if (c <= cap_) return;
    size_t nc = std::max<size_t>(c, 2 * cap_);
    T* np = static_cast<T*>(::operator new(nc * sizeof(T)));
    for (size_t i = 0; i < n_; i++) {
      new (np + i) T(std::move(p_[i]));
      p_[i].~T();
    }
    release();
    p_ = np;
    cap_ = nc;
  }
  void push_back(const T& v) {
    if (n_ == cap_) { T tmp(v); reserve(n_ + 1); new (p_ + n_) T(std::move(tmp)); }
    else new (p_ + n_) T(v);
    n_++;
  }
  void push_back(T&& v) {
    if (n_ == cap_) { T tmp(std::move(v)); reserve(n_ + 1); new (p_ + n_) T(std::move(tmp)); }
    else new (p_ + n_) T(std::move(v));
    n_++;
  }
  void resize(size_t n) {
    while (n_ > n) p_[--n_].~T();
    reserve(n);
    while (n_ < n) new (p_ + n_++) T();
  }
  void assign(size_t n, const T& v) {
    clear();
    reserve(n);
    while (n_ < n) new (p_ + n_++) T(v);
  }
  void insert_front(const T& v) {   // (the biquad cascade walks upstream: at most kMaxBiquadSections elements)
    T tmp(v);
    push_back(tmp);
    for (size_t i = n_ - 1; i > 0; i--) p_[i] = std::move(p_[i - 1]);
    p_[0] = std::move(tmp);
  }

 private:
  T* inl() { return reinterpret_cast<T*>(buf_); }
  void release() {
    if (p_ != inl()) ::operator delete(p_);
    p_ = inl();
    cap_ = N;
  }
  void take(SmallVec&& o) {   // *this is empty and inline
    if (o.p_ != o.inl()) {
      p_ = o.p_;
      cap_ = o.cap_;
      n_ = o.n_;
      o.p_ = o.inl();
      o.cap_ = N;
      o.n_ = 0;
    } else {
      for (size_t i = 0; i < o.n_; i++) {
        new (p_ + i) T(std::move(o.p_[i]));
        o.p_[i].~T();
      }
      n_ = o.n_;
      o.n_ = 0;
    }
  }
  alignas(T) unsigned char buf_[sizeof(T) * N];
  T* p_ = inl();
  uint32_t n_ = 0, cap_ = N;
};
using Views = SmallVec<const float*, 4>;   // per-channel views (device pointers; nullptr = silent) of a node output / a mixed input

// one evaluated control state of a node within a segment
struct TermS {
  int node, out, ch;
  // The producer was still being processed when this input pulled it (a feedback cycle): ProcessInternal's memo check returns at
  // once (Nodes/AudioNode.cs:153-156), so the consumer mixes the buffer the producer's output STILL holds -- its previous block
  // (an implicit one-block delay on the edge that closes the loop).  The device keeps that block per producer (NodeS::staleBuf).
  bool stale = false;
};
struct InSeg {
  int bufCh = 0;
  bool silent = true;
  bool zero = false;   // non-silent, but every contributing buffer is known to hold exact zeros (OutputS::zero)
  SmallVec<TermS, 2> terms;  // non-silent contributors in connection order
};
struct NodeSeg {
  int id = 0;
  int type = 0;   // (the node's type, so that a replayed record is dispatched without touching the node: Context::chunkSimulate)
  // ... and what the planner's sorting / cascade-fusion sweeps ask of every node, so that only the planning pass itself touches the
  // 1.2 KB node records (config 4: 28,672 of them, three sweeps of cache misses per chunk): set when the record is made, valid
  // while the graph stands (a replayed record belongs to the same graph version)
  int16_t level = 0, depth = 0;
  bool fan1 = false;   // output 0 feeds exactly one input
  SmallVec<InSeg, 1> ins;
  // AudioParam modulation inputs (AudioParam.cs:97-101), one per param -- sized only when some parameter of the node HAS a
  // modulation input (rare); otherwise empty, which reads as "every pin silent" (pinSilent)
  SmallVec<InSeg, 1> pins;
  bool pinSilent(int p) const { return p >= (int)pins.size() || pins[p].silent; }
  bool bqDynamic = false;   // biquad with automated parameters
  int outCh = 0;
  bool outSilent = true;
  bool outZero = false;     // flagged non-silent, but exact zeros in the reference (OutputS::zero): consumers are handed the zero page
  int srcPhase = SRC_IDLE;
  int64_t srcPos = 0;
  int64_t srcBlk = 0;
  int srcBuf = -1;  // buffer id at evaluation time (a later Dispose clears the node's reference)
  bool bqActive = false;
  float b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0;
  uint32_t outMask = 0;     // ChannelSplitterNode: outputs that carry audio
  float panGL = 0, panGR = 0, pan = 0;   // StereoPannerNode: gains in force in this segment
  int panMode = 0;          // 1 = mono law, 2 = stereo law
  bool panDyn = false;      // automated pan: per-sample gains on the device
  bool delayAudible = false;
};
struct Segment {
  int64_t b0 = 0, b1 = 0;  // chunk-relative block range
  std::vector<NodeSeg> nodes;  // processing (post) order
  uint64_t hash = 0;
};

struct ConvGroup {
  std::shared_ptr<IrSpectra> ir;
  int irCh = 0;
  int depth = 0;                          // convolver depth the group is executed at (one pass per chunk)
  int P = 0;
  std::vector<std::pair<int, int>> rows;  // (node id, row slot) ; node id < 0 = free row
  int rp = 0;                             // allocated (padded) rows of the state arrays
  float* histR = nullptr;                 // [129][P-1][rp]
  float* histI = nullptr;
  float* overlap[2] = {nullptr, nullptr}; // [rp][128], double buffered
  int ovCur = 0;
  bool histZero = true;
};

// FFT length of formulation C for P partitions: N2 = 4 * 2^ceil(log2 P), so that the valid part of a segment is
// (N2 - P + 1) / N2 >= 75 %; the kernels exist for 1024, 2048 and 4096 points (65 <= P <= 128 runs with 1024)
inline int tapFftSize(int P) {
  int n = 1;
  while (n < P) n <<= 1;
  return std::max(1024, 4 * n);
}

// which statistics bucket a recorded launch belongs to: the GA_STAGE_* indices of ga_stats
enum LaunchKind { LK_OTHER = GA_STAGE_OTHER, LK_MIX = GA_STAGE_MIX, LK_FFT = GA_STAGE_RFFT_FWD, LK_MAC = GA_STAGE_MAC,
                  LK_IFFT = GA_STAGE_RFFT_INV, LK_CFWD = GA_STAGE_COARSE_FWD, LK_CMAC = GA_STAGE_COARSE_MAC,
                  LK_CINV = GA_STAGE_COARSE_INV, LK_CHIST = GA_STAGE_COARSE_HIST, LK_CPREMIX = GA_STAGE_COARSE_PREMIX };

struct DevArena {  // grow-only device scratch
  void* p = nullptr;
  size_t bytes = 0;
};

struct Resampler {  // host replay of CubicResampler's position recurrence (CubicResampler.cs:40-60) for one rate
  double rate = 1.0;
  std::vector<ResampleBlock> blocks;  // state at the start of each played block (unbounded input)
  int64_t consumedEnd = 0;            // consumption after the last computed block
  double posEnd = 0.0;
  int readyEnd = 0;
  int devOffset = -1;                 // offset in the per-chunk device trajectory table
  // per OUTPUT SAMPLE (resample_fast_kernel): where the window ends and the interpolation fraction.  Produced by extend() next to the
  // block states, moved to a device table that only grows (1 KB per block and rate, shared by every voice of the rate)
  std::vector<ResampleSample> pending;   // samples of blocks [devBlocks, devBlocks + pending.size() / 128): not on the device yet
  ResampleSample* devSamples = nullptr;
  int64_t devBlocks = 0, devCapBlocks = 0;
  bool samplesOk = true;                 // false once a consumption count no longer fits the table's 32-bit index
  void extend(int64_t nblocks);
};

struct ChunkRun;   // ga_chunk_internal.hpp
struct Exec;       // ga_chunk_internal.hpp
struct NodePlanCtx;   // ga_chunk_internal.hpp
struct ConvPlanCtx;   // ga_chunk_internal.hpp

void biquadCoefficients(int filterType, float sampleRate, float frequency, float q, float gain, float o[5]);   // BiQuadFilterNode.cs:149-258

struct Context {
  int sampleRate;
  int device = 0;
  hipStream_t stream = nullptr;
  bool ownStream = true;
  int64_t currentBlock = 0;
  double currentTime = 0.0;
  bool disposed = false;
  bool latched = false;  // _renderThreadId != -1 (AudioContextBase.cs:59-62)
  bool inRender = false;
  std::deque<std::function<void()>> pending;
  std::vector<std::unique_ptr<NodeS>> nodes;
  std::vector<std::unique_ptr<PlayBuf>> buffers;
  std::map<std::pair<int, int>, std::shared_ptr<IrSpectra>> irCache;  // (buffer id, normalize)
  std::vector<std::unique_ptr<ConvGroup>> groups;
  std::map<std::tuple<IrSpectra*, int, int>, ConvGroup*> groupOf;   // (IR, IR channel, depth)
  std::map<uint64_t, std::unique_ptr<Resampler>> resamplers;          // keyed by rate bits
  std::string lastError;
  ga_stats stats{};
  // options
  int64_t maxChunkBlocks = 4096;
  double memBudgetFraction = 0.7;
  bool profile = false;
  // option "profile_every" = k: only every k-th chunk records its HIP events (they cost ~50 us of device time per chunk: 2 % of a
  // 1024-voice step, 10 % of a 128-voice one); ga_stats.profiled_chunks counts the chunks the stage times come from
  int profileEvery = 1;
  int64_t profileSeq = 0;
  bool profileNow = false;
  // device resources
  double2* w128 = nullptr;
  double2* w256 = nullptr;
  std::map<int, float2*> twC;   // float32 twiddles exp(-2 pi i j / N2) for the block-axis FFTs
  float* zeros = nullptr;  // zero page (chunk frames)
  int64_t zerosLen = 0;
  DevArena planes[4];      // xr, xi, yr, yi scratch shared by all groups
  DevArena planesB[4];     // formulation B scratch: x planes of pair 0 (re, im), y planes (re, im)
  DevArena planesBalt[2];  // x planes of pair 1: chunks alternate between the pairs, so the previous chunk's spectra stay readable
  int bPairCur = 0;        // pair written by the last chunk that ran a formulation B/C convolver
  int bPairWrite = 0;      // pair this chunk writes
  size_t bRowX = 0, bRowY = 0;        // next free x / y row of this chunk (convolver depths share the planes)
  std::vector<int> bResidents[2];     // nodes whose history lives in pair p
  float* xPlane(int pair, int im) { return (float*)(pair == 0 ? planesB[im].p : planesBalt[im].p); }
  void flushPlaneHistories(int pair); // moves them to the nodes' private stores (before the pair is rewritten)
  DevArena tables;         // per-chunk job tables
  bool tableUploadKernel = true;   // option "table_upload_kernel": the job tables reach the device by a kernel, not by the DMA engine
  void* tablesHost = nullptr;      // pinned staging of the job tables (buffer 0)
  size_t tablesHostBytes = 0;
  // option "async": render calls return after the work is enqueued (the host simulation and planning of the next call
  // overlap the device execution of this one); ga_synchronize waits.  The host may run one chunk ahead: the job tables are
  // staged in two pinned buffers and chunk k + 1 is not planned before chunk k - 1 has finished.
  bool asyncMode = false;
  void* tablesHostB = nullptr;     // buffer 1 (async mode)
  size_t tablesHostBBytes = 0;
  uint64_t chunkSeq = 0;
  hipEvent_t chunkDone[2] = {nullptr, nullptr};
  struct ProfBatch {
    hipEvent_t begin, end;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
    std::vector<int> kinds;
    std::vector<double> bytes;   // necessary HBM bytes of each launch (0 = not accounted)
  };
  std::deque<ProfBatch> pendingProf;
  void harvestProfile(bool wait);   // folds finished event batches into `stats`
  void synchronize();
  std::vector<float*> slabFree, slabAll;
  int64_t slabFrames = 0;
  std::vector<void*> slabBlocks;
  std::vector<void*> bqBlocks;
  size_t bqUsed = 0;
  int64_t devBytes = 0;
  // destination output of the last block (for Render(int) and the leftover cache)
  int destOutCh = 0;
  float* cacheDev = nullptr;  // [32][128]
  int cachedFrames = 0, cachedCh = 0;

  explicit Context(int sr) : sampleRate(sr) {}
  ~Context();
  void init_device(int device);
  void* dalloc(size_t bytes);
  void dfree(void* p, size_t bytes);
  void ensure(DevArena& a, size_t bytes);
  // Rows that many workgroups walk side by side (the voices' sample buffers, their input histories): separate allocations
  // start at multiples of 2 MiB, so the same offset of every row would sit on the same HBM channel and bank -- measured 5.0
  // TB/s against 6.1 for the pre-mix kernel's access pattern (tools/proto/hbm_peak.hip).  Each row therefore starts 1 KiB
  // further into its allocation than the one before (mod 64).  `*base` / `*total` are what dfree() takes.
  float* dallocSkewed(size_t bytes, void** base, size_t* total);
  // id of the live PlayableAudioBuffer whose device storage holds all of [p, p + n), -1 if none (a sorted table, rebuilt when buffers
  // come or go: bufVersion)
  int persistentBuffer(const float* p, int64_t n);
  struct BufSpan { const float* lo; const float* hi; int id; };
  std::vector<BufSpan> bufSpans;
  uint64_t bufVersion = 1, bufSpansVersion = 0;
  bool coarseExtHist = true;   // option "coarse_ext_history"
  bool coarseMfma = true;      // option "coarse_mfma": 16-column jobs on the matrix cores (coarse_mfma16_kernel); 0 = the register-tiled instance
  bool coarseWide = true;      // option "coarse_wide": 16-column multiply-accumulate jobs for multi-channel private impulse responses
  bool gainFold = true;        // option "gain_fold": a GainNode with any other constant gain and one consumer is multiplied inside the consumer's mix
  bool gainPassThrough = true; // option "gain_pass_through": a GainNode with a constant gain of exactly 1 hands its input views on (no kernel)
  // constant-coefficient biquad cascades split along time (ga_kernels.hpp, BiquadScanJob): A^K per (coefficients, K), float64 on
  // the host, remembered; the pieces' states live in blocks that are handed out per chunk and kept
  int biquadTimeSplit = 1;            // option "biquad_time_split": 0 never, 1 where the predicted deviation is small (below), 2 always
  int64_t biquadSplitMinFrames = 16384;   // option "biquad_split_min_frames": segments shorter than this stay one walk
  // biquadDeviation: predicted RMS difference between two float32 evaluations of the cascade that round differently (the one walk
  // and the split): rounding noise injected at every section's W, shaped by the rest of the cascade.  Direct form II at low cut-offs is the hard case (large W, cancelling output taps): e.g. the 100 Hz low shelf of
  // config 4 carries ~1e-4 of such noise IN THE REFERENCE'S OWN ARITHMETIC, so no re-association can stay within 1e-5 of it.
  struct BqTransition { std::vector<float> key; std::vector<float> M; };
  std::unordered_map<uint64_t, std::vector<std::pair<std::vector<float>, double>>> bqDeviations;
  double biquadDeviation(const float* coefs, int nsec);
  std::unordered_map<uint64_t, std::vector<BqTransition>> bqTransitions;
  const BqTransition& biquadTransition(const float* coefs, int nsec, int64_t K);   // coefs: [nsec][5] = b0 b1 b2 a1 a2
  // mode 1 splits a cascade when that prediction (absolute, unit-variance white input) is below this: with audio at sigma <= 0.25
  // a thousand such cascades summed incoherently stay within north_star's 1e-5
  double biquadSplitMaxDeviation = 2.5e-6;   // option "biquad_split_max_deviation"
  std::vector<float*> bqSplitBlocks;   // blocks of kBqSplitBlock bytes
  size_t bqSplitUsed = 0;              // bytes handed out in the current chunk
  static constexpr size_t kBqSplitBlock = (size_t)4 << 20;
  float* bqSplitAlloc(size_t floats);
  unsigned skewSeq = 0;

  // command queue (AudioContextBase.cs:266-305)
  void executeOrPost(std::function<void()> cmd);
  void post(std::function<void()> cmd);
  void drain();

  NodeS* node(int id);
  ParamS* param(int node, int p);
  PlayBuf* buffer(int id);
  std::vector<int> releasedPending;   // released buffers whose storage is still held
  int64_t chunksSinceGc = 0;
  void collectGarbage();              // frees released, unreferenced buffers and impulse-response spectra nobody uses
  InputS* inputOf(const InRef& r);

  // graph edits (Nodes/AudioNode.cs:109-150,207-238; AudioNodeOutput.cs:42-70; AudioNodeInput.cs:60-83)
  void connectTo(int src, int out, InRef in);
  void disconnectFrom(int src, int out, InRef in);
  void outputDisconnectAll(int src, int out);
  void inputDisconnectAll(InRef in);
  void doDispose(int id);

  std::shared_ptr<IrSpectra> irSpectra(int bufId, bool normalize);
  void releaseConvState(NodeS& n);
  void refOrderSensitivity(const std::vector<int>& topo);
  bool ensureResampleSamples(Exec& ex, Resampler& rs, int64_t upto);
  bool resampleFast = true;   // option "resample_fast": one lane per output sample from the trajectory's per-sample table
  void assignConvPaths(const std::vector<int>& topo, int64_t chunkBlocks);
  // formulation D
  // Option `coarse_overlap` (default 0): run the forward transforms and the multiply-accumulate CONCURRENTLY -- the signals
  // are cut into groups, group g's multiply-accumulate jobs run on a second stream while the main stream transforms group
  // g + 1, joined before the inverse transforms.  Measured on config 3: 3.37 ms of device time per step against 3.18 ms one
  // after the other (both stages slow down by ~50 % when they share the chip: they are bound by the same memory system),
  // so it stays off; kept as a switch for other shapes.
  // Output hand-over of an asynchronous render into host memory: the bus is copied (device to device, a few microseconds) into
  // one of two staging buffers on the context's stream and leaves for the host on a copy stream of its own, so that the
  // next chunk's kernels do not queue behind ~0.15 ms of PCIe transfer per 10 s of stereo bus.  Only with the context's own
  // stream: a caller-supplied stream keeps the one-stream ordering its owner expects.  OFF by default (option
  // "host_copy_stream"): measured on config 3 it LOSES 5 % (2.84-2.89 ms per step against 2.70-2.72): the copy's blit kernels
  // start beside the next chunk's forward transforms and those run 0.1-0.3 ms longer for it (tools/trace_gaps.sh).
  hipStream_t copyStream = nullptr;
  bool hostCopyStream = false;   // option "host_copy_stream"
  hipEvent_t outReady[2] = {nullptr, nullptr}, outCopied[2] = {nullptr, nullptr};
  float* outStage[2] = {nullptr, nullptr};
  size_t outStageBytes[2] = {0, 0};
  bool outPending[2] = {false, false};
  int outCur = 0;
  void handOverToHost(const float* const* src, float* const* out, int channels, int64_t offset, int64_t frames);
  void waitHostCopies();
  hipStream_t stream2 = nullptr;
  hipEvent_t dGroupEv[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t dJoinEv = nullptr;
  bool coarseOverlap = false;
  bool coarseTail = true;    // option "coarse_tail": outputs carry their tails from chunk to chunk (0: input histories only)
  int convRefOrder = 1;               // option "conv_reference_order": 0 = never, 1 = where refOrderSensitivity() asks for it, 2 = every convolver
  double convRefMinDeviation = 2.5e-6;  // option "conv_ref_min_deviation": a downstream biquad counts as sensitive above this predicted deviation
  std::vector<char> refSensScratch;
  bool coarseTailPrivate = false;  // option "coarse_tail_private": also groups whose members have impulse responses of their own (measured: the forward
                                   // stage saves P' - 1 windows per signal, the multiply-accumulate and inverse stages pay for the tail blocks -- 2.95 vs 2.97 ms at
                                   // 1024 voices per 10 s step, 0.50 vs 0.43 ms at 64 voices per 2.5 s call: off)
  bool coarseCarry = true;   // option "coarse_carry": the forward kernel writes the next chunk's history (0: always the copy kernel)
  bool coarsePremix = true;  // option "coarse_premix": fused groups on ONE impulse response are summed in the time domain, in front of
                             // one set of transforms (0: every member is transformed, the spectra are summed -- coarse_sum_kernel)
  void ensureOverlapStream();
  // event pairs recorded by launches that time their own pieces (several kernels, two streams); folded into the chunk's profile batch
  void noteKernel(int kind, const char* name) {   // ga_stats.stage_kernel: the kernel instance a stage's latest launch ran
    if (kind < 0 || kind >= 16 || !name || !*name) return;
    std::strncpy(stats.stage_kernel[kind], name, sizeof(stats.stage_kernel[kind]) - 1);
  }
  struct ExtraProf { hipEvent_t e0, e1; int kind; double bytes; };
  std::vector<ExtraProf> extraProf;
  // host-side storage of a chunk's per-node records, handed from chunk to chunk (a fresh allocation of megabytes per chunk costs
  // its page faults again every time)
  std::vector<std::vector<NodeSeg>> segNodePool;
  std::vector<std::vector<Views>> viewsPool;
  DevArena coarseM;                 // mixed input signals of pre-mixed groups
  DevArena coarseX, coarseY;        // spectra frames of a convolver stage (shared by the stages of a chunk, which run in order)
  float2* coarseTw = nullptr;       // combine-pass twiddles [2][2049]: W_8192^k, W_16384^k
  const float2* coarseTwab();
  float2* tw16pw = nullptr;
  const float2* twiddles16pw();
  void ensureCoarseSpectra(IrSpectra& ir);
  void planCoarseFusion(const std::vector<int>& topo, const std::vector<Segment>& segs);
  uint64_t fusionKey = 0;      // what the current dLeader assignment was derived from (planCoarseFusion)
  bool fusionKeyValid = false;
  void aliasBusToLeader(ChunkRun& r);
  ConvRowRef addGroupRow(const std::shared_ptr<IrSpectra>& ir, int ch, int depth, int nodeId);
  void ensureGroupState(ConvGroup& g);
  const float2* twiddlesC(int N2);
  const float2* twiddles16(int N2);
  std::map<int, float2*> tw16;
  bool useRadix16 = true;   // option `tconv_radix16`
  bool useCoarse = true;          // option `coarse`: formulation D (coarse partitions, voice sum fused) for long impulse responses
  int64_t coarseMinBlocks = 256;  // option `coarse_min_blocks`: a convolver takes formulation D when its first chunk has at least this many blocks
  int debugTconvN2 = 0;     // option `debug_tconv_n2` (tests of the error path only)
  const float2* ensureTapSpectra(IrSpectra& ir, int N2);
  struct TconvLaunch { int N2, tbase, nseg; };
  std::vector<TconvLaunch> tconvPlan(int nblocks, int P) const;   // FFT lengths of the segments that cover a chunk
  bool fft64 = false;            // option "fft64": double-precision 256-point transforms in the B-layout kernels (reference-like)
  bool useTimeFft = true;        // option "time_fft": formulation C for 64 < P <= 1024
  void updateBiquadCoefficients(NodeS& n, float frequency, float q, float gain);

  void render(float* const* out, int channels, int64_t frames, int64_t start, bool deviceOut);
  void runChunk(int64_t nblocks, float* const* bus);
  void runChunkImpl(int64_t nblocks, float* const* bus);
  // the passes of one chunk (ga_chunk.cpp, ga_plan_nodes.cpp, ga_plan_conv.cpp; ChunkRun holds what they share)
  void chunkTopology(ChunkRun& r);
  void chunkSimulate(ChunkRun& r);
  void chunkResources(ChunkRun& r);
  void chunkParamCurves(ChunkRun& r);
  void chunkConvScratch(ChunkRun& r);
  void chunkPlanNodes(ChunkRun& r, int depth);
  void planConstantSource(NodePlanCtx& k);   // (chunkPlanNodes, one per node type)
  void planOscillator(NodePlanCtx& k);
  void planDelay(NodePlanCtx& k);
  void planStereoPanner(NodePlanCtx& k);
  void planBufferSource(NodePlanCtx& k);
  void planStreamSource(NodePlanCtx& k);
  void planGain(NodePlanCtx& k);
  void planBiquad(NodePlanCtx& k);
  void chunkPlanConvolvers(ChunkRun& r, int depth);
  void planConvolversShared(ChunkRun& r, int depth, ConvPlanCtx& k);    // formulation A groups
  void planConvolversPrivate(ChunkRun& r, int depth, ConvPlanCtx& k, bool refOrder);   // formulations B / C, and R on their state layout
  void chunkDelayCommit(ChunkRun& r);
  void chunkStaleSeed(ChunkRun& r);
  void chunkStaleCommit(ChunkRun& r);   // feedback cycles: the block every stale producer put out becomes what its consumers read next
  void chunkExecute(ChunkRun& r);
  void chunkCommit(ChunkRun& r);
  void ensureBiquadState(NodeS& bn);
  // AudioStreamNodeBase.Process replayed on indices for `nblocks` blocks from the node's current state: fills the node's per-chunk
  // tables (commit = false) or moves the node's state to the end of the replayed blocks (commit = true)
  void streamReplay(NodeS& s, int64_t nblocks, const std::vector<double>& bt, bool commit);
  void streamFlushToProcessed(NodeS& s);
  bool faulted = false;      // a render failed after control state had moved: the context refuses further renders
  std::string faultMsg;
  int chunkPhase = 0;        // 0 = checks only (a failure leaves the context usable), 1 = state is moving

  // sharded render (ga_comm_*, ga_render_reduce): the RCCL communicator of this rank and its device-side bus staging
  void* comm = nullptr;          // ncclComm_t (RCCL is loaded with dlopen on first use: ga_comm.cpp)
  int commRanks = 0, commRank = 0;
  float* reduceBuf = nullptr;    // [channels][frames] contiguous
  size_t reduceBytes = 0;
  bool commDead = false;         // aborted after a local failure or a failed / timed-out collective: ga_comm_destroy + ga_comm_init
  double commTimeoutS = 120.0;   // option "comm_timeout_s"
  void commInit(const void* id, int nRanks, int rank);
  void commDestroy();
  void commInfo(int* nRanks, int* rank, int* usesRccl);
  void commAbort();
  void commWait();               // wait for the stream; a dead peer becomes an error code, not a hang
  void renderReduce(float* const* out, int channels, int64_t frames, int64_t start, int root);

  int64_t busCapFrames = 0;
  std::vector<float*> busSlabs;
  // a render into device memory whose chunk covers whole blocks lets the destination mix straight into the caller's rows
  // (no copy of the bus afterwards): set by Context::render around runChunk, consulted where the destination's input is resolved
  // Deferred hand-over (option "host_defer"): an ASYNCHRONOUS render whose rows are page-locked host memory leaves its bus in device
  // staging rows, and the rows cross PCIe inside the NEXT chunk's first long kernel (extra one-term jobs of the pre-mix launch:
  // their workgroups write over PCIe while the others stream HBM) instead of at the end of this chunk's last kernel, where
  // nothing else runs.  Without a next chunk: plain copies on the stream, from ga_synchronize or whatever touches the stream next.
  struct HandOver { const float* src; float* dst_dev; float* dst_host; int64_t n; };
  std::vector<HandOver> pendingHandOver;
  // device memory that has been replaced while launches that read it may still be in flight (a per-rate sample table that doubled):
  // freed at the next point where the stream is known to be idle (synchronize, destruction) -- hipFree would wait for the device
  std::vector<std::pair<void*, size_t>> retired;
  void freeRetired();
  float* deferStage = nullptr;
  size_t deferStageBytes = 0;
  bool hostDefer = true;
  void flushHandOver();
  float* busTarget[32] = {};
  bool hostDirect = true;   // option "host_direct": page-locked host rows are such a target too (0: always the copy kernels)
  struct SegCh { int64_t b0, b1; int ch; };
  std::vector<SegCh> chunkSegCh;   // destination buffer channel count of every segment of the last chunk
  float* ilvDev = nullptr;         // device staging for interleaved output
  size_t ilvBytes = 0;
  void processBlocks(float* const* outPlanar, float* outInterleaved, int channels, int64_t blockCount, bool deviceOut);
  int64_t chunkLimit(int64_t nblk);
  // per-chunk scratch of the biquad fusion pre-pass (dense, stamp-validated)
  std::vector<uint32_t> fuseStamp;
  std::vector<const NodeSeg*> fuseSeg;
  std::vector<int> fuseAbs, fuseLen;
  uint32_t fuseEpoch = 0;
  uint64_t graphVersion = 1, topoVersion = 0;   // connections / disposals / IR changes bump graphVersion
  std::vector<int> topoCache;
  int topoMaxDepth = 0, topoMaxLevel = 0;
  std::deque<int> endedQueue;  // sources whose Ended was raised and not yet reported through ga_poll_ended
  uint64_t lastHash = 0;       // control-state hash of the last block of the previous chunk
  // Steady renders: the control-plane records of the previous chunk's LAST segment (kept instead of being recycled).  When nothing
  // can have moved since -- no API call (apiEpoch), no queued command, the same graph, the segment was a fixpoint (two equal hashes
  // in a row), no node whose control state depends on time (delay lines, stream sources), every source in the phase the records
  // say -- the first block of the next chunk is not traversed again: the records are taken over and only the sources' positions
  // are brought up to date (chunkSimulate).  28,672 nodes: 6.5 ms of pull-model traversal per chunk become a pass over a dense array.
  std::vector<NodeSeg> lastSegNodes;
  uint64_t lastSegHash = 0, lastSegEpoch = 0, lastSegGraphVersion = 0;
  bool lastSegStable = false;
  uint64_t apiEpoch = 0;       // bumped by every API call that can change what a render computes (ga_api.cpp guard) and by drained commands
  bool simReplay = true;       // option "sim_replay"
  bool twinChannels = true;    // option "twin_channels": channels that carry the same signal from the same state are evaluated once
  bool topoHasTimeNodes = false, topoHasConvolvers = false, topoHasOscillators = false, topoHasStreams = false;
  bool topoHasCycles = false;   // (chunkTopology) some node is pulled while it is being processed: chunks of ONE block (the reference's own granularity)
  std::vector<int> staleProducers;
  std::vector<int> staleLeavers;   // nodes that were evaluated in the previous chunk and are not in this one (an edit took them out of
                                   // the graph): their output buffers keep the last block, which a later loop through them would read
  double loopGainBound = 0.0;   // (chunkTopology) the largest estimated gain of a feedback loop: differences that enter it grow by 1 / (1 - gain)
  std::vector<int> topoRefOrder;   // the reference-order walk of a graph with feedback (the planning order may cut loops at DelayNodes)
  int cycleBlocks = 1;          // blocks per chunk of a graph with feedback (1 unless every loop is cut at a DelayNode)
  bool cycleDelaySplit = true;  // option "cycle_delay_split"
  // the output views of the previous chunk's last segment (and the gains folded into them): when an edit closes a cycle, the block
  // the new stale producer put out LAST is still in those slabs (Context::chunkStaleSeed)
  std::vector<Views> lastViews;
  std::vector<float> lastViewScale;
  std::vector<const float*> lastViewCurve;   // (a GainNode folded into its consumer with a gain CURVE: its output is view x curve)
  int64_t lastViewFrames = 0;
  uint64_t slabGen = 0, lastViewSlabGen = ~0ull;   // (slabGen: bumped when the slab pool is reallocated -- old views dangle)
  uint64_t topoStatsVersion = ~0ull;
  size_t topoStatsSize = 0;
  std::vector<std::pair<int, int>> curveList;   // (node, parameter) pairs with a timeline among the reachable nodes (chunkParamCurves)
  uint64_t curveListEpoch = ~0ull, curveListGraphVersion = ~0ull;
  size_t curveListTopoSize = 0;
  std::vector<int> deviceStateNodes;   // nodes whose pan gains / biquad coefficients live on the device (panOnDevice / coefOnDevice)   // (chunkTopology) delay / stream-source nodes ; convolvers with an impulse response
  int chunkMinDestCh = 0;      // smallest destination channel count over the blocks of the last chunk
  int64_t chunkBlocksDone = 0; // blocks actually executed by the last runChunk
};

void commUniqueId(void* out);   // ga_comm.cpp: ncclGetUniqueId through the run-time loaded RCCL

}  // namespace ga

// ga_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain C++17 restatement of the reference's offline render path (the-byte-bender/GraphAudio,
// GraphAudio.Core), written from reading the C# sources: same pull-model traversal, same per-block
// processing, float32 exactly where the reference is float32, double where it is double, same loop
// and accumulation order, same silent-flag propagation, same block clock.  Every function cites the
// reference file:line it follows.  Build flags (oracle/Makefile): -O3 -mavx2 -ffp-contract=off
// -fno-fast-math, so no fused multiply-adds are introduced (the .NET JIT does not contract either).
//
// PARITY UNPINNED by the reference: the reference ships no tests, golden vectors or fixtures for this
// path and .NET is not available in the build environment, so this
// restatement is pinned only by independent analytic models (numpy/scipy; tests/test_oracle_*.py and
// tests/golden/).  It is labelled "restatement", never "the reference".
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.  The
// product (graphaudio_amd/, libgraphaudio_hip.so) never links, imports or calls it.
//
// The library exports the surface of include/graphaudio_hip.h under the gao_ prefix.

#define GA_FN(name) gao_##name
#include "../include/graphaudio_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

// The reference's MathF.Cos / Sin / Pow are the C runtime's single-precision functions: std::cos(float) etc. here.  A diagnostic
// build (-DGAO_DOUBLE_TRIG, tools/fuzz_libm_class.py -- never the oracle the tests use) evaluates them in double and rounds once,
// which is what the device does: a deviation between device and oracle that vanishes against that build is the last bit of libm.
#ifdef GAO_DOUBLE_TRIG
static inline float gaoCos(float x) { return (float)std::cos((double)x); }
static inline float gaoSin(float x) { return (float)std::sin((double)x); }
static inline float gaoPow(float a, float b) { return (float)std::pow((double)a, (double)b); }
#else
static inline float gaoCos(float x) { return std::cos(x); }
static inline float gaoSin(float x) { return std::sin(x); }
static inline float gaoPow(float a, float b) { return std::pow(a, b); }
#endif

namespace {

constexpr int kBlock = 128;  // AudioBuffer.FramesPerBlock, AudioBuffer.cs:10

struct Err {
  int code;
  std::string msg;
};
[[noreturn]] void fail(int code, const std::string& m) { throw Err{code, m}; }

// ------------------------------------------------------------------------------------------------
// Real FFT, double precision.  The reference vendors Ooura's fftsg (FftFlat/fftsg.cs); only its
// conventions matter: forward X[k] = sum x[n] e^{-2 pi i k n / N}, k = 0..N/2,
// unscaled (RealFourierTransform.cs:62-88); inverse includes the 2/N scale (:46,:129), i.e. it is
// the exact inverse.  This is an independent radix-2 implementation of the same transform.
// ------------------------------------------------------------------------------------------------
struct RealFft {
  int n;                       // real length (256 on the hot path, PartitionedConvolver.cs:40-42)
  int h;                       // n / 2 complex points
  std::vector<double> cw, sw;  // e^{-2 pi i j / h}, j < h/2  (complex FFT twiddles)
  std::vector<double> cr, sr;  // e^{-2 pi i k / n}, k <= h/2.. (real post-pass twiddles), k < h
  std::vector<int> rev;
  std::vector<double> zr, zi;

  explicit RealFft(int len) : n(len), h(len / 2) {
    const double pi = 3.14159265358979323846264338327950288;
    cw.resize(h / 2 > 0 ? h / 2 : 1);
    sw.resize(cw.size());
    for (int j = 0; j < h / 2; j++) {
      cw[j] = std::cos(2.0 * pi * j / h);
      sw[j] = -std::sin(2.0 * pi * j / h);
    }
    cr.resize(h + 1);
    sr.resize(h + 1);
    for (int k = 0; k <= h; k++) {
      cr[k] = std::cos(2.0 * pi * k / n);
      sr[k] = -std::sin(2.0 * pi * k / n);
    }
    rev.resize(h);
    int bits = 0;
    while ((1 << bits) < h) bits++;
    for (int i = 0; i < h; i++) {
      int r = 0;
      for (int b = 0; b < bits; b++)
        if (i & (1 << b)) r |= 1 << (bits - 1 - b);
      rev[i] = r;
    }
    zr.resize(h);
    zi.resize(h);
  }

  // in-place complex FFT of size h on (zr, zi); sign = -1 forward, +1 inverse (unscaled)
  void cfft(int sign) {
    for (int i = 0; i < h; i++) {
      int r = rev[i];
      if (r > i) {
        std::swap(zr[i], zr[r]);
        std::swap(zi[i], zi[r]);
      }
    }
    for (int len = 2; len <= h; len <<= 1) {
      int half = len >> 1, step = h / len;
      for (int base = 0; base < h; base += len) {
        for (int j = 0; j < half; j++) {
          double wr = cw[j * step], wi = sign < 0 ? sw[j * step] : -sw[j * step];
          int a = base + j, b = a + half;
          double tr = zr[b] * wr - zi[b] * wi;
          double ti = zr[b] * wi + zi[b] * wr;
          zr[b] = zr[a] - tr;
          zi[b] = zi[a] - ti;
          zr[a] = zr[a] + tr;
          zi[a] = zi[a] + ti;
        }
      }
    }
  }

  // x[n] real -> re/im[0..h]  (RealFourierTransform.Forward, RealFourierTransform.cs:62-88)
  void forward(const double* x, double* re, double* im) {
    for (int i = 0; i < h; i++) {
      zr[i] = x[2 * i];
      zi[i] = x[2 * i + 1];
    }
    cfft(-1);
    for (int k = 0; k <= h; k++) {
      int a = k % h, b = (h - k) % h;
      double ar = zr[a], ai = zi[a], br = zr[b], bi = -zi[b];  // b = conj(Z[h-k])
      double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);       // even part
      double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);       // (Z - conj Z')/2
      // odd part = -i * w^k * d ,  w^k = cr + i sr
      double pr = dr * cr[k] - di * sr[k], pi_ = dr * sr[k] + di * cr[k];
      re[k] = er + pi_;
      im[k] = ei - pr;
    }
    im[0] = 0.0;  // RealFourierTransform.cs:76-78: DC and Nyquist imaginary parts are exactly zero
    im[h] = 0.0;
  }

  // re/im[0..h] -> x[n] real, scaled so that inverse(forward(x)) == x  (RealFourierTransform.cs:101-131)
  void inverse(const double* re, const double* im, double* x) {
    for (int k = 0; k < h; k++) {
      double ar = re[k], ai = im[k];
      double br = re[h - k], bi = -im[h - k];  // conj(X[h-k])
      if (k == 0) {                             // imaginary parts of DC / Nyquist are ignored by rdft
        ai = 0.0;
        bi = 0.0;
      }
      double er = ar + br, ei = ai + bi;
      double dr = ar - br, di = ai - bi;
      // odd part = i * conj(w^k) * d
      double cwk = cr[k], swk = -sr[k];
      double pr = dr * cwk - di * swk, pi_ = dr * swk + di * cwk;
      zr[k] = er - pi_;
      zi[k] = ei + pr;
    }
    cfft(+1);
    const double scale = 1.0 / n;  // = (2/n) * (1/2): inverseScaling 2/n of RealFourierTransform.cs:46,129
    for (int i = 0; i < h; i++) {
      x[2 * i] = zr[i] * scale;
      x[2 * i + 1] = zi[i] * scale;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// PartitionedConvolver (PartitionedConvolver.cs:12-224)
// ------------------------------------------------------------------------------------------------
struct PartitionedConvolver {
  int blockSize, fftSize, complexCount, nPartitions;
  std::vector<float> irReal, irImag, delayReal, delayImag, overlap, tempReal, tempImag, accReal, accImag;
  std::vector<double> fftInput, re, im, timeOut;
  int writeIndex = 0;
  RealFft fft;

  // PartitionedConvolver.cs:37-63
  PartitionedConvolver(const float* ir, int irLen, int block, bool normalize)
      : blockSize(block), fftSize(block * 2), complexCount(block + 1), fft(block * 2) {
    nPartitions = (int)std::ceil((double)irLen / blockSize);
    size_t total = (size_t)nPartitions * complexCount;
    irReal.assign(total, 0.f);
    irImag.assign(total, 0.f);
    delayReal.assign(total, 0.f);
    delayImag.assign(total, 0.f);
    fftInput.assign(fftSize + 2, 0.0);
    overlap.assign(blockSize, 0.f);
    tempReal.assign(complexCount, 0.f);
    tempImag.assign(complexCount, 0.f);
    accReal.assign(complexCount, 0.f);
    accImag.assign(complexCount, 0.f);
    re.assign(complexCount, 0.0);
    im.assign(complexCount, 0.0);
    timeOut.assign(fftSize, 0.0);
    prepare(ir, irLen, normalize);
  }

  // CalculateNormalizationScale, PartitionedConvolver.cs:93-102
  static float normalizationScale(const float* r, int len) {
    const float GainCalibration = -58;
    const float MinPower = 0.000125f;
    double sumSquared = 0;
    for (int i = 0; i < len; i++) {
      float p = r[i] * r[i];  // float * float -> float, then widened (:98)
      sumSquared += p;
    }
    float power = (float)std::sqrt(sumSquared / len);
    if (std::isnan(power) || std::isinf(power) || power < MinPower) power = MinPower;
    float e = GainCalibration * 0.05f;
    return (1.0f / power) * (float)std::pow(10.0, (double)e);
  }

  // PrepareImpulseResponse, PartitionedConvolver.cs:65-91
  void prepare(const float* ir, int irLen, bool normalize) {
    float scale = 1.0f;
    if (normalize) scale = normalizationScale(ir, irLen);
    std::vector<double> tempTime(fftSize + 2);
    for (int p = 0; p < nPartitions; p++) {
      std::fill(tempTime.begin(), tempTime.end(), 0.0);
      int offset = p * blockSize;
      int len = std::min(blockSize, irLen - offset);
      for (int i = 0; i < len; i++) {
        float v = ir[offset + i] * scale;  // float multiply, then widened to double (:80)
        tempTime[i] = v;
      }
      fft.forward(tempTime.data(), re.data(), im.data());
      size_t po = (size_t)p * complexCount;
      for (int i = 0; i < complexCount; i++) {
        irReal[po + i] = (float)re[i];
        irImag[po + i] = (float)im[i];
      }
    }
  }

  // Process, PartitionedConvolver.cs:104-152
  void process(const float* input, float* output) {
    if (nPartitions == 0)  // Array.Copy into a zero-length delay line throws (:123)
      fail(GA_ERR_INVALID_ARGUMENT, "PartitionedConvolver: empty impulse response");
    for (int i = 0; i < blockSize; i++) fftInput[i] = input[i];
    std::fill(fftInput.begin() + blockSize, fftInput.begin() + 2 * blockSize, 0.0);
    fft.forward(fftInput.data(), re.data(), im.data());
    for (int i = 0; i < complexCount; i++) {
      tempReal[i] = (float)re[i];
      tempImag[i] = (float)im[i];
    }
    size_t cur = (size_t)writeIndex * complexCount;
    std::memcpy(&delayReal[cur], tempReal.data(), sizeof(float) * complexCount);
    std::memcpy(&delayImag[cur], tempImag.data(), sizeof(float) * complexCount);
    spectralConvolution();
    writeIndex--;
    if (writeIndex < 0) writeIndex = nPartitions - 1;
    for (int i = 0; i < complexCount; i++) {
      re[i] = accReal[i];
      im[i] = accImag[i];
    }
    fft.inverse(re.data(), im.data(), timeOut.data());
    for (int i = 0; i < blockSize; i++) {
      output[i] = (float)timeOut[i] + overlap[i];
      overlap[i] = (float)timeOut[i + blockSize];
    }
  }

  // ProcessSpectralConvolution, PartitionedConvolver.cs:154-223.  The AVX body and the scalar tail do
  // the same unfused arithmetic (mul, mul, sub / mul, mul, add, then add into the accumulator), in
  // partition order p = 0..P-1, so one scalar loop restates both.
  void spectralConvolution() {
    std::fill(accReal.begin(), accReal.end(), 0.f);
    std::fill(accImag.begin(), accImag.end(), 0.f);
    const int count = complexCount;
    float* __restrict ar = accReal.data();
    float* __restrict ai = accImag.data();
    for (int p = 0; p < nPartitions; p++) {
      int delayPos = writeIndex + p;
      if (delayPos >= nPartitions) delayPos -= nPartitions;
      const float* __restrict dr = &delayReal[(size_t)delayPos * count];
      const float* __restrict di = &delayImag[(size_t)delayPos * count];
      const float* __restrict hr = &irReal[(size_t)p * count];
      const float* __restrict hi = &irImag[(size_t)p * count];
      for (int i = 0; i < count; i++) {
        float ac = dr[i] * hr[i];
        float bd = di[i] * hi[i];
        float ad = dr[i] * hi[i];
        float bc = di[i] * hr[i];
        ar[i] = ar[i] + (ac - bd);
        ai[i] = ai[i] + (ad + bc);
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// CubicResampler (CubicResampler.cs:19-98)
// ------------------------------------------------------------------------------------------------
struct CubicResampler {
  float S0 = 0, S1 = 0, S2 = 0, S3 = 0;
  double Pos = 0;
  uint8_t Ready = 0;
  void shift(float s) {  // :91-97
    S0 = S1;
    S1 = S2;
    S2 = S3;
    S3 = s;
  }
  void clear() {  // :66-71
    S0 = S1 = S2 = S3 = 0;
    Pos = 0;
    Ready = 0;
  }
  // Process, CubicResampler.cs:26-63
  void process(const float* input, int inLen, float* output, int outLen, double rate, int& consumed, int& produced) {
    int inPos = 0, outPos = 0;
    while (Ready < 4 && inPos < inLen) {
      shift(input[inPos++]);
      Ready++;
    }
    if (Ready < 4) {
      consumed = inPos;
      produced = outPos;
      return;
    }
    while (outPos < outLen) {
      int consume = (int)Pos;
      if (inPos + consume > inLen) break;
      for (int i = 0; i < consume; i++) shift(input[inPos++]);
      Pos -= consume;
      float t = (float)Pos;
      output[outPos++] =
          S1 + t * (0.5f * (S2 - S0) + t * ((S0 - 2.5f * S1 + 2.f * S2 - 0.5f * S3) + t * (0.5f * (S3 - S0) + 1.5f * (S1 - S2))));
      Pos += rate;
    }
    consumed = inPos;
    produced = outPos;
  }
};

// ------------------------------------------------------------------------------------------------
// AudioBuffer (AudioBuffer.cs:8-181): planar float[ch][128] + IsSilent flag
// ------------------------------------------------------------------------------------------------
struct AudioBuffer {
  int channelCount;
  bool silent = true;
  std::vector<float> data;  // [ch][128]
  explicit AudioBuffer(int ch) : channelCount(ch), data((size_t)ch * kBlock, 0.f) {
    if (ch < 1 || ch > 32) fail(GA_ERR_OUT_OF_RANGE, "channelCount");  // AudioBuffer.cs:18
  }
  float* span(int ch) {
    if (ch < 0 || ch >= channelCount) fail(GA_ERR_OUT_OF_RANGE, "channelIndex");  // AudioBuffer.cs:40
    return &data[(size_t)ch * kBlock];
  }
  void clear() {  // :60-67
    std::fill(data.begin(), data.end(), 0.f);
    silent = true;
  }
  void markNonSilent() { silent = false; }  // :73-76
  void copyFrom(AudioBuffer& src) {         // :81-105
    if (src.silent) {
      clear();
      return;
    }
    int m = std::min(channelCount, src.channelCount);
    for (int ch = 0; ch < m; ch++) std::memcpy(span(ch), src.span(ch), sizeof(float) * kBlock);
    for (int ch = m; ch < channelCount; ch++) std::fill(span(ch), span(ch) + kBlock, 0.f);
    silent = false;
  }
};
using BufPtr = std::shared_ptr<AudioBuffer>;
// BufferPool.Rent returns a cleared buffer (BufferPool.cs:66-84); pooling itself has no effect on values.
BufPtr rent(int ch) { return std::make_shared<AudioBuffer>(ch); }

struct Node;
struct Output;
struct Context;

// ------------------------------------------------------------------------------------------------
// AudioNodeInput (AudioNodeInput.cs:11-273)
// ------------------------------------------------------------------------------------------------
struct Input {
  Node* owner;
  int index;
  std::vector<Output*> connected;
  BufPtr buffer;
  bool dirty = true;
  int channelCount = 2;                   // :19
  int interpretation = GA_INTERP_SPEAKERS;  // :20
  int mode = GA_COUNT_MODE_MAX;           // :21
  Input(Node* o, int i) : owner(o), index(i) {}

  void setChannelCount(int c) {  // :41-48
    if (c < 1 || c > 32) fail(GA_ERR_OUT_OF_RANGE, "Channel count must be between 1 and 32");
    channelCount = c;
    dirty = true;
  }
  void addConnection(Output* o) {  // :60-67
    if (std::find(connected.begin(), connected.end(), o) == connected.end()) {
      connected.push_back(o);
      dirty = true;
    }
  }
  void removeConnection(Output* o) {  // :69-73
    auto it = std::find(connected.begin(), connected.end(), o);
    if (it != connected.end()) connected.erase(it);
    dirty = true;
  }
  void disconnectAll();  // :75-83
  void dispose() { buffer.reset(); }  // :88-95
  void ensureBuffer() {  // :170-180
    if (!buffer || dirty) {
      buffer = rent(channelCount);
      dirty = false;
    }
  }
  int computeOutputChannelCount();  // :140-168
  void pull(int blockNumber, double blockTime);  // :100-138
  static void mixBuffer(AudioBuffer& src, AudioBuffer& dst);  // :182-244
};

// AudioNodeOutput (AudioNodeOutput.cs:10-79)
struct Output {
  Node* owner;
  int index;
  std::vector<Input*> connectedInputs;
  BufPtr buffer;  // :16 ; null until the owner's first Process()
  Output(Node* o, int i) : owner(o), index(i) {}
  void connectTo(Input* in);       // :42-52
  void disconnectFrom(Input* in) {  // :54-60
    auto it = std::find(connectedInputs.begin(), connectedInputs.end(), in);
    if (it != connectedInputs.end()) {
      connectedInputs.erase(it);
      in->removeConnection(this);
    }
  }
  void disconnectAll() {  // :62-70
    std::vector<Input*> ins = connectedInputs;
    connectedInputs.clear();
    for (Input* in : ins) in->removeConnection(this);
  }
};

void Input::disconnectAll() {
  std::vector<Output*> outs = connected;
  for (Output* o : outs) o->disconnectFrom(this);
  dirty = true;
}

// ------------------------------------------------------------------------------------------------
// AudioParam (AudioParam.cs:11-392)
// ------------------------------------------------------------------------------------------------
enum EvType { EvSetValue = 0, EvLinearRamp = 1, EvExponentialRamp = 2, EvSetTarget = 3 };  // :369-375
struct Event {  // :360-367
  int type;
  float value = 0, target = 0;
  double time = 0, timeConstant = 0;
};

struct Param {
  Node* owner;
  std::unique_ptr<Input> input;  // modulation input, 1 channel explicit (:68-70)
  float defaultValue, minValue, maxValue;
  bool aRate;
  float value;
  float computed[kBlock];
  std::vector<Event> events;
  double currentTime = 0;

  Param(Node* o, float def, float mn, float mx, bool arate)
      : owner(o), defaultValue(def), minValue(mn), maxValue(mx), aRate(arate), value(def) {
    std::fill(computed, computed + kBlock, 0.f);
    input = std::make_unique<Input>(o, -1);
    input->setChannelCount(1);
    input->mode = GA_COUNT_MODE_EXPLICIT;
  }
  static float clampf(float v, float mn, float mx) {  // Math.Clamp(float)
    if (v < mn) return mn;
    if (v > mx) return mx;
    return v;
  }
  void setValue(float v) {  // Value setter, :37-48: clamps and cancels all events
    value = clampf(v, minValue, maxValue);
    events.clear();
  }
  void addEvent(const Event& e) {  // AddEvent, :333-352: insert after all events with time <= e.time
    size_t lo = 0, hi = events.size();
    while (lo < hi) {
      size_t mid = (lo + hi) >> 1;
      if (e.time < events[mid].time) hi = mid; else lo = mid + 1;
    }
    events.insert(events.begin() + lo, e);
  }
  void setValueAtTime(float v, double t) {  // :252-261
    Event e;
    e.type = EvSetValue;
    e.value = clampf(v, minValue, maxValue);
    e.time = t;
    addEvent(e);
  }
  void linearRamp(float v, double t) {  // :266-275
    Event e;
    e.type = EvLinearRamp;
    e.value = clampf(v, minValue, maxValue);
    e.time = t;
    addEvent(e);
  }
  void exponentialRamp(float v, double t) {  // :280-292
    v = clampf(v, minValue, maxValue);
    if (v <= 0.f) fail(GA_ERR_INVALID_ARGUMENT, "Exponential ramp target must be > 0");
    Event e;
    e.type = EvExponentialRamp;
    e.value = v;
    e.time = t;
    addEvent(e);
  }
  void setTarget(float target, double t, double tc) {  // :297-307
    Event e;
    e.type = EvSetTarget;
    e.target = clampf(target, minValue, maxValue);
    e.time = t;
    e.timeConstant = tc;
    addEvent(e);
  }
  void cancelScheduled(double cancelTime) {  // :312-331
    size_t survivors = 0;
    for (size_t i = 0; i < events.size(); i++) {
      if (events[i].time < cancelTime) survivors++; else break;
    }
    events.resize(survivors);
  }

  static float interpLinear(float v0, double t0, float v1, double t1, double t) {  // :220-225
    double u = (t - t0) / (t1 - t0);
    u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
    float d = v1 - v0;
    return (float)((double)v0 + (double)d * u);
  }
  static float interpExp(float v0, double t0, float v1, double t1, double t) {  // :228-237
    if (v0 <= 0 || v1 <= 0) return interpLinear(v0, t0, v1, t1, t);
    double u = (t - t0) / (t1 - t0);
    u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
    float ratio = v1 / v0;
    return (float)((double)v0 * std::pow((double)ratio, u));
  }
  static float setTargetFromBaseline(const Event& evt, float baseline, double time) {  // :240-247
    double elapsed = time - evt.time;
    if (elapsed <= 0) return baseline;
    double tc = std::max(evt.timeConstant, 0.001);
    float d = baseline - evt.target;
    return (float)((double)evt.target + (double)d * std::exp(-elapsed / tc));
  }
  float valueAtTime(double time) const {  // ComputeValueAtTime, :169-217
    size_t count = events.size();
    if (count == 0) return value;
    float valueAtBoundary = value;
    for (size_t i = 0; i < count; i++) {
      const Event& evt = events[i];
      if (time < evt.time) {
        if (i == 0) return valueAtBoundary;
        const Event& prev = events[i - 1];
        if (evt.type == EvLinearRamp) return interpLinear(prev.value, prev.time, evt.value, evt.time, time);
        if (evt.type == EvExponentialRamp) return interpExp(prev.value, prev.time, evt.value, evt.time, time);
        if (prev.type == EvSetTarget) return setTargetFromBaseline(prev, valueAtBoundary, time);
        return prev.value;
      }
      switch (evt.type) {
        case EvSetValue:
        case EvLinearRamp:
        case EvExponentialRamp: valueAtBoundary = evt.value; break;
        default: break;
      }
    }
    const Event& last = events[count - 1];
    if (last.type == EvSetTarget) return setTargetFromBaseline(last, valueAtBoundary, time);
    return last.value;
  }
  void computeValues(int blockNumber, double blockTime);  // :93-166
};

// ------------------------------------------------------------------------------------------------
// PlayableAudioBuffer (PlayableAudioBuffer.cs:11-175)
// ------------------------------------------------------------------------------------------------
struct PlayableBuffer {
  int channels = 0;
  int64_t length = 0;
  int sampleRate = 0;
  std::vector<std::vector<float>> ch;
};

// ------------------------------------------------------------------------------------------------
// AudioNode (Nodes/AudioNode.cs:10-239)
// ------------------------------------------------------------------------------------------------
struct Node {
  Context* ctx;
  int id;
  int type;
  std::vector<std::unique_ptr<Input>> inputs;
  std::vector<std::unique_ptr<Output>> outputs;
  std::vector<std::unique_ptr<Param>> params;
  int lastProcessedBlock = -1;
  bool isProcessing = false;
  bool disposed = false;
  bool endedRaised = false, endedReported = false;  // scheduled sources (Ended event)

  Node(Context* c, int id_, int type_, int nIn, int nOut) : ctx(c), id(id_), type(type_) {
    for (int i = 0; i < nIn; i++) inputs.push_back(std::make_unique<Input>(this, i));
    for (int i = 0; i < nOut; i++) outputs.push_back(std::make_unique<Output>(this, i));
  }
  virtual ~Node() {}
  Param* createParam(float def, float mn, float mx, bool arate) {  // :52-62
    params.push_back(std::make_unique<Param>(this, def, mn, mx, arate));
    return params.back().get();
  }
  void processInternal(int blockNumber, double blockTime) {  // :152-183
    if (lastProcessedBlock == blockNumber) return;
    if (isProcessing) fail(GA_ERR_CYCLE, "Audio graph cycle detected at node " + std::to_string(id));
    isProcessing = true;
    lastProcessedBlock = blockNumber;
    try {
      for (auto& p : params) p->computeValues(blockNumber, blockTime);
      for (auto& in : inputs) in->pull(blockNumber, blockTime);
      process();
    } catch (...) {
      isProcessing = false;
      throw;
    }
    isProcessing = false;
  }
  virtual void process() = 0;
  virtual void onDispose() {}
  void doDispose() {  // DoDispose, :212-235
    if (disposed) return;
    disposed = true;
    for (auto& o : outputs) o->disconnectAll();
    for (auto& in : inputs) {
      in->disconnectAll();
      in->dispose();
    }
    for (auto& p : params) {  // AudioParam.Dispose, AudioParam.cs:354-358
      p->input->disconnectAll();
      p->input->dispose();
    }
    onDispose();
  }
};

void Output::connectTo(Input* in) {
  if (in->owner == owner) fail(GA_ERR_INVALID_OPERATION, "Cannot connect a node to itself");
  if (std::find(connectedInputs.begin(), connectedInputs.end(), in) == connectedInputs.end()) {
    connectedInputs.push_back(in);
    in->addConnection(this);
  }
}

int Input::computeOutputChannelCount() {
  switch (mode) {
    case GA_COUNT_MODE_EXPLICIT: return channelCount;
    case GA_COUNT_MODE_CLAMPED_MAX: {
      int maxCh = 0;
      for (Output* o : connected)
        if (o->buffer) maxCh = std::max(maxCh, o->buffer->channelCount);
      return std::min(maxCh == 0 ? channelCount : maxCh, channelCount);
    }
    default: {
      int mx = channelCount;
      for (Output* o : connected)
        if (o->buffer) mx = std::max(mx, o->buffer->channelCount);
      return mx;
    }
  }
}

void Input::pull(int blockNumber, double blockTime) {
  if (connected.empty()) {
    ensureBuffer();
    buffer->clear();
    return;
  }
  int outCh = computeOutputChannelCount();  // looks at upstream buffers of the PREVIOUS block
  ensureBuffer();
  if (buffer->channelCount != outCh) buffer = rent(outCh);
  buffer->clear();
  bool mixedAny = false;
  for (size_t i = 0; i < connected.size(); i++) {
    Output* o = connected[i];
    o->owner->processInternal(blockNumber, blockTime);  // AudioNodeOutput.ProcessIfNeeded, AudioNodeOutput.cs:75
    AudioBuffer* src = o->buffer.get();
    if (src && !src->silent) {
      mixBuffer(*src, *buffer);
      mixedAny = true;
    }
  }
  if (mixedAny) buffer->markNonSilent();
}

void Input::mixBuffer(AudioBuffer& src, AudioBuffer& dst) {
  int sc = src.channelCount, dc = dst.channelCount;
  if (sc == dc) {
    for (int ch = 0; ch < sc; ch++) {
      float* s = src.span(ch);
      float* d = dst.span(ch);
      for (int i = 0; i < kBlock; i++) d[i] += s[i];
    }
  } else if (sc == 1 && dc > 1) {
    float* s = src.span(0);
    for (int ch = 0; ch < dc; ch++) {
      float* d = dst.span(ch);
      for (int i = 0; i < kBlock; i++) d[i] += s[i];
    }
  } else if (sc > 1 && dc == 1) {
    float* d = dst.span(0);
    float scale = 1.0f / std::sqrt((float)sc);  // 1.0f / MathF.Sqrt(srcChannels), :217
    for (int i = 0; i < kBlock; i++) {
      float sum = 0;
      for (int ch = 0; ch < sc; ch++) sum += src.span(ch)[i];
      d[i] += sum * scale;
    }
  } else {
    int m = std::min(sc, dc);
    for (int ch = 0; ch < m; ch++) {
      float* s = src.span(ch);
      float* d = dst.span(ch);
      for (int i = 0; i < kBlock; i++) d[i] += s[i];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// AudioContextBase + OfflineAudioContext (AudioContextBase.cs:14-306, OfflineAudioContext.cs:8-158)
// ------------------------------------------------------------------------------------------------
struct DestinationNode;
struct Context {
  int sampleRate;
  int currentBlock = 0;
  double currentTime = 0.0;
  bool disposed = false;
  bool renderThreadLatched = false;  // _renderThreadId != -1 (:59-62); the API is single threaded
  bool inRender = false;
  std::deque<std::function<void()>> pending;
  std::vector<std::unique_ptr<Node>> nodes;
  std::vector<std::unique_ptr<PlayableBuffer>> buffers;
  std::string lastError;
  // OfflineAudioContext leftover-frame cache (:53-75,:89-100)
  std::vector<std::vector<float>> cache;
  int cachedFrames = 0;

  explicit Context(int sr) : sampleRate(sr) {}
  DestinationNode* destination();

  void executeOrPost(std::function<void()> cmd) {  // ExecuteOrPost, AudioContextBase.cs:291-305
    if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
    if (renderThreadLatched && !inRender) cmd(); else pending.push_back(std::move(cmd));
  }
  void post(std::function<void()> cmd) {  // Post, :266-270
    if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
    pending.push_back(std::move(cmd));
  }
  void drainCommands() {  // :272-284 -- exceptions thrown by commands are swallowed
    while (!pending.empty()) {
      auto cmd = std::move(pending.front());
      pending.pop_front();
      try {
        cmd();
      } catch (...) {
      }
    }
  }
  AudioBuffer* processBlock();  // :52-81
};

void Param::computeValues(int blockNumber, double blockTime) {
  currentTime = blockTime;
  bool hasMod = !input->connected.empty();
  if (hasMod) input->pull(blockNumber, blockTime);
  AudioBuffer* mb = hasMod ? input->buffer.get() : nullptr;
  if (aRate) {  // ComputeARate, :114-141
    double deltaTime = 1.0 / owner->ctx->sampleRate;
    for (int i = 0; i < kBlock; i++) {
      double sampleTime = currentTime + i * deltaTime;
      float intrinsic = valueAtTime(sampleTime);
      if (hasMod && mb && !mb->silent)
        computed[i] = clampf(intrinsic + mb->span(0)[i], minValue, maxValue);
      else
        computed[i] = intrinsic;
    }
  } else {  // ComputeKRate, :144-166
    float intrinsic = valueAtTime(currentTime);
    float v = intrinsic;
    if (hasMod && mb && !mb->silent) v = clampf(intrinsic + mb->span(0)[0], minValue, maxValue);
    std::fill(computed, computed + kBlock, v);
  }
}

// ---- AudioDestinationNode (Nodes/AudioDestinationNode.cs:9-75) ----
struct DestinationNode : Node {
  BufPtr outputBuffer;
  explicit DestinationNode(Context* c) : Node(c, 0, GA_NODE_DESTINATION, 1, 0) { inputs[0]->setChannelCount(2); }
  void process() override {  // :42-64 -- the destination aliases its input buffer
    outputBuffer = inputs[0]->buffer;
    if (!outputBuffer) {
      outputBuffer = rent(inputs[0]->channelCount);
      outputBuffer->clear();
    }
  }
  void onDispose() override { outputBuffer.reset(); }
};
DestinationNode* Context::destination() { return static_cast<DestinationNode*>(nodes[0].get()); }

AudioBuffer* Context::processBlock() {
  if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
  drainCommands();
  renderThreadLatched = true;
  int nextBlock = currentBlock + 1;
  currentBlock = nextBlock;
  double blockTime = currentTime;
  inRender = true;
  try {
    destination()->processInternal(nextBlock, blockTime);
  } catch (...) {
    inRender = false;
    throw;
  }
  inRender = false;
#ifdef GAO_DEBUG_DUMP   // diagnostic build only (tools/fuzz_node_dump.py): every node's output buffers of the blocks GAO_DUMP_FROM..GAO_DUMP_TO
  if (const char* f0 = getenv("GAO_DUMP_FROM")) {
    const int from = atoi(f0), to = getenv("GAO_DUMP_TO") ? atoi(getenv("GAO_DUMP_TO")) : from;
    if (nextBlock - 1 >= from && nextBlock - 1 <= to) {
      FILE* f = fopen(getenv("GAO_DUMP_FILE") ? getenv("GAO_DUMP_FILE") : "gao_dump.txt", "a");
      if (f) {
        for (auto& n : nodes) {
          if (!n) continue;
          for (auto& o : n->outputs) {
            if (!o->buffer) continue;
            fprintf(f, "block %d node %d type %d processed %d out %d ch %d silent %d\n", nextBlock - 1, n->id, n->type, n->lastProcessedBlock == nextBlock,
                    o->index, o->buffer->channelCount, (int)o->buffer->silent);
            for (int ch = 0; ch < o->buffer->channelCount; ch++) {
              const float* p = o->buffer->span(ch);
              for (int i = 0; i < kBlock; i++) fprintf(f, "%.9g ", p[i]);
              fprintf(f, "\n");
            }
          }
        }
        fclose(f);
      }
    }
  }
#endif
  double increment = (double)kBlock / sampleRate;
  currentTime = blockTime + increment;  // accumulated, not block * 128 / sr (:78-79)
  return destination()->outputBuffer.get();
}

// ---- GainNode (Nodes/GainNode.cs:9-71) ----
struct GainNode : Node {
  BufPtr out;
  explicit GainNode(Context* c, int id) : Node(c, id, GA_NODE_GAIN, 1, 1) {
    createParam(1.0f, std::numeric_limits<float>::lowest(), std::numeric_limits<float>::max(), true);
  }
  void process() override {  // :29-61
    AudioBuffer* in = inputs[0]->buffer.get();
    if (!out || out->channelCount != in->channelCount) out = rent(in->channelCount);
    if (in->silent) {
      out->clear();
      outputs[0]->buffer = out;
      return;
    }
    const float* g = params[0]->computed;
    out->copyFrom(*in);
    for (int ch = 0; ch < in->channelCount; ch++) {
      float* s = out->span(ch);
      for (int i = 0; i < kBlock; i++) s[i] *= g[i];
    }
    outputs[0]->buffer = out;
  }
  void onDispose() override { out.reset(); }
};

// ---- BiQuadFilterNode (Nodes/BiQuadFilterNode.cs:10-298) ----
struct BiquadNode : Node {
  int filterType = GA_FILTER_LOWPASS;
  float lastFrequency = 1000.f, lastQ = 1.0f, lastGain = 0.f;  // never updated after construction (:13-15)
  float b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0;
  bool coefficientsDirty = true;
  struct St {
    float W1 = 0, W2 = 0;
  };
  std::vector<St> states;
  BufPtr out;
  explicit BiquadNode(Context* c, int id) : Node(c, id, GA_NODE_BIQUAD, 1, 1) {
    states.resize(2);
    createParam(1000.f, 1.f, c->sampleRate / 2.f, true);  // frequency (:63-68)
    createParam(1.0f, 0.001f, 1000.f, true);              // Q (:70-75)
    createParam(0.f, -60.f, 60.f, false);                 // gain dB, k-rate (:77-82)
    updateCoefficients(lastFrequency, lastQ, lastGain);
  }
  void process() override {  // :87-147
    const float* freqValues = params[0]->computed;
    const float* qValues = params[1]->computed;
    float gainDb = params[2]->computed[0];
    AudioBuffer* in = inputs[0]->buffer.get();
    int channels = in->channelCount;
    if (channels > (int)states.size()) states.resize(channels);  // EnsureChannelStates, :260-270
    if (!out || out->channelCount != channels) out = rent(channels);
    if (in->silent) {  // :103-108 -- state is NOT advanced
      out->clear();
      outputs[0]->buffer = out;
      return;
    }
    float lastB0 = b0, lastB1 = b1, lastB2 = b2, lastA1 = a1, lastA2 = a2;
    float usedFreq = lastFrequency;
    float usedQ = lastQ;
    float usedGain = gainDb;
    const float nyq = ctx->sampleRate / 2.f;
    for (int ch = 0; ch < channels; ch++) {
      const float* x = in->span(ch);
      float* y = out->span(ch);
      St& st = states[ch];
      for (int i = 0; i < kBlock; i++) {
        float f = Param::clampf(freqValues[i], 1.f, nyq);
        float q = std::max(0.001f, qValues[i]);
        if (coefficientsDirty || std::fabs(f - usedFreq) > 0.001f || std::fabs(q - usedQ) > 0.0001f ||
            std::fabs(gainDb - usedGain) > 0.001f) {
          updateCoefficients(f, q, gainDb);
          usedFreq = f;
          usedQ = q;
          usedGain = gainDb;
          coefficientsDirty = false;
          lastB0 = b0;
          lastB1 = b1;
          lastB2 = b2;
          lastA1 = a1;
          lastA2 = a2;
        }
        float xv = x[i];
        float w = xv - lastA1 * st.W1 - lastA2 * st.W2;
        float yv = lastB0 * w + lastB1 * st.W1 + lastB2 * st.W2;
        st.W2 = st.W1;
        st.W1 = w;
        y[i] = yv;
      }
    }
    out->markNonSilent();
    outputs[0]->buffer = out;
  }
  void updateCoefficients(float frequency, float q, float gain) {  // :149-258
    const float PI = 3.14159274f;  // MathF.PI
    float w0 = 2.f * PI * frequency / ctx->sampleRate;
    float cosW0 = gaoCos(w0);
    float sinW0 = gaoSin(w0);
    float alpha = sinW0 / (2.f * q);
    float a0, A1, A2, B0, B1, B2;
    switch (filterType) {
      case GA_FILTER_LOWPASS:
        B0 = (1.f - cosW0) / 2.f; B1 = 1.f - cosW0; B2 = (1.f - cosW0) / 2.f;
        a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha;
        break;
      case GA_FILTER_HIGHPASS:
        B0 = (1.f + cosW0) / 2.f; B1 = -(1.f + cosW0); B2 = (1.f + cosW0) / 2.f;
        a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha;
        break;
      case GA_FILTER_BANDPASS:
        B0 = alpha; B1 = 0.f; B2 = -alpha;
        a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha;
        break;
      case GA_FILTER_NOTCH:
        B0 = 1.f; B1 = -2.f * cosW0; B2 = 1.f;
        a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha;
        break;
      case GA_FILTER_ALLPASS:
        B0 = 1.f - alpha; B1 = -2.f * cosW0; B2 = 1.f + alpha;
        a0 = 1.f + alpha; A1 = -2.f * cosW0; A2 = 1.f - alpha;
        break;
      case GA_FILTER_PEAKING: {
        float A = gaoPow(10.f, gain / 40.f);
        B0 = 1.f + alpha * A; B1 = -2.f * cosW0; B2 = 1.f - alpha * A;
        a0 = 1.f + alpha / A; A1 = -2.f * cosW0; A2 = 1.f - alpha / A;
        break;
      }
      case GA_FILTER_LOWSHELF: {
        float A = gaoPow(10.f, gain / 40.f);
        float sqrtA = std::sqrt(A);
        float beta = sqrtA / q;  // non-standard RBJ variant, kept as in the reference (:221)
        B0 = A * ((A + 1.f) - (A - 1.f) * cosW0 + beta * sinW0);
        B1 = 2.f * A * ((A - 1.f) - (A + 1.f) * cosW0);
        B2 = A * ((A + 1.f) - (A - 1.f) * cosW0 - beta * sinW0);
        a0 = (A + 1.f) + (A - 1.f) * cosW0 + beta * sinW0;
        A1 = -2.f * ((A - 1.f) + (A + 1.f) * cosW0);
        A2 = (A + 1.f) + (A - 1.f) * cosW0 - beta * sinW0;
        break;
      }
      case GA_FILTER_HIGHSHELF: {
        float A = gaoPow(10.f, gain / 40.f);
        float sqrtA = std::sqrt(A);
        float beta = sqrtA / q;
        B0 = A * ((A + 1.f) + (A - 1.f) * cosW0 + beta * sinW0);
        B1 = -2.f * A * ((A - 1.f) + (A + 1.f) * cosW0);
        B2 = A * ((A + 1.f) + (A - 1.f) * cosW0 - beta * sinW0);
        a0 = (A + 1.f) - (A - 1.f) * cosW0 + beta * sinW0;
        A1 = 2.f * ((A - 1.f) - (A + 1.f) * cosW0);
        A2 = (A + 1.f) - (A - 1.f) * cosW0 - beta * sinW0;
        break;
      }
      default:
        B0 = 1.f; B1 = 0.f; B2 = 0.f; a0 = 1.f; A1 = 0.f; A2 = 0.f;
        break;
    }
    b0 = B0 / a0;
    b1 = B1 / a0;
    b2 = B2 / a0;
    a1 = A1 / a0;
    a2 = A2 / a0;
  }
  void onDispose() override { out.reset(); }
};

// ---- ConvolverNode (Nodes/ConvolverNode.cs:10-176) ----
struct ConvolverNode : Node {
  PlayableBuffer* buffer = nullptr;
  std::shared_ptr<std::vector<std::unique_ptr<PartitionedConvolver>>> convolvers;
  BufPtr out;
  int effectiveOutputChannels = 0;
  bool isTrueStereo = false;
  bool normalize = true;         // :87
  bool enableTrueStereo = true;  // :95
  explicit ConvolverNode(Context* c, int id) : Node(c, id, GA_NODE_CONVOLVER, 1, 1) {}

  void setBuffer(PlayableBuffer* value) {  // Buffer setter, :25-79
    if (buffer == value) return;
    if (!value) {
      ctx->post([this]() {
        buffer = nullptr;
        convolvers.reset();
        effectiveOutputChannels = 0;
        isTrueStereo = false;
        inputs[0]->mode = GA_COUNT_MODE_MAX;
      });
      return;
    }
    if (value->sampleRate != ctx->sampleRate)
      fail(GA_ERR_INVALID_OPERATION, "Impulse response buffer sample rate must match the audio context sample rate");
    auto convs = std::make_shared<std::vector<std::unique_ptr<PartitionedConvolver>>>();
    for (int i = 0; i < value->channels; i++)  // eager, with the CURRENT Normalize (:51-56)
      convs->push_back(std::make_unique<PartitionedConvolver>(value->ch[i].data(), (int)value->length, kBlock, normalize));
    ctx->post([this, value, convs]() {  // :58-77
      buffer = value;
      convolvers = convs;
      int channels = value->channels;
      isTrueStereo = (channels == 4 && enableTrueStereo);  // EnableTrueStereo read when the command runs (:64)
      effectiveOutputChannels = isTrueStereo ? 2 : channels;
      inputs[0]->setChannelCount(isTrueStereo ? 2 : channels);
      inputs[0]->mode = GA_COUNT_MODE_EXPLICIT;
    });
  }
  void process() override {  // :102-155
    AudioBuffer* in = inputs[0]->buffer.get();
    auto convs = convolvers;
    if (!convs) {
      int ch = in->channelCount;
      if (!out || out->channelCount != ch) out = rent(ch);
      out->clear();
      outputs[0]->buffer = out;
      return;
    }
    if (!out || out->channelCount != effectiveOutputChannels) out = rent(effectiveOutputChannels);
    if (isTrueStereo) {
      float temp1[kBlock], temp2[kBlock];
      float* outL = out->span(0);
      float* outR = out->span(1);
      (*convs)[0]->process(in->span(0), temp1);
      (*convs)[2]->process(in->span(1), temp2);
      for (int i = 0; i < kBlock; i++) outL[i] = temp1[i] + temp2[i];
      (*convs)[1]->process(in->span(0), temp1);
      (*convs)[3]->process(in->span(1), temp2);
      for (int i = 0; i < kBlock; i++) outR[i] = temp1[i] + temp2[i];
    } else {
      for (int ch = 0; ch < effectiveOutputChannels; ch++) (*convs)[ch]->process(in->span(ch), out->span(ch));
    }
    out->markNonSilent();  // always, even for silent input (:153)
    outputs[0]->buffer = out;
  }
  void onDispose() override {
    out.reset();
    convolvers.reset();
    buffer = nullptr;
  }
};

// ---- AudioBufferSourceNode (Nodes/AudioBufferSourceNode.cs:13-415) ----
struct SourceNode : Node {
  PlayableBuffer* buffer = nullptr;
  bool hasStarted = false, hasStopped = false;
  double startTime = std::nan(""), stopTime = std::nan("");
  double offset = 0, duration = std::numeric_limits<double>::infinity();
  int64_t playbackPosition = 0;
  bool loop = false;
  double loopStart = 0, loopEnd = 0;
  BufPtr out;
  std::vector<CubicResampler> resamplers;
  std::vector<float> loopWrap;

  explicit SourceNode(Context* c, int id) : Node(c, id, GA_NODE_BUFFER_SOURCE, 0, 1) {
    createParam(1.f, 0.001f, 1000.f, false);  // playbackRate, k-rate (:76)
  }
  void start(double when, double off, double dur) {  // :79-114
    ctx->executeOrPost([this, when, off, dur]() {
      if (hasStarted) fail(GA_ERR_INVALID_OPERATION, "AudioBufferSourceNode can only be started once.");
      if (!buffer) fail(GA_ERR_INVALID_OPERATION, "Cannot start without a buffer set");
      hasStarted = true;
      startTime = std::max(0.0, when);
      offset = std::max(0.0, off);
      duration = dur;
      playbackPosition = (int64_t)(offset * buffer->sampleRate);
      for (auto& r : resamplers) r.clear();
      if (!(std::isinf(dur) && dur > 0) && dur >= 0) {
        stopTime = startTime + dur;
        hasStopped = true;
      }
    });
  }
  void stop(double when) {  // :116-129
    ctx->executeOrPost([this, when]() {
      if (hasStopped) return;
      double at = std::max(0.0, when);
      stopTime = std::isnan(stopTime) ? at : std::min(stopTime, at);
      hasStopped = true;
    });
  }
  void produceSilence() {  // :391-402
    if (!out || out->channelCount != 1) out = rent(1);
    out->clear();
    outputs[0]->buffer = out;
  }
  void process() override {  // :131-376
    double t0 = ctx->currentTime;
    double t1 = t0 + (double)kBlock / ctx->sampleRate;
    bool shouldPlay = false;
    if (hasStarted) {
      if (t1 > startTime && (std::isnan(stopTime) || t0 < stopTime)) shouldPlay = true;
    }
    if (!shouldPlay) {
      produceSilence();
      return;
    }
    if (!buffer) {
      produceSilence();
      return;
    }
    int outputChannels = buffer->channels;
    if (!out || out->channelCount != outputChannels) out = rent(outputChannels);

    float playbackRate = params[0]->computed[0];
    const int framesToRender = kBlock;
    double sampleRateRatio = buffer->sampleRate / (double)ctx->sampleRate;
    double effectiveRate = sampleRateRatio * playbackRate;
    const int64_t bufLen = buffer->length;

    int64_t loopStartFrame = (int64_t)(loopStart * buffer->sampleRate);
    int64_t loopEndFrame = loopEnd > 0 ? (int64_t)(loopEnd * buffer->sampleRate) : bufLen;
    loopEndFrame = std::min(loopEndFrame, bufLen);
    loopStartFrame = std::min(loopStartFrame, loopEndFrame);
    int64_t durationEndFrame = duration < std::numeric_limits<double>::infinity()
                                   ? (int64_t)(offset * buffer->sampleRate) + (int64_t)(duration * buffer->sampleRate)
                                   : bufLen;
    durationEndFrame = std::min(durationEndFrame, bufLen);
    bool hasMoreData = false;

    if (effectiveRate == 1.0) {  // :186-235
      for (int ch = 0; ch < outputChannels; ch++) {
        const float* channelData = buffer->ch[ch].data();
        float* o = out->span(ch);
        int64_t pos = playbackPosition;
        int outIdx = 0;
        while (outIdx < framesToRender) {
          if (loop && pos >= loopEndFrame) pos = loopStartFrame;
          if (pos >= durationEndFrame && !loop) {
            std::fill(o + outIdx, o + kBlock, 0.f);
            break;
          }
          int64_t endFrame = loop ? loopEndFrame : std::min(durationEndFrame, bufLen);
          int available = (int)std::min<int64_t>(endFrame - pos, framesToRender - outIdx);
          if (available <= 0) {
            std::fill(o + outIdx, o + kBlock, 0.f);
            break;
          }
          std::memcpy(o + outIdx, channelData + pos, sizeof(float) * available);
          pos += available;
          outIdx += available;
          hasMoreData = true;
        }
      }
      playbackPosition += framesToRender;
      if (loop && playbackPosition >= loopEndFrame) {
        int64_t loopLength = loopEndFrame - loopStartFrame;
        if (loopLength > 0) {
          int64_t overshoot = playbackPosition - loopEndFrame;
          playbackPosition = loopStartFrame + (overshoot % loopLength);
        }
      }
    } else {  // :236-358
      if ((int)resamplers.size() != outputChannels) {
        resamplers.assign(outputChannels, CubicResampler());
      }
      if (loopWrap.empty()) loopWrap.assign(512, 0.f);
      int64_t totalInputConsumed = 0;
      for (int ch = 0; ch < outputChannels; ch++) {
        const float* channelData = buffer->ch[ch].data();
        float* o = out->span(ch);
        int64_t pos = playbackPosition;
        int64_t inputConsumedThisChannel = 0;
        CubicResampler& rs = resamplers[ch];
        int outIdx = 0;
        while (outIdx < framesToRender) {
          if (loop && pos >= loopEndFrame) pos = loopStartFrame;
          if (pos >= durationEndFrame && !loop) {
            std::fill(o + outIdx, o + kBlock, 0.f);
            break;
          }
          int64_t endFrame = loop ? loopEndFrame : std::min(durationEndFrame, bufLen);
          int available = (int)std::min<int64_t>(endFrame - pos, bufLen - pos);
          if (available <= 0) {
            if (loop) {
              pos = loopStartFrame;
              inputConsumedThisChannel = pos - playbackPosition;
              continue;
            } else {
              std::fill(o + outIdx, o + kBlock, 0.f);
              break;
            }
          }
          int consumed = 0, produced = 0;
          if (loop && pos + available >= loopEndFrame - 4) {  // :297-314
            int64_t loopLength = loopEndFrame - loopStartFrame;
            int samplesFromEnd = (int)(loopEndFrame - pos);
            int samplesNeeded = std::min(framesToRender - outIdx + 4, (int)loopWrap.size());
            int copied = 0;
            for (int i = 0; i < samplesFromEnd && copied < samplesNeeded; i++) loopWrap[copied++] = channelData[pos + i];
            for (int64_t i = 0; copied < samplesNeeded && i < loopLength; i++) loopWrap[copied++] = channelData[loopStartFrame + i];
            rs.process(loopWrap.data(), copied, o + outIdx, kBlock - outIdx, effectiveRate, consumed, produced);
          } else {
            rs.process(channelData + pos, available, o + outIdx, kBlock - outIdx, effectiveRate, consumed, produced);
          }
          if (produced > 0) hasMoreData = true;
          int64_t newPos = pos + consumed;
          if (loop && newPos >= loopEndFrame) {
            int64_t overshoot = newPos - loopEndFrame;
            newPos = loopStartFrame + overshoot;
          }
          inputConsumedThisChannel += (newPos >= pos) ? (newPos - pos) : (loopEndFrame - pos + newPos - loopStartFrame);
          pos = newPos;
          outIdx += produced;
          if (consumed == 0 && produced == 0) {
            std::fill(o + outIdx, o + kBlock, 0.f);
            break;
          }
        }
        if (ch == 0) totalInputConsumed = inputConsumedThisChannel;
      }
      playbackPosition += totalInputConsumed;
      if (loop && playbackPosition >= loopEndFrame) {
        int64_t loopLength = loopEndFrame - loopStartFrame;
        if (loopLength > 0) {
          int64_t overshoot = playbackPosition - loopEndFrame;
          playbackPosition = loopStartFrame + (overshoot % loopLength);
        }
      }
    }

    if (!hasMoreData || (!loop && playbackPosition >= durationEndFrame)) {  // :360-368 whole block dropped
      out->clear();
      if (std::isnan(stopTime)) {
        stopTime = t1;
        hasStopped = true;
      }
    } else {
      out->markNonSilent();
    }
    outputs[0]->buffer = out;
    // TryRaiseEndedEvent(t1), :378-389
    if (hasStarted && !std::isnan(stopTime) && t1 >= stopTime) {
      if (!endedRaised) {
        endedRaised = true;
        ctx->executeOrPost([this]() { doDispose(); });  // Dispose() queues: we are inside the render
      }
    }
  }
  void onDispose() override {
    out.reset();
    buffer = nullptr;
  }
};

// ---- ChannelSplitterNode (Nodes/ChannelSplitterNode.cs:9-71) ----
struct SplitterNode : Node {
  std::vector<BufPtr> outs;
  SplitterNode(Context* c, int id, int nOut) : Node(c, id, GA_NODE_CHANNEL_SPLITTER, 1, nOut), outs(nOut) {}
  void process() override {  // :24-59
    AudioBuffer* in = inputs[0]->buffer.get();
    const int n = (int)outs.size();
    if (!in || in->silent) {
      for (int i = 0; i < n; i++) {
        if (!outs[i]) outs[i] = rent(1);
        outs[i]->clear();
        outputs[i]->buffer = outs[i];
      }
      return;
    }
    for (int i = 0; i < n; i++) {
      if (!outs[i]) outs[i] = rent(1);
      if (i < in->channelCount) {
        std::memcpy(outs[i]->span(0), in->span(i), sizeof(float) * kBlock);  // CopyChannelFrom marks non-silent (AudioBuffer.cs:110-121)
        outs[i]->silent = false;
      } else {
        outs[i]->clear();
      }
      outputs[i]->buffer = outs[i];
    }
  }
  void onDispose() override { for (auto& o : outs) o.reset(); }
};

// ---- ChannelMergerNode (Nodes/ChannelMergerNode.cs:9-65) ----
struct MergerNode : Node {
  BufPtr out;
  int nIn;
  MergerNode(Context* c, int id, int nIn_) : Node(c, id, GA_NODE_CHANNEL_MERGER, nIn_, 1), nIn(nIn_) {}
  void process() override {  // :23-55
    if (!out || out->channelCount != nIn) out = rent(nIn);
    out->clear();
    bool hasAudio = false;
    for (int i = 0; i < nIn; i++) {
      AudioBuffer* in = inputs[i]->buffer.get();
      if (in && !in->silent) {
        std::memcpy(out->span(i), in->span(0), sizeof(float) * kBlock);  // source channel 0 only (:38-43)
        out->silent = false;
        hasAudio = true;
      }
    }
    if (hasAudio) out->markNonSilent();
    outputs[0]->buffer = out;
  }
  void onDispose() override { out.reset(); }
};

// scheduling shared by ConstantSourceNode and OscillatorNode (ConstantSourceNode.cs:44-74,83-110; OscillatorNode.cs:54-88,97-118)
struct ScheduledNode : Node {
  bool hasStarted = false, hasStopped = false;
  double startTime = std::nan(""), stopTime = std::nan("");
  BufPtr out;
  using Node::Node;
  void stop(double when) {
    ctx->executeOrPost([this, when]() {
      if (hasStopped) return;
      double at = std::max(0.0, when);
      stopTime = std::isnan(stopTime) ? at : std::min(stopTime, at);
      hasStopped = true;
    });
  }
  // returns shouldPlay and the frame window of this block
  bool window(double t0, double t1, int& startFrame, int& endFrame) {
    startFrame = 0;
    endFrame = kBlock;
    if (!hasStarted) return false;
    if (!(t1 > startTime && (std::isnan(stopTime) || t0 < stopTime))) return false;
    if (t0 < startTime && startTime < t1)
      startFrame = (int)std::clamp(std::ceil((startTime - t0) * ctx->sampleRate), 0.0, (double)kBlock);
    if (!std::isnan(stopTime) && t0 < stopTime && stopTime < t1)
      endFrame = (int)std::clamp(std::floor((stopTime - t0) * ctx->sampleRate), 0.0, (double)kBlock);
    return true;
  }
  void tryRaiseEnded(double blockEndTime) {  // ConstantSourceNode.cs:143-152, OscillatorNode.cs:160-169
    if (hasStarted && hasStopped && !endedRaised && !std::isnan(stopTime) && blockEndTime >= stopTime) {
      endedRaised = true;
      ctx->executeOrPost([this]() { doDispose(); });
    }
  }
  void onDispose() override { out.reset(); }
};

// ---- AudioStreamNodeBase (GraphAudio.IO/AudioStreamSourceNodeBase.cs:19-329) with the queue filled by the host ----
struct StreamNode : Node {
  std::deque<PlayableBuffer*> queued, processed;   // _queuedBuffers / _processedBuffers (:21-22)
  PlayableBuffer* current = nullptr;                // _currentBuffer
  int64_t currentPos = 0;                           // _currentBufferPosition
  int lastRate = 0;                                 // _lastBufferSampleRate
  BufPtr out;
  std::vector<CubicResampler> resamplers;
  bool haveResamplers = false;                      // `_resamplers is null` until the first playing block
  int state = GA_STREAM_STOPPED;

  StreamNode(Context* c, int id) : Node(c, id, GA_NODE_STREAM_SOURCE, 0, 1) {
    createParam(1.f, 0.001f, 1000.f, false);  // playbackRate, k-rate (:66)
  }
  void setState(int st) {  // State setter (:37-49): immediate, not a posted command
    const int old = state;
    state = st;
    if (st == GA_STREAM_STOPPED && old != GA_STREAM_STOPPED) flushToProcessed();
  }
  void flushToProcessed() {  // :95-116
    if (current) processed.push_back(current);
    current = nullptr;
    while (!queued.empty()) {
      processed.push_back(queued.front());
      queued.pop_front();
    }
    if (haveResamplers)
      for (auto& r : resamplers) r.clear();
    currentPos = 0;
    lastRate = 0;
  }
  void produceSilence() {  // :303-313
    if (!out || out->channelCount != 1) out = rent(1);
    out->clear();
    outputs[0]->buffer = out;
  }
  void process() override {  // :132-301
    if (state != GA_STREAM_PLAYING) {
      produceSilence();
      return;
    }
    if (!current) {
      if (queued.empty()) {
        produceSilence();
        return;
      }
      current = queued.front();
      queued.pop_front();
      currentPos = 0;
    }
    const int channelCount = current->channels;
    if (!out || out->channelCount != channelCount) out = rent(channelCount);
    const int framesToRender = kBlock;
    int framesRendered = 0;
    if (!haveResamplers || (int)resamplers.size() != channelCount) {
      resamplers.assign(channelCount, CubicResampler());
      haveResamplers = true;
    }
    auto clearRest = [&]() {
      for (int ch = 0; ch < channelCount; ch++) std::fill(out->span(ch) + framesRendered, out->span(ch) + kBlock, 0.f);
    };
    while (framesRendered < framesToRender) {
      if (!current) {
        if (queued.empty()) {
          clearRest();
          break;
        }
        current = queued.front();
        queued.pop_front();
        currentPos = 0;
        if (current->channels != channelCount) {  // :189-198: back to the END of the queue
          clearRest();
          queued.push_back(current);
          current = nullptr;
          break;
        }
      }
      const int bufferSampleRate = current->sampleRate;
      if (bufferSampleRate != lastRate && lastRate != 0 && haveResamplers)
        for (auto& r : resamplers) r.clear();
      lastRate = bufferSampleRate;
      const float playbackRate = params[0]->computed[0];
      const double sampleRateRatio = bufferSampleRate / (double)ctx->sampleRate;
      const double effectiveRate = sampleRateRatio * playbackRate;
      if (effectiveRate == 1.0) {
        const int remainingInBuffer = (int)current->length - (int)currentPos;
        const int remainingInOutput = framesToRender - framesRendered;
        const int framesToCopy = std::min(remainingInBuffer, remainingInOutput);
        for (int ch = 0; ch < channelCount; ch++)
          std::copy(current->ch[ch].begin() + currentPos, current->ch[ch].begin() + currentPos + framesToCopy, out->span(ch) + framesRendered);
        currentPos += framesToCopy;
        framesRendered += framesToCopy;
        if (currentPos >= current->length) {
          processed.push_back(current);
          current = nullptr;
          currentPos = 0;
        }
      } else {
        int64_t minInputConsumed = std::numeric_limits<int64_t>::max();
        int outputProduced = 0;
        for (int ch = 0; ch < channelCount; ch++) {
          const int available = (int)current->length - (int)currentPos;
          if (available <= 0) break;
          int consumed = 0, produced = 0;
          resamplers[ch].process(current->ch[ch].data() + currentPos, available, out->span(ch) + framesRendered, framesToRender - framesRendered,
                                 effectiveRate, consumed, produced);
          if (ch == 0) {
            minInputConsumed = consumed;
            outputProduced = produced;
          } else {
            minInputConsumed = std::min<int64_t>(minInputConsumed, consumed);
          }
        }
        currentPos += minInputConsumed;
        framesRendered += outputProduced;
        if (currentPos >= current->length - 4) {
          processed.push_back(current);
          current = nullptr;
          currentPos = 0;
        }
        if (minInputConsumed == 0) {
          clearRest();
          break;
        }
      }
    }
    if (framesRendered > 0) out->markNonSilent();
    else out->clear();
    outputs[0]->buffer = out;
  }
  void onDispose() override {  // :315-327
    if (state != GA_STREAM_STOPPED) setState(GA_STREAM_STOPPED);
    out.reset();
  }
};

// ---- ConstantSourceNode (Nodes/ConstantSourceNode.cs:15-163) ----
struct ConstantSourceNode : ScheduledNode {
  ConstantSourceNode(Context* c, int id) : ScheduledNode(c, id, GA_NODE_CONSTANT_SOURCE, 0, 1) {
    createParam(1.f, std::numeric_limits<float>::lowest(), std::numeric_limits<float>::max(), true);  // offset (:32-37)
  }
  void start(double when, double, double duration) {  // :44-63: a second Start is ignored
    ctx->executeOrPost([this, when, duration]() {
      if (hasStarted) return;
      hasStarted = true;
      startTime = std::max(0.0, when);
      if (!std::isnan(duration) && duration >= 0) {
        stopTime = startTime + duration;
        hasStopped = true;
      }
    });
  }
  void process() override {  // :76-141
    if (!out) out = rent(1);
    double t0 = ctx->currentTime;
    double t1 = t0 + (double)kBlock / ctx->sampleRate;
    int startFrame, endFrame;
    if (!window(t0, t1, startFrame, endFrame)) {
      out->clear();
      outputs[0]->buffer = out;
      tryRaiseEnded(t1);
      return;
    }
    float* o = out->span(0);
    const float* v = params[0]->computed;
    for (int i = 0; i < startFrame; i++) o[i] = 0.f;
    for (int i = startFrame; i < endFrame; i++) o[i] = v[i];
    for (int i = std::max(endFrame, 0); i < kBlock; i++) o[i] = 0.f;
    out->markNonSilent();
    outputs[0]->buffer = out;
    tryRaiseEnded(t1);
  }
};

// ---- OscillatorNode (Nodes/OscillatorNode.cs:12-214) ----
struct OscillatorNode : ScheduledNode {
  int oscType = 0;  // Sine, Square, Sawtooth, Triangle (:207-213)
  double phase = 0.0;
  OscillatorNode(Context* c, int id) : ScheduledNode(c, id, GA_NODE_OSCILLATOR, 0, 1) {
    createParam(440.f, 0.f, c->sampleRate / 2.f, true);  // frequency (:46-51)
  }
  void start(double when, double, double duration) {  // :54-73
    ctx->executeOrPost([this, when, duration]() {
      if (hasStarted) fail(GA_ERR_INVALID_OPERATION, "OscillatorNode can only be started once.");
      hasStarted = true;
      phase = 0.0;
      startTime = std::max(0.0, when);
      if (!std::isnan(duration) && duration >= 0) {
        stopTime = startTime + duration;
        hasStopped = true;
      }
    });
  }
  static float generate(double ph, int type) {  // :171-195
    const double PI = 3.14159265358979323846;
    switch (type) {
      case 0: return (float)std::sin(ph);
      case 1: return ph < PI ? 1.0f : -1.0f;
      case 2: return (float)(2.0 * (ph / (2.0 * PI)) - 1.0);
      case 3: {
        double t = ph / (2.0 * PI);
        return (float)(4.0 * std::fabs(t - std::floor(t + 0.5)) - 1.0);
      }
      default: return 0.f;
    }
  }
  void process() override {  // :91-158
    const double PI = 3.14159265358979323846;
    if (!out) out = rent(1);
    double t0 = ctx->currentTime;
    double t1 = t0 + (double)kBlock / ctx->sampleRate;
    int startFrame, endFrame;
    if (!window(t0, t1, startFrame, endFrame)) {
      out->clear();
      outputs[0]->buffer = out;
      tryRaiseEnded(t1);
      return;
    }
    float* o = out->span(0);
    const float* f = params[0]->computed;
    for (int i = 0; i < startFrame; i++) o[i] = 0.f;
    for (int i = startFrame; i < endFrame; i++) {
      o[i] = generate(phase, oscType);
      double phaseIncrement = (2.0 * PI * f[i]) / ctx->sampleRate;
      phase += phaseIncrement;
      if (phase >= 2.0 * PI) phase -= 2.0 * PI;
    }
    for (int i = std::max(endFrame, 0); i < kBlock; i++) o[i] = 0.f;
    out->markNonSilent();
    outputs[0]->buffer = out;
    tryRaiseEnded(t1);
  }
};

// ---- StereoPannerNode (Nodes/StereoPannerNode.cs:9-163) ----
struct StereoPannerNode : Node {
  BufPtr out;
  float lastPan = std::nanf(""), lastGainL = 0.5f, lastGainR = 0.5f;
  StereoPannerNode(Context* c, int id) : Node(c, id, GA_NODE_STEREO_PANNER, 1, 1) {
    inputs[0]->setChannelCount(2);                      // :24-26
    inputs[0]->mode = GA_COUNT_MODE_CLAMPED_MAX;
    inputs[0]->interpretation = GA_INTERP_SPEAKERS;
    createParam(0.f, -1.f, 1.f, true);                  // pan (:28-33)
  }
  void process() override {  // :36-74
    AudioBuffer* in = inputs[0]->buffer.get();
    int inputChannels = in->channelCount;
    if (!out || out->channelCount != 2) out = rent(2);
    if (in->silent) {
      out->clear();
      outputs[0]->buffer = out;
      return;
    }
    const float PIf = 3.14159265358979323846f;  // MathF.PI
    const float* pv = params[0]->computed;
    float* oL = out->span(0);
    float* oR = out->span(1);
    float gainL = lastGainL, gainR = lastGainR, lp = lastPan;
    if (inputChannels == 1) {  // ProcessMono, :76-109
      const float* x = in->span(0);
      for (int i = 0; i < kBlock; i++) {
        float pan = std::min(std::max(pv[i], -1.0f), 1.0f);
        if (pan != lp) {
          float xx = (pan + 1.0f) * 0.5f;
          gainL = gaoCos(xx * PIf / 2.0f);
          gainR = gaoSin(xx * PIf / 2.0f);
          lp = pan;
        }
        float sm = x[i];
        oL[i] = sm * gainL;
        oR[i] = sm * gainR;
      }
      lastPan = lp; lastGainL = gainL; lastGainR = gainR;
    } else if (inputChannels >= 2) {  // ProcessStereo, :111-153
      const float* xl = in->span(0);
      const float* xr = in->span(1);
      for (int i = 0; i < kBlock; i++) {
        float pan = std::min(std::max(pv[i], -1.0f), 1.0f);
        if (pan != lp) {
          float xx = pan <= 0.0f ? pan + 1.0f : pan;
          gainL = gaoCos(xx * PIf / 2.0f);
          gainR = gaoSin(xx * PIf / 2.0f);
          lp = pan;
        }
        float inL = xl[i], inR = xr[i];
        if (pan <= 0.0f) {
          oL[i] = inL + inR * gainL;
          oR[i] = inR * gainR;
        } else {
          oL[i] = inL * gainL;
          oR[i] = inR + inL * gainR;
        }
      }
      lastPan = lp; lastGainL = gainL; lastGainR = gainR;
    }
    out->markNonSilent();
    outputs[0]->buffer = out;
  }
  void onDispose() override { out.reset(); }
};

// ---- DelayNode (Nodes/DelayNode.cs:9-150) ----
struct DelayNode : Node {
  struct Ring {  // CircularBuffer, :125-149
    std::vector<float> buf;
    int writePos = 0;
    explicit Ring(int size) : buf((size_t)size, 0.f) {}
    void write(float v) {
      buf[writePos] = v;
      writePos = (writePos + 1) % (int)buf.size();
    }
    float read(int d) const {
      if (d <= 0 || d > (int)buf.size()) return 0.f;
      int rp = (writePos - d + (int)buf.size()) % (int)buf.size();
      return buf[rp];
    }
  };
  std::vector<Ring> rings;
  BufPtr out;
  int maxDelaySamples;
  DelayNode(Context* c, int id, double maxDelayTime) : Node(c, id, GA_NODE_DELAY, 1, 1) {
    if (maxDelayTime <= 0 || maxDelayTime > 10) fail(GA_ERR_OUT_OF_RANGE, "maxDelayTime");  // :25-26
    maxDelaySamples = (int)(maxDelayTime * c->sampleRate);
    if (maxDelaySamples < 1) fail(GA_ERR_OUT_OF_RANGE, "maxDelayTime");  // (new float[0] + modulo by zero in the reference)
    for (int i = 0; i < 2; i++) rings.emplace_back(maxDelaySamples);
    createParam(0.f, 0.f, (float)maxDelayTime, true);  // delayTime (:35-40)
  }
  void process() override {  // :43-100
    AudioBuffer* in = inputs[0]->buffer.get();
    int ch = in ? in->channelCount : 2;
    while ((int)rings.size() < ch) rings.emplace_back(maxDelaySamples);
    if (!out || out->channelCount != ch) out = rent(ch);
    bool hasAudio = false;
    const float* dt = params[0]->computed;
    const bool silentIn = !in || in->silent;
    for (int c = 0; c < ch; c++) {
      const float* x = silentIn ? nullptr : in->span(c);
      float* o = out->span(c);
      for (int i = 0; i < kBlock; i++) {
        int d = (int)(dt[i] * ctx->sampleRate);   // float * int -> float, truncated (:68,:88)
        d = std::min(std::max(d, 0), maxDelaySamples);
        o[i] = rings[c].read(d);
        rings[c].write(x ? x[i] : 0.f);
        if (o[i] != 0.f) hasAudio = true;
      }
    }
    // the rented buffer keeps its silent flag from the previous block unless audio appears (:96-97): it is never re-cleared
    if (hasAudio) out->markNonSilent();
    outputs[0]->buffer = out;
  }
  void onDispose() override { out.reset(); }
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// C surface (include/graphaudio_hip.h under the gao_ prefix)
// ------------------------------------------------------------------------------------------------
struct ga_context {
  Context c;
  explicit ga_context(int sr) : c(sr) {}
};

namespace {
template <class F>
int guard(ga_context* h, F&& f) {
  if (!h) return GA_ERR_INVALID_ARGUMENT;
  try {
    f(h->c);
    return GA_OK;
  } catch (const Err& e) {
    h->c.lastError = e.msg;
    return e.code;
  } catch (const std::bad_alloc&) {
    h->c.lastError = "out of memory";
    return GA_ERR_OUT_OF_MEMORY;
  } catch (...) {
    h->c.lastError = "unknown error";
    return GA_ERR_INVALID_OPERATION;
  }
}
Node* getNode(Context& c, int id) {
  if (id < 0 || id >= (int)c.nodes.size()) fail(GA_ERR_INVALID_ARGUMENT, "bad node id");
  return c.nodes[id].get();
}
Param* getParam(Context& c, int node, int param) {
  Node* n = getNode(c, node);
  if (param < 0 || param >= (int)n->params.size()) fail(GA_ERR_INVALID_ARGUMENT, "bad param index");
  return n->params[param].get();
}
PlayableBuffer* getBuffer(Context& c, int id) {
  if (id < 0) return nullptr;
  if (id >= (int)c.buffers.size() || !c.buffers[id]) fail(GA_ERR_INVALID_ARGUMENT, "bad buffer id");
  return c.buffers[id].get();
}
template <class T>
T* as(Node* n, int type) {
  if (n->type != type) fail(GA_ERR_INVALID_ARGUMENT, "node has the wrong type for this call");
  return static_cast<T*>(n);
}
}  // namespace

extern "C" {

const char* gao_strerror(int code) {
  switch (code) {
    case GA_OK: return "ok";
    case GA_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GA_ERR_OUT_OF_RANGE: return "argument out of range";
    case GA_ERR_INVALID_OPERATION: return "invalid operation";
    case GA_ERR_DISPOSED: return "object disposed";
    case GA_ERR_CYCLE: return "audio graph cycle detected";
    case GA_ERR_UNSUPPORTED: return "unsupported";
    case GA_ERR_DEVICE: return "device error";
    case GA_ERR_OUT_OF_MEMORY: return "out of memory";
    case GA_ERR_NO_DEVICE: return "no device";
    default: return "unknown error code";
  }
}
const char* gao_version(void) { return "graphaudio-oracle 0.1 (CPU restatement, test infrastructure)"; }
int gao_device_count(void) { return 0; }

int gao_context_create(int sample_rate, int device_ordinal, ga_context** out) {
  (void)device_ordinal;
  if (!out) return GA_ERR_INVALID_ARGUMENT;
  if (sample_rate <= 0) return GA_ERR_OUT_OF_RANGE;  // AudioContextBase.cs:37-38
  ga_context* h = new ga_context(sample_rate);
  h->c.nodes.push_back(std::make_unique<DestinationNode>(&h->c));
  *out = h;
  return GA_OK;
}
int gao_context_destroy(ga_context* ctx) {
  delete ctx;
  return GA_OK;
}
const char* gao_last_error(ga_context* ctx) { return ctx ? ctx->c.lastError.c_str() : ""; }
double gao_current_time(ga_context* ctx) { return ctx ? ctx->c.currentTime : 0.0; }
int64_t gao_current_block(ga_context* ctx) { return ctx ? ctx->c.currentBlock : 0; }
int gao_set_option(ga_context*, const char*, double) { return GA_OK; }
int gao_get_stats(ga_context* ctx, ga_stats* out) {
  if (!ctx || !out) return GA_ERR_INVALID_ARGUMENT;
  std::memset(out, 0, sizeof(*out));
  out->blocks_rendered = ctx->c.currentBlock;
  out->n_nodes = (int)ctx->c.nodes.size();
  return GA_OK;
}
int gao_context_set_stream(ga_context*, void*) { return GA_ERR_UNSUPPORTED; }
int gao_synchronize(ga_context* ctx) { return ctx ? GA_OK : GA_ERR_INVALID_ARGUMENT; }   // the CPU restatement is synchronous

int gao_buffer_create(ga_context* ctx, const float* const* planar, int channels, int64_t frames, int sample_rate,
                      int* out_id) {
  return guard(ctx, [&](Context& c) {
    if (!planar || !out_id) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "Channel count must be between 1 and 32");
    if (frames < 0) fail(GA_ERR_OUT_OF_RANGE, "Length must be non-negative");
    if (sample_rate <= 0) fail(GA_ERR_OUT_OF_RANGE, "Sample rate must be positive");
    auto b = std::make_unique<PlayableBuffer>();
    b->channels = channels;
    b->length = frames;
    b->sampleRate = sample_rate;
    b->ch.resize(channels);
    for (int i = 0; i < channels; i++) b->ch[i].assign(planar[i], planar[i] + frames);
    c.buffers.push_back(std::move(b));
    *out_id = (int)c.buffers.size() - 1;
  });
}
int gao_buffer_release(ga_context* ctx, int) { return ctx ? GA_OK : GA_ERR_INVALID_ARGUMENT; }  // kept alive: nodes hold raw pointers

int gao_node_create_ex(ga_context* ctx, int node_type, double arg, int* out_id) {
  return guard(ctx, [&](Context& c) {
    if (!out_id) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    int id = (int)c.nodes.size();
    switch (node_type) {
      case GA_NODE_BUFFER_SOURCE: c.nodes.push_back(std::make_unique<SourceNode>(&c, id)); break;
      case GA_NODE_GAIN: c.nodes.push_back(std::make_unique<GainNode>(&c, id)); break;
      case GA_NODE_BIQUAD: c.nodes.push_back(std::make_unique<BiquadNode>(&c, id)); break;
      case GA_NODE_CONVOLVER: c.nodes.push_back(std::make_unique<ConvolverNode>(&c, id)); break;
      case GA_NODE_CHANNEL_SPLITTER:
      case GA_NODE_CHANNEL_MERGER: {
        int n = (int)arg;
        if (n < 1 || n > 32) fail(GA_ERR_OUT_OF_RANGE, node_type == GA_NODE_CHANNEL_SPLITTER ? "numberOfOutputs" : "numberOfInputs");
        if (node_type == GA_NODE_CHANNEL_SPLITTER) c.nodes.push_back(std::make_unique<SplitterNode>(&c, id, n));
        else c.nodes.push_back(std::make_unique<MergerNode>(&c, id, n));
        break;
      }
      case GA_NODE_CONSTANT_SOURCE: c.nodes.push_back(std::make_unique<ConstantSourceNode>(&c, id)); break;
      case GA_NODE_STEREO_PANNER: c.nodes.push_back(std::make_unique<StereoPannerNode>(&c, id)); break;
      case GA_NODE_OSCILLATOR: c.nodes.push_back(std::make_unique<OscillatorNode>(&c, id)); break;
      case GA_NODE_DELAY: c.nodes.push_back(std::make_unique<DelayNode>(&c, id, arg)); break;
      case GA_NODE_STREAM_SOURCE: c.nodes.push_back(std::make_unique<StreamNode>(&c, id)); break;
      default: fail(GA_ERR_INVALID_ARGUMENT, "unknown node type");
    }
    *out_id = id;
  });
}
int gao_node_create(ga_context* ctx, int node_type, int* out_id) {  // constructor defaults: 2 outputs / 2 inputs / 1.0 s
  double arg = node_type == GA_NODE_DELAY ? 1.0 : 2.0;
  return gao_node_create_ex(ctx, node_type, arg, out_id);
}
int gao_node_dispose(ga_context* ctx, int node) {
  return guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    if (n->disposed) return;  // Nodes/AudioNode.cs:209-210
    c.executeOrPost([n]() { n->doDispose(); });
  });
}
int gao_node_connect(ga_context* ctx, int src, int dst, int output_index, int input_index) {
  return guard(ctx, [&](Context& c) {
    Node* s = getNode(c, src);
    Node* d = getNode(c, dst);
    c.executeOrPost([s, d, output_index, input_index]() {  // DoConnect, Nodes/AudioNode.cs:111-120
      if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
      if (input_index < 0 || input_index >= (int)d->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
      s->outputs[output_index]->connectTo(d->inputs[input_index].get());
    });
  });
}
int gao_node_disconnect(ga_context* ctx, int src, int dst, int output_index, int input_index) {
  return guard(ctx, [&](Context& c) {
    Node* s = getNode(c, src);
    Node* d = dst < 0 ? nullptr : getNode(c, dst);
    c.executeOrPost([s, d, output_index, input_index]() {  // DoDisconnect, Nodes/AudioNode.cs:131-147
      if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
      if (!d) {
        s->outputs[output_index]->disconnectAll();
      } else {
        if (input_index < 0 || input_index >= (int)d->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
        s->outputs[output_index]->disconnectFrom(d->inputs[input_index].get());
      }
    });
  });
}
int gao_node_connect_param(ga_context* ctx, int src, int dst_node, int dst_param, int output_index) {
  return guard(ctx, [&](Context& c) {
    Node* s = getNode(c, src);
    Param* p = getParam(c, dst_node, dst_param);
    if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");  // :88-89
    Output* o = s->outputs[output_index].get();
    c.executeOrPost([o, p]() { o->connectTo(p->input.get()); });  // AudioParam.ConnectFrom, AudioParam.cs:73-77
  });
}
int gao_node_disconnect_param(ga_context* ctx, int src, int dst_node, int dst_param, int output_index) {
  return guard(ctx, [&](Context& c) {
    Node* s = getNode(c, src);
    Param* p = getParam(c, dst_node, dst_param);
    if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
    Output* o = s->outputs[output_index].get();
    c.executeOrPost([o, p]() { o->disconnectFrom(p->input.get()); });
  });
}
int gao_node_has_ended(ga_context* ctx, int node) {
  int r = 0;
  int rc = guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    r = n->endedRaised ? 1 : 0;
  });
  return rc < 0 ? rc : r;
}

int gao_poll_ended(ga_context* ctx, int* out_ids, int capacity) {
  int n = 0;
  int rc = guard(ctx, [&](Context& c) {
    if (!out_ids || capacity < 0) fail(GA_ERR_INVALID_ARGUMENT, "bad buffer");
    for (auto& np : c.nodes) {
      if (n >= capacity) break;
      Node* s = np.get();
      if (s->endedRaised && !s->endedReported) {
        s->endedReported = true;
        out_ids[n++] = np->id;
      }
    }
  });
  return rc < 0 ? rc : n;
}

int gao_input_set_channel_count(ga_context* ctx, int node, int input_index, int count) {
  return guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    if (input_index < 0 || input_index >= (int)n->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
    n->inputs[input_index]->setChannelCount(count);
  });
}
int gao_input_set_channel_count_mode(ga_context* ctx, int node, int input_index, int mode) {
  return guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    if (input_index < 0 || input_index >= (int)n->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
    if (mode < 0 || mode > 2) fail(GA_ERR_INVALID_ARGUMENT, "mode");
    n->inputs[input_index]->mode = mode;
  });
}
int gao_input_set_channel_interpretation(ga_context* ctx, int node, int input_index, int interp) {
  return guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    if (input_index < 0 || input_index >= (int)n->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
    n->inputs[input_index]->interpretation = interp;
  });
}
int gao_destination_set_channel_count(ga_context* ctx, int channels) {
  return guard(ctx, [&](Context& c) {
    if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "channels");  // AudioDestinationNode.cs:25-26
    Node* d = c.nodes[0].get();
    c.executeOrPost([d, channels]() { d->inputs[0]->setChannelCount(channels); });
  });
}
int gao_destination_output_channels(ga_context* ctx) {
  if (!ctx) return GA_ERR_INVALID_ARGUMENT;
  auto* d = ctx->c.destination();
  return d->outputBuffer ? d->outputBuffer->channelCount : 2;  // OfflineAudioContext.cs:113-114
}

int gao_param_set_value(ga_context* ctx, int node, int param, float value) {
  return guard(ctx, [&](Context& c) { getParam(c, node, param)->setValue(value); });
}
int gao_param_get_value(ga_context* ctx, int node, int param, float* out) {
  return guard(ctx, [&](Context& c) {
    if (!out) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    *out = getParam(c, node, param)->value;
  });
}
int gao_param_set_value_at_time(ga_context* ctx, int node, int param, float value, double t) {
  return guard(ctx, [&](Context& c) { getParam(c, node, param)->setValueAtTime(value, t); });
}
int gao_param_linear_ramp_to_value_at_time(ga_context* ctx, int node, int param, float value, double t) {
  return guard(ctx, [&](Context& c) { getParam(c, node, param)->linearRamp(value, t); });
}
int gao_param_exponential_ramp_to_value_at_time(ga_context* ctx, int node, int param, float value, double t) {
  return guard(ctx, [&](Context& c) { getParam(c, node, param)->exponentialRamp(value, t); });
}
int gao_param_set_target_at_time(ga_context* ctx, int node, int param, float target, double t, double tc) {
  return guard(ctx, [&](Context& c) { getParam(c, node, param)->setTarget(target, t, tc); });
}
int gao_param_cancel_scheduled_values(ga_context* ctx, int node, int param, double t) {
  return guard(ctx, [&](Context& c) { getParam(c, node, param)->cancelScheduled(t); });
}

int gao_source_set_buffer(ga_context* ctx, int node, int buffer_id) {
  return guard(ctx, [&](Context& c) {
    as<SourceNode>(getNode(c, node), GA_NODE_BUFFER_SOURCE)->buffer = getBuffer(c, buffer_id);
  });
}
int gao_source_set_loop(ga_context* ctx, int node, int loop, double loop_start, double loop_end) {
  return guard(ctx, [&](Context& c) {
    auto* s = as<SourceNode>(getNode(c, node), GA_NODE_BUFFER_SOURCE);
    s->loop = loop != 0;
    s->loopStart = std::max(0.0, loop_start);  // :51
    s->loopEnd = std::max(0.0, loop_end);      // :60
  });
}
int gao_source_start(ga_context* ctx, int node, double when, double offset, double duration) {
  return guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    if (n->type == GA_NODE_CONSTANT_SOURCE) static_cast<ConstantSourceNode*>(n)->start(when, offset, duration);
    else if (n->type == GA_NODE_OSCILLATOR) static_cast<OscillatorNode*>(n)->start(when, offset, duration);
    else as<SourceNode>(n, GA_NODE_BUFFER_SOURCE)->start(when, offset, duration);
  });
}
int gao_source_stop(ga_context* ctx, int node, double when) {
  return guard(ctx, [&](Context& c) {
    Node* n = getNode(c, node);
    if (n->type == GA_NODE_CONSTANT_SOURCE || n->type == GA_NODE_OSCILLATOR) static_cast<ScheduledNode*>(n)->stop(when);
    else as<SourceNode>(n, GA_NODE_BUFFER_SOURCE)->stop(when);
  });
}
int gao_stream_queue_buffer(ga_context* ctx, int node, int buffer_id) {
  return guard(ctx, [&](Context& c) {
    PlayableBuffer* b = getBuffer(c, buffer_id);
    if (!b) fail(GA_ERR_INVALID_ARGUMENT, "Buffer must be initialized");
    as<StreamNode>(getNode(c, node), GA_NODE_STREAM_SOURCE)->queued.push_back(b);
  });
}
int gao_stream_set_state(ga_context* ctx, int node, int state) {
  return guard(ctx, [&](Context& c) {
    if (state < GA_STREAM_PLAYING || state > GA_STREAM_STOPPED) fail(GA_ERR_OUT_OF_RANGE, "stream state");
    as<StreamNode>(getNode(c, node), GA_NODE_STREAM_SOURCE)->setState(state);
  });
}
int gao_stream_dequeue_processed(ga_context* ctx, int node, int* buffer_id_out) {
  int got = 0;
  int rc = guard(ctx, [&](Context& c) {
    auto* s = as<StreamNode>(getNode(c, node), GA_NODE_STREAM_SOURCE);
    if (s->processed.empty()) return;
    PlayableBuffer* b = s->processed.front();
    s->processed.pop_front();
    got = 1;
    if (buffer_id_out) {
      *buffer_id_out = -1;
      for (size_t i = 0; i < c.buffers.size(); i++)
        if (c.buffers[i].get() == b) *buffer_id_out = (int)i;
    }
  });
  return rc < 0 ? rc : got;
}
int gao_stream_queued_count(ga_context* ctx, int node) {
  int n = 0;
  int rc = guard(ctx, [&](Context& c) { n = (int)as<StreamNode>(getNode(c, node), GA_NODE_STREAM_SOURCE)->queued.size(); });
  return rc < 0 ? rc : n;
}
int gao_stream_processed_count(ga_context* ctx, int node) {
  int n = 0;
  int rc = guard(ctx, [&](Context& c) { n = (int)as<StreamNode>(getNode(c, node), GA_NODE_STREAM_SOURCE)->processed.size(); });
  return rc < 0 ? rc : n;
}
int gao_oscillator_set_type(ga_context* ctx, int node, int oscillator_type) {
  return guard(ctx, [&](Context& c) {
    auto* o = as<OscillatorNode>(getNode(c, node), GA_NODE_OSCILLATOR);
    if (oscillator_type < 0 || oscillator_type > 3) fail(GA_ERR_INVALID_ARGUMENT, "oscillator type");
    c.executeOrPost([o, oscillator_type]() { o->oscType = oscillator_type; });  // OscillatorNode.cs:33-42
  });
}
int gao_biquad_set_type(ga_context* ctx, int node, int filter_type) {
  return guard(ctx, [&](Context& c) {
    auto* b = as<BiquadNode>(getNode(c, node), GA_NODE_BIQUAD);
    if (filter_type < 0 || filter_type > GA_FILTER_HIGHSHELF) fail(GA_ERR_INVALID_ARGUMENT, "filter type");
    c.executeOrPost([b, filter_type]() {  // BiQuadFilterNode.cs:24-36
      if (b->filterType != filter_type) {
        b->filterType = filter_type;
        b->coefficientsDirty = true;
      }
    });
  });
}
int gao_convolver_set_normalize(ga_context* ctx, int node, int normalize) {
  return guard(ctx, [&](Context& c) { as<ConvolverNode>(getNode(c, node), GA_NODE_CONVOLVER)->normalize = normalize != 0; });
}
int gao_convolver_set_enable_true_stereo(ga_context* ctx, int node, int enable) {
  return guard(ctx, [&](Context& c) { as<ConvolverNode>(getNode(c, node), GA_NODE_CONVOLVER)->enableTrueStereo = enable != 0; });
}
int gao_convolver_set_buffer(ga_context* ctx, int node, int buffer_id) {
  return guard(ctx, [&](Context& c) { as<ConvolverNode>(getNode(c, node), GA_NODE_CONVOLVER)->setBuffer(getBuffer(c, buffer_id)); });
}

// OfflineAudioContext.Render(float[][] output, int frameCount, int startIndex), OfflineAudioContext.cs:30-102
int gao_render(ga_context* ctx, float* const* output, int channels, int64_t frameCount, int64_t startIndex) {
  return guard(ctx, [&](Context& c) {
    if (channels == 0 || !output) fail(GA_ERR_INVALID_ARGUMENT, "Output buffer must have at least one channel.");
    if (frameCount <= 0) fail(GA_ERR_OUT_OF_RANGE, "Frame count must be positive.");
    if (startIndex < 0) fail(GA_ERR_OUT_OF_RANGE, "Start index must be non-negative.");
    for (int ch = 0; ch < channels; ch++)
      if (!output[ch]) fail(GA_ERR_INVALID_ARGUMENT, "Channel buffer is null.");
    int64_t framesWritten = 0;
    if (c.cachedFrames > 0) {
      int toCopy = (int)std::min<int64_t>(c.cachedFrames, frameCount);
      for (int ch = 0; ch < channels; ch++) std::memcpy(output[ch] + startIndex, c.cache[ch].data(), sizeof(float) * toCopy);
      if (toCopy < c.cachedFrames) {
        int remaining = c.cachedFrames - toCopy;
        for (int ch = 0; ch < channels; ch++) std::memmove(c.cache[ch].data(), c.cache[ch].data() + toCopy, sizeof(float) * remaining);
      }
      framesWritten = toCopy;
      c.cachedFrames -= toCopy;
    }
    while (framesWritten < frameCount) {
      AudioBuffer* buffer = c.processBlock();
      int toCopy = (int)std::min<int64_t>(kBlock, frameCount - framesWritten);
      for (int ch = 0; ch < channels; ch++)  // GetChannelSpan(ch) throws when ch >= buffer.ChannelCount (:82-85)
        std::memcpy(output[ch] + startIndex + framesWritten, buffer->span(ch), sizeof(float) * toCopy);
      framesWritten += toCopy;
      int excess = kBlock - toCopy;
      if (excess > 0) {
        if ((int)c.cache.size() < channels) c.cache.resize(channels);
        for (int ch = 0; ch < channels; ch++) {
          if ((int)c.cache[ch].size() < c.cachedFrames + excess) c.cache[ch].resize(std::max(512, c.cachedFrames + excess));
          std::memcpy(c.cache[ch].data() + c.cachedFrames, buffer->span(ch) + toCopy, sizeof(float) * excess);
        }
        c.cachedFrames += excess;
      }
    }
  });
}
int gao_render_device(ga_context*, float* const*, int, int64_t, int64_t) { return GA_ERR_UNSUPPORTED; }
// sharded render (include/graphaudio_hip.h): the CPU restatement is a single rank; shard_range is the same arithmetic
int gao_comm_unique_id(void*) { return GA_ERR_UNSUPPORTED; }
int gao_comm_init(ga_context* ctx, const void*, int n_ranks, int rank) {
  if (!ctx) return GA_ERR_INVALID_ARGUMENT;
  return (n_ranks == 1 && rank == 0) ? GA_OK : GA_ERR_UNSUPPORTED;
}
int gao_comm_destroy(ga_context* ctx) { return ctx ? GA_OK : GA_ERR_INVALID_ARGUMENT; }
int gao_comm_info(ga_context* ctx, int* n_ranks, int* rank, int* uses_rccl) {
  if (!ctx || !n_ranks || !rank || !uses_rccl) return GA_ERR_INVALID_ARGUMENT;
  *n_ranks = 1;
  *rank = 0;
  *uses_rccl = 0;
  return GA_OK;
}
int gao_shard_range(int64_t n_voices, int n_ranks, int rank, int64_t* first, int64_t* count) {
  if (!first || !count) return GA_ERR_INVALID_ARGUMENT;
  if (n_voices < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return GA_ERR_OUT_OF_RANGE;
  const int64_t base = n_voices / n_ranks, extra = n_voices % n_ranks;
  *first = rank * base + (rank < extra ? rank : extra);
  *count = base + (rank < extra ? 1 : 0);
  return GA_OK;
}
int gao_render_reduce(ga_context* ctx, float* const* out, int channels, int64_t frames, int64_t start, int root) {
  if (root != 0) return GA_ERR_OUT_OF_RANGE;
  return gao_render(ctx, out, channels, frames, start);
}

// AudioContextBase.ProcessBlocks, AudioContextBase.cs:163-186
int gao_process_blocks(ga_context* ctx, float* const* outputBuffers, int nOut, int64_t blockCount, int onDevice) {
  return guard(ctx, [&](Context& c) {
    if (onDevice) fail(GA_ERR_UNSUPPORTED, "the oracle has no device memory");
    if (blockCount < 0) fail(GA_ERR_OUT_OF_RANGE, "blockCount");
    if (nOut < 0 || (nOut > 0 && !outputBuffers)) fail(GA_ERR_INVALID_ARGUMENT, "outputBuffers");
    for (int64_t block = 0; block < blockCount; block++) {
      AudioBuffer* buffer = c.processBlock();
      int channels = std::min(nOut, buffer->channelCount);
      for (int ch = 0; ch < channels; ch++)
        if (outputBuffers[ch]) std::memcpy(outputBuffers[ch] + block * kBlock, buffer->span(ch), sizeof(float) * kBlock);
    }
  });
}
// AudioContextBase.ProcessBlockInterleaved, AudioContextBase.cs:88-157, repeated blockCount times
int gao_process_blocks_interleaved(ga_context* ctx, float* interleaved, int channels, int64_t blockCount, int onDevice) {
  return guard(ctx, [&](Context& c) {
    if (onDevice) fail(GA_ERR_UNSUPPORTED, "the oracle has no device memory");
    if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "channels");
    if (!interleaved) fail(GA_ERR_INVALID_ARGUMENT, "Buffer too small for interleaved output.");
    if (blockCount < 0) fail(GA_ERR_OUT_OF_RANGE, "blockCount");
    for (int64_t block = 0; block < blockCount; block++) {
      AudioBuffer* buffer = c.processBlock();
      float* out = interleaved + block * kBlock * channels;
      int used = std::min(channels, buffer->channelCount);
      for (int ch = 0; ch < used; ch++) {
        const float* src = buffer->span(ch);
        for (int f = 0; f < kBlock; f++) out[(size_t)f * channels + ch] = src[f];
      }
      for (int ch = used; ch < channels; ch++)
        for (int f = 0; f < kBlock; f++) out[(size_t)f * channels + ch] = 0.f;
    }
  });
}

// ---- test-only extras (not part of the product ABI): direct access to the DSP primitives so that
// tests/ can pin them against numpy/scipy without building a graph ----
int gao_test_rfft256(const double* x, double* re, double* im) {
  static RealFft f(256);
  f.forward(x, re, im);
  return 0;
}
int gao_test_irfft256(const double* re, const double* im, double* x) {
  static RealFft f(256);
  f.inverse(re, im, x);
  return 0;
}
float gao_test_normalization_scale(const float* ir, int len) { return PartitionedConvolver::normalizationScale(ir, len); }
// Runs one PartitionedConvolver over nBlocks*128 input frames.
int gao_test_convolve(const float* ir, int irLen, int normalize, const float* in, float* out, int nBlocks) {
  try {
    PartitionedConvolver pc(ir, irLen, kBlock, normalize != 0);
    for (int b = 0; b < nBlocks; b++) pc.process(in + (size_t)b * kBlock, out + (size_t)b * kBlock);
    return 0;
  } catch (...) {
    return -1;
  }
}
int gao_test_resample(const float* in, int inLen, float* out, int outLen, double rate, int* consumed, int* produced) {
  CubicResampler r;
  r.process(in, inLen, out, outLen, rate, *consumed, *produced);
  return 0;
}

}  // extern "C"

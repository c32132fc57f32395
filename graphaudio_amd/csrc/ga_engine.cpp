// ga_engine.cpp -- graph state, command queue, device resources.  See ga_engine.hpp for the design overview and
// ga_chunk.cpp (+ ga_sources / ga_plan_nodes / ga_plan_conv) for the control-plane simulation + chunk executor.
#include "ga_engine.hpp"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace ga {

void launch_fail(const char* what) { fail(GA_ERR_INVALID_OPERATION, std::string("internal: ") + what); }

// ------------------------------------------------------------------------------------------------------
// device resources
// ------------------------------------------------------------------------------------------------------
void* Context::dalloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 256;
  GA_HIP(hipMalloc(&p, bytes));
  devBytes += (int64_t)bytes;
  return p;
}
float* Context::dallocSkewed(size_t bytes, void** base, size_t* total) {
  *total = bytes + 64 * 1024;
  *base = dalloc(*total);
  return (float*)((char*)*base + (size_t)(skewSeq++ % 64) * 1024);
}
// Predicted RMS difference between two float32 evaluations of a cascade that round differently, for white input of unit variance.
double Context::biquadDeviation(const float* coefs, int nsec) {
  std::vector<float> key(coefs, coefs + 5 * nsec);
  uint64_t h = 1469598103934665603ull;
  for (float f : key) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    h = (h ^ u) * 1099511628211ull;
  }
  auto& bucket = bqDeviations[h];
  for (auto& e : bucket)
    if (e.first == key) return e.second;
  double dev = 1.0;
  // Rounding-noise model (white input): the rounding of section q's  w = x - a1 w1 - a2 w2  is an error of ~ 2^-24 / sqrt(3) |w|
  // entering at that section's input; it reaches the output through sections q .. nsec-1.  With g_q = energy of the impulse response
  // input -> w_q and t_q = energy of (input of section q -> output), the noise-to-signal ratio is eps^2 sum_q g_q t_q / energy of the
  // whole cascade; two evaluations that round differently differ by sqrt(2) of that.
  {
    const int N = 1 << 14;
    std::vector<double> sig(N, 0.0);
    sig[0] = 1.0;
    std::vector<std::vector<double>> wq(nsec);
    for (int q = 0; q < nsec; q++) {   // impulse response of the first q sections + 1/A_q: the W of section q
      const double b0 = coefs[5 * q], b1 = coefs[5 * q + 1], b2 = coefs[5 * q + 2], a1 = coefs[5 * q + 3], a2 = coefs[5 * q + 4];
      wq[q].resize(N);
      double w1 = 0, w2 = 0;
      for (int i = 0; i < N; i++) {
        const double w = sig[i] - a1 * w1 - a2 * w2;
        wq[q][i] = w;
        sig[i] = b0 * w + b1 * w1 + b2 * w2;
        w2 = w1;
        w1 = w;
      }
    }
    double eOut = 0;
    for (double v : sig) eOut += v * v;
    double acc = 0;
    for (int q = 0; q < nsec; q++) {
      std::vector<double> tail(N, 0.0);
      tail[0] = 1.0;
      for (int r = q; r < nsec; r++) {
        const double b0 = coefs[5 * r], b1 = coefs[5 * r + 1], b2 = coefs[5 * r + 2], a1 = coefs[5 * r + 3], a2 = coefs[5 * r + 4];
        double w1 = 0, w2 = 0;
        for (int i = 0; i < N; i++) {
          const double w = tail[i] - a1 * w1 - a2 * w2;
          tail[i] = b0 * w + b1 * w1 + b2 * w2;
          w2 = w1;
          w1 = w;
        }
      }
      double g = 0, tt = 0;
      for (int i = 0; i < N; i++) {
        g += wq[q][i] * wq[q][i];
        tt += tail[i] * tail[i];
      }
      acc += g * tt;
    }
    const double eps = 5.96e-8 / 1.7320508;
    // absolute, for white input of unit variance; x 3: what the measured deviations are above this model (tests/test_gpu_biquad_split.py)
    (void)eOut;
    dev = 3.0 * std::sqrt(2.0 * eps * eps * acc);
  }
  bucket.push_back({std::move(key), dev});
  return dev;
}
// A^K of a cascade's state recursion (state = W1, W2 of every section, direct form II as in BiQuadFilterNode.cs:137-141), float64.
const Context::BqTransition& Context::biquadTransition(const float* coefs, int nsec, int64_t K) {
  const int D = 2 * nsec;
  std::vector<float> key(coefs, coefs + 5 * nsec);
  key.push_back((float)K);
  uint64_t h = 1469598103934665603ull;
  for (float f : key) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    h = (h ^ u) * 1099511628211ull;
  }
  auto& bucket = bqTransitions[h];
  for (auto& t : bucket)
    if (t.key == key) return t;
  // one step from the unit states with zero input gives the columns of A
  std::vector<double> A((size_t)D * D, 0.0), R((size_t)D * D, 0.0), T((size_t)D * D);
  for (int j = 0; j < D; j++) {
    std::vector<double> st(D, 0.0);
    st[j] = 1.0;
    double x = 0.0;
    for (int q = 0; q < nsec; q++) {
      const double b0 = coefs[5 * q], b1 = coefs[5 * q + 1], b2 = coefs[5 * q + 2], a1 = coefs[5 * q + 3], a2 = coefs[5 * q + 4];
      const double w1 = st[2 * q], w2 = st[2 * q + 1];
      const double w = x - a1 * w1 - a2 * w2;
      x = b0 * w + b1 * w1 + b2 * w2;
      st[2 * q + 1] = w1;
      st[2 * q] = w;
    }
    for (int r = 0; r < D; r++) A[(size_t)r * D + j] = st[r];
  }
  for (int i = 0; i < D; i++) R[(size_t)i * D + i] = 1.0;
  auto mul = [&](std::vector<double>& out, const std::vector<double>& a, const std::vector<double>& b) {
    for (int i = 0; i < D; i++)
      for (int j = 0; j < D; j++) {
        double acc = 0.0;
        for (int k = 0; k < D; k++) acc += a[(size_t)i * D + k] * b[(size_t)k * D + j];
        T[(size_t)i * D + j] = acc;
      }
    out = T;
  };
  for (int64_t e = K; e > 0; e >>= 1) {
    if (e & 1) mul(R, R, A);
    if (e > 1) mul(A, A, A);
  }
  BqTransition t;
  t.key = std::move(key);
  t.M.resize((size_t)D * D);
  for (size_t i = 0; i < t.M.size(); i++) t.M[i] = (float)R[i];
  bucket.push_back(std::move(t));
  return bucket.back();
}
float* Context::bqSplitAlloc(size_t floats) {
  const size_t bytes = (floats * sizeof(float) + 63) & ~(size_t)63;
  if (bytes > kBqSplitBlock) fail(GA_ERR_INVALID_OPERATION, "internal: biquad split state larger than a block");
  if (bqSplitUsed / kBqSplitBlock != (bqSplitUsed + bytes - 1) / kBqSplitBlock) bqSplitUsed = (bqSplitUsed / kBqSplitBlock + 1) * kBqSplitBlock;
  const size_t blk = bqSplitUsed / kBqSplitBlock;
  while (bqSplitBlocks.size() <= blk) bqSplitBlocks.push_back((float*)dalloc(kBqSplitBlock));
  float* p = (float*)((char*)bqSplitBlocks[blk] + bqSplitUsed % kBqSplitBlock);
  bqSplitUsed += bytes;
  return p;
}
int Context::persistentBuffer(const float* p, int64_t n) {
  if (!p || n <= 0) return -1;
  if (bufSpansVersion != bufVersion) {
    bufSpans.clear();
    for (size_t i = 0; i < buffers.size(); i++)
      if (buffers[i] && buffers[i]->dev) bufSpans.push_back(BufSpan{buffers[i]->dev, buffers[i]->dev + (size_t)buffers[i]->stride * buffers[i]->channels, (int)i});
    std::sort(bufSpans.begin(), bufSpans.end(), [](const BufSpan& a, const BufSpan& b) { return a.lo < b.lo; });
    bufSpansVersion = bufVersion;
  }
  auto it = std::upper_bound(bufSpans.begin(), bufSpans.end(), p, [](const float* q, const BufSpan& s) { return q < s.lo; });
  if (it == bufSpans.begin()) return -1;
  --it;
  return (p >= it->lo && p + n <= it->hi) ? it->id : -1;
}
void Context::dfree(void* p, size_t bytes) {
  if (!p) return;
  (void)hipFree(p);
  devBytes -= (int64_t)(bytes == 0 ? 256 : bytes);
}
void Context::ensure(DevArena& a, size_t bytes) {
  if (a.bytes >= bytes) return;
  if (a.p) {
    GA_HIP(hipStreamSynchronize(stream));
    dfree(a.p, a.bytes);
    a.p = nullptr;
    a.bytes = 0;
  }
  size_t nb = bytes + bytes / 8;
  a.p = dalloc(nb);
  a.bytes = nb;
  GA_HIP(hipMemsetAsync(a.p, 0, nb, stream));
}

void Context::init_device(int dev) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    fail(GA_ERR_NO_DEVICE, "no HIP device visible: libgraphaudio_hip has no CPU fallback");
  if (dev < 0 || dev >= count) fail(GA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  device = dev;
  GA_HIP(hipSetDevice(dev));
  GA_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  ownStream = true;
  // twiddle tables in double precision, computed like the oracle does (libm cos/sin)
  const double pi = 3.14159265358979323846264338327950288;
  std::vector<double2> t128(64), t256(129);
  for (int j = 0; j < 64; j++) t128[j] = make_double2(std::cos(2.0 * pi * j / 128.0), -std::sin(2.0 * pi * j / 128.0));
  for (int k = 0; k <= 128; k++) t256[k] = make_double2(std::cos(2.0 * pi * k / 256.0), -std::sin(2.0 * pi * k / 256.0));
  w128 = (double2*)dalloc(sizeof(double2) * 64);
  w256 = (double2*)dalloc(sizeof(double2) * 129);
  GA_HIP(hipMemcpy(w128, t128.data(), sizeof(double2) * 64, hipMemcpyHostToDevice));
  GA_HIP(hipMemcpy(w256, t256.data(), sizeof(double2) * 129, hipMemcpyHostToDevice));
  cacheDev = (float*)dalloc(sizeof(float) * 32 * kBlock);
}

Context::~Context() {
  if (stream) (void)hipStreamSynchronize(stream);
  try {
    commDestroy();
  } catch (...) {
  }
  for (auto& b : buffers)
    if (b && b->devBase) (void)hipFree(b->devBase);
  for (auto& g : groups) {
    if (g->histR) (void)hipFree(g->histR);
    if (g->histI) (void)hipFree(g->histI);
    if (g->overlap[0]) (void)hipFree(g->overlap[0]);
    if (g->overlap[1]) (void)hipFree(g->overlap[1]);
  }
  for (auto& a : planes)
    if (a.p) (void)hipFree(a.p);
  for (auto& a : planesB)
    if (a.p) (void)hipFree(a.p);
  for (auto& a : planesBalt)
    if (a.p) (void)hipFree(a.p);
  for (auto& np : nodes) {
    if (np && np->bHistR) (void)hipFree(np->bHistR);
    if (np && np->bHistI) (void)hipFree(np->bHistI);
    if (np && np->bOverlap) (void)hipFree(np->bOverlap);
    if (np && np->dHistBase[0]) (void)hipFree(np->dHistBase[0]);
    if (np && np->dHistBase[1]) (void)hipFree(np->dHistBase[1]);
    if (np && np->dTail[0]) (void)hipFree(np->dTail[0]);
    if (np && np->dTail[1]) (void)hipFree(np->dTail[1]);
    if (np && np->stWin[0]) (void)hipFree(np->stWin[0]);
    if (np && np->stWin[1]) (void)hipFree(np->stWin[1]);
    if (np && np->delayHist) (void)hipFree(np->delayHist);
    if (np && np->delayLine) (void)hipFree(np->delayLine);
    if (np && np->oscPhase) (void)hipFree(np->oscPhase);
    if (np && np->panDev) (void)hipFree(np->panDev);
  }
  for (auto& kv : tw16) (void)hipFree(kv.second);
  if (coarseTw) (void)hipFree(coarseTw);
  if (tw16pw) (void)hipFree(tw16pw);
  if (copyStream) {
    (void)hipStreamSynchronize(copyStream);
    (void)hipStreamDestroy(copyStream);
    for (int k = 0; k < 2; k++) {
      if (outReady[k]) (void)hipEventDestroy(outReady[k]);
      if (outCopied[k]) (void)hipEventDestroy(outCopied[k]);
      if (outStage[k]) (void)hipFree(outStage[k]);
    }
  }
  if (stream2) {
    (void)hipStreamSynchronize(stream2);
    (void)hipStreamDestroy(stream2);
    for (auto& e : dGroupEv)
      if (e) (void)hipEventDestroy(e);
    if (dJoinEv) (void)hipEventDestroy(dJoinEv);
  }
  for (auto& x : extraProf) { (void)hipEventDestroy(x.e0); (void)hipEventDestroy(x.e1); }
  if (coarseX.p) (void)hipFree(coarseX.p);
  if (coarseY.p) (void)hipFree(coarseY.p);
  if (coarseM.p) (void)hipFree(coarseM.p);
  if (deferStage) (void)hipFree(deferStage);
  for (auto& r : retired) (void)hipFree(r.first);
  for (auto& kv : resamplers)
    if (kv.second->devSamples) (void)hipFree(kv.second->devSamples);
  for (auto& np : nodes)
    if (np && np->staleBuf) {
      (void)hipFree(np->staleBuf);
      (void)hipFree(np->staleNext);
    }
  for (float* p : bqSplitBlocks) (void)hipFree(p);
  if (ilvDev) (void)hipFree(ilvDev);
  if (tables.p) (void)hipFree(tables.p);
  if (tablesHost) (void)hipHostFree(tablesHost);
  if (tablesHostB) (void)hipHostFree(tablesHostB);
  for (auto& e : chunkDone)
    if (e) (void)hipEventDestroy(e);
  for (auto& b : pendingProf) {
    (void)hipEventDestroy(b.begin);
    (void)hipEventDestroy(b.end);
    for (auto& e : b.evs) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  }
  for (void* p : slabBlocks) (void)hipFree(p);
  for (void* p : bqBlocks) (void)hipFree(p);
  for (float* p : busSlabs) (void)hipFree(p);
  if (zeros) (void)hipFree(zeros);
  if (w128) (void)hipFree(w128);
  if (w256) (void)hipFree(w256);
  for (auto& kv : twC) (void)hipFree(kv.second);
  if (cacheDev) (void)hipFree(cacheDev);
  if (stream && ownStream) (void)hipStreamDestroy(stream);
}

// ------------------------------------------------------------------------------------------------------
// command queue (AudioContextBase.cs:266-305)
// ------------------------------------------------------------------------------------------------------
void Context::executeOrPost(std::function<void()> cmd) {
  if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
  if (latched && !inRender) cmd(); else pending.push_back(std::move(cmd));
}
void Context::post(std::function<void()> cmd) {
  if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
  pending.push_back(std::move(cmd));
}
void Context::drain() {  // exceptions thrown by queued commands are swallowed (:276-282)
  while (!pending.empty()) {
    auto cmd = std::move(pending.front());
    pending.pop_front();
    apiEpoch++;   // (whatever the command does, the next chunk simulates from scratch)
    try {
      cmd();
    } catch (const Err&) {
    }
  }
}

NodeS* Context::node(int id) {
  if (id < 0 || id >= (int)nodes.size()) fail(GA_ERR_INVALID_ARGUMENT, "bad node id");
  return nodes[id].get();
}
ParamS* Context::param(int nid, int p) {
  NodeS* n = node(nid);
  if (p < 0 || p >= (int)n->params.size()) fail(GA_ERR_INVALID_ARGUMENT, "bad param index");
  return &n->params[p];
}
PlayBuf* Context::buffer(int id) {
  if (id < 0) return nullptr;
  if (id >= (int)buffers.size() || !buffers[id]) fail(GA_ERR_INVALID_ARGUMENT, "bad buffer id");
  return buffers[id].get();
}
// Storage lives as long as something can still play it: PlayableAudioBuffer objects are garbage-collected in the reference, here
// the host says when it dropped its handle (ga_buffer_release) and nodes keep what they use alive.  Runs between renders.
void Context::collectGarbage() {
  chunksSinceGc = 0;
  if (!releasedPending.empty()) {
    std::vector<char> used(buffers.size(), 0);
    for (auto& np : nodes) {
      if (!np || np->disposed) continue;
      if (np->bufId >= 0 && np->bufId < (int)used.size()) used[np->bufId] = 1;
      if (np->irBuf >= 0 && np->irBuf < (int)used.size()) used[np->irBuf] = 1;
      for (auto& e : np->dHistExt)   // a convolver's input history that is a span of the buffer (NodeS::dHistExt)
        if (e.first && e.second >= 0 && e.second < (int)used.size()) used[e.second] = 1;
      if (np->type == GA_NODE_STREAM_SOURCE) {   // queued, current and processed (not yet handed back) buffers of a stream
        if (np->stCurrent >= 0 && np->stCurrent < (int)used.size()) used[np->stCurrent] = 1;
        for (int id : np->stQueued) if (id >= 0 && id < (int)used.size()) used[id] = 1;
        for (int id : np->stProcessed) if (id >= 0 && id < (int)used.size()) used[id] = 1;
      }
    }
    std::vector<int> keep;
    bool synced = false;
    for (int id : releasedPending) {
      if (id < 0 || id >= (int)buffers.size() || !buffers[id]) continue;
      if (used[id]) {
        keep.push_back(id);
        continue;
      }
      if (!synced && stream) {
        (void)hipStreamSynchronize(stream);
        synced = true;
      }
      PlayBuf& b = *buffers[id];
      if (b.devBase) dfree(b.devBase, b.devBytes);
      bufVersion++;
      for (auto it = irCache.begin(); it != irCache.end();)   // spectra built from it stay with the convolvers that hold them
        it = it->first.first == id ? irCache.erase(it) : std::next(it);
      buffers[id].reset();
    }
    releasedPending.swap(keep);
  }
  // impulse-response spectra that only the cache still holds (their convolvers are gone or use another buffer)
  for (auto it = irCache.begin(); it != irCache.end();) {
    if (it->second.use_count() == 1) {
      if (stream) (void)hipStreamSynchronize(stream);
      it = irCache.erase(it);
    } else {
      ++it;
    }
  }
}

InputS* Context::inputOf(const InRef& r) {
  NodeS* n = nodes[r.node].get();
  if (r.input >= 0) return &n->inputs[r.input];
  return nullptr;  // param modulation inputs keep their list in ParamS::modulation
}

void Context::harvestProfile(bool wait) {
  while (!pendingProf.empty()) {
    ProfBatch& b = pendingProf.front();
    if (wait) GA_HIP(hipEventSynchronize(b.end));
    else if (hipEventQuery(b.end) != hipSuccess) { (void)hipGetLastError(); break; }
    float ms = 0;
    GA_HIP(hipEventElapsedTime(&ms, b.begin, b.end));
    stats.device_ms_total += ms;
    stats.profiled_chunks++;
    for (size_t i = 0; i < b.evs.size(); i++) {
      GA_HIP(hipEventElapsedTime(&ms, b.evs[i].first, b.evs[i].second));
      const int k = b.kinds[i];
      if (k == LK_MAC || k == LK_CMAC) stats.mac_ms_total += ms;
      else if (k == LK_FFT || k == LK_IFFT || k == LK_CFWD || k == LK_CINV || k == LK_CHIST) stats.fft_ms_total += ms;
      else if (k != GA_STAGE_COARSE_SECTION) stats.other_ms_total += ms;
      if (k >= 0 && k < 16) stats.stage_ms[k] += ms;   // (launches and bytes are counted when the launch is enqueued)
      (void)hipEventDestroy(b.evs[i].first);
      (void)hipEventDestroy(b.evs[i].second);
    }
    (void)hipEventDestroy(b.begin);
    (void)hipEventDestroy(b.end);
    pendingProf.pop_front();
  }
}
void Context::flushHandOver() {
  for (const HandOver& h : pendingHandOver)
    GA_HIP(hipMemcpyAsync(h.dst_host, h.src, sizeof(float) * (size_t)h.n, hipMemcpyDeviceToHost, stream));
  pendingHandOver.clear();
}
void Context::synchronize() {
  GA_HIP(hipSetDevice(device));
  flushHandOver();
  commWait();   // (= hipStreamSynchronize without a communicator)
  waitHostCopies();
  harvestProfile(true);
  freeRetired();
}
void Context::freeRetired() {
  for (auto& r : retired) dfree(r.first, r.second);
  retired.clear();
}
void Context::waitHostCopies() {
  if (!copyStream) return;
  GA_HIP(hipStreamSynchronize(copyStream));
  outPending[0] = outPending[1] = false;
}
// device rows src[ch][0, frames) -> out[ch] + offset (host memory), off the context's stream (ga_engine.hpp)
void Context::handOverToHost(const float* const* src, float* const* out, int channels, int64_t offset, int64_t frames) {
  if (!copyStream) {
    GA_HIP(hipStreamCreateWithFlags(&copyStream, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
      GA_HIP(hipEventCreateWithFlags(&outReady[k], hipEventDisableTiming));
      GA_HIP(hipEventCreateWithFlags(&outCopied[k], hipEventDisableTiming));
    }
  }
  const int k = outCur;
  outCur ^= 1;
  const size_t need = (size_t)channels * (size_t)frames * sizeof(float);
  if (outStageBytes[k] < need) {
    if (outPending[k]) GA_HIP(hipEventSynchronize(outCopied[k]));
    outPending[k] = false;
    if (outStage[k]) {
      GA_HIP(hipStreamSynchronize(stream));   // (a device-to-device copy into it may still be queued)
      dfree(outStage[k], outStageBytes[k]);
      outStage[k] = nullptr;
      outStageBytes[k] = 0;
    }
    outStage[k] = (float*)dalloc(need);
    outStageBytes[k] = need;
  }
  if (outPending[k]) GA_HIP(hipStreamWaitEvent(stream, outCopied[k], 0));   // the buffer's previous contents have left
  for (int ch = 0; ch < channels; ch++)
    GA_HIP(hipMemcpyAsync(outStage[k] + (size_t)ch * frames, src[ch], sizeof(float) * (size_t)frames, hipMemcpyDeviceToDevice, stream));
  GA_HIP(hipEventRecord(outReady[k], stream));
  GA_HIP(hipStreamWaitEvent(copyStream, outReady[k], 0));
  for (int ch = 0; ch < channels; ch++)
    GA_HIP(hipMemcpyAsync(out[ch] + offset, outStage[k] + (size_t)ch * frames, sizeof(float) * (size_t)frames, hipMemcpyDeviceToHost, copyStream));
  GA_HIP(hipEventRecord(outCopied[k], copyStream));
  outPending[k] = true;
}

// ---- connections: AudioNodeOutput.ConnectTo / DisconnectFrom / DisconnectAll (AudioNodeOutput.cs:42-70) and
//      AudioNodeInput.AddConnection / RemoveConnection / DisconnectAll (AudioNodeInput.cs:60-83) ----
void Context::connectTo(int src, int out, InRef in) {
  graphVersion++;
  if (in.node == src) fail(GA_ERR_INVALID_OPERATION, "Cannot connect a node to itself");
  OutputS& o = nodes[src]->outputs[out];
  if (std::find(o.connectedInputs.begin(), o.connectedInputs.end(), in) != o.connectedInputs.end()) return;
  o.connectedInputs.push_back(in);
  if (in.input >= 0) {
    InputS& i = nodes[in.node]->inputs[in.input];
    Conn c{src, out};
    if (std::find(i.connected.begin(), i.connected.end(), c) == i.connected.end()) {
      i.connected.push_back(c);
      i.dirty = true;
    }
  } else {
    auto& m = nodes[in.node]->params[-1 - in.input].modulation;
    std::pair<int, int> c{src, out};
    if (std::find(m.begin(), m.end(), c) == m.end()) m.push_back(c);
  }
}
static void removeFromInput(Context& c, int src, int out, InRef in) {
  if (in.input >= 0) {
    InputS& i = c.nodes[in.node]->inputs[in.input];
    Conn cc{src, out};
    auto it = std::find(i.connected.begin(), i.connected.end(), cc);
    if (it != i.connected.end()) i.connected.erase(it);
    i.dirty = true;
  } else {
    auto& m = c.nodes[in.node]->params[-1 - in.input].modulation;
    std::pair<int, int> cc{src, out};
    auto it = std::find(m.begin(), m.end(), cc);
    if (it != m.end()) m.erase(it);
  }
}
void Context::disconnectFrom(int src, int out, InRef in) {
  graphVersion++;
  OutputS& o = nodes[src]->outputs[out];
  auto it = std::find(o.connectedInputs.begin(), o.connectedInputs.end(), in);
  if (it != o.connectedInputs.end()) {
    o.connectedInputs.erase(it);
    removeFromInput(*this, src, out, in);
  }
}
void Context::outputDisconnectAll(int src, int out) {
  graphVersion++;
  OutputS& o = nodes[src]->outputs[out];
  std::vector<InRef> ins = o.connectedInputs;
  o.connectedInputs.clear();
  for (const InRef& in : ins) removeFromInput(*this, src, out, in);
}
void Context::inputDisconnectAll(InRef in) {
  std::vector<Conn> outs;
  if (in.input >= 0) {
    outs = nodes[in.node]->inputs[in.input].connected;
  } else {
    for (auto& m : nodes[in.node]->params[-1 - in.input].modulation) outs.push_back(Conn{m.first, m.second});
  }
  for (const Conn& c : outs) disconnectFrom(c.node, c.out, in);
  if (in.input >= 0) nodes[in.node]->inputs[in.input].dirty = true;
}
// DoDispose, Nodes/AudioNode.cs:212-235
void Context::doDispose(int id) {
  graphVersion++;
  NodeS& n = *nodes[id];
  if (n.disposed) return;
  n.disposed = true;
  for (int o = 0; o < (int)n.outputs.size(); o++) outputDisconnectAll(id, o);
  for (int i = 0; i < (int)n.inputs.size(); i++) {
    inputDisconnectAll(InRef{id, i});
    n.inputs[i].bufCh = 0;  // AudioNodeInput.Dispose returns the buffer (:88-95)
  }
  for (int p = 0; p < (int)n.params.size(); p++) inputDisconnectAll(InRef{id, -1 - p});
  // OnDispose
  if (n.type == GA_NODE_DELAY) {   // the delay lines go with the node
    if (stream) (void)hipStreamSynchronize(stream);
    if (n.delayHist) dfree(n.delayHist, (size_t)n.maxDelaySamples * n.delayHistRings * sizeof(float));
    if (n.delayLine) dfree(n.delayLine, ((size_t)n.maxDelaySamples + (size_t)n.delayCap) * n.delayLineRings * sizeof(float));
    n.delayHist = n.delayLine = nullptr;
    n.delayHistRings = n.delayLineRings = 0;
    n.delayCap = 0;
  }
  if (n.staleBuf) {   // (feedback cycles: the kept output block goes with the node)
    if (stream) (void)hipStreamSynchronize(stream);
    dfree(n.staleBuf, (size_t)n.staleRows * kBlock * sizeof(float));
    dfree(n.staleNext, (size_t)n.staleRows * kBlock * sizeof(float));
    n.staleBuf = n.staleNext = nullptr;
    n.staleRows = 0;
  }
  if (n.type == GA_NODE_BUFFER_SOURCE) n.bufId = -1;  // AudioBufferSourceNode.cs:412
  if (n.type == GA_NODE_STREAM_SOURCE && n.stState != GA_STREAM_STOPPED) {   // AudioStreamSourceNodeBase.cs:315-327
    n.stState = GA_STREAM_STOPPED;
    streamFlushToProcessed(n);
  }
  if (n.type == GA_NODE_CONVOLVER) {                   // ConvolverNode.cs:166-175
    releaseConvState(n);   // while n.ir still tells the size of the private history (device byte accounting)
    n.ir.reset();
    n.irBuf = -1;
  }
}

// a node loses its PartitionedConvolver instances (Buffer reassigned / disposed): rows and private state are released
void Context::releaseConvState(NodeS& n) {
  graphVersion++;   // the node's impulse response (hence its convolver depth contribution) changes
  for (auto& r : n.convRows)
    if (r.group) r.group->rows[r.idx] = {-1, 0};
  n.convRows.clear();
  if (n.bHistR || n.bOverlap) {
    if (stream) (void)hipStreamSynchronize(stream);
    if (n.ir) {
      size_t hb = (size_t)n.bInCh * kBins * std::max(n.ir->P - 1, 1) * sizeof(float);
      dfree(n.bHistR, hb);
      dfree(n.bHistI, hb);
    } else {
      (void)hipFree(n.bHistR);
      (void)hipFree(n.bHistI);
    }
    dfree(n.bOverlap, (size_t)n.bSlots * 2 * kBlock * sizeof(float));
  }
  if (n.dHist[0]) {
    if (stream) (void)hipStreamSynchronize(stream);
    dfree(n.dHistBase[0], n.dHistBytes);
    dfree(n.dHistBase[1], n.dHistBytes);
    n.dHistBase[0] = n.dHistBase[1] = nullptr;
  }
  if (n.dTail[0]) {
    if (stream) (void)hipStreamSynchronize(stream);
    for (int b = 0; b < 2; b++) dfree(n.dTail[b], (size_t)n.dTailLen * n.dTailCh * sizeof(float));
  }
  n.dTail[0] = n.dTail[1] = nullptr;
  n.dTailLen = 0;
  n.dTailCh = n.dTailCur = 0;
  n.dTailSig = 0;
  n.dTailSeq = ~0ull - 1;
  n.dHist[0] = n.dHist[1] = nullptr;
  n.dHistExt.clear();
  n.dHistLen = 0;
  n.dHistCur = 0;
  n.dHistZero = true;
  n.dLeader = -1;
  n.dGroupSize = 0;
  n.everFed = false;   // (new PartitionedConvolver instances: an empty delay line, exact zeros out until something arrives)
  fusionKeyValid = false;
  n.bHistR = n.bHistI = n.bOverlap = nullptr;
  n.bHistPlane = -1;
  n.convPath = 0;
  n.bShared = true;
  n.bHistZero = true;
  n.bOvCur = 0;
}

// Formulation B/C keeps the last P-1 input spectra of a node in the x planes of the chunk that produced them; the next
// chunk writes the OTHER pair of planes and reads them from there.  A node that did not run in that next chunk would lose
// them one chunk later, so before a pair is rewritten its remaining residents move to their private stores.
void Context::flushPlaneHistories(int pair) {
  std::vector<HistJobB> jobs;
  int maxh = 1;
  for (int id : bResidents[pair]) {
    NodeS* n = node(id);
    if (!n || n->bHistPlane != pair || !n->ir || !n->bHistR) continue;
    const int h = n->ir->P - 1;
    const size_t hstride = (size_t)kBins * std::max(h, 1);
    for (int c = 0; c < n->bHistNx && h > 0; c++) {
      const size_t off = (size_t)(n->bHistRow + c) * kBins * n->bHistTxb + n->bHistOff;
      jobs.push_back(HistJobB{n->bHistR + c * hstride, xPlane(pair, 0) + off, h, n->bHistTxb, h, 0});
      jobs.push_back(HistJobB{n->bHistI + c * hstride, xPlane(pair, 1) + off, h, n->bHistTxb, h, 0});
    }
    maxh = std::max(maxh, h);
    n->bHistPlane = -1;
  }
  bResidents[pair].clear();
  if (jobs.empty()) return;
  HistJobB* tab = nullptr;
  GA_HIP(hipHostMalloc((void**)&tab, jobs.size() * sizeof(HistJobB), hipHostMallocDefault));
  memcpy(tab, jobs.data(), jobs.size() * sizeof(HistJobB));
  launch_hist_copy_b(stream, tab, (int)jobs.size(), maxh);
  GA_HIP(hipGetLastError());
  GA_HIP(hipStreamSynchronize(stream));
  GA_HIP(hipHostFree(tab));
}

// one row (= one PartitionedConvolver instance) in the shared-IR group of (IR, IR channel, convolver depth)
ConvRowRef Context::addGroupRow(const std::shared_ptr<IrSpectra>& ir, int ch, int depth, int nodeId) {
  auto key = std::make_tuple(ir.get(), ch, depth);
  ConvGroup* g;
  auto it = groupOf.find(key);
  if (it == groupOf.end()) {
    auto ng = std::make_unique<ConvGroup>();
    ng->ir = ir;
    ng->irCh = ch;
    ng->depth = depth;
    ng->P = ir->P;
    g = ng.get();
    groups.push_back(std::move(ng));
    groupOf[key] = g;
  } else {
    g = it->second;
  }
  int idx = (int)g->rows.size();   // rows are append-only while a group holds live state
  g->rows.push_back({nodeId, ch});
  if (idx < g->rp && g->histR) {   // the column may hold stale scratch data: a new convolver starts from zero state
    const int hist = g->P - 1;
    if (hist > 0) {
      GA_HIP(hipMemset2DAsync(g->histR + idx, (size_t)g->rp * 4, 0, 4, (size_t)kBins * hist, stream));
      GA_HIP(hipMemset2DAsync(g->histI + idx, (size_t)g->rp * 4, 0, 4, (size_t)kBins * hist, stream));
    }
    GA_HIP(hipMemsetAsync(g->overlap[0] + (size_t)idx * kBlock, 0, kBlock * 4, stream));
    GA_HIP(hipMemsetAsync(g->overlap[1] + (size_t)idx * kBlock, 0, kBlock * 4, stream));
  }
  return ConvRowRef{g, idx};
}

// Decide, for convolver nodes that do not have DSP state yet, which formulation serves them: nodes sharing an impulse
// response with at least 7 others become rows of the shared-IR GEMM (A); nodes with a (nearly) private IR use the
// per-node formulation (B), whose state is O(P) per input channel instead of O(P * 128 rows) per IR channel.
// Which convolvers have to be evaluated in the reference's own order (formulation R, launch_refmac)?  Every device formulation of
// the partition sum (A/B: matrix-core fma chains, C: FFT along the block axis, D: coarse partitions) associates the reference's
// sequential float32 sum differently, so a convolver's output differs from the reference's in its last bits (~1e-7 relative).
// That is far inside the 1e-5 contract -- unless something downstream amplifies or quantises it:
//  * a parameter input (AudioParam.cs:97-101,123-135): DelayNode turns the value into a sample index ((int)(delayTime * sampleRate),
//    DelayNode.cs:66,86), StereoPannerNode compares it with the previous value (StereoPannerNode.cs:92-99), BiQuadFilterNode
//    thresholds it (BiQuadFilterNode.cs:126) -- a last-bit difference moves a whole sample / a coefficient set;
//  * a biquad whose direct-form-II recursion has a large round-off noise gain (a resonant section at a low cut-off frequency): two
//    float32 evaluations whose inputs differ in the last bit decorrelate and differ by ~sqrt(2) x that noise (biquadDeviation).
// A convolver is sensitive when its output reaches such a sink through any chain of nodes (mixes included: they are bit-exact on
// the device).  One reverse sweep over the processing order, per chunk.
void Context::refOrderSensitivity(const std::vector<int>& topo) {
  bool anyConv = false;
  for (int id : topo) anyConv = anyConv || (nodes[id]->type == GA_NODE_CONVOLVER && nodes[id]->ir);
  if (!anyConv) return;
  if (convRefOrder != 1) {
    for (int id : topo)
      if (nodes[id]->type == GA_NODE_CONVOLVER) nodes[id]->refSens = convRefOrder == 2;
    return;
  }
  // (a graph with feedback: the sweep follows the reference's processing order -- the planning order may have cut loops at DelayNodes)
  const std::vector<int>& order = (topoHasCycles && topoRefOrder.size() == topo.size()) ? topoRefOrder : topo;
  // A loop multiplies what enters it by 1 / (1 - gain): only where the estimated loop gain (Context::chunkTopology: constant gains,
  // filter boosts) comes near 1 does a last-bit difference grow -- a master echo at 0.5 doubles it and is left alone.
  const bool wildLoops = topoHasCycles && loopGainBound >= 0.7;
  std::vector<char>& sens = refSensScratch;
  sens.assign(nodes.size(), 0);
  for (auto it = order.rbegin(); it != order.rend(); ++it) {   // consumers before producers
    NodeS& nd = *nodes[*it];
    // * a feedback loop whose gain may come near 1: what enters it is multiplied by 1 / (1 - gain) (fuzz session 50178: 2.7e-5 with the
    //   loop's convolver on formulation B, exactly 0 in the reference's order).  Everything that feeds a node some consumer reads one
    //   block late (a stale producer: the loop's entry) counts -- unless the loops are tame (wildLoops above).
    bool s = nd.staleProducer && wildLoops;
    if (nd.type == GA_NODE_BIQUAD) {
      bool moving = false;
      for (auto& p : nd.params) moving = moving || !p.events.empty() || !p.modulation.empty();
      if (moving) {
        s = true;   // (coefficients on a timeline or modulated: no single noise gain to look at)
      } else {
        const float nyq = sampleRate / 2.f;
        float f = nd.params[0].value;
        f = f < 1.f ? 1.f : (f > nyq ? nyq : f);
        const float q = std::max(0.001f, nd.params[1].value);
        float o[5];
        biquadCoefficients(nd.filterType, (float)sampleRate, f, q, nd.params[2].value, o);
        s = biquadDeviation(o, 1) > convRefMinDeviation;
      }
    }
    for (const OutputS& out : nd.outputs)
      for (const InRef& r : out.connectedInputs) {
        if (r.node < 0 || r.node >= (int)sens.size() || !nodes[r.node] || !nodes[r.node]->reachable) continue;
        if (r.input < 0 || sens[r.node]) s = true;
      }
    sens[*it] = s ? 1 : 0;
    if (nd.type == GA_NODE_CONVOLVER) nd.refSens = s;
  }
}

void Context::assignConvPaths(const std::vector<int>& topo, int64_t chunkBlocks) {
  std::map<IrSpectra*, int> users;
  std::map<IrSpectra*, bool> hasA;
  // a group is executed once per chunk at ONE convolver depth, for all of its rows; a graph edit that moved a node to another
  // depth moves its rows (FDL column + overlap tail) to the group of that depth.  A node that is no longer reachable from the
  // destination is not processed at all (pull model): its rows wait in a group of their own (depth -1, never executed), so
  // that the delay line keeps its content, and come back the same way when the node is connected again.
  auto moveRows = [&](NodeS& nd, int id, int depth) {
      for (size_t slot = 0; slot < nd.convRows.size(); slot++) {
        ConvRowRef old = nd.convRows[slot];
        if (old.group->depth == depth) continue;
        ConvRowRef nw = addGroupRow(nd.ir, old.group->irCh, depth, id);
        ensureGroupState(*nw.group);
        ConvGroup& og = *old.group;
        ConvGroup& ng = *nw.group;
        const int hist = og.P - 1;
        if (hist > 0 && !og.histZero && og.histR) {
          GA_HIP(hipMemcpy2DAsync(ng.histR + nw.idx, (size_t)ng.rp * 4, og.histR + old.idx, (size_t)og.rp * 4, 4,
                                  (size_t)kBins * hist, hipMemcpyDeviceToDevice, stream));
          GA_HIP(hipMemcpy2DAsync(ng.histI + nw.idx, (size_t)ng.rp * 4, og.histI + old.idx, (size_t)og.rp * 4, 4,
                                  (size_t)kBins * hist, hipMemcpyDeviceToDevice, stream));
          ng.histZero = false;   // (a fresh group's arrays are zero-filled, so the other columns stay correct)
        }
        if (og.overlap[og.ovCur])
          GA_HIP(hipMemcpyAsync(ng.overlap[ng.ovCur] + (size_t)nw.idx * kBlock, og.overlap[og.ovCur] + (size_t)old.idx * kBlock,
                                kBlock * 4, hipMemcpyDeviceToDevice, stream));
        og.rows[old.idx] = {-1, 0};
        nd.convRows[slot] = nw;
      }
  };
  for (auto& np : nodes) {
    if (!np) continue;
    NodeS& nd = *np;
    if (nd.type == GA_NODE_CONVOLVER && nd.ir && nd.convPath == 1 && !nd.reachable && !nd.disposed) moveRows(nd, nd.id, -1);
  }
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_CONVOLVER || !nd.ir) continue;
    users[nd.ir.get()]++;
    if (nd.convPath == 1) {
      hasA[nd.ir.get()] = true;
      moveRows(nd, id, nd.depth);
    }
  }
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.convPath != 0) continue;
    IrSpectra* ir = nd.ir.get();
    const int channels = ir->nch;
    const bool pathC = useTimeFft && ir->P > 64 && ir->P <= 1024;   // FFT along the block axis (N2 <= 4096)
    // coarse partitions (formulation D): impulse responses of 8,193 .. 131,072 taps when the render comes in long chunks --
    // a short chunk would pay the transforms of the whole input history for a few blocks of output (the node keeps the
    // formulation it starts with: its state is formulation specific)
    const int coarseParts = (int)(((int64_t)ir->P * kBlock + kCoarseBlock - 1) / kCoarseBlock);
    // (a node that has to be evaluated in the reference's order takes the B / C state layout, which formulation R shares)
    // ... unless the node is one of many that share the impulse response and feed one consumer each: such a group is summed in the
    // time domain in front of ONE set of transforms and carries its tail from chunk to chunk (ga_plan_conv.cpp, classifyGroups), which
    // costs little at any chunk length -- the headline graph behind a master echo renders in chunks of the echo's delay (1024 voices,
    // 0.25 s echo: 69 ms per 10 s on formulation C, 22 ms on D).  coarse_min_blocks >= 2^29 still means "never".
    const bool manyShare = users[ir] >= 8 && nd.outputs.size() == 1 && nd.outputs[0].connectedInputs.size() == 1 && coarsePremix &&
                           coarseMinBlocks < ((int64_t)1 << 29);
    const bool pathD = !nd.refSens && useCoarse && useTimeFft && ir->P > 64 && coarseParts <= kCoarseMaxP &&
                       (chunkBlocks >= coarseMinBlocks || manyShare);
    const bool pathA = !nd.refSens && !pathC && !pathD && (hasA[ir] || users[ir] >= 8);
    if (pathD) {
      nd.bInCh = nd.isTrueStereo ? 2 : channels;
      nd.bSlots = nd.isTrueStereo ? 4 : channels;
      ensureCoarseSpectra(*ir);
      nd.dHistLen = (int64_t)ir->coarseP * kCoarseBlock;
      const size_t hb = (size_t)nd.bInCh * (size_t)nd.dHistLen * sizeof(float);
      nd.dHist[0] = dallocSkewed(hb, &nd.dHistBase[0], &nd.dHistBytes);
      nd.dHist[1] = dallocSkewed(hb, &nd.dHistBase[1], &nd.dHistBytes);
      nd.dHistCur = 0;
      nd.dHistZero = true;   // (never read while the flag is set)
      nd.bShared = true;
      nd.convPath = 4;
    } else if (pathA) {
      for (int ch = 0; ch < channels; ch++) nd.convRows.push_back(addGroupRow(nd.ir, ch, nd.depth, id));
      nd.convPath = 1;
    } else {
      nd.bInCh = nd.isTrueStereo ? 2 : channels;
      nd.bSlots = nd.isTrueStereo ? 4 : channels;
      size_t hb = (size_t)nd.bInCh * kBins * std::max(ir->P - 1, 1) * sizeof(float);
      size_t ob = (size_t)nd.bSlots * 2 * kBlock * sizeof(float);
      nd.bHistR = (float*)dalloc(hb);
      nd.bHistI = (float*)dalloc(hb);
      nd.bOverlap = (float*)dalloc(ob);
      GA_HIP(hipMemsetAsync(nd.bHistR, 0, hb, stream));
      GA_HIP(hipMemsetAsync(nd.bHistI, 0, hb, stream));
      GA_HIP(hipMemsetAsync(nd.bOverlap, 0, ob, stream));
      nd.bShared = true;
      nd.bHistZero = true;
      nd.bOvCur = 0;
      nd.convPath = pathC ? 3 : 2;
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// BiQuadFilterNode.UpdateCoefficients (BiQuadFilterNode.cs:149-258), float32, host libm
// ------------------------------------------------------------------------------------------------------
// the RBJ table itself (BiQuadFilterNode.cs:149-258), float32, host libm: o = {b0, b1, b2, a1, a2} / a0
void biquadCoefficients(int filterType, float sampleRate, float frequency, float q, float gain, float o[5]) {
  const float PI = 3.14159274f;
  float w0 = 2.f * PI * frequency / sampleRate;
  float cosW0 = std::cos(w0);
  float sinW0 = std::sin(w0);
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
      float A = std::pow(10.f, gain / 40.f);
      B0 = 1.f + alpha * A; B1 = -2.f * cosW0; B2 = 1.f - alpha * A;
      a0 = 1.f + alpha / A; A1 = -2.f * cosW0; A2 = 1.f - alpha / A;
      break;
    }
    case GA_FILTER_LOWSHELF: {
      float A = std::pow(10.f, gain / 40.f);
      float sqrtA = std::sqrt(A);
      float beta = sqrtA / q;
      B0 = A * ((A + 1.f) - (A - 1.f) * cosW0 + beta * sinW0);
      B1 = 2.f * A * ((A - 1.f) - (A + 1.f) * cosW0);
      B2 = A * ((A + 1.f) - (A - 1.f) * cosW0 - beta * sinW0);
      a0 = (A + 1.f) + (A - 1.f) * cosW0 + beta * sinW0;
      A1 = -2.f * ((A - 1.f) + (A + 1.f) * cosW0);
      A2 = (A + 1.f) + (A - 1.f) * cosW0 - beta * sinW0;
      break;
    }
    case GA_FILTER_HIGHSHELF: {
      float A = std::pow(10.f, gain / 40.f);
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
  o[0] = B0 / a0;
  o[1] = B1 / a0;
  o[2] = B2 / a0;
  o[3] = A1 / a0;
  o[4] = A2 / a0;
}
void Context::updateBiquadCoefficients(NodeS& n, float frequency, float q, float gain) {
  // The reference recomputes the coefficients at sample 0 of every block (BiQuadFilterNode.cs:111-126: its "last" values are
  // never updated) -- with unchanged parameters that is the same arithmetic on the same inputs: remember the result.
  if (n.coefMemoValid && n.coefMemoType == n.filterType && n.coefMemoIn[0] == frequency && n.coefMemoIn[1] == q && n.coefMemoIn[2] == gain) {
    n.b0 = n.coefMemoOut[0]; n.b1 = n.coefMemoOut[1]; n.b2 = n.coefMemoOut[2]; n.a1 = n.coefMemoOut[3]; n.a2 = n.coefMemoOut[4];
    return;
  }
  float o[5];
  biquadCoefficients(n.filterType, (float)sampleRate, frequency, q, gain, o);
  n.b0 = o[0];
  n.b1 = o[1];
  n.b2 = o[2];
  n.a1 = o[3];
  n.a2 = o[4];
  n.coefMemoValid = true;
  n.coefMemoType = n.filterType;
  n.coefMemoIn[0] = frequency; n.coefMemoIn[1] = q; n.coefMemoIn[2] = gain;
  n.coefMemoOut[0] = n.b0; n.coefMemoOut[1] = n.b1; n.coefMemoOut[2] = n.b2; n.coefMemoOut[3] = n.a1; n.coefMemoOut[4] = n.a2;
}

// ------------------------------------------------------------------------------------------------------
// IR preparation (PartitionedConvolver..ctor / PrepareImpulseResponse / CalculateNormalizationScale,
// PartitionedConvolver.cs:37-102): scale on the host in the reference's mixed precision, spectra with the same
// rfft256 kernel the render path uses.
// ------------------------------------------------------------------------------------------------------
static float normalizationScale(const float* r, int64_t len) {
  const float GainCalibration = -58;
  const float MinPower = 0.000125f;
  double sumSquared = 0;
  for (int64_t i = 0; i < len; i++) {
    float p = r[i] * r[i];
    sumSquared += p;
  }
  float power = (float)std::sqrt(sumSquared / (double)len);
  if (std::isnan(power) || std::isinf(power) || power < MinPower) power = MinPower;
  float e = GainCalibration * 0.05f;
  return (1.0f / power) * (float)std::pow(10.0, (double)e);
}

std::shared_ptr<IrSpectra> Context::irSpectra(int bufId, bool normalize) {
  auto key = std::make_pair(bufId, normalize ? 1 : 0);
  auto it = irCache.find(key);
  if (it != irCache.end()) return it->second;
  PlayBuf* b = buffer(bufId);
  if (b->length <= 0) fail(GA_ERR_INVALID_ARGUMENT, "impulse response buffer is empty");
  const int P = (int)((b->length + kBlock - 1) / kBlock);  // ceil(len / blockSize), :44
  const int nch = b->channels;
  auto sp = std::make_shared<IrSpectra>();
  sp->P = P;
  sp->nch = nch;
  const int64_t padded = (int64_t)P * kBlock;
  std::vector<float> scaled((size_t)nch * padded, 0.f);
  for (int c = 0; c < nch; c++) {
    float scale = normalize ? normalizationScale(b->host[c].data(), b->length) : 1.0f;
    float* dst = &scaled[(size_t)c * padded];
    const float* src = b->host[c].data();
    for (int64_t i = 0; i < b->length; i++) dst[i] = src[i] * scale;  // float multiply (:80)
  }
  GA_HIP(hipSetDevice(device));
  const int rp = 128;  // nch <= 32
  float* dIn = (float*)dalloc(scaled.size() * sizeof(float));
  size_t planeBytes = (size_t)kBins * P * rp * sizeof(float);
  float* xr = (float*)dalloc(planeBytes);
  float* xi = (float*)dalloc(planeBytes);
  size_t hBytes = (size_t)nch * kBins * P * sizeof(float);
  sp->hBytes = hBytes;
  sp->devBytesRef = &devBytes;
  sp->hr = (float*)dalloc(hBytes);
  sp->hi = (float*)dalloc(hBytes);
  ConvRowIO* rowsDev = (ConvRowIO*)dalloc(sizeof(ConvRowIO) * nch);
  std::vector<ConvRowIO> rows(nch);
  for (int c = 0; c < nch; c++) rows[c] = ConvRowIO{dIn + (size_t)c * padded, nullptr};
  GA_HIP(hipMemcpyAsync(dIn, scaled.data(), scaled.size() * sizeof(float), hipMemcpyHostToDevice, stream));
  GA_HIP(hipMemcpyAsync(rowsDev, rows.data(), sizeof(ConvRowIO) * nch, hipMemcpyHostToDevice, stream));
  GA_HIP(hipMemsetAsync(xr, 0, planeBytes, stream));
  GA_HIP(hipMemsetAsync(xi, 0, planeBytes, stream));
  ConvPlanes pl{xr, xi, nullptr, nullptr, P, 0, rp};
  // grid.y is limited to 65535 blocks per launch
  for (int t0 = 0; t0 < P; t0 += 32768) {
    int nb = std::min(32768, P - t0);
    std::vector<ConvRowIO> r2(nch);
    for (int c = 0; c < nch; c++) r2[c] = ConvRowIO{dIn + (size_t)c * padded + (size_t)t0 * kBlock, nullptr};
    if (t0 > 0) {
      GA_HIP(hipStreamSynchronize(stream));
      GA_HIP(hipMemcpy(rowsDev, r2.data(), sizeof(ConvRowIO) * nch, hipMemcpyHostToDevice));
    }
    launch_rfft_fwd(stream, rowsDev, nch, nb, t0, pl, Twiddles{w128, w256});
  }
  launch_extract_ir(stream, sp->hr, sp->hi, xr, xi, P, rp, P, nch);
  GA_HIP(hipStreamSynchronize(stream));
  GA_HIP(hipGetLastError());
  sp->taps = dIn;   // kept: formulation D transforms the scaled taps with its own partition size (ensureCoarseSpectra)
  sp->tapsStride = padded;
  sp->tapsBytes = scaled.size() * sizeof(float);
  dfree(xr, planeBytes);
  dfree(xi, planeBytes);
  dfree(rowsDev, sizeof(ConvRowIO) * nch);
  irCache[key] = sp;
  return sp;
}

const float2* Context::twiddles16(int N2) {
  auto it = tw16.find(N2);
  if (it != tw16.end()) return it->second;
  const double pi = 3.14159265358979323846264338327950288;
  const int R3 = N2 / 256;
  std::vector<float2> t;
  for (int m = 1; m < 16; m++)
    for (int kk = 0; kk < 16; kk++) t.push_back(make_float2((float)std::cos(2.0 * pi * kk * m / 256), (float)-std::sin(2.0 * pi * kk * m / 256)));
  for (int m = 1; m < R3; m++)
    for (int j = 0; j < 256; j++) t.push_back(make_float2((float)std::cos(2.0 * pi * j * m / N2), (float)-std::sin(2.0 * pi * j * m / N2)));
  float2* d = (float2*)dalloc(sizeof(float2) * t.size());
  GA_HIP(hipMemcpy(d, t.data(), sizeof(float2) * t.size(), hipMemcpyHostToDevice));
  tw16[N2] = d;
  return d;
}

// AudioStreamNodeBase.FlushToProcessed (AudioStreamSourceNodeBase.cs:95-116)
void Context::streamFlushToProcessed(NodeS& s) {
  if (s.stCurrent >= 0) s.stProcessed.push_back(s.stCurrent);
  s.stCurrent = -1;
  while (!s.stQueued.empty()) {
    s.stProcessed.push_back(s.stQueued.front());
    s.stQueued.pop_front();
  }
  if (s.stChannels >= 0) {   // resamplers.Clear()
    s.stRsPos = 0.0;
    s.stRsReady = 0;
    s.stWinValid = false;
  }
  s.stPos = 0;
  s.stLastRate = 0;
}

// ---- formulation D (ga_coarse.hip) ----
void Context::ensureOverlapStream() {
  if (stream2) return;
  GA_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
  for (auto& e : dGroupEv) GA_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  GA_HIP(hipEventCreateWithFlags(&dJoinEv, hipEventDisableTiming));
}
const float2* Context::coarseTwab() {
  if (coarseTw) return coarseTw;
  const double pi = 3.14159265358979323846264338327950288;
  std::vector<float2> t(2 * 2049);
  for (int k = 0; k <= 2048; k++) {
    t[k] = make_float2((float)std::cos(2.0 * pi * k / 8192), (float)-std::sin(2.0 * pi * k / 8192));
    t[2049 + k] = make_float2((float)std::cos(2.0 * pi * k / 16384), (float)-std::sin(2.0 * pi * k / 16384));
  }
  coarseTw = (float2*)dalloc(sizeof(float2) * t.size());
  GA_HIP(hipMemcpy(coarseTw, t.data(), sizeof(float2) * t.size(), hipMemcpyHostToDevice));
  return coarseTw;
}
// packed 16,384-point spectra of the coarse partitions [h_p | 0] of every channel, with the power-of-two factors of the
// transform chain folded in: the forward / inverse kernels skip the 1/2 of their even / odd splits (x 2 on X, x 2 on H, x 4
// in the inverse) and the 1 / 4096 of the inverse complex transform, so H carries 1 / (2 * 2 * 4 * 4096) = 2^-16 -- exact.
void Context::ensureCoarseSpectra(IrSpectra& ir) {
  if (ir.coarse) return;
  if (!ir.taps) fail(GA_ERR_INVALID_OPERATION, "internal: impulse response without device taps");
  const int P = (int)((ir.tapsStride + kCoarseBlock - 1) / kCoarseBlock);
  ir.coarseP = P;
  ir.coarseBytes = (size_t)ir.nch * P * kCoarseBins * sizeof(float2);
  ir.coarse = (float2*)dalloc(ir.coarseBytes);
  std::vector<CoarseXRow> rows(ir.nch);
  for (int c = 0; c < ir.nch; c++) {
    CoarseXRow& r = rows[c];
    r.hist = nullptr;
    r.in = ir.taps + (size_t)c * ir.tapsStride;
    r.nvalid = ir.tapsStride;
    r.frame0 = c * P;
    r.n_frames = P;
    r.u0 = 1;            // window p + 1 starts at partition p; its second half is forced to zero
    r.hist_len = 0;
    r.flags = 1;
    r.scale = 1.0f / 65536.0f;
    r.carry = nullptr;
    r.carry_from = 0;
  }
  CoarseXRow* rd = (CoarseXRow*)dalloc(sizeof(CoarseXRow) * rows.size());
  GA_HIP(hipMemcpy(rd, rows.data(), sizeof(CoarseXRow) * rows.size(), hipMemcpyHostToDevice));
  launch_coarse_fwd(stream, rd, ir.nch, P, P, ir.coarse, twiddles16pw(), coarseTwab());
  GA_HIP(hipGetLastError());
  GA_HIP(hipStreamSynchronize(stream));
  dfree(rd, sizeof(CoarseXRow) * rows.size());
}

// forward transform of formulation D: the W_256 table of the second pass followed by the POWER twiddles of the last pass,
// W_4096^(j m) for m = 1, 2, 4, 8 (ga_fft16.hpp, fft16_own<4096, true>)
const float2* Context::twiddles16pw() {
  if (tw16pw) return tw16pw;
  const double pi = 3.14159265358979323846264338327950288;
  std::vector<float2> t;
  for (int m = 1; m < 16; m++)
    for (int kk = 0; kk < 16; kk++) t.push_back(make_float2((float)std::cos(2.0 * pi * kk * m / 256), (float)-std::sin(2.0 * pi * kk * m / 256)));
  for (int m : {1, 2, 4, 8})
    for (int j = 0; j < 256; j++) t.push_back(make_float2((float)std::cos(2.0 * pi * j * m / 4096), (float)-std::sin(2.0 * pi * j * m / 4096)));
  tw16pw = (float2*)dalloc(sizeof(float2) * t.size());
  GA_HIP(hipMemcpy(tw16pw, t.data(), sizeof(float2) * t.size(), hipMemcpyHostToDevice));
  return tw16pw;
}

const float2* Context::twiddlesC(int N2) {
  auto it = twC.find(N2);
  if (it != twC.end()) return it->second;
  const double pi = 3.14159265358979323846264338327950288;
  std::vector<float2> t(N2);
  for (int j = 0; j < N2; j++) t[j] = make_float2((float)std::cos(2.0 * pi * j / N2), (float)-std::sin(2.0 * pi * j / N2));
  float2* d = (float2*)dalloc(sizeof(float2) * N2);
  GA_HIP(hipMemcpy(d, t.data(), sizeof(float2) * N2, hipMemcpyHostToDevice));
  twC[N2] = d;
  return d;
}
const float2* Context::ensureTapSpectra(IrSpectra& ir, int N2) {
  if (N2 != 1024 && N2 != 2048 && N2 != 4096) fail(GA_ERR_INVALID_OPERATION, "internal: no block-axis FFT of " + std::to_string(N2) + " points");
  float2*& h = ir.hspecN[IrSpectra::n2Index(N2)];
  if (h) return h;
  size_t bytes = (size_t)ir.nch * kBins * N2 * sizeof(float2);
  ir.hspecBytes += bytes;
  h = (float2*)dalloc(bytes);
  launch_tap_spectra(stream, h, ir.hr, ir.hi, ir.nch, ir.P, N2, twiddlesC(N2));
  GA_HIP(hipGetLastError());
  return h;
}

// Segments of the overlap-save convolution along the block axis: a segment of N2 points yields N2 - P + 1 blocks.  The chunk
// is covered by the cheapest mix of the three kernel sizes.  Cost model fitted to MI355X measurements at 1024 rows (config 3
// and its 32,768 / 131,072-tap variants): time = alpha(N2) x points + beta x valid blocks; beta is the same for every plan, so
// only alpha counts (ms per 1000 points: 0.58 for 1024 and 2048, 0.70 for 4096, which runs one transform per workgroup).
// P = 512, 3750 blocks -> one segment of 4096 + one of 1024 instead of three of 2048 (measured 4.46 vs 4.56 ms); a 128-block
// render with P = 512 takes one 1024-point segment instead of a 2048-point one.  Larger sizes first, equal sizes in one launch.
std::vector<Context::TconvLaunch> Context::tconvPlan(int nblocks, int P) const {
  static const int sizes[3] = {1024, 2048, 4096};
  static const double perPoint[3] = {0.577, 0.583, 0.70};
  std::vector<double> best(nblocks + 1, 0.0);
  std::vector<int> pick(nblocks + 1, -1);
  for (int rem = 1; rem <= nblocks; rem++) {
    best[rem] = 1e300;
    for (int i = 0; i < 3; i++) {
      const int L = sizes[i] - (P - 1);
      if (L < sizes[i] / 4) continue;   // at least a quarter of a segment has to be output
      const double c = sizes[i] * perPoint[i] + best[std::max(0, rem - L)];
      if (c < best[rem]) { best[rem] = c; pick[rem] = i; }
    }
    if (pick[rem] < 0) {   // P too large for every size (cannot happen for P <= 1024): one size fits all
      pick[rem] = 2;
      best[rem] = 0;
    }
  }
  int count[3] = {0, 0, 0};
  for (int rem = nblocks; rem > 0;) {
    const int i = pick[rem];
    count[i]++;
    rem -= std::min(rem, sizes[i] - (P - 1));
  }
  if (const char* e = expenv("GA_TCONV_PLAN")) {   // measurement only: "n1024,n2048,n4096" segment counts (must cover the chunk)
    int c0 = 0, c1 = 0, c2 = 0;
    if (sscanf(e, "%d,%d,%d", &c0, &c1, &c2) == 3 &&
        (long long)c0 * (1024 - (P - 1)) + (long long)c1 * (2048 - (P - 1)) + (long long)c2 * (4096 - (P - 1)) >= nblocks) {
      count[0] = c0; count[1] = c1; count[2] = c2;
    }
  }
  std::vector<TconvLaunch> plan;
  int tbase = 0;
  for (int i = 2; i >= 0; i--)
    if (count[i] > 0) {
      plan.push_back(TconvLaunch{sizes[i], tbase, count[i]});
      tbase += count[i] * (sizes[i] - (P - 1));
    }
  return plan;
}

// host replay of the CubicResampler position recurrence for unbounded input (CubicResampler.cs:31-60)
void Resampler::extend(int64_t nblocks) {
  while ((int64_t)blocks.size() < nblocks) {
    ResampleBlock rb;
    rb.consumed = consumedEnd;
    rb.pos = posEnd;
    rb.ready = readyEnd;
    rb.produced = kBlock;
    blocks.push_back(rb);
    int64_t in = consumedEnd;
    double Pos = posEnd;
    int ready = readyEnd;
    while (ready < 4) {
      in++;
      ready++;
    }
    for (int o = 0; o < kBlock; o++) {
      int consume = (int)Pos;
      in += consume;
      Pos -= consume;
      if (in > 0xFFFFFFF0ll || pending.size() > ((size_t)1 << 24)) {   // (32-bit index exhausted, or nobody takes the samples: 128 MB of them)
        samplesOk = false;
        pending.clear();
        pending.shrink_to_fit();
      }
      if (samplesOk) pending.push_back(ResampleSample{(uint32_t)in, (float)Pos});   // `t = (float)Pos`, CubicResampler.cs:50
      Pos += rate;
    }
    consumedEnd = in;
    posEnd = Pos;
    readyEnd = ready;
  }
}

}  // namespace ga

// ga_api.cpp -- the C ABI of include/graphaudio_hip.h over the engine (ga_engine.hpp).
// Every entry point validates its arguments like the reference member it replaces, never throws across the
// boundary, and returns 0 or a negative GA_ERR_* code (message via ga_last_error).
#include <algorithm>
#include <cstring>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include "ga_engine.hpp"

using namespace ga;

// GA_BACKTRACE=1 in the environment: print a native backtrace on SIGSEGV (debug aid, off by default)
static void ga_segv_handler(int sig) {
  void* frames[64];
  int n = backtrace(frames, 64);
  const char msg[] = "\n[graphaudio_hip] fatal signal, native backtrace:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void ga_install_handler() {
  const char* e = getenv("GA_BACKTRACE");
  if (e && e[0] == '1') signal(SIGSEGV, ga_segv_handler);
}

struct ga_context {
  Context c;
  explicit ga_context(int sr) : c(sr) {}
};

namespace {
template <class F>
int guardRO(ga_context* h, F&& f) {   // renders and queries: calls that do not change what a later render computes
  if (!h) return GA_ERR_INVALID_ARGUMENT;
  try {
    f(h->c);
    return GA_OK;
  } catch (const Err& e) {
    h->c.lastError = e.msg;
    return e.code;
  } catch (const std::bad_alloc&) {
    h->c.lastError = "out of host memory";
    return GA_ERR_OUT_OF_MEMORY;
  } catch (const std::exception& e) {
    h->c.lastError = e.what();
    return GA_ERR_INVALID_OPERATION;
  } catch (...) {
    h->c.lastError = "unknown error";
    return GA_ERR_INVALID_OPERATION;
  }
}
template <class F>
int guard(ga_context* h, F&& f) {   // everything else: the next chunk simulates its first block from scratch (Context::apiEpoch)
  if (h) h->c.apiEpoch++;
  return guardRO(h, std::forward<F>(f));
}
NodeS* typed(Context& c, int id, int type) {
  NodeS* n = c.node(id);
  if (n->type != type) fail(GA_ERR_INVALID_ARGUMENT, "node has the wrong type for this call");
  return n;
}
float clampf(float v, float mn, float mx) {  // Math.Clamp(float, float, float)
  if (v < mn) return mn;
  if (v > mx) return mx;
  return v;
}
void addEvent(ParamS& p, const ParamEvent& e) {  // AudioParam.AddEvent, AudioParam.cs:333-352
  size_t lo = 0, hi = p.events.size();
  while (lo < hi) {
    size_t mid = (lo + hi) >> 1;
    if (e.time < p.events[mid].time) hi = mid; else lo = mid + 1;
  }
  p.events.insert(p.events.begin() + lo, e);
}
ParamS makeParam(float def, float mn, float mx, bool arate) {
  ParamS p;
  p.def = def;
  p.minv = mn;
  p.maxv = mx;
  p.arate = arate;
  p.value = def;
  return p;
}
}  // namespace

// ------------------------------------------------------------------------------------------------------
// OfflineAudioContext.Render (OfflineAudioContext.cs:30-102) on top of runChunk
// ------------------------------------------------------------------------------------------------------
// chunk size: bounded by the option and by device memory (slabs + convolver planes scale with the block count)
int64_t Context::chunkLimit(int64_t) {
    int64_t limit = maxChunkBlocks;
    {
      size_t freeB = 0, totalB = 0;
      if (hipMemGetInfo(&freeB, &totalB) == hipSuccess) {
        double perBlock = 0, fixedBytes = 0;   // what a chunk needs per block, and whatever its length
        int convRowsMax = 0;
        for (auto& g : groups) convRowsMax = std::max(convRowsMax, (int)((g->rows.size() + 127) / 128 * 128));
        perBlock += (double)convRowsMax * kBins * 4.0 * 4.0;
        for (auto& np : nodes)   // private-IR convolvers (formulations B / C): y rows + x rows in both plane pairs
          if (np->type == GA_NODE_CONVOLVER && np->ir && np->convPath == 4)
            // formulation D: X frames of the input channels + (at worst, nothing fused) Y frames of the slots: 64 KB per 8192 samples
          {
            perBlock += (double)((np->isTrueStereo ? 2 : np->ir->nch) + (np->isTrueStereo ? 4 : np->ir->nch)) * 1024.0;
            // ... and per input row the P' history windows in front of the chunk + the window behind it, per slot the P' blocks of a
            // carried tail: 64 KB each, whatever the chunk's length (1 GB at 1024 voices x 8 partitions)
            const double inCh = np->isTrueStereo ? 2 : np->ir->nch, slots = np->isTrueStereo ? 4 : np->ir->nch;
            fixedBytes += (inCh * (np->ir->coarseP + 1) + slots * (double)kCoarseMaxP) * 65536.0;
          }
          else if (np->type == GA_NODE_CONVOLVER && np->ir && np->convPath != 1)
            perBlock += (double)(np->ir->nch + 2 * (np->isTrueStereo ? 2 : np->ir->nch)) * kBins * 8.0;
        perBlock += ((double)nodes.size() * 2.0 + 64.0) * kBlock * 4.0;
        double budget = ((double)freeB + (double)slabBlocks.size() * (double)((size_t)1 << 30) * 0.0) * memBudgetFraction;
        // memory already held by slabs / planes is reused, so add it back to the budget
        budget += (double)(planes[0].bytes + planes[1].bytes + planes[2].bytes + planes[3].bytes);
        budget += (double)(planesB[0].bytes + planesB[1].bytes + planesB[2].bytes + planesB[3].bytes + planesBalt[0].bytes + planesBalt[1].bytes);
        budget += (double)slabAll.size() * (double)slabFrames * 4.0;
        budget += (double)(coarseX.bytes + coarseY.bytes + coarseM.bytes);
        int64_t byMem = (int64_t)(std::max(budget - fixedBytes, 0.0) / std::max(perBlock, 1.0));
        limit = std::max<int64_t>(1, std::min<int64_t>(limit, byMem));
      }
    }
  return limit;
}

// AudioContextBase.ProcessBlocks (AudioContextBase.cs:163-186) / ProcessBlockInterleaved (:88-157) over whole chunks
void Context::processBlocks(float* const* outPlanar, float* outInterleaved, int channels, int64_t blockCount, bool deviceOut) {
  if (blockCount < 0) fail(GA_ERR_OUT_OF_RANGE, "blockCount");
  if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
  flushHandOver();
  if (outInterleaved || outPlanar == nullptr) {
    if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "channels");
    if (!outInterleaved) fail(GA_ERR_INVALID_ARGUMENT, "Buffer too small for interleaved output.");
  } else if (channels < 0 || channels > 32) {
    fail(GA_ERR_OUT_OF_RANGE, "channels");
  }
  GA_HIP(hipSetDevice(device));
  const hipMemcpyKind kind = deviceOut ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  int64_t doneBlocks = 0;
  while (doneBlocks < blockCount) {
    int64_t nblk = std::min(blockCount - doneBlocks, chunkLimit(blockCount - doneBlocks));
    runChunk(nblk, nullptr);
    const int64_t done = chunkBlocksDone;
    if (done <= 0) fail(GA_ERR_INVALID_OPERATION, "render made no progress");
    const int64_t base = doneBlocks * kBlock;
    if (outInterleaved) {
      float* stage = outInterleaved + base * channels;
      if (!deviceOut) {
        size_t need = (size_t)done * kBlock * channels * sizeof(float);
        if (ilvBytes < need) {
          if (ilvDev) {
            GA_HIP(hipStreamSynchronize(stream));
            dfree(ilvDev, ilvBytes);
          }
          ilvBytes = need + need / 4;
          ilvDev = (float*)dalloc(ilvBytes);
        }
        stage = ilvDev;
      }
      for (const SegCh& sg : chunkSegCh) {
        InterleaveSrc src{};
        const int used = std::min(channels, sg.ch);
        for (int ch = 0; ch < used; ch++) src.ch[ch] = busSlabs[ch];
        launch_interleave(stream, stage, src, channels, used, sg.b0 * kBlock, (sg.b1 - sg.b0) * kBlock);
      }
      if (!deviceOut)
        GA_HIP(hipMemcpyAsync(outInterleaved + base * channels, ilvDev, (size_t)done * kBlock * channels * sizeof(float), kind, stream));
    } else {
      for (const SegCh& sg : chunkSegCh) {   // `channels = Math.Min(outputBuffers.Length, buffer.ChannelCount)` per block (:173)
        const int used = std::min(channels, sg.ch);
        for (int ch = 0; ch < used; ch++)
          if (outPlanar[ch])
            GA_HIP(hipMemcpyAsync(outPlanar[ch] + base + sg.b0 * kBlock, busSlabs[ch] + sg.b0 * kBlock,
                                  sizeof(float) * (size_t)(sg.b1 - sg.b0) * kBlock, kind, stream));
      }
    }
    GA_HIP(hipStreamSynchronize(stream));
    doneBlocks += done;
  }
  stats.device_bytes_in_use = devBytes;
}

void Context::render(float* const* out, int channels, int64_t frameCount, int64_t startIndex, bool deviceOut) {
  if (channels == 0 || !out) fail(GA_ERR_INVALID_ARGUMENT, "Output buffer must have at least one channel.");
  if (frameCount <= 0) fail(GA_ERR_OUT_OF_RANGE, "Frame count must be positive.");
  if (startIndex < 0) fail(GA_ERR_OUT_OF_RANGE, "Start index must be non-negative.");
  if (channels < 0 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "channelIndex");
  for (int ch = 0; ch < channels; ch++)
    if (!out[ch]) fail(GA_ERR_INVALID_ARGUMENT, "Channel buffer is null.");
  if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
  GA_HIP(hipSetDevice(device));
  const hipMemcpyKind kind = deviceOut ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  // A render that spans several chunks is pipelined inside the call even when the context is synchronous: chunk k + 1 is
  // simulated and planned while chunk k executes, and the call waits once, at its end.
  struct Pipelined {
    Context& c;
    const bool callerAsync;
    explicit Pipelined(Context& cc) : c(cc), callerAsync(cc.asyncMode) { c.asyncMode = true; }
    ~Pipelined() { c.asyncMode = callerAsync; }
  } pipelined(*this);
  struct Drain {   // a blocking render never leaves work in flight, not even when it throws
    Context& c;
    const bool blocking;
    ~Drain() {
      if (blocking && c.stream) (void)hipStreamSynchronize(c.stream);
      if (blocking && c.copyStream) (void)hipStreamSynchronize(c.copyStream);
    }
  } drain{*this, !pipelined.callerAsync};
  int64_t written = 0;
  if (cachedFrames > 0) {  // leftover frames of the previous call's last block (:55-75)
    if (channels > cachedCh) fail(GA_ERR_OUT_OF_RANGE, "channelIndex");
    int toCopy = (int)std::min<int64_t>(cachedFrames, frameCount);
    for (int ch = 0; ch < channels; ch++)
      GA_HIP(hipMemcpyAsync(out[ch] + startIndex, cacheDev + (size_t)ch * kBlock + (kBlock - cachedFrames), sizeof(float) * toCopy,
                            kind, stream));
    if (!asyncMode) GA_HIP(hipStreamSynchronize(stream));
    written = toCopy;
    cachedFrames -= toCopy;
  }
  while (written < frameCount) {
    int64_t need = frameCount - written;
    int64_t nblk = (need + kBlock - 1) / kBlock;
    const int64_t limit = chunkLimit(nblk);
    nblk = std::min(nblk, limit);
    // device output, whole blocks, 16-byte aligned rows: the destination mixes straight into the caller's memory
    // ... and so does page-locked host memory the device can address (hipHostMalloc / hipHostRegister: what an asynchronous
    // render needs anyway): the last kernel of the chunk writes the bus over PCIe itself instead of two copy kernels behind it
    bool direct = (deviceOut || hostDirect) && need >= nblk * kBlock;
    float* tgt[32];
    for (int ch = 0; ch < channels && direct; ch++) {
      float* p = out[ch] + startIndex + written;
      if (!deviceOut) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost || !at.devicePointer) {
          (void)hipGetLastError();   // pageable memory: copies
          direct = false;
          break;
        }
        p = (float*)at.devicePointer;
      }
      tgt[ch] = p;
      direct = (((uintptr_t)p) & 15) == 0;
    }
    struct Target {   // (cleared on every way out of the chunk)
      Context& c;
      ~Target() { std::memset(c.busTarget, 0, sizeof(c.busTarget)); }
    } target{*this};
    // an asynchronous render into page-locked rows: the bus stays on the device and crosses PCIe under the next chunk's kernels
    // (only on the context's OWN stream: a caller who supplied the stream (ga_context_set_stream) may wait on that stream or on an
    // event of their own instead of calling ga_synchronize, and must then find the rows complete -- graphaudio_hip.h, "Pipelined rendering")
    const bool defer = direct && !deviceOut && pipelined.callerAsync && hostDefer && ownStream;
    if (defer) {
      const size_t rowBytes = ((size_t)nblk * kBlock * sizeof(float) + 255) & ~(size_t)255;
      if (deferStageBytes < rowBytes * 32) {
        flushHandOver();
        if (deferStage) {
          GA_HIP(hipStreamSynchronize(stream));
          dfree(deferStage, deferStageBytes);
        }
        deferStageBytes = rowBytes * 32;
        deferStage = (float*)dalloc(deferStageBytes);
      }
      for (int ch = 0; ch < channels; ch++) busTarget[ch] = (float*)((char*)deferStage + (size_t)ch * (deferStageBytes / 32));
    } else if (direct) {
      for (int ch = 0; ch < channels; ch++) busTarget[ch] = tgt[ch];
    }
    runChunk(nblk, nullptr);   // (takes a pending hand-over with it: in its pre-mix launch, or as copies in front of everything else)
    if (defer)
      for (int ch = 0; ch < channels; ch++)
        pendingHandOver.push_back(HandOver{busTarget[ch], tgt[ch], out[ch] + startIndex + written, chunkBlocksDone * kBlock});
    std::memset(busTarget, 0, sizeof(busTarget));
    const int64_t done = chunkBlocksDone;
    if (done <= 0) fail(GA_ERR_INVALID_OPERATION, "render made no progress");
    // buffer.GetChannelSpan(ch) throws for ch >= destination channel count (:82-85)
    if (channels > chunkMinDestCh) fail(GA_ERR_OUT_OF_RANGE, "channelIndex");
    const int64_t chunkFrames = done * kBlock;
    const int64_t toCopy = std::min(chunkFrames, need);
    if (direct) {
      // nothing to copy: the bus of these `done` blocks is where the caller wants it
    } else if (!deviceOut && ownStream && pipelined.callerAsync && hostCopyStream) {
      handOverToHost(busSlabs.data(), out, channels, startIndex + written, toCopy);
    } else {
      for (int ch = 0; ch < channels; ch++)
        GA_HIP(hipMemcpyAsync(out[ch] + startIndex + written, busSlabs[ch], sizeof(float) * toCopy, kind, stream));
    }
    const int64_t excess = chunkFrames - toCopy;  // < 128: only the last block can be partial (:89-100)
    if (excess > 0) {
      cachedCh = chunkMinDestCh;
      for (int ch = 0; ch < cachedCh; ch++)
        GA_HIP(hipMemcpyAsync(cacheDev + (size_t)ch * kBlock, busSlabs[ch] + chunkFrames - kBlock, sizeof(float) * kBlock,
                              hipMemcpyDeviceToDevice, stream));
      cachedFrames = (int)excess;
    }
    if (!asyncMode) GA_HIP(hipStreamSynchronize(stream));
    written += toCopy;
  }
  if (!pipelined.callerAsync) {
    GA_HIP(hipStreamSynchronize(stream));
    harvestProfile(true);
  }
  stats.device_bytes_in_use = devBytes;
}

extern "C" {

const char* ga_strerror(int code) {
  switch (code) {
    case GA_OK: return "ok";
    case GA_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GA_ERR_OUT_OF_RANGE: return "argument out of range";
    case GA_ERR_INVALID_OPERATION: return "invalid operation";
    case GA_ERR_DISPOSED: return "object disposed";
    case GA_ERR_CYCLE: return "audio graph cycle detected";
    case GA_ERR_UNSUPPORTED: return "unsupported on the device path";
    case GA_ERR_DEVICE: return "HIP device error";
    case GA_ERR_OUT_OF_MEMORY: return "out of memory";
    case GA_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown error code";
  }
}
const char* ga_version(void) { return "graphaudio-hip 0.1 (gfx950)"; }
int ga_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ga_context_create(int sample_rate, int device_ordinal, ga_context** out) {
  if (!out) return GA_ERR_INVALID_ARGUMENT;
  if (sample_rate <= 0) return GA_ERR_OUT_OF_RANGE;  // AudioContextBase.cs:37-38
  ga_context* h = nullptr;
  try {
    h = new ga_context(sample_rate);
    h->c.init_device(device_ordinal);
    auto d = std::make_unique<NodeS>();  // AudioDestinationNode: 1 input with channelCount 2, no outputs (:17-21)
    d->id = 0;
    d->type = GA_NODE_DESTINATION;
    d->inputs.resize(1);
    d->inputs[0].channelCount = 2;
    h->c.nodes.push_back(std::move(d));
    *out = h;
    return GA_OK;
  } catch (const Err& e) {
    delete h;
    return e.code;
  } catch (...) {
    delete h;
    return GA_ERR_DEVICE;
  }
}
int ga_context_destroy(ga_context* ctx) {
  delete ctx;
  return GA_OK;
}
const char* ga_last_error(ga_context* ctx) { return ctx ? ctx->c.lastError.c_str() : ""; }
double ga_current_time(ga_context* ctx) { return ctx ? ctx->c.currentTime : 0.0; }
int64_t ga_current_block(ga_context* ctx) { return ctx ? ctx->c.currentBlock : 0; }
int ga_set_option(ga_context* ctx, const char* key, double value) {
  return guard(ctx, [&](Context& c) {
    std::string k = key ? key : "";
    if (k == "max_chunk_blocks") c.maxChunkBlocks = std::max<int64_t>(1, std::min<int64_t>(32768, (int64_t)value));
    else if (k == "profile") c.profile = value != 0;
    else if (k == "profile_every") c.profileEvery = (int)std::max<int64_t>(1, value);
    else if (k == "time_fft") c.useTimeFft = value != 0;
    else if (k == "fft64") c.fft64 = value != 0;
    else if (k == "tconv_radix16") c.useRadix16 = value != 0;
    else if (k == "async") {
      if (c.asyncMode && value == 0) c.synchronize();
      c.asyncMode = value != 0;
    }
    else if (k == "coarse") c.useCoarse = value != 0;
    else if (k == "coarse_overlap") c.coarseOverlap = value != 0;
    else if (k == "coarse_carry") c.coarseCarry = value != 0;
    else if (k == "coarse_tail") c.coarseTail = value != 0;
    else if (k == "coarse_tail_private") c.coarseTailPrivate = value != 0;
    else if (k == "sim_replay") c.simReplay = value != 0;
    else if (k == "twin_channels") c.twinChannels = value != 0;
    else if (k == "cycle_delay_split") c.cycleDelaySplit = value != 0;
    else if (k == "resample_fast") c.resampleFast = value != 0;
    else if (k == "conv_reference_order") c.convRefOrder = (int)std::min(2.0, std::max(0.0, value));
    else if (k == "conv_ref_min_deviation") c.convRefMinDeviation = std::max(0.0, value);
    else if (k == "coarse_premix") c.coarsePremix = value != 0;
    else if (k == "coarse_ext_history") c.coarseExtHist = value != 0;
    else if (k == "coarse_wide") c.coarseWide = value != 0;
    else if (k == "gain_pass_through") c.gainPassThrough = value != 0;
    else if (k == "gain_fold") c.gainFold = value != 0;
    else if (k == "coarse_mfma") c.coarseMfma = value != 0;
    else if (k == "biquad_time_split") c.biquadTimeSplit = (int)std::min(2.0, std::max(0.0, value));
    else if (k == "biquad_split_max_deviation") c.biquadSplitMaxDeviation = std::max(0.0, value);
    else if (k == "biquad_split_min_frames") c.biquadSplitMinFrames = std::max<int64_t>(2048, (int64_t)value);
    else if (k == "host_direct") c.hostDirect = value != 0;
    else if (k == "host_defer") c.hostDefer = value != 0;
    else if (k == "table_upload_kernel") c.tableUploadKernel = value != 0;
    else if (k == "comm_timeout_s") c.commTimeoutS = std::max(1.0, value);
    else if (k == "host_copy_stream") c.hostCopyStream = value != 0;
    else if (k == "coarse_min_blocks") c.coarseMinBlocks = std::max<int64_t>(1, (int64_t)value);
    else if (k == "debug_tconv_n2") c.debugTconvN2 = (int)value;   // tests only: plan the block-axis FFT with this (possibly unsupported) length
    else if (k == "mem_budget_fraction") c.memBudgetFraction = std::min(0.95, std::max(0.05, value));
    else fail(GA_ERR_INVALID_ARGUMENT, "unknown option " + k);
  });
}
int ga_get_stats(ga_context* ctx, ga_stats* out) {
  return guardRO(ctx, [&](Context& c) {
    if (!out) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    c.harvestProfile(true);
    c.stats.device_bytes_in_use = c.devBytes;
    c.stats.n_nodes = (int)c.nodes.size();
    int rows = 0;
    for (auto& g : c.groups) rows += (int)g->rows.size();
    c.stats.n_conv_rows = rows;
    *out = c.stats;
  });
}
int ga_synchronize(ga_context* ctx) {
  return guardRO(ctx, [&](Context& c) { c.synchronize(); });
}
int ga_context_set_stream(ga_context* ctx, void* hip_stream) {
  return guard(ctx, [&](Context& c) {
    c.flushHandOver();
    GA_HIP(hipStreamSynchronize(c.stream));
    c.waitHostCopies();
    if (c.ownStream && c.stream) (void)hipStreamDestroy(c.stream);
    c.stream = (hipStream_t)hip_stream;
    c.ownStream = false;
  });
}

int ga_buffer_create(ga_context* ctx, const float* const* planar, int channels, int64_t frames, int sample_rate, int* out_id) {
  return guard(ctx, [&](Context& c) {
    if (!planar || !out_id) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "Channel count must be between 1 and 32");
    if (frames < 0) fail(GA_ERR_OUT_OF_RANGE, "Length must be non-negative");
    if (sample_rate <= 0) fail(GA_ERR_OUT_OF_RANGE, "Sample rate must be positive");
    auto b = std::make_unique<PlayBuf>();
    b->channels = channels;
    b->length = frames;
    b->sampleRate = sample_rate;
    b->stride = (frames + 8 + 63) / 64 * 64;  // a few floats of slack: the resampler window looks back 4 samples
    b->host.resize(channels);
    GA_HIP(hipSetDevice(c.device));
    size_t bytes = (size_t)b->stride * channels * sizeof(float);
    b->dev = c.dallocSkewed(bytes, &b->devBase, &b->devBytes);
    GA_HIP(hipMemsetAsync(b->dev, 0, bytes, c.stream));
    for (int i = 0; i < channels; i++) {
      if (!planar[i]) fail(GA_ERR_INVALID_ARGUMENT, "null channel pointer");
      b->host[i].assign(planar[i], planar[i] + frames);
      if (frames)
        GA_HIP(hipMemcpyAsync(b->dev + (size_t)i * b->stride, planar[i], sizeof(float) * frames, hipMemcpyHostToDevice, c.stream));
    }
    GA_HIP(hipStreamSynchronize(c.stream));
    c.buffers.push_back(std::move(b));
    c.bufVersion++;
    *out_id = (int)c.buffers.size() - 1;
  });
}
int ga_buffer_release(ga_context* ctx, int buffer_id) {
  return guard(ctx, [&](Context& c) {   // the host dropped its handle: the storage goes once no node plays / convolves with it
    PlayBuf* b = c.buffer(buffer_id);
    if (!b || b->released) return;
    b->released = true;
    c.releasedPending.push_back(buffer_id);
  });
}

int ga_node_create(ga_context* ctx, int node_type, int* out_id) {  // constructor defaults: 2 outputs / 2 inputs / 1.0 s
  return ga_node_create_ex(ctx, node_type, node_type == GA_NODE_DELAY ? 1.0 : 2.0, out_id);
}
int ga_node_create_ex(ga_context* ctx, int node_type, double arg, int* out_id) {
  return guard(ctx, [&](Context& c) {
    if (!out_id) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    auto n = std::make_unique<NodeS>();
    n->id = (int)c.nodes.size();
    n->type = node_type;
    const float FMAX = std::numeric_limits<float>::max();
    switch (node_type) {
      case GA_NODE_BUFFER_SOURCE:
        n->outputs.resize(1);
        n->params.push_back(makeParam(1.f, 0.001f, 1000.f, false));  // AudioBufferSourceNode.cs:76
        break;
      case GA_NODE_GAIN:
        n->inputs.resize(1);
        n->outputs.resize(1);
        n->params.push_back(makeParam(1.f, -FMAX, FMAX, true));  // GainNode.cs:21-26
        break;
      case GA_NODE_BIQUAD:
        n->inputs.resize(1);
        n->outputs.resize(1);
        n->params.push_back(makeParam(1000.f, 1.f, c.sampleRate / 2.f, true));  // BiQuadFilterNode.cs:63-82
        n->params.push_back(makeParam(1.f, 0.001f, 1000.f, true));
        n->params.push_back(makeParam(0.f, -60.f, 60.f, false));
        c.updateBiquadCoefficients(*n, 1000.f, 1.0f, 0.f);  // constructor call (:84)
        n->coefDirty = true;
        break;
      case GA_NODE_CONVOLVER:
        n->inputs.resize(1);
        n->outputs.resize(1);
        break;
      case GA_NODE_CHANNEL_SPLITTER:   // ChannelSplitterNode.cs:14-22
      case GA_NODE_CHANNEL_MERGER: {   // ChannelMergerNode.cs:14-21
        int cnt = (int)arg;
        if (cnt < 1 || cnt > 32) fail(GA_ERR_OUT_OF_RANGE, node_type == GA_NODE_CHANNEL_SPLITTER ? "numberOfOutputs" : "numberOfInputs");
        n->inputs.resize(node_type == GA_NODE_CHANNEL_SPLITTER ? 1 : cnt);
        n->outputs.resize(node_type == GA_NODE_CHANNEL_SPLITTER ? cnt : 1);
        break;
      }
      case GA_NODE_CONSTANT_SOURCE:    // ConstantSourceNode.cs:29-38
        n->outputs.resize(1);
        n->params.push_back(makeParam(1.f, -FMAX, FMAX, true));
        break;
      case GA_NODE_STEREO_PANNER:      // StereoPannerNode.cs:21-34
        n->inputs.resize(1);
        n->outputs.resize(1);
        n->inputs[0].channelCount = 2;
        n->inputs[0].mode = GA_COUNT_MODE_CLAMPED_MAX;
        n->inputs[0].interp = GA_INTERP_SPEAKERS;
        n->params.push_back(makeParam(0.f, -1.f, 1.f, true));
        break;
      case GA_NODE_OSCILLATOR:         // OscillatorNode.cs:43-52
        n->outputs.resize(1);
        n->params.push_back(makeParam(440.f, 0.f, c.sampleRate / 2.f, true));
        break;
      case GA_NODE_DELAY: {            // DelayNode.cs:22-41
        if (!(arg > 0) || arg > 10) fail(GA_ERR_OUT_OF_RANGE, "maxDelayTime");
        n->maxDelaySamples = (int)(arg * c.sampleRate);
        if (n->maxDelaySamples < 1) fail(GA_ERR_OUT_OF_RANGE, "maxDelayTime");
        n->delayRings = 2;
        n->inputs.resize(1);
        n->outputs.resize(1);
        n->params.push_back(makeParam(0.f, 0.f, (float)arg, true));
        break;
      }
      case GA_NODE_STREAM_SOURCE:      // AudioStreamSourceNodeBase.cs:59-67
        n->outputs.resize(1);
        n->params.push_back(makeParam(1.f, 0.001f, 1000.f, false));
        break;
      default: fail(GA_ERR_INVALID_ARGUMENT, "unknown node type");
    }
    *out_id = n->id;
    c.nodes.push_back(std::move(n));
  });
}
int ga_node_dispose(ga_context* ctx, int node) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = c.node(node);
    if (n->disposed) return;
    Context* cp = &c;
    c.executeOrPost([cp, node]() { cp->doDispose(node); });
  });
}
int ga_node_connect(ga_context* ctx, int src, int dst, int output_index, int input_index) {
  return guard(ctx, [&](Context& c) {
    c.node(src);
    c.node(dst);
    Context* cp = &c;
    c.executeOrPost([cp, src, dst, output_index, input_index]() {  // DoConnect, Nodes/AudioNode.cs:111-120
      NodeS* s = cp->nodes[src].get();
      NodeS* d = cp->nodes[dst].get();
      if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
      if (input_index < 0 || input_index >= (int)d->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
      cp->connectTo(src, output_index, InRef{dst, input_index});
    });
  });
}
int ga_node_disconnect(ga_context* ctx, int src, int dst, int output_index, int input_index) {
  return guard(ctx, [&](Context& c) {
    c.node(src);
    if (dst >= 0) c.node(dst);
    Context* cp = &c;
    c.executeOrPost([cp, src, dst, output_index, input_index]() {  // DoDisconnect, Nodes/AudioNode.cs:131-147
      NodeS* s = cp->nodes[src].get();
      if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
      if (dst < 0) {
        cp->outputDisconnectAll(src, output_index);
      } else {
        NodeS* d = cp->nodes[dst].get();
        if (input_index < 0 || input_index >= (int)d->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
        cp->disconnectFrom(src, output_index, InRef{dst, input_index});
      }
    });
  });
}
int ga_node_connect_param(ga_context* ctx, int src, int dst_node, int dst_param, int output_index) {
  return guard(ctx, [&](Context& c) {
    NodeS* s = c.node(src);
    c.param(dst_node, dst_param);
    if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
    Context* cp = &c;
    c.executeOrPost([cp, src, dst_node, dst_param, output_index]() { cp->connectTo(src, output_index, InRef{dst_node, -1 - dst_param}); });
  });
}
int ga_node_disconnect_param(ga_context* ctx, int src, int dst_node, int dst_param, int output_index) {
  return guard(ctx, [&](Context& c) {
    NodeS* s = c.node(src);
    c.param(dst_node, dst_param);
    if (output_index < 0 || output_index >= (int)s->outputs.size()) fail(GA_ERR_OUT_OF_RANGE, "outputIndex");
    Context* cp = &c;
    c.executeOrPost([cp, src, dst_node, dst_param, output_index]() { cp->disconnectFrom(src, output_index, InRef{dst_node, -1 - dst_param}); });
  });
}
int ga_node_has_ended(ga_context* ctx, int node) {
  int r = 0;
  int rc = guardRO(ctx, [&](Context& c) {
    NodeS* n = c.node(node);
    r = n->endedRaised ? 1 : 0;
  });
  return rc < 0 ? rc : r;
}

int ga_poll_ended(ga_context* ctx, int* out_ids, int capacity) {
  int n = 0;
  int rc = guardRO(ctx, [&](Context& c) {
    if (!out_ids || capacity < 0) fail(GA_ERR_INVALID_ARGUMENT, "bad buffer");
    while (n < capacity && !c.endedQueue.empty()) {
      out_ids[n++] = c.endedQueue.front();
      c.endedQueue.pop_front();
    }
  });
  return rc < 0 ? rc : n;
}

int ga_input_set_channel_count(ga_context* ctx, int node, int input_index, int count) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = c.node(node);
    if (input_index < 0 || input_index >= (int)n->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
    if (count < 1 || count > 32) fail(GA_ERR_OUT_OF_RANGE, "Channel count must be between 1 and 32");  // AudioNodeInput.cs:43-44
    n->inputs[input_index].channelCount = count;
    n->inputs[input_index].dirty = true;
  });
}
int ga_input_set_channel_count_mode(ga_context* ctx, int node, int input_index, int mode) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = c.node(node);
    if (input_index < 0 || input_index >= (int)n->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
    if (mode < 0 || mode > 2) fail(GA_ERR_INVALID_ARGUMENT, "mode");
    n->inputs[input_index].mode = mode;
  });
}
int ga_input_set_channel_interpretation(ga_context* ctx, int node, int input_index, int interp) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = c.node(node);
    if (input_index < 0 || input_index >= (int)n->inputs.size()) fail(GA_ERR_OUT_OF_RANGE, "inputIndex");
    n->inputs[input_index].interp = interp;  // stored; MixBuffer ignores it (AudioNodeInput.cs:182-244)
  });
}
int ga_destination_set_channel_count(ga_context* ctx, int channels) {
  return guard(ctx, [&](Context& c) {
    if (channels < 1 || channels > 32) fail(GA_ERR_OUT_OF_RANGE, "channels");  // AudioDestinationNode.cs:25-26
    Context* cp = &c;
    c.executeOrPost([cp, channels]() {
      cp->nodes[0]->inputs[0].channelCount = channels;
      cp->nodes[0]->inputs[0].dirty = true;
    });
  });
}
int ga_destination_output_channels(ga_context* ctx) {
  if (!ctx) return GA_ERR_INVALID_ARGUMENT;
  return ctx->c.destOutCh > 0 ? ctx->c.destOutCh : 2;  // OfflineAudioContext.cs:113-114
}

int ga_param_set_value(ga_context* ctx, int node, int param, float value) {
  return guard(ctx, [&](Context& c) {  // Value setter: clamp + cancel all events (AudioParam.cs:37-48)
    ParamS* p = c.param(node, param);
    p->value = clampf(value, p->minv, p->maxv);
    p->events.clear();
  });
}
int ga_param_get_value(ga_context* ctx, int node, int param, float* out) {
  return guardRO(ctx, [&](Context& c) {
    if (!out) fail(GA_ERR_INVALID_ARGUMENT, "null pointer");
    *out = c.param(node, param)->value;
  });
}
int ga_param_set_value_at_time(ga_context* ctx, int node, int param, float value, double t) {
  return guard(ctx, [&](Context& c) {  // AudioParam.cs:252-261
    ParamS* p = c.param(node, param);
    ParamEvent e{};
    e.type = 0;
    e.value = clampf(value, p->minv, p->maxv);
    e.time = t;
    addEvent(*p, e);
  });
}
int ga_param_linear_ramp_to_value_at_time(ga_context* ctx, int node, int param, float value, double t) {
  return guard(ctx, [&](Context& c) {  // AudioParam.cs:266-275
    ParamS* p = c.param(node, param);
    ParamEvent e{};
    e.type = 1;
    e.value = clampf(value, p->minv, p->maxv);
    e.time = t;
    addEvent(*p, e);
  });
}
int ga_param_exponential_ramp_to_value_at_time(ga_context* ctx, int node, int param, float value, double t) {
  return guard(ctx, [&](Context& c) {  // AudioParam.cs:280-292
    ParamS* p = c.param(node, param);
    float v = clampf(value, p->minv, p->maxv);
    if (v <= 0.f) fail(GA_ERR_INVALID_ARGUMENT, "Exponential ramp target must be > 0");
    ParamEvent e{};
    e.type = 2;
    e.value = v;
    e.time = t;
    addEvent(*p, e);
  });
}
int ga_param_set_target_at_time(ga_context* ctx, int node, int param, float target, double t, double tc) {
  return guard(ctx, [&](Context& c) {  // AudioParam.cs:297-307
    ParamS* p = c.param(node, param);
    ParamEvent e{};
    e.type = 3;
    e.target = clampf(target, p->minv, p->maxv);
    e.time = t;
    e.time_constant = tc;
    addEvent(*p, e);
  });
}
int ga_param_cancel_scheduled_values(ga_context* ctx, int node, int param, double cancel_time) {
  return guard(ctx, [&](Context& c) {  // AudioParam.cs:312-331
    ParamS* p = c.param(node, param);
    size_t survivors = 0;
    for (size_t i = 0; i < p->events.size(); i++) {
      if (p->events[i].time < cancel_time) survivors++; else break;
    }
    p->events.resize(survivors);
  });
}

int ga_source_set_buffer(ga_context* ctx, int node, int buffer_id) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = typed(c, node, GA_NODE_BUFFER_SOURCE);
    if (buffer_id >= 0) c.buffer(buffer_id);
    n->bufId = buffer_id < 0 ? -1 : buffer_id;
  });
}
int ga_source_set_loop(ga_context* ctx, int node, int loop, double loop_start, double loop_end) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = typed(c, node, GA_NODE_BUFFER_SOURCE);
    n->loop = loop != 0;
    n->loopStart = std::max(0.0, loop_start);
    n->loopEnd = std::max(0.0, loop_end);
  });
}
int ga_source_start(ga_context* ctx, int node, double when, double offset, double duration) {
  return guard(ctx, [&](Context& c) {
    Context* cp = &c;
    const int ty = c.node(node)->type;
    if (ty == GA_NODE_CONSTANT_SOURCE || ty == GA_NODE_OSCILLATOR) {
      c.executeOrPost([cp, node, when, duration, ty]() {  // ConstantSourceNode.cs:44-63, OscillatorNode.cs:54-73
        NodeS& s = *cp->nodes[node];
        if (s.hasStarted) {
          if (ty == GA_NODE_OSCILLATOR) fail(GA_ERR_INVALID_OPERATION, "OscillatorNode can only be started once.");
          return;   // a second ConstantSourceNode.Start is ignored
        }
        s.hasStarted = true;
        s.oscPhaseReset = true;
        s.startTime = std::max(0.0, when);
        if (!std::isnan(duration) && duration >= 0) {
          s.stopTime = s.startTime + duration;
          s.hasStopped = true;
        }
      });
      return;
    }
    typed(c, node, GA_NODE_BUFFER_SOURCE);
    c.executeOrPost([cp, node, when, offset, duration]() {  // AudioBufferSourceNode.cs:81-111
      NodeS& s = *cp->nodes[node];
      if (s.hasStarted) fail(GA_ERR_INVALID_OPERATION, "AudioBufferSourceNode can only be started once.");
      if (s.bufId < 0) fail(GA_ERR_INVALID_OPERATION, "Cannot start without a buffer set");
      s.hasStarted = true;
      s.startTime = std::max(0.0, when);
      s.offset = std::max(0.0, offset);
      s.duration = duration;
      s.playbackPosition = (int64_t)(s.offset * cp->buffers[s.bufId]->sampleRate);
      s.rsBlocks = 0;
      if (!(std::isinf(duration) && duration > 0) && duration >= 0) {
        s.stopTime = s.startTime + duration;
        s.hasStopped = true;
      }
    });
  });
}
int ga_source_stop(ga_context* ctx, int node, double when) {
  return guard(ctx, [&](Context& c) {
    const int ty = c.node(node)->type;
    if (ty != GA_NODE_CONSTANT_SOURCE && ty != GA_NODE_OSCILLATOR) typed(c, node, GA_NODE_BUFFER_SOURCE);
    Context* cp = &c;
    c.executeOrPost([cp, node, when]() {  // AudioBufferSourceNode.cs:118-126
      NodeS& s = *cp->nodes[node];
      if (s.hasStopped) return;
      double at = std::max(0.0, when);
      s.stopTime = std::isnan(s.stopTime) ? at : std::min(s.stopTime, at);
      s.hasStopped = true;
    });
  });
}
int ga_oscillator_set_type(ga_context* ctx, int node, int oscillator_type) {
  return guard(ctx, [&](Context& c) {  // OscillatorNode.cs:33-42
    typed(c, node, GA_NODE_OSCILLATOR);
    if (oscillator_type < 0 || oscillator_type > 3) fail(GA_ERR_INVALID_ARGUMENT, "oscillator type");
    Context* cp = &c;
    c.executeOrPost([cp, node, oscillator_type]() { cp->nodes[node]->oscType = oscillator_type; });
  });
}
int ga_biquad_set_type(ga_context* ctx, int node, int filter_type) {
  return guard(ctx, [&](Context& c) {
    typed(c, node, GA_NODE_BIQUAD);
    if (filter_type < 0 || filter_type > GA_FILTER_HIGHSHELF) fail(GA_ERR_INVALID_ARGUMENT, "filter type");
    Context* cp = &c;
    c.executeOrPost([cp, node, filter_type]() {  // BiQuadFilterNode.cs:24-36
      NodeS& b = *cp->nodes[node];
      if (b.filterType != filter_type) {
        b.filterType = filter_type;
        b.coefDirty = true;
      }
    });
  });
}
int ga_convolver_set_normalize(ga_context* ctx, int node, int normalize) {
  return guard(ctx, [&](Context& c) { typed(c, node, GA_NODE_CONVOLVER)->normalize = normalize != 0; });
}
int ga_convolver_set_enable_true_stereo(ga_context* ctx, int node, int enable) {
  return guard(ctx, [&](Context& c) { typed(c, node, GA_NODE_CONVOLVER)->enableTrueStereo = enable != 0; });
}
int ga_convolver_set_buffer(ga_context* ctx, int node, int buffer_id) {
  return guard(ctx, [&](Context& c) {  // ConvolverNode.Buffer setter, ConvolverNode.cs:25-79
    NodeS* n = typed(c, node, GA_NODE_CONVOLVER);
    Context* cp = &c;
    // `if (_buffer == value) return;` (:30) compares with the buffer of the last EXECUTED swap: a swap that is still
    // queued does not count, so A -> B -> A between two blocks ends on B
    if (n->irBuf == (buffer_id < 0 ? -1 : buffer_id)) {
      if (buffer_id >= 0) (void)c.buffer(buffer_id);
      return;
    }
    if (buffer_id < 0) {
      c.post([cp, node]() {
        NodeS& nd = *cp->nodes[node];
        cp->releaseConvState(nd);
        nd.irBuf = -1;
        nd.ir.reset();
        nd.effectiveOutCh = 0;
        nd.isTrueStereo = false;
        nd.inputs[0].mode = GA_COUNT_MODE_MAX;
      });
      return;
    }
    PlayBuf* b = c.buffer(buffer_id);
    if (b->sampleRate != c.sampleRate)
      fail(GA_ERR_INVALID_OPERATION, "Impulse response buffer sample rate must match the audio context sample rate.");
    // spectra are built eagerly with the CURRENT Normalize flag (:51-56); the swap is posted (:58-77)
    std::shared_ptr<IrSpectra> sp = c.irSpectra(buffer_id, n->normalize);
    c.post([cp, node, buffer_id, sp]() {
      NodeS& nd = *cp->nodes[node];
      // new PartitionedConvolver instances: fresh (zero) delay line and overlap; the formulation (shared-IR rows or
      // private state) is chosen at the next render, when it is known how many nodes share this impulse response
      cp->releaseConvState(nd);
      nd.irBuf = buffer_id;
      nd.ir = sp;
      cp->graphVersion++;
      int channels = sp->nch;
      nd.isTrueStereo = (channels == 4 && nd.enableTrueStereo);
      nd.effectiveOutCh = nd.isTrueStereo ? 2 : channels;
      nd.inputs[0].channelCount = nd.isTrueStereo ? 2 : channels;
      nd.inputs[0].dirty = true;
      nd.inputs[0].mode = GA_COUNT_MODE_EXPLICIT;
    });
  });
}

int ga_render(ga_context* ctx, float* const* out_planar, int out_channels, int64_t frame_count, int64_t start_index) {
  return guardRO(ctx, [&](Context& c) { c.render(out_planar, out_channels, frame_count, start_index, false); });
}
int ga_process_blocks(ga_context* ctx, float* const* out_planar, int out_channels, int64_t block_count, int out_on_device) {
  return guardRO(ctx, [&](Context& c) {
    if (out_channels > 0 && !out_planar) fail(GA_ERR_INVALID_ARGUMENT, "outputBuffers");
    static float* const none[1] = {nullptr};
    c.processBlocks(out_planar ? out_planar : none, nullptr, out_channels, block_count, out_on_device != 0);
  });
}
int ga_process_blocks_interleaved(ga_context* ctx, float* interleaved, int channels, int64_t block_count, int out_on_device) {
  return guardRO(ctx, [&](Context& c) { c.processBlocks(nullptr, interleaved, channels, block_count, out_on_device != 0); });
}
int ga_render_device(ga_context* ctx, float* const* out_planar_dev, int out_channels, int64_t frame_count, int64_t start_index) {
  return guardRO(ctx, [&](Context& c) { c.render(out_planar_dev, out_channels, frame_count, start_index, true); });
}

// ---- AudioStreamNodeBase (GraphAudio.IO/AudioStreamSourceNodeBase.cs) ----
int ga_stream_queue_buffer(ga_context* ctx, int node, int buffer_id) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = typed(c, node, GA_NODE_STREAM_SOURCE);
    PlayBuf* b = c.buffer(buffer_id);
    if (b->channels < 1 || b->length < 1) fail(GA_ERR_INVALID_ARGUMENT, "Buffer must be initialized");
    if (c.inRender) fail(GA_ERR_INVALID_OPERATION, "QueueBuffer during a render");
    n->stQueued.push_back(buffer_id);
  });
}
int ga_stream_set_state(ga_context* ctx, int node, int state) {
  return guard(ctx, [&](Context& c) {
    NodeS* n = typed(c, node, GA_NODE_STREAM_SOURCE);
    if (state < GA_STREAM_PLAYING || state > GA_STREAM_STOPPED) fail(GA_ERR_OUT_OF_RANGE, "stream state");
    const int old = n->stState;   // State setter (:37-49): immediate (Interlocked in the reference), not a posted command
    n->stState = state;
    if (state == GA_STREAM_STOPPED && old != GA_STREAM_STOPPED) c.streamFlushToProcessed(*n);
  });
}
int ga_stream_dequeue_processed(ga_context* ctx, int node, int* buffer_id_out) {
  int got = 0;
  int rc = guard(ctx, [&](Context& c) {
    NodeS* n = typed(c, node, GA_NODE_STREAM_SOURCE);
    if (n->stProcessed.empty()) return;
    if (buffer_id_out) *buffer_id_out = n->stProcessed.front();
    n->stProcessed.pop_front();
    got = 1;
  });
  return rc < 0 ? rc : got;
}
int ga_stream_queued_count(ga_context* ctx, int node) {
  int v = 0;
  int rc = guardRO(ctx, [&](Context& c) { v = (int)typed(c, node, GA_NODE_STREAM_SOURCE)->stQueued.size(); });
  return rc < 0 ? rc : v;
}
int ga_stream_processed_count(ga_context* ctx, int node) {
  int v = 0;
  int rc = guardRO(ctx, [&](Context& c) { v = (int)typed(c, node, GA_NODE_STREAM_SOURCE)->stProcessed.size(); });
  return rc < 0 ? rc : v;
}

// ---- sharded render (ga_comm.cpp) ----
int ga_comm_unique_id(void* id_out) {
  try {
    ga::commUniqueId(id_out);
    return GA_OK;
  } catch (const Err& e) {
    return e.code;
  } catch (...) {
    return GA_ERR_INVALID_OPERATION;
  }
}
int ga_comm_init(ga_context* ctx, const void* id, int n_ranks, int rank) {
  return guard(ctx, [&](Context& c) { c.commInit(id, n_ranks, rank); });
}
int ga_comm_destroy(ga_context* ctx) {
  return guard(ctx, [&](Context& c) { c.commDestroy(); });
}
int ga_comm_info(ga_context* ctx, int* n_ranks, int* rank, int* uses_rccl) {
  return guardRO(ctx, [&](Context& c) { c.commInfo(n_ranks, rank, uses_rccl); });
}
int ga_shard_range(int64_t n_voices, int n_ranks, int rank, int64_t* first, int64_t* count) {
  if (!first || !count) return GA_ERR_INVALID_ARGUMENT;
  if (n_voices < 0 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return GA_ERR_OUT_OF_RANGE;
  const int64_t base = n_voices / n_ranks, extra = n_voices % n_ranks;
  *first = rank * base + std::min<int64_t>(rank, extra);
  *count = base + (rank < extra ? 1 : 0);
  return GA_OK;
}
int ga_render_reduce(ga_context* ctx, float* const* out_planar, int out_channels, int64_t frame_count, int64_t start_index, int root) {
  return guardRO(ctx, [&](Context& c) { c.renderReduce(out_planar, out_channels, frame_count, start_index, root); });
}

}  // extern "C"

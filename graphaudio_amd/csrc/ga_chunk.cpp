// ga_chunk.cpp -- control-plane simulation and the per-chunk device executor (see ga_engine.hpp).
#include <time.h>

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "ga_engine.hpp"

namespace ga {

static inline uint64_t hmix(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  return h;
}
static inline int64_t roundup(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// ======================================================================================================
// job tables: built on the host while planning, uploaded once, then every recorded launch runs in order
// ======================================================================================================
struct Plan {
  std::vector<uint8_t> host;
  struct L {
    std::function<void(uint8_t*)> fn;
    int kind;
    double bytes;   // HBM bytes the launch has to move (inputs once + outputs once); 0 = not accounted
    double flops;   // floating-point operations it executes; 0 = not accounted
  };
  std::vector<L> launches;
  size_t put(const void* p, size_t bytes) {
    size_t off = (host.size() + 15) & ~(size_t)15;
    host.resize(off + bytes);
    if (bytes) std::memcpy(&host[off], p, bytes);
    return off;
  }
  template <class T>
  size_t putv(const std::vector<T>& v) {
    return put(v.data(), v.size() * sizeof(T));
  }
  void add(int kind, std::function<void(uint8_t*)> fn, double bytes = 0.0, double flops = 0.0) {
    launches.push_back(L{std::move(fn), kind, bytes, flops});
  }
};

// ======================================================================================================
// slabs: chunk-frame indexed float arrays handed to node outputs / mixed inputs for the duration of a chunk
// ======================================================================================================
static float* getSlab(Context& c) {
  if (c.slabFree.empty()) {
    size_t slabBytes = (size_t)c.slabFrames * sizeof(float);
    size_t count = std::max<size_t>(8, std::min<size_t>(1024, ((size_t)1 << 30) / slabBytes));
    char* blk = (char*)c.dalloc(slabBytes * count);
    c.slabBlocks.push_back(blk);
    for (size_t i = 0; i < count; i++) {
      float* p = (float*)(blk + i * slabBytes);
      c.slabAll.push_back(p);
      c.slabFree.push_back(p);
    }
  }
  float* p = c.slabFree.back();
  c.slabFree.pop_back();
  return p;
}
static void resetSlabs(Context& c, int64_t frames) {
  int64_t need = roundup(frames, 256);
  if (need > c.slabFrames) {
    GA_HIP(hipStreamSynchronize(c.stream));
    size_t oldBytes = (size_t)c.slabFrames * sizeof(float);
    size_t perBlock = oldBytes ? std::max<size_t>(8, std::min<size_t>(1024, ((size_t)1 << 30) / oldBytes)) : 0;
    for (void* p : c.slabBlocks) c.dfree(p, oldBytes * perBlock);
    c.slabBlocks.clear();
    c.slabAll.clear();
    c.slabFrames = need;
    c.slabGen++;
  }
  c.slabFree = c.slabAll;
}

// ======================================================================================================
// source scheduling (AudioBufferSourceNode.Process control flow, AudioBufferSourceNode.cs:131-389) as a per-chunk
// timeline of phases.  bt[i] = block start times, bt[i+1] = t1 of block i (AudioContextBase.cs:78-79).
// ======================================================================================================
struct SrcGeom {
  int64_t loopStartFrame, loopEndFrame, durationEndFrame;
  double effectiveRate;
};
static SrcGeom sourceGeom(Context& c, NodeS& s, PlayBuf& b) {
  SrcGeom g;
  float playbackRate = s.params[0].value;  // k-rate; a timeline on it is handled by the general replay (gsrReplayBlock)
  double sampleRateRatio = b.sampleRate / (double)c.sampleRate;
  g.effectiveRate = sampleRateRatio * playbackRate;
  g.loopStartFrame = (int64_t)(s.loopStart * b.sampleRate);
  g.loopEndFrame = s.loopEnd > 0 ? (int64_t)(s.loopEnd * b.sampleRate) : b.length;
  g.loopEndFrame = std::min(g.loopEndFrame, b.length);
  g.loopStartFrame = std::min(g.loopStartFrame, g.loopEndFrame);
  g.durationEndFrame = s.duration < std::numeric_limits<double>::infinity()
                           ? (int64_t)(s.offset * b.sampleRate) + (int64_t)(s.duration * b.sampleRate)
                           : b.length;
  g.durationEndFrame = std::min(g.durationEndFrame, b.length);
  return g;
}

static Resampler& resamplerFor(Context& c, double rate) {
  uint64_t key;
  std::memcpy(&key, &rate, 8);
  auto it = c.resamplers.find(key);
  if (it == c.resamplers.end()) {
    auto r = std::make_unique<Resampler>();
    r->rate = rate;
    it = c.resamplers.emplace(key, std::move(r)).first;
  }
  return *it->second;
}

// bounded replay of ONE block of CubicResampler.Process (CubicResampler.cs:26-63) from a trajectory state
static void resampleBlockBounded(const ResampleBlock& st, double rate, int64_t avail, int& produced, int64_t& consumedAfter,
                                 double* posAfter = nullptr, int* readyAfter = nullptr) {
  int64_t in = st.consumed;
  double Pos = st.pos;
  int ready = st.ready;
  while (ready < 4 && in < avail) {
    in++;
    ready++;
  }
  produced = 0;
  if (ready == 4) {
    while (produced < kBlock) {
      int consume = (int)Pos;
      if (in + consume > avail) break;
      in += consume;
      Pos -= consume;
      produced++;
      Pos += rate;
    }
  }
  consumedAfter = in;
  if (posAfter) *posAfter = Pos;
  if (readyAfter) *readyAfter = ready;
}

// ---- general source replay: AudioBufferSourceNode.Process for ONE block on indices only (see GsrBlock) ----
struct GsrState {
  int64_t w[4];
  double pos;
  int ready;
  int64_t pp;
};
static inline void gsrFeed(GsrState& st, int64_t idx) {  // CubicResampler.Shift, :91-97
  st.w[0] = st.w[1];
  st.w[1] = st.w[2];
  st.w[2] = st.w[3];
  st.w[3] = idx;
}
// CubicResampler.Process (:26-63) on an index stream at(k), k < inLen
template <class At>
static void gsrProcess(GsrState& st, At at, int inLen, int outLen, double rate, int& consumed, int& produced) {
  int inPos = 0, outPos = 0;
  while (st.ready < 4 && inPos < inLen) {
    gsrFeed(st, at(inPos++));
    st.ready++;
  }
  if (st.ready < 4) {
    consumed = inPos;
    produced = 0;
    return;
  }
  while (outPos < outLen) {
    int consume = (int)st.pos;
    if (inPos + consume > inLen) break;
    for (int i = 0; i < consume; i++) gsrFeed(st, at(inPos++));
    st.pos -= consume;
    outPos++;
    st.pos += rate;
  }
  consumed = inPos;
  produced = outPos;
}
// returns true when the block is an END block (`!hasMoreData || (!_loop && _playbackPosition >= durationEndFrame)`, :360)
static bool gsrReplayBlock(NodeS& s, const SrcGeom& g, PlayBuf& b, Context& c, float playbackRate, GsrState& st, GsrBlock& d) {
  const double effectiveRate = (b.sampleRate / (double)c.sampleRate) * playbackRate;
  const int64_t loopStart = g.loopStartFrame, loopEnd = g.loopEndFrame, durEnd = g.durationEndFrame, len = b.length;
  const bool loop = s.loop;
  bool hasMore = false;
  int64_t first = -1;
  int outIdx = 0;
  d.pp = st.pp;
  d.rate = effectiveRate;
  d.pad_ = 0;
  auto snap = [&]() {
    for (int k = 0; k < 4; k++) d.w[k] = st.w[k];
    d.pos = st.pos;
    d.ready = st.ready;
  };
  if (effectiveRate == 1.0) {  // :186-235
    d.copy = 1;
    snap();
    int64_t pos = st.pp;
    while (outIdx < kBlock) {
      if (loop && pos >= loopEnd) pos = loopStart;
      if (pos >= durEnd && !loop) break;
      int64_t endFrame = loop ? loopEnd : std::min(durEnd, len);
      int available = (int)std::min<int64_t>(endFrame - pos, kBlock - outIdx);
      if (available <= 0) break;
      if (first < 0) first = pos;
      pos += available;
      outIdx += available;
      hasMore = true;
    }
    st.pp += kBlock;
  } else {  // :236-358
    d.copy = 0;
    if (s.rsChannels != b.channels) {  // `_resamplers` (re)created and cleared (:238-245)
      st.w[0] = st.w[1] = st.w[2] = st.w[3] = -1;
      st.pos = 0.0;
      st.ready = 0;
      s.rsChannels = b.channels;
    }
    snap();
    int64_t pos = st.pp, consumedThis = 0;
    int guard = 0;
    while (outIdx < kBlock) {
      if (++guard > 4096) fail(GA_ERR_UNSUPPORTED, "source loop of zero length with resampling never finishes a block in the reference");
      if (loop && pos >= loopEnd) pos = loopStart;
      if (pos >= durEnd && !loop) break;
      int64_t endFrame = loop ? loopEnd : std::min(durEnd, len);
      int available = (int)std::min<int64_t>(endFrame - pos, len - pos);
      if (available <= 0) {
        if (loop) {
          pos = loopStart;
          consumedThis = pos - st.pp;
          continue;
        }
        break;
      }
      if (first < 0) first = pos;
      int consumed = 0, produced = 0;
      if (loop && pos + available >= loopEnd - 4) {  // the 512-sample wrap buffer (:297-314)
        const int64_t loopLength = loopEnd - loopStart;
        const int fromEnd = (int)(loopEnd - pos);
        const int needed = std::min(kBlock - outIdx + 4, 512);
        const int head = std::min(fromEnd, needed);
        const int tail = (int)std::min<int64_t>(std::max(needed - head, 0), loopLength);
        gsrProcess(st, [&](int k) { return k < head ? pos + k : loopStart + (k - head); }, head + tail, kBlock - outIdx,
                   effectiveRate, consumed, produced);
      } else {
        gsrProcess(st, [&](int k) { return pos + k; }, available, kBlock - outIdx, effectiveRate, consumed, produced);
      }
      if (produced > 0) hasMore = true;
      int64_t newPos = pos + consumed;
      if (loop && newPos >= loopEnd) newPos = loopStart + (newPos - loopEnd);
      consumedThis += (newPos >= pos) ? (newPos - pos) : (loopEnd - pos + newPos - loopStart);
      pos = newPos;
      outIdx += produced;
      if (consumed == 0 && produced == 0) break;
    }
    st.pp += consumedThis;
  }
  if (loop && st.pp >= loopEnd) {  // :226-234, :349-357
    int64_t loopLength = loopEnd - loopStart;
    if (loopLength > 0) st.pp = loopStart + ((st.pp - loopEnd) % loopLength);
  }
  d.next = first < 0 ? 0 : first;
  d.produced = outIdx;
  return !hasMore || (!loop && st.pp >= durEnd);
}

struct SrcPlanOut {
  int64_t playedBlocks = 0;  // PLAY + END blocks inside the chunk (advance of the node's state)
  bool reachedEnd = false;   // an END block with stopTime NaN was reached (stopTime := t1)
  int64_t endBlock = -1;
  bool gone = false;         // Ended raised + Dispose queued inside the chunk
  int64_t goneAt = -1;       // first block at which the node is disconnected
  int64_t partialBlock = -1; // resampler: block with fewer than 128 outputs
  int partialProduced = 0;
};

static SrcPlanOut planSource(Context& c, NodeS& s, int64_t n, const std::vector<double>& bt) {
  SrcPlanOut po;
  s.spans.clear();
  PlayBuf* b = s.bufId >= 0 ? c.buffers[s.bufId].get() : nullptr;
  if (!s.hasStarted || !b || s.disposed) {
    s.spans.push_back(SrcSpan{0, SRC_IDLE, 0, 0});
    return po;
  }
  // first block with t1 > startTime
  int64_t bs = std::upper_bound(bt.begin() + 1, bt.begin() + 1 + n, s.startTime) - (bt.begin() + 1);
  if (bs >= n || (!std::isnan(s.stopTime) && !(bt[bs] < s.stopTime))) {
    s.spans.push_back(SrcSpan{0, SRC_IDLE, 0, 0});
    return po;
  }
  if (bs > 0) s.spans.push_back(SrcSpan{0, SRC_IDLE, 0, 0});
  SrcGeom g = sourceGeom(c, s, *b);
  const int64_t INF = std::numeric_limits<int64_t>::max() / 4;
  // kTime: relative index of the block after which Ended is raised because t1 >= stopTime
  int64_t kTime = INF;
  if (!std::isnan(s.stopTime)) {
    int64_t kb = std::lower_bound(bt.begin() + 1 + bs, bt.begin() + 1 + n, s.stopTime) - (bt.begin() + 1 + bs);
    kTime = kb;  // may be >= n - bs: not inside this chunk
  }
  // kData: relative index of the first END (cleared) block
  int64_t kData = INF;
  const bool rate1 = g.effectiveRate == 1.0;
  int64_t pos = s.playbackPosition;
  const bool hasTimeline = !s.params[0].events.empty();
  const bool resamplerLive = s.gsr ? s.gsrReady > 0 : s.rsBlocks > 0;
  if (resamplerLive && s.rsBufId != s.bufId)
    fail(GA_ERR_UNSUPPORTED, "the Buffer of a source was replaced while its resampler holds samples of the old one");
  if (!resamplerLive) s.rsBufId = s.bufId;
  bool wantGsr = s.gsr || hasTimeline || (s.loop && !rate1) || (s.rsBlocks > 0 && g.effectiveRate != s.rsRate);
  if (wantGsr) {
    if (!s.gsr) {  // leave trajectory mode: the state after rsBlocks blocks becomes explicit
      if (s.rsBlocks > 0) {
        Resampler& rs = resamplerFor(c, s.rsRate);
        rs.extend(s.rsBlocks + 2);
        ResampleBlock rb = rs.blocks[s.rsBlocks];
        // the trajectory assumes unbounded input: if the data ran out in an earlier block the true state is that block's
        // bounded replay (later END blocks find nothing to consume, AudioBufferSourceNode.cs:267-271)
        const int64_t avail0 = std::max<int64_t>(g.durationEndFrame - s.rsStartPos, 0);
        if (rb.consumed >= avail0) {
          int64_t lo = 0, hi = s.rsBlocks - 1;
          while (lo < hi) {
            int64_t mid = (lo + hi) >> 1;
            if (rs.blocks[mid + 1].consumed >= avail0) hi = mid; else lo = mid + 1;
          }
          int produced;
          int64_t consumedAfter;
          double posAfter;
          int readyAfter;
          resampleBlockBounded(rs.blocks[lo], s.rsRate, avail0, produced, consumedAfter, &posAfter, &readyAfter);
          rb.consumed = consumedAfter;
          rb.pos = posAfter;
          rb.ready = readyAfter;
        }
        s.gsrPos = rb.pos;
        s.gsrReady = rb.ready;
        for (int k = 0; k < 4; k++) s.gsrW[3 - k] = k < rb.ready ? s.rsStartPos + rb.consumed - 1 - k : -1;
        s.playbackPosition = s.rsStartPos + rb.consumed;  // `_playbackPosition += totalInputConsumed` (:347)
        s.rsChannels = b->channels;
        s.rsBlocks = 0;
      }
      s.gsr = true;
    }
    GsrState st;
    for (int k = 0; k < 4; k++) st.w[k] = s.gsrW[k];
    st.pos = s.gsrPos;
    st.ready = s.gsrReady;
    st.pp = s.playbackPosition;
    s.gsrBlocks.clear();
    s.gsrUploaded = false;
    int64_t maxRel = n - bs;
    if (kTime != INF) maxRel = std::min(maxRel, kTime + 1);
    for (int64_t rel = 0; rel < maxRel; rel++) {
      float pr = hasTimeline ? param_value_at(s.params[0].events.data(), (int)s.params[0].events.size(), s.params[0].value, bt[bs + rel])
                             : s.params[0].value;  // k-rate: GetValues()[0] at the block start (AudioParam.cs:146-165)
      GsrBlock d;
      bool end = gsrReplayBlock(s, g, *b, c, pr, st, d);
      s.gsrBlocks.push_back(d);
      // END blocks keep being processed until the stop time (their state still moves: `_playbackPosition += 128` on the
      // copy path), and with unchanged controls an END block is followed by END blocks only
      if (end && kData == INF) kData = rel;
      if (!end && kData != INF) fail(GA_ERR_UNSUPPORTED, "a source resumed after an end block inside one render chunk");
    }
    // every index the device will touch is checked here, on the host: a wrong descriptor must be an error, not a GPU fault
    for (size_t bi = 0; bi < s.gsrBlocks.size(); bi++) {
      const GsrBlock& d = s.gsrBlocks[bi];
      if (kData != INF && (int64_t)bi >= kData) break;  // END blocks: cleared, no device reads
      int64_t ip = d.next;
      int64_t feeds = 0;
      if (d.copy) {
        feeds = d.produced;
      } else if (d.produced > 0) {
        for (int k = 0; k < 4; k++)
          if (d.w[k] < -1 || d.w[k] >= b->length) fail(GA_ERR_DEVICE, "internal: source replay window index out of range");
        feeds = 4 - d.ready;
        double P = d.pos;
        for (int o = 0; o < d.produced; o++) {
          int consume = (int)P;
          if (consume > 0) feeds += consume;
          P -= consume;
          P += d.rate;
        }
      }
      for (int64_t f = 0; f < feeds; f++) {
        if (ip < 0 || ip >= b->length) fail(GA_ERR_DEVICE, "internal: source replay feed index out of range");
        ip++;
        if (s.loop && ip >= g.loopEndFrame) ip = g.loopStartFrame;
      }
    }
    GsrBlock tail{};  // state after the last replayed block
    tail.pp = st.pp;
    for (int k = 0; k < 4; k++) tail.w[k] = st.w[k];
    tail.pos = st.pos;
    tail.ready = st.ready;
    s.gsrBlocks.push_back(tail);
  } else if (s.loop) {
    int64_t loopLen = g.loopEndFrame - g.loopStartFrame;
    if (loopLen <= 0) kData = 0;  // available <= 0 on the first iteration: hasMoreData stays false
  } else if (rate1) {
    int64_t rem = g.durationEndFrame - pos;
    kData = rem <= 0 ? 0 : (rem + kBlock - 1) / kBlock - 1;
  } else {
    Resampler& rs = resamplerFor(c, g.effectiveRate);
    if (s.rsBlocks == 0) {
      s.rsStartPos = pos;
      s.rsRate = g.effectiveRate;
    }
    int64_t avail = g.durationEndFrame - s.rsStartPos;
    int64_t need = s.rsBlocks + (n - bs) + 2;
    rs.extend(need + 1);
    // first trajectory block that consumes the LAST available input sample (or would need more): a block that ends with
    // _playbackPosition == durationEndFrame is already cleared by the reference (AudioBufferSourceNode.cs:360)
    int64_t jx = s.rsBlocks;
    {
      int64_t lo = s.rsBlocks, hi = need - 1;  // consumed at the END of block j = blocks[j+1].consumed
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (rs.blocks[mid + 1].consumed >= avail) hi = mid; else lo = mid + 1;
      }
      jx = (rs.blocks[lo + 1].consumed >= avail) ? lo : INF;
    }
    if (avail <= 0) {
      kData = 0;
    } else if (jx != INF) {
      int produced;
      int64_t consumedAfter;
      resampleBlockBounded(rs.blocks[jx], g.effectiveRate, avail, produced, consumedAfter);
      if (produced == 0 || consumedAfter >= avail) {
        kData = jx - s.rsBlocks;
      } else {
        kData = jx - s.rsBlocks + 1;
        po.partialBlock = bs + (jx - s.rsBlocks);
        po.partialProduced = produced;
      }
    }
  }
  // blocks [0, min(kData, kTime+1)) PLAY ; [kData, kTime] END ; gone after min(kTime, kData if stopTime was NaN)
  int64_t kGone;  // relative index of the last processed block (Ended raised after it)
  if (std::isnan(s.stopTime)) kGone = kData; else kGone = std::max(kTime, (int64_t)-1);
  if (!std::isnan(s.stopTime) && kTime == INF) kGone = INF;
  int64_t playEnd = std::min(kData, kGone == INF ? INF : kGone + 1);  // exclusive
  int64_t rel = 0;
  if (playEnd > 0) {
    if (s.gsr) {
      s.spans.push_back(SrcSpan{bs, SRC_PLAY, pos, 0});  // blkIdx indexes gsrBlocks
    } else if (s.loop && rate1 && pos >= g.loopEndFrame && g.loopEndFrame > g.loopStartFrame) {
      // start offset beyond the loop end: the first block restarts exactly at loopStart (`pos = loopStartFrame`,
      // AudioBufferSourceNode.cs:197-200) whereas _playbackPosition itself wraps modulo the loop length afterwards
      // (:226-234).  Reading from `loopEnd` makes the loop kernel's modular map start at loopStart for that block.
      s.spans.push_back(SrcSpan{bs, SRC_PLAY, g.loopEndFrame, s.rsBlocks});
      if (playEnd > 1 && bs + 1 < n) s.spans.push_back(SrcSpan{bs + 1, SRC_PLAY, pos + kBlock, s.rsBlocks + 1});
    } else {
      s.spans.push_back(SrcSpan{bs, SRC_PLAY, pos, s.rsBlocks});
    }
    rel = playEnd;
  }
  if (kData < (kGone == INF ? INF : kGone + 1) && bs + kData < n) {
    s.spans.push_back(SrcSpan{bs + kData, SRC_END, pos + kData * kBlock, s.rsBlocks + kData});
    if (std::isnan(s.stopTime)) {
      po.reachedEnd = true;
      po.endBlock = bs + kData;
    }
  }
  (void)rel;
  if (kGone != INF && bs + kGone + 1 <= n) {
    po.gone = true;
    po.goneAt = bs + kGone + 1;
    if (po.goneAt < n) s.spans.push_back(SrcSpan{po.goneAt, SRC_GONE, 0, 0});
  }
  int64_t lastProcessed = std::min<int64_t>(n, kGone == INF ? n : bs + kGone + 1);
  po.playedBlocks = lastProcessed - bs;
  // drop spans starting at or beyond the chunk end
  while (!s.spans.empty() && s.spans.back().b0 >= n) s.spans.pop_back();
  if (po.partialBlock >= n) po.partialBlock = -1;
  return po;
}

// ConstantSourceNode / OscillatorNode scheduling (ConstantSourceNode.cs:83-110,143-152; OscillatorNode.cs:97-118,160-169):
// sample-accurate start and stop inside a block, Ended + queued Dispose after the first block whose end reaches stopTime
static SrcPlanOut planScheduled(Context& c, NodeS& s, int64_t n, const std::vector<double>& bt) {
  SrcPlanOut po;
  s.spans.clear();
  s.schedLo = s.schedHi = 0;
  const int64_t INF = std::numeric_limits<int64_t>::max() / 4;
  // kEnd: first block with t1 >= stopTime (TryRaiseEnded runs in every processed block, playing or not)
  int64_t kEnd = INF;
  if (s.hasStarted && s.hasStopped && !s.endedRaised && !std::isnan(s.stopTime))
    kEnd = std::lower_bound(bt.begin() + 1, bt.begin() + 1 + n, s.stopTime) - (bt.begin() + 1);   // may be n: not in this chunk
  s.spans.push_back(SrcSpan{0, SRC_IDLE, 0, 0});
  if (s.hasStarted && !s.disposed) {
    // first block with t1 > startTime, last block with t0 < stopTime
    int64_t bs = std::upper_bound(bt.begin() + 1, bt.begin() + 1 + n, s.startTime) - (bt.begin() + 1);
    int64_t be = n - 1;
    if (!std::isnan(s.stopTime)) be = (std::lower_bound(bt.begin(), bt.begin() + n, s.stopTime) - bt.begin()) - 1;   // t0 < stop
    be = std::min(be, std::min<int64_t>(n - 1, kEnd));
    if (bs < n && bs <= be) {
      int startFrame = 0, endFrame = kBlock;
      if (bt[bs] < s.startTime && s.startTime < bt[bs + 1])
        startFrame = (int)std::min(std::max(std::ceil((s.startTime - bt[bs]) * c.sampleRate), 0.0), (double)kBlock);
      if (!std::isnan(s.stopTime) && bt[be] < s.stopTime && s.stopTime < bt[be + 1])
        endFrame = (int)std::min(std::max(std::floor((s.stopTime - bt[be]) * c.sampleRate), 0.0), (double)kBlock);
      s.schedLo = bs * kBlock + startFrame;
      s.schedHi = be * kBlock + endFrame;
      if (bs == be && endFrame < startFrame) s.schedHi = s.schedLo;   // `if (endFrame > startFrame)` (:126): nothing copied
      if (bs > 0) s.spans.push_back(SrcSpan{bs, SRC_PLAY, 0, 0}); else s.spans[0].phase = SRC_PLAY;
      if (be + 1 < n) s.spans.push_back(SrcSpan{be + 1, SRC_IDLE, 0, 0});
    }
  }
  if (kEnd < n) {
    po.gone = true;
    po.goneAt = kEnd + 1;
    if (po.goneAt < n) {
      while (!s.spans.empty() && s.spans.back().b0 >= po.goneAt) s.spans.pop_back();
      s.spans.push_back(SrcSpan{po.goneAt, SRC_GONE, 0, 0});
    }
  }
  return po;
}

static const SrcSpan& spanAt(const NodeS& s, int64_t b) {
  size_t i = s.spans.size() - 1;
  while (i > 0 && s.spans[i].b0 > b) i--;
  return s.spans[i];
}

// ======================================================================================================
// control-plane simulation
// ======================================================================================================
struct Sim {
  Context& c;
  int64_t n;
  int64_t blockNumber = 0;
  Segment* cur = nullptr;
  int64_t brel = 0;
  std::vector<int64_t>* extraBreaks = nullptr;   // chunk-relative blocks at which a node asks to be evaluated again
  const std::vector<double>* blockTimes = nullptr;   // accumulated block clock of the chunk (chunk-relative block -> time)

  int computeOutputChannelCount(InputS& in) {  // AudioNodeInput.cs:140-168
    switch (in.mode) {
      case GA_COUNT_MODE_EXPLICIT: return in.channelCount;
      case GA_COUNT_MODE_CLAMPED_MAX: {
        int mx = 0;
        for (const Conn& cn : in.connected) {
          int ch = c.nodes[cn.node]->outputs[cn.out].bufCh;
          if (ch) mx = std::max(mx, ch);
        }
        return std::min(mx == 0 ? in.channelCount : mx, in.channelCount);
      }
      default: {
        int mx = in.channelCount;
        for (const Conn& cn : in.connected) {
          int ch = c.nodes[cn.node]->outputs[cn.out].bufCh;
          if (ch) mx = std::max(mx, ch);
        }
        return mx;
      }
    }
  }

  void pull(NodeS& n_, int i, InSeg& is) {  // AudioNodeInput.Pull, AudioNodeInput.cs:100-138
    InputS& in = n_.inputs[i];
    if (in.connected.empty()) {
      in.bufCh = in.channelCount;
      in.dirty = false;
      in.silent = true;
      is.bufCh = in.bufCh;
      is.silent = true;
      return;
    }
    int outCh = computeOutputChannelCount(in);
    in.dirty = false;
    in.bufCh = outCh;
    bool mixed = false, allZero = true;
    for (size_t k = 0; k < in.connected.size(); k++) {
      Conn cn = in.connected[k];
      if (k + 2 < in.connected.size()) __builtin_prefetch(c.nodes[in.connected[k + 2].node].get());   // (tens of thousands of nodes: the walk is bound by cache misses)
      evalNode(cn.node);
      const NodeS& pn = *c.nodes[cn.node];
      const OutputS& o = pn.outputs[cn.out];
      if (o.bufCh != 0 && !o.silent) {   // (a producer that is still being processed shows the state of its PREVIOUS block: a stale term)
        is.terms.push_back(TermS{cn.node, cn.out, o.bufCh, pn.isProcessing});
        mixed = true;
        allZero = allZero && o.zero && !pn.isProcessing;
      }
    }
    in.silent = !mixed;
    is.bufCh = in.bufCh;
    is.silent = in.silent;
    is.zero = mixed && allZero;
  }

  void evalNode(int id) {  // AudioNode.ProcessInternal, Nodes/AudioNode.cs:152-183
    NodeS& n_ = *c.nodes[id];
    if (n_.lastProcessedBlock == blockNumber) return;
    if (n_.isProcessing) fail(GA_ERR_CYCLE, "Audio graph cycle detected at node " + std::to_string(id));   // (unreachable, as in the reference)
    n_.isProcessing = true;
    n_.lastProcessedBlock = blockNumber;
    NodeSeg ns;
    ns.id = id;
    ns.type = n_.type;
    ns.ins.resize(n_.inputs.size());
    // params first: ComputeValues pulls the modulation input (1 channel, explicit) before the node's inputs (:167-175)
    bool anyMod = false;
    for (auto& ps : n_.params) anyMod = anyMod || !ps.modulation.empty();
    if (anyMod) ns.pins.resize(n_.params.size());
    for (int p = 0; anyMod && p < (int)n_.params.size(); p++) {
      auto& mod = n_.params[p].modulation;
      InSeg& is = ns.pins[p];
      is.bufCh = 1;
      is.silent = true;
      for (auto& m : mod) {
        evalNode(m.first);
        const NodeS& pn = *c.nodes[m.first];
        const OutputS& o = pn.outputs[m.second];
        if (o.bufCh != 0 && !o.silent) {
          is.terms.push_back(TermS{m.first, m.second, o.bufCh, pn.isProcessing});
          is.silent = false;
        }
      }
    }
    for (int i = 0; i < (int)n_.inputs.size(); i++) pull(n_, i, ns.ins[i]);
    process(n_, ns);
    n_.isProcessing = false;
    cur->nodes.push_back(std::move(ns));
  }

  void process(NodeS& n_, NodeSeg& ns) {
    switch (n_.type) {
      case GA_NODE_DESTINATION:  // AudioDestinationNode.cs:42-64
        ns.outCh = ns.ins[0].bufCh;
        ns.outSilent = ns.ins[0].silent;
        c.destOutCh = ns.outCh;
        break;
      case GA_NODE_GAIN:  // GainNode.cs:29-61
        n_.outputs[0].bufCh = ns.ins[0].bufCh;
        n_.outputs[0].silent = ns.ins[0].silent;
        n_.outputs[0].zero = ns.ins[0].zero;   // (0 * g = 0 for every finite gain; a NaN gain is not worth a special case here)
        break;
      case GA_NODE_BIQUAD: {  // BiQuadFilterNode.cs:87-147
        n_.outputs[0].bufCh = ns.ins[0].bufCh;
        n_.outputs[0].silent = ns.ins[0].silent;
        if (!ns.ins[0].silent && !ns.ins[0].zero) n_.everFed = true;
        n_.outputs[0].zero = ns.ins[0].zero && !n_.everFed;   // zero input AND zero state
        ns.bqDynamic = !n_.params[0].events.empty() || !n_.params[1].events.empty() || !n_.params[2].events.empty() ||
                       !ns.pinSilent(0) || !ns.pinSilent(1) || !ns.pinSilent(2) ||   // a modulated parameter moves per sample
                       n_.bqDynSeq == c.chunkSeq ||   // (went dynamic earlier in this chunk: the coefficient state lives on the device until the chunk ends)
                       // the coefficient state is on the device and a signal is still connected to a parameter (silent right now):
                       // the per-sample kernel serves constants too, and the state is not fetched back per chunk (Context::chunkTopology)
                       (n_.coefOnDevice && (!n_.params[0].modulation.empty() || !n_.params[1].modulation.empty() || !n_.params[2].modulation.empty()));
        if (ns.bqDynamic && !ns.ins[0].silent) n_.bqDynSeq = c.chunkSeq;
        if (!ns.ins[0].silent && ns.bqDynamic) {
          ns.bqActive = true;   // coefficients are refreshed per sample on the device
        } else if (!ns.ins[0].silent) {
          float nyq = c.sampleRate / 2.f;
          float f = n_.params[0].value;
          f = f < 1.f ? 1.f : (f > nyq ? nyq : f);
          float q = std::max(0.001f, n_.params[1].value);
          float gainDb = n_.params[2].value;
          // usedFreq/usedQ start every block at 1000 / 1.0 (_lastFrequency/_lastQ are never updated, :13-14,111-112)
          if (n_.coefDirty || std::fabs(f - 1000.f) > 0.001f || std::fabs(q - 1.0f) > 0.0001f) {
            c.updateBiquadCoefficients(n_, f, q, gainDb);
            n_.coefDirty = false;
          }
          ns.bqActive = true;
          ns.b0 = n_.b0; ns.b1 = n_.b1; ns.b2 = n_.b2; ns.a1 = n_.a1; ns.a2 = n_.a2;
        }
        break;
      }
      case GA_NODE_CONVOLVER:  // ConvolverNode.cs:102-155
        if (!n_.ir) {
          n_.outputs[0].bufCh = ns.ins[0].bufCh;
          n_.outputs[0].silent = true;
          n_.outputs[0].zero = false;
        } else {
          n_.outputs[0].bufCh = n_.effectiveOutCh;
          n_.outputs[0].silent = false;  // MarkAsNonSilent even for silent input (:153)
          if (!ns.ins[0].silent && !ns.ins[0].zero) n_.everFed = true;
          n_.outputs[0].zero = !n_.everFed;   // nothing has reached the input yet: the flagged-non-silent output is exact zeros
        }
        break;
      case GA_NODE_BUFFER_SOURCE: {
        const SrcSpan& sp = spanAt(n_, brel);
        PlayBuf* b = n_.bufId >= 0 ? c.buffers[n_.bufId].get() : nullptr;
        ns.srcPhase = sp.phase;
        ns.srcBuf = n_.bufId;
        if (sp.phase == SRC_PLAY && b) {
          n_.outputs[0].bufCh = b->channels;
          n_.outputs[0].silent = false;
          ns.srcPos = sp.pos + (brel - sp.b0) * kBlock;
          ns.srcBlk = sp.blkIdx + (brel - sp.b0);
        } else if (sp.phase == SRC_END && b) {  // whole block cleared (:360-368)
          n_.outputs[0].bufCh = b->channels;
          n_.outputs[0].silent = true;
        } else {  // ProduceSilence: 1-channel silent buffer (:391-402)
          n_.outputs[0].bufCh = 1;
          n_.outputs[0].silent = true;
        }
        break;
      }
      case GA_NODE_CHANNEL_SPLITTER: {  // ChannelSplitterNode.cs:24-59: N mono outputs
        const InSeg& in = ns.ins[0];
        for (int o = 0; o < (int)n_.outputs.size(); o++) {
          const bool audio = !in.silent && o < in.bufCh;
          n_.outputs[o].bufCh = 1;
          n_.outputs[o].silent = !audio;
          if (audio) ns.outMask |= 1u << o;
        }
        break;
      }
      case GA_NODE_CHANNEL_MERGER: {  // ChannelMergerNode.cs:23-55: channel i = channel 0 of input i
        bool any = false;
        for (int i = 0; i < (int)ns.ins.size(); i++)
          if (!ns.ins[i].silent) {
            any = true;
            ns.outMask |= 1u << i;
          }
        n_.outputs[0].bufCh = (int)ns.ins.size();
        n_.outputs[0].silent = !any;
        break;
      }
      case GA_NODE_STREAM_SOURCE: {  // AudioStreamSourceNodeBase.cs:132-301: channel count / silence per block from the host replay
        const NodeS::StreamBlockInfo bi = brel < (int64_t)n_.stInfo.size() ? n_.stInfo[brel] : NodeS::StreamBlockInfo{1, true};
        n_.outputs[0].bufCh = bi.outCh;
        n_.outputs[0].silent = bi.silent;
        break;
      }
      case GA_NODE_CONSTANT_SOURCE:
      case GA_NODE_OSCILLATOR: {  // always a 1-channel buffer; non-silent in every block that plays (:136, :151)
        const SrcSpan& sp = spanAt(n_, brel);
        ns.srcPhase = sp.phase;
        n_.outputs[0].bufCh = 1;
        n_.outputs[0].silent = sp.phase != SRC_PLAY;
        break;
      }
      case GA_NODE_STEREO_PANNER: {  // StereoPannerNode.cs:36-74
        n_.outputs[0].bufCh = 2;
        n_.outputs[0].silent = ns.ins[0].silent;
        n_.outputs[0].zero = ns.ins[0].zero;
        ns.panMode = ns.ins[0].bufCh == 1 ? 1 : 2;
        if (!ns.ins[0].silent && (!n_.params[0].events.empty() || !ns.pinSilent(0) || n_.panDynSeq == c.chunkSeq ||
                                  (n_.panOnDevice && !n_.params[0].modulation.empty()))) {   // (state on the device, a signal still connected: Context::chunkTopology)
          n_.panDynSeq = c.chunkSeq;
          ns.panDyn = true;   // gains follow the a-rate curve on the device (stereo_panner_dynamic_kernel)
        } else if (!ns.ins[0].silent) {
          float pan = std::min(std::max(n_.params[0].value, -1.0f), 1.0f);
          if (pan != n_.panLast) {  // the gains follow the law of the path that sees the change (:92-99, :127-134)
            const float PIf = 3.14159265358979323846f;
            float x = ns.panMode == 1 ? (pan + 1.0f) * 0.5f : (pan <= 0.0f ? pan + 1.0f : pan);
            n_.panGL = std::cos(x * PIf / 2.0f);
            n_.panGR = std::sin(x * PIf / 2.0f);
            n_.panLast = pan;
          }
          ns.pan = pan;
          ns.panGL = n_.panGL;
          ns.panGR = n_.panGR;
        }
        break;
      }
      case GA_NODE_DELAY: {  // DelayNode.cs:43-100
        const InSeg& in = ns.ins[0];
        const int ch = in.bufCh;
        if (ch != n_.delayCh) {   // `_outputBuffer` re-rented: a cleared buffer is silent again (:49-55)
          n_.delayAudible = false;
          n_.delayCh = ch;
        }
        const int64_t B = c.currentBlock + brel;   // absolute block of this evaluation
        const int64_t OPEN = std::numeric_limits<int64_t>::max();
        const int maxD = n_.maxDelaySamples;
        auto& model = n_.delayModel;
        // rings that were written since the previous evaluation advanced by the blocks in between
        if (n_.delayPrevEval >= 0)
          for (int r = 0; r < std::min((int)model.size(), n_.delayPrevCh); r++) model[r].pos += (B - n_.delayPrevEval) * kBlock;
        n_.delayPrevEval = B;
        n_.delayPrevCh = ch;
        if ((int)model.size() < std::max(ch, 2)) model.resize(std::max(ch, 2));   // EnsureChannelCount (:102-113)
        for (int r = 0; r < (int)model.size(); r++) {
          auto& m = model[r];
          const bool writesAudio = r < ch && !in.silent && !in.zero;   // (exact zeros never raise the output flag)
          if (writesAudio && !m.open) {
            m.runs.push_back({m.pos, OPEN});
            m.open = true;
          } else if (!writesAudio && m.open) {
            m.runs.back().second = m.pos;
            m.open = false;
          }
          while (m.runs.size() > 1 && m.runs.front().second != OPEN && m.runs.front().second + maxD + 2 * kBlock < m.pos) m.runs.erase(m.runs.begin());
        }
        // The output buffer's non-silent flag is set by the first non-zero output SAMPLE and never cleared (:72,:92,:96-97).
        // Data is not visible to the control plane: samples that came from a non-silent input block are taken to be non-zero.
        int dmin = 1, dmax = maxD;
        if (n_.params[0].events.empty() && ns.pinSilent(0)) {
          int d = (int)(n_.params[0].value * (float)c.sampleRate);
          d = std::min(std::max(d, 0), maxD);
          dmin = dmax = d;
        }
        // A delay time on a timeline (no audio-rate modulation): the host evaluates the same per-sample curve the device does
        // (param_value_at at blockTime + i / sampleRate, DelayNode.cs:66,86) and tests every frame of this block.  With the
        // [1, maxDelay] bound used for modulated delay times the output would be flagged non-silent the moment its INPUT becomes
        // audible -- blocks before the delayed audio arrives -- and a consumer whose state was frozen by silence (a biquad with a
        // second connection that ended earlier) would wake up too early.  While audio is on its way the node is evaluated again
        // block by block.
        const bool timelineOnly = !n_.params[0].events.empty() && ns.pinSilent(0) && blockTimes && brel < (int64_t)blockTimes->size();
        if (!n_.delayAudible && timelineOnly) {
          const ParamS& pd = n_.params[0];
          const double t0 = (*blockTimes)[brel], dts = 1.0 / c.sampleRate;
          bool pending = false;
          for (int r = 0; r < ch && !n_.delayAudible; r++) {
            auto& m = model[r];
            for (auto& run : m.runs)
              if (run.second == OPEN || run.second + maxD >= m.pos) pending = true;
            if (m.runs.empty()) continue;
            for (int i = 0; i < kBlock && !n_.delayAudible; i++) {
              const float dtv = param_value_at(pd.events.data(), (int)pd.events.size(), pd.value, pd.arate ? t0 + i * dts : t0);
              int d = (int)(dtv * (float)c.sampleRate);
              d = std::min(std::max(d, 0), maxD);
              if (d == 0) continue;   // (reads nothing: delay_kernel writes 0)
              const int64_t q = m.pos + i - d;
              for (auto& run : m.runs)
                if (q >= run.first && (run.second == OPEN || q < run.second)) {
                  n_.delayAudible = true;
                  break;
                }
            }
          }
          if (!n_.delayAudible && pending && extraBreaks) extraBreaks->push_back(brel + 1);
        } else if (!n_.delayAudible && dmax > 0) {
          dmin = std::max(dmin, 1);
          int64_t nextFlip = OPEN;
          for (int r = 0; r < ch && !n_.delayAudible; r++) {
            auto& m = model[r];
            const int64_t lo = m.pos - dmax, hi = m.pos + (kBlock - 1) - dmin;   // ring frames this block can read
            for (auto& run : m.runs) {
              const int64_t rs = run.first, re = run.second == OPEN ? OPEN : run.second - 1;
              if (rs <= hi && re >= lo) {
                n_.delayAudible = true;
                break;
              }
              if (rs > hi) {   // arrives k blocks from now: pos + 128 k + 127 - dmin >= rs
                int64_t k = (rs + dmin - (kBlock - 1) - m.pos + kBlock - 1) / kBlock;
                nextFlip = std::min(nextFlip, B + std::max<int64_t>(k, 1));
              }
            }
          }
          if (!n_.delayAudible && nextFlip != OPEN && extraBreaks) extraBreaks->push_back(nextFlip - c.currentBlock);
        }
        ns.delayAudible = n_.delayAudible;
        n_.outputs[0].bufCh = ch;
        n_.outputs[0].silent = !n_.delayAudible;
        break;
      }
      default: fail(GA_ERR_UNSUPPORTED, "node type not supported on the device path");
    }
    if (!n_.outputs.empty()) {
      ns.outCh = n_.outputs[0].bufCh;
      ns.outSilent = n_.outputs[0].silent;
      ns.outZero = !n_.outputs[0].silent && n_.outputs[0].zero;
    }
  }

  uint64_t hashSeg(const Segment& s) {
    uint64_t h = 1469598103934665603ull;
    for (const NodeSeg& ns : s.nodes) {
      h = hmix(h, (uint64_t)ns.id);
      h = hmix(h, ((uint64_t)ns.outCh << 8) | (ns.outSilent ? 1 : 0) | ((uint64_t)ns.srcPhase << 4) | (ns.bqActive ? 2 : 0) |
                      ((uint64_t)ns.outMask << 16) | ((uint64_t)ns.panMode << 48) | ((uint64_t)(ns.panDyn ? 1 : 0) << 52) |
                      ((uint64_t)(ns.outZero ? 1 : 0) << 53));
      for (const InSeg& is : ns.ins) {
        h = hmix(h, ((uint64_t)is.bufCh << 1) | (is.silent ? 1 : 0));
        for (const TermS& t : is.terms) h = hmix(h, ((uint64_t)t.node << 16) | ((uint64_t)t.out << 8) | (uint64_t)t.ch | ((uint64_t)t.stale << 60));
      }
      for (const InSeg& is : ns.pins)
        for (const TermS& t : is.terms) h = hmix(h, 0x5151ull ^ (((uint64_t)t.node << 16) | ((uint64_t)t.out << 8) | (uint64_t)t.ch | ((uint64_t)t.stale << 60)));
    }
    return h;
  }
};

// ======================================================================================================
// executor
// ======================================================================================================
struct Exec {
  Context& c;
  int64_t n, frames;
  Plan plan;
  std::vector<Segment>& segs;
  std::unordered_map<uint64_t, float*> nodeSlab, inSlab;   // (node outputs: channels >= 2 only, see nodeOut)
  std::vector<float*> nodeSlab01;   // [node][channel 0, 1]: a dense table for the slabs nearly every node asks for
  std::vector<std::vector<Views>> outViews;  // [segment][node][channel]
  // per (level) batch tables
  std::vector<const float*> terms;
  // constant GainNodes folded into their consumer's mix (option "gain_fold"): the node hands its input views on and records its
  // gain per segment; the consumer's mix / down-mix job multiplies the term first -- fl(x * g), then the add, exactly the values
  // GainNode.Process (GainNode.cs:48-58) + AudioNodeInput.MixBuffer produce, without a pass over the samples in between
  std::vector<float> termGains;                  // parallel to `terms` (missing entries = 1)
  bool anyTermGain = false;                      // this level has a term with a gain != 1
  std::vector<std::vector<float>> outScale;      // [segment][node]: allocated for a segment when its first gain is folded
  float scaleOf(int si, int node) const { return (si < (int)outScale.size() && !outScale[si].empty()) ? outScale[si][node] : 1.f; }
  void setScale(int si, int node, float g) {
    if ((int)outScale.size() <= si) outScale.resize(si + 1);
    if (outScale[si].empty()) outScale[si].assign(c.nodes.size(), 1.f);
    outScale[si][node] = g;
  }
  // ... and a GainNode whose gain follows a TIMELINE (no audio-rate modulation) and has one consumer hands on its input views with
  // the curve: the consumer's mix multiplies the term by curve[f] first -- GainNode.Process's `out = in * gain[i]` (GainNode.cs:
  // 52-57), the same product, without writing and re-reading the voice (config 4: 4096 gain curves in front of the destination)
  std::vector<const float*> termCurves;          // parallel to `terms` (missing entries = null)
  bool anyTermCurve = false;
  std::vector<std::vector<const float*>> outCurve;   // [segment][node]
  const float* curveOf(int si, int node) const { return (si < (int)outCurve.size() && !outCurve[si].empty()) ? outCurve[si][node] : nullptr; }
  void setCurve(int si, int node, const float* cv) {
    if ((int)outCurve.size() <= si) outCurve.resize(si + 1);
    if (outCurve[si].empty()) outCurve[si].assign(c.nodes.size(), nullptr);
    outCurve[si][node] = cv;
  }
  void pushTerm(const float* p, float g, const float* cv = nullptr) {
    termGains.resize(terms.size(), 1.f);
    termCurves.resize(terms.size(), nullptr);
    terms.push_back(p);
    termGains.push_back(g);
    termCurves.push_back(cv);
    if (g != 1.f) anyTermGain = true;
    if (cv) anyTermGain = anyTermCurve = true;
  }
  std::vector<MixJob> mixJobs;
  std::vector<DownmixJob> dmJobs;
  std::vector<GainJob> gainJobs;
  std::vector<BiquadJob> bqJobs[kMaxBiquadSections + 1];  // by cascade length
  std::vector<BiquadSection> bqSecs;
  // cascades split along time (ga_kernels.hpp, BiquadScanJob): pass A / pass B pieces by cascade length, the scans, A^K matrices
  std::vector<BiquadScanJob> bqScans[kMaxBiquadSections + 1];          // by cascade length; all of one level share G and K
  std::vector<const std::vector<float>*> bqMats[kMaxBiquadSections + 1];   // their A^K -> m_off once the table exists
  int bqG = 0;
  int64_t bqK = 0;
  size_t bqZeroFrom = 0;   // Context::bqSplitUsed up to which the pieces' states are already covered by a zeroing launch
  std::vector<BiquadDynJob> bqDynJobs;
  std::vector<LoopJob> loopJobs;
  std::vector<ResampleJob> rsJobs;
  std::vector<ResampleFastJob> rsFastJobs;   // (full blocks of a trajectory whose per-sample table is on the device: one lane per output)
  std::vector<GsrJob> gsrJobs;
  std::vector<StreamJob> streamJobs;
  std::vector<ConstJob> constJobs;
  std::vector<OscJob> oscJobs;
  std::vector<PanJob> panJobs;
  std::vector<DelayJob> delayJobs;
  std::vector<PanDynJob> panDynJobs;
  std::vector<ParamModJob> pmodJobs;
  std::vector<ResampleBlock> traj;  // per-chunk trajectory table (all rates + custom tail blocks)
  bool mixAligned = true;
  // conv inputs: node -> slot -> per segment view
  // conv inputs: node -> per segment views of its input channels (a dense table: one lookup per convolver and pass)
  struct ConvInRow {   // the per-segment views of one node (a window of ConvIn::flat)
    Views* p = nullptr;
    size_t n = 0;
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    Views& operator[](size_t i) const { return p[i]; }
  };
  struct ConvIn {
    std::vector<int> slot;     // node id -> index of its row, -1 = the node's input was not resolved in this chunk
    std::vector<Views> flat;   // [row][segment]
    size_t nsegs = 1, used = 0;
    bool has(int id) const { return id < (int)slot.size() && slot[id] >= 0; }
    ConvInRow operator[](int id) {
      if (id >= (int)slot.size()) slot.resize(id + 1, -1);
      if (slot[id] < 0) {
        slot[id] = (int)used++;
        if (flat.size() < used * nsegs) flat.resize(std::max(used * nsegs, 2 * flat.size()));
      }
      return ConvInRow{flat.data() + (size_t)slot[id] * nsegs, nsegs};
    }
  } convIn;

  Exec(Context& c_, int64_t n_, std::vector<Segment>& s) : c(c_), n(n_), frames(n_ * kBlock), segs(s) {
    nodeSlab01.assign(c.nodes.size() * 2, nullptr);
    convIn.slot.assign(c.nodes.size(), -1);
    convIn.nsegs = std::max<size_t>(segs.size(), 1);
  }

  float* slabFor(std::unordered_map<uint64_t, float*>& m, uint64_t key) {
    auto it = m.find(key);
    if (it != m.end()) return it->second;
    float* p = getSlab(c);
    m[key] = p;
    return p;
  }
  float* nodeOut(int node, int ch) {
    if (ch < 2 && (size_t)node * 2 + 1 < nodeSlab01.size()) {
      float*& p = nodeSlab01[(size_t)node * 2 + ch];
      if (!p) p = getSlab(c);
      return p;
    }
    return slabFor(nodeSlab, ((uint64_t)node << 8) | (uint64_t)ch);
  }
  void setNodeOut(int node, int ch, float* p) {   // a node output that is produced in place somewhere else (Context::aliasBusToLeader)
    if (ch < 2 && (size_t)node * 2 + 1 < nodeSlab01.size()) nodeSlab01[(size_t)node * 2 + ch] = p;
    else nodeSlab[((uint64_t)node << 8) | (uint64_t)ch] = p;
  }
  float* inMixed(int node, int input, int ch) { return slabFor(inSlab, ((uint64_t)node << 16) | ((uint64_t)input << 8) | (uint64_t)ch); }

  void noteAlign(const float* p, int64_t f0) {
    if (((uintptr_t)(p + f0)) & 15) mixAligned = false;
  }

  // AudioNodeInput.Pull + MixBuffer (AudioNodeInput.cs:100-138,182-244) for one input over one segment
  Views resolveInput(int si, const NodeSeg& ns, int i, bool force, float* const* forcedSlabs) {
    return resolveInSeg(si, ns.id, i, ns.ins[i], force, forcedSlabs);
  }
  // i >= 0: node input i ; i < 0: modulation input of param (-1 - i)
  Views resolveInSeg(int si, int nodeId, int i, const InSeg& is, bool force, float* const* forcedSlabs) {
    const Segment& sg = segs[si];
    const int dstCh = is.bufCh;
    const int64_t f0 = sg.b0 * kBlock, nf = (sg.b1 - sg.b0) * kBlock;
    struct Tm { const float* p; float g; const float* c; };
    SmallVec<SmallVec<Tm, 2>, 4> lists((size_t)dstCh);
    for (const TermS& t : is.terms) {
      Views staleViews;
      if (t.stale) {   // feedback edge: the block the producer put out last (kept by Context::chunkStaleCommit)
        const NodeS& pn = *c.nodes[t.node];
        const int rows = pn.type == GA_NODE_CHANNEL_SPLITTER ? (int)pn.outputs.size() : t.ch;
        staleViews.assign((size_t)rows, nullptr);
        if (frames <= kBlock) {   // a chunk of one block: the kept block itself
          for (int r = 0; r < rows; r++) staleViews[r] = (pn.staleBuf && r < pn.staleRows) ? pn.staleBuf + (size_t)r * kBlock : c.zeros;
        } else {
          // A chunk of several blocks (every loop cut at a DelayNode, Context::chunkTopology): block b of this input reads block
          // b - 1 of the producer -- the kept block for the chunk's first block, the producer's own output of THIS chunk, one block
          // late, for the others (the producer is planned before this consumer: the stale edge is an ordinary forward edge of the
          // cut graph).  Materialised per row by two copy jobs (the down-mix launch of this level runs before its mix launch).
          const float gB = scaleOf(si, t.node);
          for (int r = 0; r < rows; r++) {
            float* T = getSlab(c);
            const float* srcA = nullptr;
            float gA = 1.f;
            if (sg.b0 == 0) {
              srcA = (pn.staleBuf && r < pn.staleRows) ? pn.staleBuf + (size_t)r * kBlock : nullptr;   // (indexed from frame 0)
            } else if (si > 0 && t.node < (int)outViews[si - 1].size() && r < (int)outViews[si - 1][t.node].size() && outViews[si - 1][t.node][r]) {
              srcA = outViews[si - 1][t.node][r] - kBlock;
              gA = scaleOf(si - 1, t.node);
            }
            DownmixJob a;
            a.out = T;
            a.term0 = (int)terms.size();
            a.nch = 1;
            a.scale = 1.0f;
            a.f0 = f0;
            a.n = std::min<int64_t>(nf, kBlock);
            pushTerm(srcA ? srcA : c.zeros, gA);
            dmJobs.push_back(a);
            if (nf > kBlock) {
              const auto& cv = outViews[si][t.node];
              const float* srcB = (r < (int)cv.size() && cv[r]) ? cv[r] - kBlock : nullptr;
              DownmixJob b;
              b.out = T;
              b.term0 = (int)terms.size();
              b.nch = 1;
              b.scale = 1.0f;
              b.f0 = f0 + kBlock;
              b.n = nf - kBlock;
              pushTerm(srcB ? srcB : c.zeros, srcB ? gB : 1.f);
              dmJobs.push_back(b);
            }
            staleViews[r] = T;
          }
        }
      }
      const auto& uvAll = t.stale ? staleViews : outViews[si][t.node];
      const float g = t.stale ? 1.f : scaleOf(si, t.node);   // (a folded constant GainNode: its views are its INPUT's, to be multiplied here)
      const float* gc = t.stale ? nullptr : curveOf(si, t.node);   // (... or by its gain curve)
      // a ChannelSplitterNode keeps one mono view per OUTPUT; every other node has one output with t.ch channels
      Views uvOne;
      if (c.nodes[t.node]->type == GA_NODE_CHANNEL_SPLITTER) uvOne.assign(1, t.out < (int)uvAll.size() ? uvAll[t.out] : nullptr);
      const auto& uv = c.nodes[t.node]->type == GA_NODE_CHANNEL_SPLITTER ? uvOne : uvAll;
      const int srcCh = t.ch;
      if (srcCh == dstCh) {
        for (int ch = 0; ch < dstCh; ch++)
          if (uv[ch]) lists[ch].push_back(Tm{uv[ch], g, gc});
      } else if (srcCh == 1 && dstCh > 1) {
        if (uv[0])
          for (int ch = 0; ch < dstCh; ch++) lists[ch].push_back(Tm{uv[0], g, gc});
      } else if (srcCh > 1 && dstCh == 1) {
        // (sum over channels) * 1/sqrt(N), AudioNodeInput.cs:214-228
        bool anyCh = false;
        for (int ch = 0; ch < srcCh; ch++) anyCh = anyCh || uv[ch] != nullptr;
        if (!anyCh) continue;   // e.g. a convolver whose output is carried by the leader of its fused group
        DownmixJob dj;
        dj.out = getSlab(c);
        dj.term0 = (int)terms.size();
        dj.nch = srcCh;
        dj.scale = 1.0f / std::sqrt((float)srcCh);
        dj.f0 = f0;
        dj.n = nf;
        for (int ch = 0; ch < srcCh; ch++) pushTerm(uv[ch] ? uv[ch] : c.zeros, g);
        dmJobs.push_back(dj);
        lists[0].push_back(Tm{dj.out, 1.f, nullptr});
      } else {
        int m = std::min(srcCh, dstCh);
        for (int ch = 0; ch < m; ch++)
          if (uv[ch]) lists[ch].push_back(Tm{uv[ch], g, gc});
      }
    }
    Views views((size_t)dstCh, nullptr);
    for (int ch = 0; ch < dstCh; ch++) {
      auto& l = lists[ch];
      if (!force) {
        if (l.empty()) continue;
        if (l.size() == 1 && l[0].g == 1.f && !l[0].c) {
          views[ch] = l[0].p;
          continue;
        }
      }
      float* out = forcedSlabs ? forcedSlabs[ch] : inMixed(nodeId, i + 64, ch);
      if (!out) continue;
      if (l.size() == 1 && l[0].p == out && l[0].g == 1.f && !l[0].c) {   // the only term was produced in place (Context::aliasBusToLeader)
        views[ch] = out;
        continue;
      }
      MixJob mj;
      mj.out = out;
      mj.term0 = (int)terms.size();
      mj.nterms = (int)l.size();
      mj.f0 = f0;
      mj.n = nf;
      for (const Tm& tm : l) {
        pushTerm(tm.p, tm.g, tm.c);
        noteAlign(tm.p, f0);
        if (tm.c) noteAlign(tm.c, f0);
      }
      noteAlign(out, f0);
      mixJobs.push_back(mj);
      views[ch] = out;
    }
    return views;
  }

  // the per-frame values of parameter p of node `ns` in segment si: the timeline curve, or -- when a non-silent signal is
  // connected to the parameter -- clamp(intrinsic + modulation) (AudioParam.cs:123-135,148-160); null = the constant Value
  const float* paramView(int si, const NodeSeg& ns, int p) {
    NodeS& nd = *c.nodes[ns.id];
    ParamS& ps = nd.params[p];
    if (p >= (int)ns.pins.size() || ns.pins[p].silent) return ps.curve;
    const Segment& sg = segs[si];
    auto mv = resolveInSeg(si, ns.id, -1 - p, ns.pins[p], false, nullptr);
    if (mv.empty() || !mv[0]) return ps.curve;
    ParamModJob pj;
    pj.intrinsic = ps.curve;
    pj.mod = mv[0];
    pj.out = slabFor(inSlab, ((uint64_t)ns.id << 16) | ((uint64_t)(200 + p) << 8));
    pj.value = ps.value;
    pj.vmin = ps.minv;
    pj.vmax = ps.maxv;
    pj.krate = ps.arate ? 0 : 1;
    pj.f0 = sg.b0 * kBlock;
    pj.n = (sg.b1 - sg.b0) * kBlock;
    pmodJobs.push_back(pj);
    return pj.out;
  }

  void flushLevel() {
    // order: down-mix -> mix -> sources -> gain -> biquad (everything in one level is independent)
    size_t termsOff = plan.putv(terms);
    termGains.resize(terms.size(), 1.f);
    termCurves.resize(terms.size(), nullptr);
    const bool scaled = anyTermGain;
    const size_t gainsOff = scaled ? plan.putv(termGains) : 0;
    const bool curved = anyTermCurve;
    const size_t curvesOff = curved ? plan.putv(termCurves) : 0;
    if (!dmJobs.empty()) {
      size_t off = plan.putv(dmJobs);
      int nj = (int)dmJobs.size();
      int64_t mx = 0;
      for (auto& j : dmJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) {
        launch_downmix(st, (const DownmixJob*)(base + off), nj, (const float* const*)(base + termsOff), mx, scaled ? (const float*)(base + gainsOff) : nullptr);
      });
    }
    if (!mixJobs.empty()) {
      size_t off = plan.putv(mixJobs);
      int nj = (int)mixJobs.size();
      int64_t mx = 0;
      for (auto& j : mixJobs) mx = std::max(mx, j.n);
      bool v4 = mixAligned;
      hipStream_t st = c.stream;
      double mixBytes = 0;
      for (auto& j : mixJobs) mixBytes += 4.0 * (double)(j.nterms + 1) * (double)j.n;
      plan.add(LK_MIX, [=](uint8_t* base) {
        launch_mix(st, (const MixJob*)(base + off), nj, (const float* const*)(base + termsOff), mx, v4, scaled ? (const float*)(base + gainsOff) : nullptr,
                   curved ? (const float* const*)(base + curvesOff) : nullptr);
      }, mixBytes);
    }
    if (!pmodJobs.empty()) {   // after the mixes (the modulation inputs), before the nodes that read the parameter
      size_t off = plan.putv(pmodJobs);
      int nj = (int)pmodJobs.size();
      int64_t mx = 0;
      for (auto& j : pmodJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_param_mod(st, (const ParamModJob*)(base + off), nj, mx); });
    }
    if (!loopJobs.empty()) {
      size_t off = plan.putv(loopJobs);
      int nj = (int)loopJobs.size();
      int64_t mx = 0;
      for (auto& j : loopJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_loop_source(st, (const LoopJob*)(base + off), nj, mx); });
    }
    if (!rsJobs.empty()) {
      size_t off = plan.putv(rsJobs);
      int nj = (int)rsJobs.size();
      int64_t mx = 0;
      for (auto& j : rsJobs) mx = std::max(mx, j.nblocks);
      hipStream_t st = c.stream;
      rsLaunches.push_back(RsLaunch{off, nj, mx});
      plan.add(LK_OTHER, [this, st, idx = rsLaunches.size() - 1](uint8_t* base) {
        const RsLaunch& r = rsLaunches[idx];
        launch_resample(st, (const ResampleJob*)(base + r.off), r.nj, (const ResampleBlock*)(base + trajOffFinal), r.mx);
      });
    }
    if (!rsFastJobs.empty()) {
      size_t off = plan.putv(rsFastJobs);
      int nj = (int)rsFastJobs.size();
      int64_t mx = 0;
      for (auto& j : rsFastJobs) mx = std::max(mx, j.nblocks);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_resample_fast(st, (const ResampleFastJob*)(base + off), nj, mx); });
    }
    if (!gsrJobs.empty()) {
      size_t off = plan.putv(gsrJobs);
      int nj = (int)gsrJobs.size();
      int64_t mx = 0;
      for (auto& j : gsrJobs) mx = std::max(mx, j.nblocks);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_gsr(st, (const GsrJob*)(base + off), nj, base, mx); });
    }
    if (!streamJobs.empty()) {
      size_t off = plan.putv(streamJobs);
      int nj = (int)streamJobs.size();
      int64_t mx = 0;
      for (auto& j : streamJobs) mx = std::max(mx, j.nblocks);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_stream(st, (const StreamJob*)(base + off), nj, base, mx); });
    }
    if (!constJobs.empty()) {
      size_t off = plan.putv(constJobs);
      int nj = (int)constJobs.size();
      int64_t mx = 0;
      for (auto& j : constJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_const_source(st, (const ConstJob*)(base + off), nj, mx); });
    }
    if (!oscJobs.empty()) {
      size_t off = plan.putv(oscJobs);
      int nj = (int)oscJobs.size();
      hipStream_t st = c.stream;
      bool anyCurve = false;
      for (auto& j : oscJobs) anyCurve = anyCurve || j.curve != nullptr;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_oscillator(st, (const OscJob*)(base + off), nj, anyCurve); });
    }
    if (!panJobs.empty()) {
      size_t off = plan.putv(panJobs);
      int nj = (int)panJobs.size();
      int64_t mx = 0;
      for (auto& j : panJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_stereo_panner(st, (const PanJob*)(base + off), nj, mx); });
    }
    if (!panDynJobs.empty()) {
      size_t off = plan.putv(panDynJobs);
      int nj = (int)panDynJobs.size();
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_stereo_panner_dynamic(st, (const PanDynJob*)(base + off), nj); });
    }
    if (!delayJobs.empty()) {   // after the mix jobs of this level, which append the input to the delay lines
      size_t off = plan.putv(delayJobs);
      int nj = (int)delayJobs.size();
      int64_t mx = 0;
      for (auto& j : delayJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_delay(st, (const DelayJob*)(base + off), nj, mx); });
    }
    if (!gainJobs.empty()) {
      size_t off = plan.putv(gainJobs);
      int nj = (int)gainJobs.size();
      int64_t mx = 0;
      for (auto& j : gainJobs) mx = std::max(mx, j.n);
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_gain(st, (const GainJob*)(base + off), nj, mx); });
    }
    {
      bool any = false;
      for (int k = 1; k <= kMaxBiquadSections; k++) any = any || !bqJobs[k].empty();
      if (any) {
        size_t soff = plan.putv(bqSecs);
        for (int k = 1; k <= kMaxBiquadSections; k++) {
          if (bqJobs[k].empty()) continue;
          size_t off = plan.putv(bqJobs[k]);
          int nj = (int)bqJobs[k].size();
          hipStream_t st = c.stream;
          plan.add(LK_OTHER, [=](uint8_t* base) {
            launch_biquad(st, (const BiquadJob*)(base + off), nj, (const BiquadSection*)(base + soff), k);
          });
        }
      }
    }
    if (bqG > 1) {   // cascades split along time: expand the pieces, zero their states, pass A, scan, pass B
      const size_t soff = plan.putv(bqSecs);
      hipStream_t st = c.stream;
      Context* cp = &c;
      const int G = bqG;
      const int64_t K = bqK;
      struct Grp { size_t off; int n, k; BiquadJob* pa; BiquadJob* pb; };
      std::vector<Grp> grps;
      for (int k = 1; k <= kMaxBiquadSections; k++) {
        if (bqScans[k].empty()) continue;
        for (size_t i = 0; i < bqScans[k].size(); i++) bqScans[k][i].m_off = (uint64_t)plan.putv(*bqMats[k][i]);
        const int n = (int)bqScans[k].size();
        // the expanded job tables live in the same blocks as the pieces' states (in chunks that fit a block)
        const int per = (int)std::max<size_t>(1, Context::kBqSplitBlock / sizeof(BiquadJob) / (size_t)G);
        for (int i0 = 0; i0 < n; i0 += per) {
          const int m = std::min(per, n - i0);
          std::vector<BiquadScanJob> part(bqScans[k].begin() + i0, bqScans[k].begin() + i0 + m);
          BiquadJob* pa = (BiquadJob*)c.bqSplitAlloc((size_t)m * (G - 1) * sizeof(BiquadJob) / sizeof(float));
          BiquadJob* pb = (BiquadJob*)c.bqSplitAlloc((size_t)m * G * sizeof(BiquadJob) / sizeof(float));
          grps.push_back(Grp{plan.putv(part), m, k, pa, pb});
        }
      }
      const size_t used0 = bqZeroFrom, used1 = c.bqSplitUsed;
      bqZeroFrom = used1;
      plan.add(LK_OTHER, [=](uint8_t* base) {
        // zero states for pass A: the ranges of the blocks handed out since the previous level (the job tables in them are written next)
        for (size_t b = used0 / Context::kBqSplitBlock; b * Context::kBqSplitBlock < used1; b++) {
          const size_t lo = std::max(used0, b * Context::kBqSplitBlock), hi = std::min(used1, (b + 1) * Context::kBqSplitBlock);
          if (hi > lo) GA_HIP(hipMemsetAsync((char*)cp->bqSplitBlocks[b] + lo % Context::kBqSplitBlock, 0, hi - lo, st));
        }
        for (const Grp& g : grps) launch_biquad_split_expand(st, (const BiquadScanJob*)(base + g.off), g.n, G, K, g.pa, g.pb);
        for (const Grp& g : grps) launch_biquad_lanes(st, g.pa, g.n * (G - 1), (const BiquadSection*)(base + soff), g.k);
        for (const Grp& g : grps) launch_biquad_scan(st, (const BiquadScanJob*)(base + g.off), g.n, G, (const BiquadSection*)(base + soff), base);
        for (const Grp& g : grps) launch_biquad_lanes(st, g.pb, g.n * G, (const BiquadSection*)(base + soff), g.k);
      });
      for (auto& v : bqScans) v.clear();
      for (auto& v : bqMats) v.clear();
      bqG = 0;
      bqK = 0;
    }
    if (!bqDynJobs.empty()) {
      size_t off = plan.putv(bqDynJobs);
      int nj = (int)bqDynJobs.size();
      hipStream_t st = c.stream;
      plan.add(LK_OTHER, [=](uint8_t* base) { launch_biquad_dynamic(st, (const BiquadDynJob*)(base + off), nj); });
    }
    bqDynJobs.clear();
    terms.clear();
    termGains.clear();
    termCurves.clear();
    anyTermGain = anyTermCurve = false;
    mixJobs.clear();
    dmJobs.clear();
    gainJobs.clear();
    for (auto& v : bqJobs) v.clear();
    bqSecs.clear();
    loopJobs.clear();
    rsJobs.clear();
    rsFastJobs.clear();
    gsrJobs.clear();
    streamJobs.clear();
    constJobs.clear();
    oscJobs.clear();
    panJobs.clear();
    delayJobs.clear();
    panDynJobs.clear();
    pmodJobs.clear();
    mixAligned = true;
  }
  struct RsLaunch {
    size_t off;
    int nj;
    int64_t mx;
  };
  std::vector<RsLaunch> rsLaunches;
  size_t trajOffFinal = 0;
};

// ======================================================================================================
// convolver rows <-> groups
// ======================================================================================================
// make sure the group's state arrays cover all rows; new rows start from zero state
void Context::ensureGroupState(ConvGroup& g) {
  Context& c = *this;
  int need = (int)roundup(std::max<size_t>(g.rows.size(), 1), 128);
  if (need <= g.rp) return;
  const int hist = g.P - 1;
  size_t hBytes = (size_t)kBins * std::max(hist, 1) * need * sizeof(float);
  size_t oBytes = (size_t)need * kBlock * sizeof(float);
  float* nr = (float*)c.dalloc(hBytes);
  float* ni = (float*)c.dalloc(hBytes);
  float* o0 = (float*)c.dalloc(oBytes);
  float* o1 = (float*)c.dalloc(oBytes);
  GA_HIP(hipMemsetAsync(nr, 0, hBytes, c.stream));
  GA_HIP(hipMemsetAsync(ni, 0, hBytes, c.stream));
  GA_HIP(hipMemsetAsync(o0, 0, oBytes, c.stream));
  GA_HIP(hipMemsetAsync(o1, 0, oBytes, c.stream));
  if (g.rp > 0) {
    if (hist > 0 && !g.histZero) {
      GA_HIP(hipMemcpy2DAsync(nr, (size_t)need * 4, g.histR, (size_t)g.rp * 4, (size_t)g.rp * 4, (size_t)kBins * hist,
                              hipMemcpyDeviceToDevice, c.stream));
      GA_HIP(hipMemcpy2DAsync(ni, (size_t)need * 4, g.histI, (size_t)g.rp * 4, (size_t)g.rp * 4, (size_t)kBins * hist,
                              hipMemcpyDeviceToDevice, c.stream));
    }
    GA_HIP(hipMemcpyAsync(o0, g.overlap[g.ovCur], (size_t)g.rp * kBlock * 4, hipMemcpyDeviceToDevice, c.stream));
    GA_HIP(hipStreamSynchronize(c.stream));
    size_t oldH = (size_t)kBins * std::max(hist, 1) * g.rp * sizeof(float);
    c.dfree(g.histR, oldH);
    c.dfree(g.histI, oldH);
    c.dfree(g.overlap[0], (size_t)g.rp * kBlock * 4);
    c.dfree(g.overlap[1], (size_t)g.rp * kBlock * 4);
  }
  g.histR = nr;
  g.histI = ni;
  g.overlap[0] = o0;
  g.overlap[1] = o1;
  g.ovCur = 0;
  g.rp = need;
}

// ======================================================================================================
// AudioStreamNodeBase.Process on indices (GraphAudio.IO/AudioStreamSourceNodeBase.cs:132-301)
// ======================================================================================================
void Context::streamReplay(NodeS& s, int64_t nblocks, const std::vector<double>& bt, bool commit) {
  // working copy of the node's state
  std::deque<int> queued = s.stQueued, processed = s.stProcessed;
  int cur = s.stCurrent;
  int64_t pos = s.stPos;
  int lastRate = s.stLastRate;
  int rsChannels = s.stChannels;
  struct Rs { int64_t w[4]; int wseg[4]; double pos; int ready; } rs;
  for (int k = 0; k < 4; k++) {
    rs.w[k] = k;
    rs.wseg[k] = s.stWinValid ? -2 : -1;   // -2: the value the slot holds on the device since the previous chunk
  }
  rs.pos = s.stRsPos;
  rs.ready = s.stRsReady;
  auto clearRs = [&]() {
    for (int k = 0; k < 4; k++) { rs.w[k] = 0; rs.wseg[k] = -1; }
    rs.pos = 0.0;
    rs.ready = 0;
  };
  bool fed = false;
  if (!commit) {
    s.stInfo.assign(nblocks, NodeS::StreamBlockInfo{1, true});
    s.stBlocks.assign(nblocks, StreamBlock{0, 0});
    s.stPieces.clear();
    s.stSegs.clear();
    s.stUploaded = false;
  }
  std::unordered_map<int, int> segOf;   // buffer id -> segment index of this chunk
  auto segment = [&](int bufId) {
    auto it = segOf.find(bufId);
    if (it != segOf.end()) return it->second;
    PlayBuf& b = *buffers[bufId];
    const int idx = (int)segOf.size();
    segOf[bufId] = idx;
    if (!commit) s.stSegs.push_back(StreamSeg{b.dev, b.stride});
    return idx;
  };
  const bool hasTimeline = !s.params[0].events.empty();
  for (int64_t blk = 0; blk < nblocks; blk++) {
    if (s.stState != GA_STREAM_PLAYING) continue;   // ProduceSilence (:136-140)
    if (cur < 0) {
      if (queued.empty()) continue;                  // ProduceSilence (:144-148)
      cur = queued.front();
      queued.pop_front();
      pos = 0;
    }
    const int channelCount = buffers[cur]->channels;
    if (rsChannels != channelCount) {   // `_resamplers is null || Length != channelCount` (:164-171): new, cleared resamplers
      clearRs();
      rsChannels = channelCount;
    }
    // PlaybackRate.GetValues()[0]: k-rate value at the block start (the parameter is computed once per block)
    const float playbackRate = hasTimeline ? param_value_at(s.params[0].events.data(), (int)s.params[0].events.size(), s.params[0].value, bt[blk])
                                           : s.params[0].value;
    int rendered = 0;
    const int piece0 = commit ? 0 : (int)s.stPieces.size();
    while (rendered < kBlock) {
      if (cur < 0) {
        if (queued.empty()) break;
        cur = queued.front();
        queued.pop_front();
        pos = 0;
        if (buffers[cur]->channels != channelCount) {   // :189-198: the buffer goes back to the END of the queue
          queued.push_back(cur);
          cur = -1;
          break;
        }
      }
      PlayBuf& b = *buffers[cur];
      if (b.sampleRate != lastRate && lastRate != 0) clearRs();
      lastRate = b.sampleRate;
      const double effectiveRate = (b.sampleRate / (double)sampleRate) * playbackRate;
      StreamPiece pc{};
      pc.seg = segment(cur);
      pc.next = pos;
      pc.out0 = rendered;
      pc.rate = effectiveRate;
      if (effectiveRate == 1.0) {
        const int remainingInBuffer = (int)b.length - (int)pos;
        const int framesToCopy = std::min(remainingInBuffer, kBlock - rendered);
        pc.copy = 1;
        pc.produced = framesToCopy;
        for (int k = 0; k < 4; k++) { pc.w[k] = 0; pc.wseg[k] = -1; }
        if (!commit && framesToCopy > 0) s.stPieces.push_back(pc);
        pos += framesToCopy;
        rendered += framesToCopy;
        if (pos >= b.length) {
          processed.push_back(cur);
          cur = -1;
          pos = 0;
        }
      } else {
        const int available = (int)b.length - (int)pos;
        if (available <= 0) fail(GA_ERR_UNSUPPORTED, "stream buffer without samples behind the read position");
        for (int k = 0; k < 4; k++) { pc.w[k] = rs.w[k]; pc.wseg[k] = rs.wseg[k]; }
        pc.pos = rs.pos;
        pc.ready = rs.ready;
        // CubicResampler.Process (:26-63) on indices
        int inPos = 0, outPos = 0;
        const int outLen = kBlock - rendered;
        auto feed = [&]() {
          rs.w[0] = rs.w[1]; rs.wseg[0] = rs.wseg[1];
          rs.w[1] = rs.w[2]; rs.wseg[1] = rs.wseg[2];
          rs.w[2] = rs.w[3]; rs.wseg[2] = rs.wseg[3];
          rs.w[3] = pos + inPos; rs.wseg[3] = pc.seg;
          inPos++;
          fed = true;
        };
        while (rs.ready < 4 && inPos < available) {
          feed();
          rs.ready++;
        }
        if (rs.ready == 4) {
          while (outPos < outLen) {
            const int consume = (int)rs.pos;
            if (inPos + consume > available) break;
            for (int i = 0; i < consume; i++) feed();
            rs.pos -= consume;
            outPos++;
            rs.pos += effectiveRate;
          }
        }
        pc.copy = 0;
        pc.produced = outPos;
        if (!commit && outPos > 0) s.stPieces.push_back(pc);
        pos += inPos;
        rendered += outPos;
        if (pos >= b.length - 4) {
          processed.push_back(cur);
          cur = -1;
          pos = 0;
        }
        if (inPos == 0) break;   // minInputConsumed == 0 (:285-292)
      }
    }
    if (!commit) {
      s.stInfo[blk] = NodeS::StreamBlockInfo{channelCount, rendered == 0};
      s.stBlocks[blk] = StreamBlock{piece0, (int)s.stPieces.size() - piece0};
    }
  }
  if (!commit) {
    for (int k = 0; k < 4; k++) {
      s.stWend[k] = rs.w[k];
      s.stWendSeg[k] = rs.wseg[k];
    }
    s.stFed = fed;
    return;
  }
  s.stQueued.swap(queued);
  s.stProcessed.swap(processed);
  s.stCurrent = cur;
  s.stPos = pos;
  s.stLastRate = lastRate;
  s.stChannels = rsChannels;
  s.stRsPos = rs.pos;
  s.stRsReady = rs.ready;
  bool any = false;
  for (int k = 0; k < 4; k++) any = any || rs.wseg[k] != -1;
  if (fed) {   // the device wrote the window at the end of these blocks into the other copy
    s.stWinCur ^= 1;
    s.stWinValid = any;
  } else if (!any) {
    s.stWinValid = false;   // cleared and not fed again
  }
}

// ======================================================================================================
// formulation D (ga_coarse.hip): coarse partitions, consumer sums fused in the frequency domain
// ======================================================================================================
// Which convolver outputs may be summed as spectra?  A node whose single output feeds exactly ONE input (or parameter) of
// one consumer, and is a term of that input in every segment of the chunk.  Every mixing rule of AudioNodeInput.MixBuffer
// (equal counts, 1 -> N, N -> 1 down-mix, min(N, M), AudioNodeInput.cs:182-244) is linear in the term, so the consumer
// may receive the sum of the group as ONE term -- the leader's output -- and nothing from the other members.  What changes
// is only the association of the float32 additions (the reference adds the members one by one in connection order).
void Context::planCoarseFusion(const std::vector<int>& topo, const std::vector<Segment>& segs) {
  // The grouping is a function of the graph, of the convolvers' formulations and impulse responses and of the segments' control
  // state (who is a term of which input): while none of them moved since the previous chunk the leaders stand.
  {
    uint64_t key = hmix(graphVersion, (uint64_t)segs.size());
    for (const Segment& sg : segs) key = hmix(key, sg.hash);
    for (int id : topo) {
      const NodeS& nd = *nodes[id];
      if (nd.type == GA_NODE_CONVOLVER) key = hmix(hmix(key, ((uint64_t)id << 8) | (uint64_t)nd.convPath), (uint64_t)(uintptr_t)nd.ir.get());
    }
    if (fusionKeyValid && key == fusionKey) return;
    fusionKey = key;
    fusionKeyValid = true;
  }
  std::vector<int> cand;
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_CONVOLVER) continue;
    nd.dLeader = -1;
    if (!nd.ir || nd.convPath != 4) continue;
    nd.dLeader = id;
    if (nd.outputs[0].connectedInputs.size() == 1) cand.push_back(id);
  }
  if (cand.size() < 2) return;
  std::unordered_map<int, int> seen;                 // candidate -> segments in which it is a term of its consumer's input
  std::unordered_map<int, std::vector<int>> byConsumer;
  for (int id : cand) {
    seen[id] = 0;
    byConsumer[nodes[id]->outputs[0].connectedInputs[0].node].push_back(id);
  }
  for (const Segment& sg : segs)
    for (const NodeSeg& ns : sg.nodes) {
      if (byConsumer.find(ns.id) == byConsumer.end()) continue;
      auto scan = [&](const InSeg& is, int inputIdx) {
        for (const TermS& t : is.terms) {
          auto it = seen.find(t.node);
          if (it == seen.end()) continue;
          const InRef& r = nodes[t.node]->outputs[0].connectedInputs[0];
          if (r.node == ns.id && r.input == inputIdx && t.out == 0) it->second++;
        }
      };
      for (int i = 0; i < (int)ns.ins.size(); i++) scan(ns.ins[i], i);
      for (int p = 0; p < (int)ns.pins.size(); p++) scan(ns.pins[p], -1 - p);
    }
  std::map<std::tuple<int, int, int, int, int>, int> leaderOf;   // (consumer, input, depth, output channels, partitions) -> leader
  for (int id : cand) {   // topo order: the leader is the first member the traversal reaches
    NodeS& nd = *nodes[id];
    if (seen[id] != (int)segs.size()) continue;
    const InRef& r = nd.outputs[0].connectedInputs[0];
    auto key = std::make_tuple(r.node, r.input, nd.depth, nd.effectiveOutCh, nd.ir->coarseP);
    auto it = leaderOf.find(key);
    if (it == leaderOf.end()) leaderOf.emplace(key, id);
    else nd.dLeader = it->second;
  }
}

// chunk-long view of input channel `c` of a convolver: the segment views when they agree, else a materialised copy
static const float* convChunkInput(Context& c, Exec& ex, const Exec::ConvInRow& ci, int ch) {
  const auto& segs = ex.segs;
  const float* stable = nullptr;
  bool same = true, first = true;
  for (size_t si = 0; si < segs.size(); si++) {
    const float* v = (ci[si].empty() || ch >= (int)ci[si].size()) ? nullptr : ci[si][ch];
    if (first) { stable = v; first = false; } else if (v != stable) same = false;
  }
  if (same) return stable;
  float* slab = getSlab(c);
  for (size_t si = 0; si < segs.size(); si++) {
    const float* v = (ci[si].empty() || ch >= (int)ci[si].size()) ? nullptr : ci[si][ch];
    MixJob mj;
    mj.out = slab;
    mj.term0 = (int)ex.terms.size();
    mj.nterms = v ? 1 : 0;
    mj.f0 = segs[si].b0 * kBlock;
    mj.n = (segs[si].b1 - segs[si].b0) * kBlock;
    if (v) {
      ex.terms.push_back(v);
      ex.noteAlign(v, mj.f0);
    }
    ex.mixJobs.push_back(mj);
  }
  return slab;
}

// the hist_len samples in front of the chunk of input channel `ch`: a span of a PlayableAudioBuffer, the node's own copy, or nothing yet
static const float* coarseHistory(const NodeS& nd, int ch) {
  if (ch < (int)nd.dHistExt.size() && nd.dHistExt[ch].first) return nd.dHistExt[ch].first;
  return nd.dHistZero ? nullptr : nd.dHist[nd.dHistCur] + (size_t)ch * nd.dHistLen;
}
// If the last hist_len samples of this chunk's input are device memory that stays (a PlayableAudioBuffer played zero-copy), the
// next chunk's history is that span and nothing has to be written; otherwise the span is forgotten and the caller copies.
static bool coarseHistoryStays(Context& c, NodeS& nd, int ch, const float* in, int64_t frames) {
  if ((int)nd.dHistExt.size() < nd.bInCh) nd.dHistExt.resize(nd.bInCh, {nullptr, -1});
  const int64_t hl = nd.dHistLen;
  int buf = -1;
  if (c.coarseExtHist && in && frames >= hl) buf = c.persistentBuffer(in + (frames - hl), hl);
  nd.dHistExt[ch] = buf >= 0 ? std::make_pair(in + (frames - hl), buf) : std::make_pair((const float*)nullptr, -1);
  return buf >= 0;
}

// One convolver stage of a chunk in formulation D, planned in five passes (Context::chunkPlanConvolvers calls planCoarseStage).
namespace {
struct CoarseStage {
  Context& c;
  Exec& ex;
  const std::vector<int>& dNodes;
  const int64_t n, frames;
  const int nT;
  static constexpr int kVoicesPerJob = kCoarseJobTerms;   // terms whose products one workgroup accumulates in registers
  struct Piece {   // <= 4 columns of one signal: (impulse-response channel, output channel of the group)
    int frame0, P, xrow, u0;
    IrSpectra* ir;
    int leader;
    int ncol;
    int irCh[16], outCh[16];
  };
  struct GroupInfo {   // a fused group (by leader), or a convolver on its own
    uint64_t sig = 1469598103934665603ull;
    int maxP = 0, nIn = 0, nOut = 0;
    const void* ir0 = nullptr;
    bool oneIr = true, tail = false, carried = false, fresh = true, noHist = false;
    // time-domain pre-mix (option "coarse_premix"): every member convolves with the same spectra and has the same channel
    // layout, so the group's inputs are added up in front of ONE set of transforms
    bool uniform = true, premix = false;
    int members = 0, nxr = 0, bInCh0 = 0, bSlots0 = 0;
    bool ts0 = false;
    int64_t hl0 = 0;
    struct Terms { std::vector<PremixTerm> in[32], hist[32]; };   // per input channel of the group
    std::unique_ptr<Terms> terms;                                 // (pre-mixed groups only)
  };
  std::vector<Views> chInOf;   // [position in dNodes]: chunk-long views of the node's input channels
  bool wideStrided = true;   // every 16-column term's spectra are h[0] + c x P x kCoarseBins
  std::vector<CoarseHandOver> fwdHandOver;   // a pending hand-over that rides in the first forward launch of this stage
  std::vector<PremixJob> pmJobs;
  std::vector<PremixTerm> pmTerms;
  size_t pmUsed = 0;       // bytes of the pre-mix arena handed out
  int64_t pmMaxN = 0;
  double pmBytes = 0;
  std::vector<CoarseXRow> xrows;
  std::vector<CoarseHistJob> hjobs;
  std::vector<Piece> pieces;
  std::map<int, GroupInfo> groups;   // by leader
  const bool tails;
  int frameNext = 0;
  int64_t maxHist = 0;
  double histBytes = 0;
  std::map<int, double> carryBytes;   // row -> bytes of next-chunk history its forward transform also writes
  int nxAll = 0, G = 1;
  int gBegin[9] = {};
  std::vector<CoarseTerm> terms;
  // launches: by column count (1, 2, 4) x (terms with their own impulse responses | one impulse response for all terms), and
  // by the group whose transforms complete the job's inputs.  Class index = 2 * column class + shared.
  static constexpr int kCwOf[4] = {1, 2, 4, 16};   // column classes of the multiply-accumulate launches
  std::vector<CoarseJob> jobs[8][8];
  int maxT[8] = {0, 0, 0, 0, 0, 0, 0, 0}, maxP[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int pbOf[8] = {4, 4, 4, 4, 4, 4, 4, 4};   // largest of 4, 2, 1 dividing every job's partition count (the sweep's register block)
  double macBytes[8][8] = {}, macFlops[8][8] = {};
  double pmFlops = 0, invFlops = 0;
  std::map<std::pair<int, int>, std::vector<int>> outRows;   // (leader, channel) -> Y rows to sum
  int yNext = 0;
  std::vector<CoarseOut> outs;
  std::vector<int> ylist;
  double invBytes = 0;
  int yFrames = 0, invBlocks = 0;

  CoarseStage(Context& c_, Exec& ex_, const std::vector<int>& d, int64_t n_)
      : c(c_), ex(ex_), dNodes(d), n(n_), frames(n_ * kBlock), nT((int)((n_ * kBlock + kCoarseBlock - 1) / kCoarseBlock)), tails(c_.coarseTail) {}
  void resolveInputs();    // chunk-long input views of every node (materialised where the segments disagree)
  void classifyGroups();   // which groups are pre-mixed, carry a tail, can use the one the previous chunk left
  void buildRows();        // signals to transform (pre-mixed groups: one per channel), history hand-over, pieces (signal x columns)
  void addPieces(NodeS& nd, int leader, int nxr, const int* xFrame, const int* xIndex);
  void buildJobs();        // multiply-accumulate jobs and their terms
  void buildOutputs();     // inverse-transform outputs, tail buffers
  void enqueue();          // tables into the plan, launches
};

void CoarseStage::resolveInputs() {
  chInOf.resize(dNodes.size());
  for (size_t di = 0; di < dNodes.size(); di++) {
    const int id = dNodes[di];
    NodeS& nd = *c.nodes[id];
    const int64_t hl = nd.dHistLen;
    const Exec::ConvInRow ci = ex.convIn[id];
    Views& chIn = chInOf[di];
    chIn.assign(nd.bInCh, nullptr);
    auto viewOf = [&](size_t si, int ch) { return (ci[si].empty() || ch >= (int)ci[si].size()) ? (const float*)nullptr : ci[si][ch]; };
    for (int ch = 0; ch < nd.bInCh; ch++) {
      // a channel that shows the same view as an earlier one in every segment IS that channel (a mono signal copied to all
      // channels of an explicit input, AudioNodeInput.cs:201-213): one chunk-long view -- one materialised copy -- serves both
      int same = -1;
      for (int e = 0; e < ch && same < 0; e++) {
        bool eq = true;
        for (size_t si = 0; si < ex.segs.size() && eq; si++) eq = viewOf(si, e) == viewOf(si, ch);
        if (eq) same = e;
      }
      chIn[ch] = same >= 0 ? chIn[same] : convChunkInput(c, ex, ci, ch);
    }
    bool allSame = true;
    for (int ch = 1; ch < nd.bInCh; ch++) allSame = allSame && (chIn[ch] == chIn[0]);
    if (nd.bShared && !allSame) {   // the channels start to differ: every channel inherits the (so far common) history
      // (a plan entry like every other device action of the chunk: ordered with the chunk's launches, nothing is issued at plan time)
      if (const float* h0 = coarseHistory(nd, 0))
        for (int ch = 1; ch < nd.bInCh; ch++) {
          float* dst = nd.dHist[nd.dHistCur] + (size_t)ch * hl;
          hipStream_t st = c.stream;
          ex.plan.add(LK_OTHER, [=](uint8_t*) { GA_HIP(hipMemcpyAsync(dst, h0, (size_t)hl * sizeof(float), hipMemcpyDeviceToDevice, st)); });
          if (ch < (int)nd.dHistExt.size()) nd.dHistExt[ch] = {nullptr, -1};
        }
      if (nd.dHistZero && coarseHistory(nd, 0)) nd.dHistZero = false;   // (the copies above are the channels' histories now)
      nd.bShared = false;
    }
  }
}

void CoarseStage::classifyGroups() {
  // ---- carried tails (option "coarse_tail"): every output of the stage keeps, from chunk to chunk, what the input so far adds to
  // the samples behind the chunk's end.  While a group of fused convolvers is the same as in the previous chunk its members need
  // no input history in front of the chunk: their windows start at the chunk (u = 0) and the previous chunk's tail is added to
  // the output instead -- P' - 1 fewer transforms per signal and chunk.  Any change (member set, impulse responses, channel
  // modes, a chunk in between that did not run this stage) falls back to the input histories, which are kept up to date either way.
  {
    auto mix = [](uint64_t& h, uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    for (int id : dNodes) {
      NodeS& nd = *c.nodes[id];
      GroupInfo& g = groups[nd.dLeader >= 0 ? nd.dLeader : id];
      mix(g.sig, (uint64_t)id);
      mix(g.sig, (uint64_t)(uintptr_t)nd.ir.get());
      mix(g.sig, (uint64_t)nd.bInCh | ((uint64_t)nd.bSlots << 8) | ((uint64_t)nd.isTrueStereo << 16) | ((uint64_t)nd.bShared << 17) |
                     ((uint64_t)nd.ir->coarseP << 24));
      g.maxP = std::max(g.maxP, nd.ir->coarseP);
      g.fresh = g.fresh && nd.dHistZero;   // no member has seen input yet: nothing in front of the chunk either
      g.nIn += nd.bShared ? 1 : nd.bInCh;
      if (!g.ir0) g.ir0 = nd.ir.get();
      g.oneIr = g.oneIr && g.ir0 == nd.ir.get();
      g.nOut = std::max(g.nOut, nd.isTrueStereo ? 2 : nd.bSlots);
      const int nxr = nd.bShared ? 1 : nd.bInCh;
      if (g.members++ == 0) {
        g.nxr = nxr;
        g.bInCh0 = nd.bInCh;
        g.bSlots0 = nd.bSlots;
        g.ts0 = nd.isTrueStereo;
        g.hl0 = nd.dHistLen;
      } else {
        g.uniform = g.uniform && g.nxr == nxr && g.bInCh0 == nd.bInCh && g.bSlots0 == nd.bSlots && g.ts0 == nd.isTrueStereo && g.hl0 == nd.dHistLen;
      }
    }
    for (auto& kv : groups) {
      NodeS& ld = *c.nodes[kv.first];
      // a tail costs P' more inverse transforms per output channel and chunk and saves P' - 2 forward transforms per input row:
      // worth it for sums of many signals, not for a convolver on its own.  The P' more output blocks are nearly free in the
      // reduction kernel (one impulse response for the whole group); the general kernel skips the partition blocks whose windows
      // lie behind the chunk (all zero), so a group of private impulse responses multiplies exactly the products it would have
      // multiplied with the histories in front -- but measured it does not pay (ga_engine.hpp): option "coarse_tail_private", off
      kv.second.premix = c.coarsePremix && kv.second.members >= 2 && kv.second.oneIr && kv.second.uniform && kv.second.nxr <= 32;
      // (a pre-mixed group always keeps its tail: reading every member's history again would cost members x (P' - 1) blocks)
      kv.second.tail = tails && (kv.second.oneIr || c.coarseTailPrivate) &&
                       (kv.second.premix || (int64_t)kv.second.nIn * (kv.second.maxP - 2) >= (int64_t)kv.second.nOut * kv.second.maxP);
      kv.second.carried = kv.second.tail && ld.dTail[0] && ld.dTailSeq + 1 == c.chunkSeq && ld.dTailSig == kv.second.sig &&
                          ld.dTailLen == (int64_t)(kv.second.maxP + 1) * kCoarseBlock;
      kv.second.noHist = kv.second.carried || (kv.second.tail && kv.second.fresh);
    }
  }
}

// columns of a node's input rows: discrete -> slot c reads input c, IR channel c, output c ; true stereo -> (L,h0,outL) (L,h1,outR)
// (R,h2,outL) (R,h3,outR)  (ConvolverNode.cs:127-151).  Pieces of 4, 2, 1 columns per row.
void CoarseStage::addPieces(NodeS& nd, int leader, int nxr, const int* xFrame, const int* xIndex) {
  IrSpectra& ir = *nd.ir;
  for (int xc = 0; xc < nxr; xc++) {
    int cols[32][2], ncols = 0;
    for (int slot = 0; slot < nd.bSlots; slot++) {
      const int inc = nd.isTrueStereo ? (slot >> 1) : slot;
      if (!(nd.bShared || inc == xc)) continue;
      cols[ncols][0] = slot;                               // slot index == IR channel index in both modes
      cols[ncols][1] = nd.isTrueStereo ? (slot & 1) : slot;
      ncols++;
    }
    // 16 columns of one signal at once where the group's terms have impulse responses of their own (the general kernel's 16-column
    // instance: the signal's frames are staged once for all of them) and its double-buffered spectra fit the LDS (P' <= 4)
    const GroupInfo& gi = groups[leader];
    const bool wide = !gi.oneIr && ir.coarseP <= 4 && c.coarseWide;
    for (int c0 = 0; c0 < ncols;) {
      const int left = ncols - c0;
      const int w = (wide && left >= 16) ? 16 : (left >= 4 ? 4 : (left >= 2 ? 2 : 1));
      Piece pc{};
      pc.frame0 = xFrame[xc];
      pc.xrow = xIndex[xc];
      pc.u0 = xrows[xIndex[xc]].u0;
      pc.P = ir.coarseP;
      pc.ir = &ir;
      pc.leader = leader;
      pc.ncol = w;
      for (int j = 0; j < w; j++) {
        pc.irCh[j] = cols[c0 + j][0];
        pc.outCh[j] = cols[c0 + j][1];
      }
      pieces.push_back(pc);
      c0 += w;
    }
  }
}

void CoarseStage::buildRows() {
  for (size_t di = 0; di < dNodes.size(); di++) {
    const int id = dNodes[di];
    NodeS& nd = *c.nodes[id];
    IrSpectra& ir = *nd.ir;
    const int P = ir.coarseP;
    if (P < 1 || P > kCoarseMaxP) fail(GA_ERR_INVALID_OPERATION, "internal: coarse partition count out of range");
    const int64_t hl = nd.dHistLen;
    GroupInfo& gi0 = groups[nd.dLeader >= 0 ? nd.dLeader : id];
    const bool carried = gi0.noHist;   // no windows in front of the chunk
    const Views& chIn = chInOf[di];
    const int nxr = nd.bShared ? 1 : nd.bInCh;
    c.stats.mac_flops_total += 8.0 * ir.P * kBins * (double)nd.bSlots * (double)n;
    c.stats.mac_bytes_total += ((double)ir.P * kBins * 8.0 + kBins * 8.0 + 512.0) * nd.bSlots * (double)n;
    if (gi0.premix) {
      // a member of a pre-mixed group: nothing to transform for it; its samples join the group's sum and its own history of the
      // next chunk is written on the way (or by a copy job where that is not possible)
      for (int ch = 0; ch < nxr; ch++) {
        const float* oldHist = coarseHistory(nd, ch);
        float* nextHist = nd.dHist[nd.dHistCur ^ 1] + (size_t)ch * hl;
        PremixTerm t{chIn[ch], nullptr};
        if (coarseHistoryStays(c, nd, ch, chIn[ch], frames)) {
          // (the next chunk's history is a span of the member's sample buffer)
        } else if (c.coarseCarry && chIn[ch] && frames >= hl && ((uintptr_t)nextHist & 15) == 0) {
          t.carry = nextHist;
          pmBytes += (double)hl * 4.0;
        } else {
          hjobs.push_back(CoarseHistJob{oldHist, chIn[ch], nextHist, hl, frames});
          maxHist = std::max(maxHist, hl);
          histBytes += 2.0 * (double)hl * 4.0;
        }
        if (!gi0.terms) gi0.terms = std::make_unique<GroupInfo::Terms>();
        gi0.terms->in[ch].push_back(t);
        if (!carried) gi0.terms->hist[ch].push_back(PremixTerm{oldHist, nullptr});
      }
      nd.dHistCur ^= 1;
      nd.dHistZero = false;
      continue;
    }
    int xFrame[32], xIndex[32];
    for (int ch = 0; ch < nxr; ch++) {
      CoarseXRow r;
      const float* oldHist = coarseHistory(nd, ch);   // (kept up to date in every mode)
      r.hist = carried ? nullptr : oldHist;
      r.in = chIn[ch];
      r.nvalid = frames;
      r.frame0 = frameNext;
      // windows u0 .. u_last: with carried tails the last one is u = nT ([last block | nothing yet]: it feeds the outputs behind
      // the chunk's end), and a group that continues needs none in front of the chunk
      r.u0 = carried ? 0 : -(P - 1);
      r.n_frames = (gi0.tail ? nT + 1 : nT) - r.u0;
      r.hist_len = (int)hl;
      r.flags = 0;
      r.scale = 1.0f;
      xFrame[ch] = frameNext;
      xIndex[ch] = (int)xrows.size();
      frameNext += r.n_frames;
      // the next chunk's history: the last hl samples of [history | input].  When they all come from this chunk's input the
      // forward kernel writes them while it holds the samples (ga_kernels.hpp, CoarseXRow::carry); otherwise a copy job.
      float* nextHist = nd.dHist[nd.dHistCur ^ 1] + (size_t)ch * hl;
      r.carry = nullptr;
      r.carry_from = 0;
      if (coarseHistoryStays(c, nd, ch, r.in, frames)) {
        // (the next chunk's history is a span of the sample buffer the input aliases)
      } else if (c.coarseCarry && r.in && frames >= hl && (((uintptr_t)r.in | (uintptr_t)nextHist) & 15) == 0) {
        r.carry = nextHist;
        r.carry_from = frames - hl;
        carryBytes[(int)xrows.size()] = (double)hl * 4.0;
      } else {
        hjobs.push_back(CoarseHistJob{oldHist, r.in, nextHist, hl, frames});
        maxHist = std::max(maxHist, hl);
        histBytes += 2.0 * (double)hl * 4.0;
      }
      xrows.push_back(r);
    }
    nd.dHistCur ^= 1;
    nd.dHistZero = false;
    addPieces(nd, nd.dLeader >= 0 ? nd.dLeader : id, nxr, xFrame, xIndex);
  }
  // ---- pre-mixed groups: one mixed signal [history | chunk] per input channel, transformed like a single convolver's input ----
  for (auto& kv : groups) {
    GroupInfo& g = kv.second;
    if (!g.premix) continue;
    NodeS& ld = *c.nodes[kv.first];
    const int64_t hl = g.hl0;
    int xFrame[32], xIndex[32];
    for (int ch = 0; ch < g.nxr; ch++) {
      const size_t bytes = (size_t)(hl + frames) * sizeof(float);
      if (pmUsed + bytes > c.coarseM.bytes) fail(GA_ERR_INVALID_OPERATION, "internal: the pre-mix arena is too small for the plan");
      float* mixed = (float*)((char*)c.coarseM.p + pmUsed);
      pmUsed += (bytes + 255) & ~(size_t)255;
      auto job = [&](float* out, const std::vector<PremixTerm>& tv, int64_t len, int64_t carryFrom) {
        PremixJob j{out, (int)pmTerms.size(), 0, len, carryFrom, 1, 0};
        for (const PremixTerm& t : tv) {
          if (!t.in) continue;   // (silent: adds nothing, and has no carry)
          pmTerms.push_back(t);
          j.nterms++;
          if ((uintptr_t)t.in & 15) j.flags &= ~1;
          if (t.carry) j.flags |= 2;
          pmBytes += (double)len * 4.0;
          pmFlops += (double)len * 4.0;   // (compensated summation: four operations per sample)
        }
        pmJobs.push_back(j);
        pmMaxN = std::max(pmMaxN, len);
        pmBytes += (double)len * 4.0;
      };
      if (!g.terms) g.terms = std::make_unique<GroupInfo::Terms>();
      if (!g.noHist) job(mixed, g.terms->hist[ch], hl, hl);
      job(mixed + hl, g.terms->in[ch], frames, std::max<int64_t>(0, frames - hl));
      CoarseXRow r{};
      r.hist = g.noHist ? nullptr : mixed;
      r.in = mixed + hl;
      r.nvalid = frames;
      r.frame0 = frameNext;
      r.u0 = g.noHist ? 0 : -(g.maxP - 1);
      r.n_frames = (g.tail ? nT + 1 : nT) - r.u0;
      r.hist_len = (int)hl;
      r.flags = 0;
      r.scale = 1.0f;
      r.carry = nullptr;
      r.carry_from = 0;
      xFrame[ch] = frameNext;
      xIndex[ch] = (int)xrows.size();
      frameNext += r.n_frames;
      xrows.push_back(r);
    }
    addPieces(ld, kv.first, g.nxr, xFrame, xIndex);
    c.stats.coarse_premixed_signals += (int64_t)g.members * g.nxr;
  }
  // the previous chunk's bus on its way to the caller's page-locked rows (Context::pendingHandOver): one-term jobs at the head of
  // this launch -- their workgroups write over PCIe while the others stream the members' samples from HBM
  if (!pmJobs.empty() && !c.pendingHandOver.empty()) {
    std::vector<PremixJob> head;
    for (const Context::HandOver& h : c.pendingHandOver) {
      head.push_back(PremixJob{h.dst_dev, (int)pmTerms.size(), 1, h.n, h.n, 1 | 4, 0});
      pmTerms.push_back(PremixTerm{h.src, nullptr});
      pmMaxN = std::max(pmMaxN, h.n);
      pmBytes += 2.0 * (double)h.n * 4.0;
    }
    pmJobs.insert(pmJobs.begin(), head.begin(), head.end());
    c.pendingHandOver.clear();
    c.stats.deferred_handovers++;
  } else if (!c.pendingHandOver.empty() && !xrows.empty()) {   // no pre-mix launch: they ride in the stage's first forward launch
    for (const Context::HandOver& h : c.pendingHandOver) fwdHandOver.push_back(CoarseHandOver{h.src, h.dst_dev, h.n});
    c.pendingHandOver.clear();
    c.stats.deferred_handovers++;
  }
  ex.flushLevel();   // (materialised inputs)
}

void CoarseStage::buildJobs() {
  // ---- jobs: pieces with the same (leader, output channels, partitions) accumulate into the same Y rows ----
  struct Key {
    int leader, P, ncol;
    std::array<int, 16> out;
    bool operator<(const Key& o) const { return std::tie(leader, P, ncol, out) < std::tie(o.leader, o.P, o.ncol, o.out); }
  };
  std::map<Key, std::vector<const Piece*>> byKey;
  for (const Piece& pc : pieces) {
    Key k{pc.leader, pc.P, pc.ncol, {}};
    k.out.fill(-1);
    for (int j = 0; j < pc.ncol; j++) k.out[j] = pc.outCh[j];
    byKey[k].push_back(&pc);
  }
  // groups of signals: the multiply-accumulate jobs of group g run (second stream) while group g + 1 is transformed
  nxAll = (int)xrows.size();
  G = !c.coarseOverlap ? 1 : (nxAll >= 256 ? 4 : (nxAll >= 64 ? 2 : 1));
  auto groupOf = [&](int xrow) { return std::min(G - 1, (int)((int64_t)xrow * G / std::max(nxAll, 1))); };
  for (int g = 0; g <= G; g++) gBegin[g] = 0;
  for (int x = 0; x < nxAll; x++) gBegin[groupOf(x) + 1] = x + 1;
  for (int g = 1; g <= G; g++) gBegin[g] = std::max(gBegin[g], gBegin[g - 1]);
  for (auto& kv : byKey) {
    const Key& k = kv.first;
    const int cw = k.ncol, ci = cw == 1 ? 0 : (cw == 2 ? 1 : (cw == 4 ? 2 : 3));
    const auto& pv = kv.second;
    for (size_t p0 = 0; p0 < pv.size(); p0 += kVoicesPerJob) {
      const size_t p1 = std::min(pv.size(), p0 + kVoicesPerJob);
      const int term0 = (int)terms.size();
      bool shared = true;
      int lastX = 0;
      for (size_t i = p0; i < p1; i++) {
        const Piece& pc = *pv[i];
        CoarseTerm t{};
        t.frame0 = pc.frame0 - (pc.u0 + (pc.P - 1));   // frame the window u = -(P - 1) would have (the kernels index from there)
        for (int j = 0; j < 16; j++) t.h[j] = nullptr;
        for (int j = 0; j < cw; j++) t.h[j] = pc.ir->coarse + (size_t)pc.irCh[j] * pc.P * kCoarseBins;
        if (i > p0)
          for (int j = 0; j < cw; j++) shared = shared && (t.h[j] == terms[term0].h[j]);
        terms.push_back(t);
        lastX = std::max(lastX, pc.xrow);
      }
      if (cw == 16) {
        shared = false;   // (the reduction kernel has no 16-column instance)
        for (size_t i = term0; i < terms.size(); i++)   // the matrix-core kernel addresses the columns' spectra from h[0]
          for (int j = 1; j < 16; j++) wideStrided = wideStrided && terms[i].h[j] == terms[i].h[0] + (size_t)j * k.P * kCoarseBins;
      }
      const int grp = groupOf(lastX);
      const int yrow0 = yNext;
      yNext += cw;
      for (int j = 0; j < cw; j++) outRows[{k.leader, k.out[j]}].push_back(yrow0 + j);
      const int jb = shared ? kCoarseSumJobBlocks(cw) : kCoarseJobBlocks(cw);   // (the two kernels' job sizes)
      const GroupInfo& gi = groups[k.leader];
      const int nTo = gi.tail ? nT + gi.maxP : nT;   // output blocks: the chunk's, and with a tail those the chunk's input still reaches
      for (int t0 = 0; t0 < nTo; t0 += jb) {
        CoarseJob jb_{};
        jb_.term0 = term0;
        jb_.n_terms = (int)(p1 - p0);
        jb_.P = k.P;
        jb_.t0 = t0;
        jb_.n_t = std::min(jb, nTo - t0);
        jb_.yrow0 = yrow0;
        jb_.shared_h = shared ? 1 : 0;
        jb_.u_lo = gi.noHist ? 0 : -(k.P - 1);
        jb_.u_hi = gi.tail ? nT : nT - 1;
        const int cj = 2 * ci + (shared ? 1 : 0);
        jobs[cj][grp].push_back(jb_);
        maxT[cj] = std::max(maxT[cj], jb_.n_t);
        maxP[cj] = std::max(maxP[cj], k.P);
        while (k.P % pbOf[cj]) pbOf[cj] >>= 1;
        const int fread = std::max(0, std::min(jb_.u_hi, t0 + jb_.n_t - 1) - std::max(jb_.u_lo, t0 - (k.P - 1)) + 1);   // frames that exist
        // (the spectra of the terms are necessary bytes ONCE: the block ranges of one group of terms are neighbours in the grid and
        // the second range finds them in the L2 -- PMC: profiles/r03_config5_pmc_hbm_traffic.json)
        macBytes[cj][grp] += (double)jb_.n_terms * fread * kCoarseBins * 8.0 +
                             (t0 == 0 ? (double)(shared ? 1 : jb_.n_terms) * k.P * cw * kCoarseBins * 8.0 : 0.0) + (double)cw * jb_.n_t * kCoarseBins * 8.0;
        // complex multiply-adds (8 flops): every term's products in the general kernel; in the reduction the terms' frames are
        // added up first (2 flops per complex value) and the sum is multiplied once
        macFlops[cj][grp] += shared ? ((double)jb_.n_terms * fread * 2.0 + (double)k.P * jb_.n_t * cw * 8.0) * kCoarseBins
                                    : (double)jb_.n_terms * k.P * jb_.n_t * cw * 8.0 * kCoarseBins;
      }
    }
  }
}

void CoarseStage::buildOutputs() {
  int maxPAll = 0;
  for (auto& kv : groups)
    if (kv.second.tail) maxPAll = std::max(maxPAll, kv.second.maxP);
  yFrames = nT + maxPAll;   // coarse blocks per Y row (rows of groups with a shorter or no tail leave their end unused)
  invBlocks = nT;
  if (tails) {   // tail buffers live with the group's leader: [2][channels][tail_len], read one, write the other
    std::map<int, int> chOf;
    for (auto& kv : outRows) chOf[kv.first.first] = std::max(chOf[kv.first.first], kv.first.second + 1);
    for (auto& kv : groups) {
      if (!kv.second.tail) continue;
      NodeS& ld = *c.nodes[kv.first];
      const int64_t len = (int64_t)(kv.second.maxP + 1) * kCoarseBlock;
      const int nch = chOf.count(kv.first) ? chOf[kv.first] : 0;
      if (nch == 0) continue;
      if (!ld.dTail[0] || ld.dTailLen != len || ld.dTailCh != nch) {
        if (kv.second.carried) fail(GA_ERR_INVALID_OPERATION, "internal: a carried tail changed its shape");
        GA_HIP(hipStreamSynchronize(c.stream));
        for (int b = 0; b < 2; b++) {
          if (ld.dTail[b]) c.dfree(ld.dTail[b], (size_t)ld.dTailLen * ld.dTailCh * sizeof(float));
          ld.dTail[b] = (float*)c.dalloc((size_t)len * nch * sizeof(float));
        }
        ld.dTailLen = len;
        ld.dTailCh = nch;
        ld.dTailCur = 0;
      }
    }
  }
  for (auto& kv : outRows) {
    CoarseOut o{};
    o.out = ex.nodeOut(kv.first.first, kv.first.second);
    o.nvalid = frames;
    o.y0 = (int)ylist.size();
    o.ny = (int)kv.second.size();
    o.n_y = nT;
    if (groups[kv.first.first].tail) {
      NodeS& ld = *c.nodes[kv.first.first];
      const GroupInfo& gi = groups[kv.first.first];
      o.n_y = nT + gi.maxP;
      o.tail_len = ld.dTailLen;
      o.tail_in = gi.carried ? ld.dTail[ld.dTailCur] + (size_t)kv.first.second * ld.dTailLen : nullptr;
      o.tail_out = ld.dTail[ld.dTailCur ^ 1] + (size_t)kv.first.second * ld.dTailLen;
      invBlocks = std::max(invBlocks, (int)((frames + o.tail_len + kCoarseBlock - 1) / kCoarseBlock));
      invBytes += (double)o.tail_len * 4.0 * (gi.carried ? 2.0 : 1.0);
      if (gi.carried) c.stats.coarse_carried_outputs++;
    }
    ylist.insert(ylist.end(), kv.second.begin(), kv.second.end());
    outs.push_back(o);
    invBytes += (double)o.ny * o.n_y * kCoarseBins * 8.0 + (double)frames * 4.0;
    invFlops += (double)o.n_y * (kCoarseTransformFlops + (double)o.ny * 2.0 * kCoarseBins);
  }
  if (tails)
    for (auto& kv : groups) {   // this chunk's tails are the next chunk's, if the group is still the same then
      NodeS& ld = *c.nodes[kv.first];
      if (!kv.second.tail || !ld.dTail[0]) continue;
      ld.dTailCur ^= 1;
      ld.dTailSig = kv.second.sig;
      ld.dTailSeq = c.chunkSeq;
    }
  if ((size_t)frameNext * kCoarseBins * sizeof(float2) > c.coarseX.bytes || (size_t)yNext * yFrames * kCoarseBins * sizeof(float2) > c.coarseY.bytes)
    fail(GA_ERR_INVALID_OPERATION, "internal: coarse spectra arenas are too small for the plan");

}

void CoarseStage::enqueue() {
  // (the launches below run after this object is gone: they capture locals, never members)
  const int G = this->G, yFrames = this->yFrames, invBlocks = this->invBlocks, nxAll = this->nxAll;
  const double invBytes = this->invBytes, histBytes = this->histBytes;
  const size_t xo = ex.plan.putv(xrows), ho = ex.plan.putv(hjobs), to = ex.plan.putv(terms), oo = ex.plan.putv(outs), yo = ex.plan.putv(ylist);
  const size_t pjo = ex.plan.putv(pmJobs), pto = ex.plan.putv(pmTerms), fho = ex.plan.putv(fwdHandOver);
  const int nfh = (int)fwdHandOver.size();
  const int npm = (int)pmJobs.size();
  const int64_t pmMaxN = this->pmMaxN;
  const double pmBytes = this->pmBytes, pmFlops = this->pmFlops, invFlops = this->invFlops;
  struct MacLaunch { size_t off; int nj, cw, mt, mp, pb, grp; bool ap; double bytes, flops; };
  const bool matrixCores = c.coarseMfma && wideStrided;
  std::vector<MacLaunch> macs;
  for (int g = 0; g < G; g++)
    for (int i = 0; i < 8; i++) {
      if (jobs[i][g].empty()) continue;
      macs.push_back(MacLaunch{ex.plan.putv(jobs[i][g]), (int)jobs[i][g].size(), kCwOf[i >> 1], maxT[i], maxP[i], pbOf[i], g, (i & 1) == 0,
                               macBytes[i][g], macFlops[i][g]});
      c.stats.mac_launches += 1;
    }
  hipStream_t st = c.stream;
  float2* X = (float2*)c.coarseX.p;
  float2* Y = (float2*)c.coarseY.p;
  const float2* tw16 = c.twiddles16(4096);
  const float2* twFwd = c.twiddles16pw();
  const float2* twab = c.coarseTwab();
  const int nh = (int)hjobs.size(), no = (int)outs.size();
  // per group: rows, longest row, bytes, windows per workgroup (long runs fetch every input sample once; keep >= ~4
  // workgroups per CU's worth of parallelism)
  struct FwdLaunch { int x0, nx, maxFrames, run; double bytes, flops; };
  std::vector<FwdLaunch> fwds;
  for (int g = 0; g < G; g++) {
    FwdLaunch f{this->gBegin[g], this->gBegin[g + 1] - this->gBegin[g], 0, 1, 0.0, 0.0};
    for (int x = f.x0; x < f.x0 + f.nx; x++) {
      f.maxFrames = std::max(f.maxFrames, xrows[x].n_frames);
      f.bytes += (double)(xrows[x].n_frames + 1) * kCoarseBlock * 4.0 + (double)xrows[x].n_frames * kCoarseBins * 8.0;
      f.flops += (double)xrows[x].n_frames * kCoarseTransformFlops;
      if (auto it = carryBytes.find(x); it != carryBytes.end()) f.bytes += it->second;
    }
    while (f.run < 16 && (int64_t)nxAll * ((f.maxFrames + 2 * f.run - 1) / (2 * f.run)) >= 1024) f.run *= 2;
    if (f.run >= 8) {   // equal runs of about 12 windows (measured on config 3: 10, 12 and 20 per run beat 16 + a short last run by 2.5 %)
      const int k = (f.maxFrames + 11) / 12;
      f.run = (f.maxFrames + k - 1) / k;
    }
    if (const char* e = expenv("GA_COARSE_RUN")) f.run = std::max(1, atoi(e));   // measurements only
    fwds.push_back(f);
  }
  if (G > 1) c.ensureOverlapStream();
  Context* cp = &c;
  ex.plan.add(GA_STAGE_COARSE_SECTION, [=](uint8_t* base) {
    // one piece of the section: a launch with its own profile events (the two stages overlap on two streams)
    auto timed = [&](hipStream_t sx, int kind, double bytes, double flops, const std::function<const char*()>& launch) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (cp->profileNow) {
        GA_HIP(hipEventCreate(&e0));
        GA_HIP(hipEventCreate(&e1));
        GA_HIP(hipEventRecord(e0, sx));
      }
      cp->noteKernel(kind, launch());
      if (cp->profileNow) {
        GA_HIP(hipEventRecord(e1, sx));
        cp->extraProf.push_back(Context::ExtraProf{e0, e1, kind, bytes});
      }
      cp->stats.kernel_launches++;
      cp->stats.stage_launches[kind]++;
      cp->stats.stage_bytes[kind] += bytes;
      cp->stats.stage_flops[kind] += flops;
    };
    hipStream_t s2 = G > 1 ? cp->stream2 : st;
    if (npm > 0)
      timed(st, LK_CPREMIX, pmBytes, pmFlops, [&] { return launch_coarse_premix(st, (const PremixJob*)(base + pjo), npm, (const PremixTerm*)(base + pto), pmMaxN); });
    for (int g = 0; g < G; g++) {
      const FwdLaunch& f = fwds[g];
      if (f.nx > 0)
        timed(st, LK_CFWD, f.bytes, f.flops, [&] {
          return launch_coarse_fwd(st, (const CoarseXRow*)(base + xo) + f.x0, f.nx, f.maxFrames, f.run, X, twFwd, twab,
                                   (const CoarseHandOver*)(base + fho), g == 0 ? nfh : 0);
        });
      if (G > 1) {
        GA_HIP(hipEventRecord(cp->dGroupEv[g], st));
        GA_HIP(hipStreamWaitEvent(s2, cp->dGroupEv[g], 0));
      }
      for (const MacLaunch& m : macs)
        if (m.grp == g)
          timed(s2, LK_CMAC, m.bytes, m.flops, [&] {
            return launch_coarse_mac(s2, (const CoarseJob*)(base + m.off), m.nj, (const CoarseTerm*)(base + to), X, Y, yFrames, m.cw, m.mt, m.mp, m.ap, m.pb,
                                     matrixCores);
          });
    }
    if (G > 1) {   // join: the inverse transforms (and the next chunk's forward transforms, which reuse X) wait for every job
      GA_HIP(hipEventRecord(cp->dJoinEv, s2));
      GA_HIP(hipStreamWaitEvent(st, cp->dJoinEv, 0));
    }
  });
  ex.plan.add(LK_CINV, [=](uint8_t* base) {
    cp->noteKernel(LK_CINV, launch_coarse_inv(st, (const CoarseOut*)(base + oo), no, invBlocks, (const int*)(base + yo), Y, yFrames, tw16, twab));
  }, invBytes, invFlops);
  const int64_t mh = maxHist;
  if (nh > 0) ex.plan.add(LK_CHIST, [=](uint8_t* base) { launch_coarse_hist(st, (const CoarseHistJob*)(base + ho), nh, mh); }, histBytes);
}
}  // namespace

static void planCoarseStage(Context& c, Exec& ex, const std::vector<int>& dNodes, int64_t n) {
  CoarseStage s(c, ex, dNodes, n);
  s.resolveInputs();
  s.classifyGroups();
  s.buildRows();
  s.buildJobs();
  s.buildOutputs();
  s.enqueue();
}

// ======================================================================================================
// runChunk
// ======================================================================================================
static const bool timing = getenv("GA_TIMING") != nullptr;   // measurements only: host phase times per chunk on stderr
static double nowMs() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// A chunk is planned in passes that advance persistent control state (queued disposals, lagged channel counts and silence
// flags, overlap / history double buffers) before the first launch, so a failure after the simulation has started leaves
// the context between two blocks.  Such a failure is STICKY: every later render on this context returns
// GA_ERR_INVALID_OPERATION until the context is recreated (include/graphaudio_hip.h, "Errors").  Failures of the argument
// and graph checks that run before any state moves (disposed context, cycle, unsupported node) leave the context usable.
void Context::runChunk(int64_t n, float* const* bus) {
  if (faulted) fail(GA_ERR_INVALID_OPERATION, "context is faulted by an earlier render error (" + faultMsg + "); create a new context");
  chunkPhase = 0;
  try {
    const double t0 = timing ? nowMs() : 0.0;
    runChunkImpl(n, bus);
    if (timing) fprintf(stderr, "[ga]   chunk total on the host (incl. destructors): %.3f ms\n", nowMs() - t0);
  } catch (const Err& e) {
    if (chunkPhase > 0) {
      faulted = true;
      faultMsg = e.msg;
    }
    throw;
  } catch (...) {
    if (chunkPhase > 0) {
      faulted = true;
      faultMsg = "unexpected exception";
    }
    throw;
  }
}
// Everything the passes of one chunk share.  runChunkImpl is the sequence of these passes; every pass is a member function of
// Context so that its body reads the graph state directly.
struct ChunkRun {
  int64_t n = 0;                       // blocks of the chunk (the simulation may shorten it)
  std::vector<int> topo;               // reachable nodes in processing (post) order
  int maxDepth = 0, maxLevel = 0;
  std::vector<double> bt;              // accumulated block clock
  std::vector<int> srcIds;
  std::vector<int> srcIndex;           // node id -> position in srcIds / srcPlans, -1 = not a source of this chunk
  std::vector<SrcPlanOut> srcPlans;
  std::vector<int> streamIds;          // AudioStreamSourceNodes of the chunk
  std::vector<Segment> segs;
  std::unique_ptr<Exec> ex;
  int bHistMax = 0;
  double tm0 = 0, tmTopo = 0, tmSrc = 0, tmSim = 0, tmRes = 0, tmPre = 0, tmPlan = 0, tmLaunch = 0;
};

// pass 1: reachability, level and convolver depth of every node; state handed back by automated runs that ended
void Context::chunkTopology(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  // ---- reachability, level, convolver depth on the graph as it stands after the queued commands ----
  // (cached while no connection, disposal or impulse response changed since the last chunk)
  // (a graph with feedback is walked again every chunk: whether its loops can be cut at their DelayNodes depends on the delay
  // times, which are parameters, not graph structure)
  if (topoVersion == graphVersion && !topoCache.empty() && !topoHasCycles) {
    topo = topoCache;
  } else {
  for (auto& np : nodes) {
    np->reachable = false;
    np->isProcessing = false;
    np->level = 0;
    np->depth = 0;
    np->delaySplit = false;
  }
  cycleBlocks = 1;
  loopGainBound = 0.0;
  std::vector<int> color(nodes.size(), 0);
  std::vector<char> candidate;   // DelayNodes on a loop whose delay is a constant of at least two blocks
  bool unbreakable = false;
  {
    // The traversal order is the reference's (parameters first, then the inputs, connections in order: Nodes/AudioNode.cs:167-175).
    // A node met again while it is still being processed closes a feedback cycle.  The reference does not refuse that: its memo
    // check (Nodes/AudioNode.cs:153-156) returns before the "cycle detected" test can fire, and the consumer mixes the producer's
    // PREVIOUS block.  Such an edge carries no ordering constraint -- the producer is processed later in the block, as there.
    for (int id : staleProducers)
      if (id < (int)nodes.size() && nodes[id]) nodes[id]->staleProducer = false;
    staleProducers.clear();
    std::vector<int> stack;
    auto splittable = [&](const NodeS& d) {
      if (d.type != GA_NODE_DELAY || !d.params[0].events.empty() || !d.params[0].modulation.empty()) return 0;
      int dl = (int)(d.params[0].value * (float)sampleRate);   // DelayNode.cs:66 (float * int -> float, truncated)
      dl = std::min(std::max(dl, 0), d.maxDelaySamples);
      return dl / kBlock;   // whole blocks of delay
    };
    std::function<bool(int)> dfs = [&](int id) {   // false: `id` is being processed (the edge that led here is a feedback edge)
      if (color[id] == 2) return true;
      if (color[id] == 1) {
        if (!nodes[id]->staleProducer) staleProducers.push_back(id);
        nodes[id]->staleProducer = true;
        // the loop this edge closes: the nodes on the stack from `id` up.  It can be cut where a DelayNode delays by >= 2 blocks.
        if (candidate.empty()) candidate.assign(nodes.size(), 0);
        bool any = false;
        double bound = 1.0;   // an upper estimate of the loop's gain: what a last-bit difference that enters it is multiplied by per turn
        for (size_t q = stack.size(); q-- > 0;) {
          const int m = stack[q];
          const NodeS& mn = *nodes[m];
          if (splittable(mn) >= 2) {
            candidate[m] = 1;
            any = true;
          }
          bool moving = false;
          for (auto& p : mn.params) moving = moving || !p.events.empty() || !p.modulation.empty();
          switch (mn.type) {
            case GA_NODE_GAIN: bound *= moving ? 1e9 : std::fabs((double)mn.params[0].value); break;
            case GA_NODE_BIQUAD:
              if (moving) bound *= 1e9;
              else if (mn.filterType == GA_FILTER_PEAKING || mn.filterType == GA_FILTER_LOWSHELF || mn.filterType == GA_FILTER_HIGHSHELF)
                bound *= std::max(1.0, std::pow(10.0, (double)mn.params[2].value / 20.0));
              else bound *= std::max(1.0, (double)mn.params[1].value);   // (the resonance peak of a low / high / band pass is ~Q)
              break;
            case GA_NODE_CONVOLVER: bound *= mn.normalize ? 2.0 : 1e9; break;   // (normalised responses: broadband gain well below 1, peaks unknown)
            case GA_NODE_STEREO_PANNER: bound *= 2.0; break;                    // (oL = inL + inR * gainL)
            default: break;
          }
          if (m == id) break;
        }
        loopGainBound = std::max(loopGainBound, bound);
        if (!any) unbreakable = true;
        return false;
      }
      color[id] = 1;
      stack.push_back(id);
      NodeS& nd = *nodes[id];
      nd.reachable = true;
      int lvl = 0, dep = 0;
      for (auto& p : nd.params)
        if (!p.modulation.empty()) {
          if (nd.type == GA_NODE_BUFFER_SOURCE)   // a modulated playbackRate makes the resampler's consumption depend on audio data
            fail(GA_ERR_UNSUPPORTED, "audio-rate modulation of AudioBufferSourceNode.playbackRate is not on the device path");
          for (auto& m : p.modulation) {
            if (!dfs(m.first)) continue;
            NodeS& up = *nodes[m.first];
            lvl = std::max(lvl, up.level + 1);
            dep = std::max(dep, up.depth + ((up.type == GA_NODE_CONVOLVER && up.ir) ? 1 : 0));
          }
        }
      for (auto& in : nd.inputs)
        for (const Conn& cn : in.connected) {
          if (!dfs(cn.node)) continue;
          NodeS& up = *nodes[cn.node];
          lvl = std::max(lvl, up.level + 1);
          dep = std::max(dep, up.depth + ((up.type == GA_NODE_CONVOLVER && up.ir) ? 1 : 0));
        }
      nd.level = lvl;
      nd.depth = dep;
      color[id] = 2;
      stack.pop_back();
      topo.push_back(id);
      return true;
    };
    dfs(0);
    topoRefOrder = staleProducers.empty() ? std::vector<int>() : topo;   // (the reference's processing order: Context::refOrderSensitivity)
    // ---- loops that can be cut at a DelayNode (option "cycle_delay_split") ----
    // A DelayNode whose delay is a constant of d >= 128 K samples reads, for any K consecutive blocks, only samples its ring held
    // BEFORE those blocks: its output for the whole K-block chunk can be produced first (a gather from the history: the READER, a node
    // without inputs), and its input appended afterwards (the WRITER).  With every loop cut that way the chunk's graph is acyclic; the
    // reference's stale edge becomes an ordinary edge that reads its producer ONE BLOCK LATE (Exec::resolveInSeg).  Chunks of K blocks
    // instead of one: a 0.25 s echo renders 93 blocks per chunk.
    if (!staleProducers.empty() && !unbreakable && cycleDelaySplit) {
      int K = 1 << 30;
      for (size_t m = 0; m < candidate.size(); m++)
        if (candidate[m]) K = std::min(K, splittable(*nodes[m]));
      // second walk, edges OUT of a cut DelayNode carry no ordering: is anything still cyclic?
      std::vector<int> color2(nodes.size(), 0), order, deferred;
      std::vector<int> lvl2(nodes.size(), 0), dep2(nodes.size(), 0);
      bool cyclic = false;
      std::function<void(int)> dfs2 = [&](int id) {
        if (color2[id] == 2 || cyclic) return;
        if (color2[id] == 1) {
          cyclic = true;
          return;
        }
        color2[id] = 1;
        NodeS& nd = *nodes[id];
        int lvl = 0, dep = 0;
        auto edge = [&](int up) {
          if (candidate[up]) {   // the reader: a source (planned in front of everything); its writer is walked as a root of its own
            if (color2[up] == 0) deferred.push_back(up);
            return;
          }
          dfs2(up);
          lvl = std::max(lvl, lvl2[up] + 1);
          dep = std::max(dep, dep2[up] + ((nodes[up]->type == GA_NODE_CONVOLVER && nodes[up]->ir) ? 1 : 0));
        };
        for (auto& p : nd.params)
          for (auto& m : p.modulation) edge(m.first);
        for (auto& in : nd.inputs)
          for (const Conn& cn : in.connected) edge(cn.node);
        lvl2[id] = lvl;
        dep2[id] = dep;
        color2[id] = 2;
        order.push_back(id);
      };
      dfs2(0);
      for (size_t q = 0; q < deferred.size() && !cyclic; q++) dfs2(deferred[q]);
      if (!cyclic && K >= 2 && order.size() == topo.size()) {
        topo = order;
        for (int id : topo) {
          nodes[id]->level = lvl2[id];
          nodes[id]->depth = dep2[id];
          nodes[id]->delaySplit = candidate[id] != 0;
        }
        cycleBlocks = K;
      }
    }
  }
  topoCache = topo;
  topoVersion = graphVersion;
  topoHasCycles = !staleProducers.empty();
  }
  // Feedback: the loop closes through the block a producer put out LAST.  Unless every loop can be cut at a DelayNode (above: chunks
  // of `cycleBlocks` blocks) nothing can be batched along time -- the chunk is one block, the reference's own granularity (a 10 s
  // render = 3,750 chunks: launch bound, ~0.1 - 0.3 ms each)
  if (topoHasCycles) r.n = std::min<int64_t>(r.n, cycleBlocks);
  if (topoStatsVersion != graphVersion || topoStatsSize != topo.size()) {   // (cached with the order: a sweep over 28,672 node records is 0.5 ms)
    topoMaxDepth = topoMaxLevel = 0;
    topoHasTimeNodes = topoHasConvolvers = topoHasOscillators = false;
    for (int id : topo) {
      const NodeS& nd = *nodes[id];
      topoMaxDepth = std::max(topoMaxDepth, nd.depth);
      topoMaxLevel = std::max(topoMaxLevel, nd.level);
      if (nd.type == GA_NODE_DELAY || nd.type == GA_NODE_STREAM_SOURCE) topoHasTimeNodes = true;
      if (nd.type == GA_NODE_CONVOLVER) topoHasConvolvers = true;   // (with or without an impulse response: Buffer setters run in drain())
      if (nd.type == GA_NODE_OSCILLATOR) topoHasOscillators = true;
    }
    topoStatsVersion = graphVersion;
    topoStatsSize = topo.size();
  }
  maxDepth = topoMaxDepth;
  maxLevel = topoMaxLevel;
  // automated runs that ended hand their state back to the host: only nodes whose state went to the device are looked at
  // (Context::deviceStateNodes; "this chunk ran the per-sample kernel" is a stamp, NodeS::bqDynSeq / panDynSeq, not a flag to reset)
  for (size_t i = 0; i < deviceStateNodes.size();) {
    const int id = deviceStateNodes[i];
    NodeS* np = id < (int)nodes.size() ? nodes[id].get() : nullptr;
    if (!np || np->disposed || (!np->panOnDevice && !np->coefOnDevice)) {
      deviceStateNodes[i] = deviceStateNodes.back();
      deviceStateNodes.pop_back();
      continue;
    }
    i++;
    NodeS& nd = *np;
    if (!nd.reachable) continue;
    // (while a signal is connected to the parameter the node stays on its per-sample kernel, which serves a silent modulation input
    // as a constant too -- Sim::process -- so the state stays where it is: no stall of the pipeline per modulated node and chunk)
    if (nd.type == GA_NODE_STEREO_PANNER && nd.panOnDevice && nd.params[0].events.empty() && nd.params[0].modulation.empty()) {
      PanState tmp;   // back to a constant pan: the gains the automated run left on the device are the node's state
      GA_HIP(hipStreamSynchronize(stream));   // the state the previous chunk left
      GA_HIP(hipMemcpy(&tmp, nd.panDev, sizeof(PanState), hipMemcpyDeviceToHost));
      nd.panLast = tmp.last_pan;
      nd.panGL = tmp.gain_l;
      nd.panGR = tmp.gain_r;
      nd.panOnDevice = false;
      apiEpoch++;   // (host-tracked state changed: the next first block is traversed)
    }
    if (nd.type == GA_NODE_BIQUAD && nd.coefOnDevice && nd.bqDyn) {
      bool automated = false;
      for (auto& p : nd.params) automated = automated || !p.events.empty() || !p.modulation.empty();
      if (!automated) {  // back to constant parameters: fetch the coefficients the automated run left on the device
        BiquadDynState tmp;
        GA_HIP(hipStreamSynchronize(stream));
        GA_HIP(hipMemcpy(&tmp, nd.bqDyn, 24, hipMemcpyDeviceToHost));
        nd.b0 = tmp.b0; nd.b1 = tmp.b1; nd.b2 = tmp.b2; nd.a1 = tmp.a1; nd.a2 = tmp.a2;
        nd.coefDirty = tmp.dirty != 0;
        nd.coefOnDevice = false;
        apiEpoch++;
      }
    }
  }

}

// pass 2: block clock, source timelines, control-plane simulation -> segments (from here on control state moves)
void Context::chunkSimulate(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  r.tmTopo = nowMs();
  // ---- block clock (accumulated, AudioContextBase.cs:78-79) ----
  bt.assign(n + 1, 0.0);
  bt[0] = currentTime;
  const double increment = (double)kBlock / sampleRate;
  for (int64_t i = 0; i < n; i++) bt[i + 1] = bt[i] + increment;

  // ---- plan sources ----
  std::vector<int64_t> breaks;
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    const bool scheduled = nd.type == GA_NODE_CONSTANT_SOURCE || nd.type == GA_NODE_OSCILLATOR;
    if (nd.type != GA_NODE_BUFFER_SOURCE && !scheduled) continue;
    if (r.srcIndex.empty()) r.srcIndex.assign(nodes.size(), -1);
    r.srcIndex[id] = (int)srcIds.size();
    srcIds.push_back(id);
    srcPlans.push_back(scheduled ? planScheduled(*this, nd, n, bt) : planSource(*this, nd, n, bt));
    for (const SrcSpan& sp : nd.spans)
      if (sp.b0 > 0 && sp.b0 < n) breaks.push_back(sp.b0);
    if (srcPlans.back().partialBlock >= 0) {
      breaks.push_back(srcPlans.back().partialBlock);
      if (srcPlans.back().partialBlock + 1 < n) breaks.push_back(srcPlans.back().partialBlock + 1);
    }
  }
  for (int id : topo) {   // AudioStreamNodeBase: replay on indices; a change of channel count / silence is a segment break
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_STREAM_SOURCE) continue;
    for (auto& p : nd.params)
      if (!p.modulation.empty()) fail(GA_ERR_UNSUPPORTED, "audio-rate modulation of AudioStreamSourceNode.playbackRate is not on the device path");
    streamReplay(nd, n, bt, false);
    r.streamIds.push_back(id);
    for (int64_t b = 1; b < n; b++)
      if (nd.stInfo[b].outCh != nd.stInfo[b - 1].outCh || nd.stInfo[b].silent != nd.stInfo[b - 1].silent) breaks.push_back(b);
  }
  std::sort(breaks.begin(), breaks.end());
  breaks.erase(std::unique(breaks.begin(), breaks.end()), breaks.end());
  std::unordered_map<int64_t, std::vector<int>> goneAt;
  for (size_t i = 0; i < srcIds.size(); i++)
    if (srcPlans[i].gone && srcPlans[i].goneAt < n) goneAt[srcPlans[i].goneAt].push_back(srcIds[i]);

  r.tmSrc = nowMs();
  chunkPhase = 1;   // from here on persistent control state moves: a failure is sticky (see runChunk)
  // ---- simulate ----
  Sim sim{*this, n};
  std::vector<int64_t> extraBreaks;
  sim.extraBreaks = &extraBreaks;
  sim.blockTimes = &bt;
  int minDestCh = 32;
  {
    int64_t b = 0;
    uint64_t prevHash = lastHash;
    const size_t kMaxSegs = 96;
    while (b < n) {
      if (segs.size() >= kMaxSegs) {  // too fragmented: stop the chunk here, the caller continues with a new one
        n = b;
        break;
      }
      auto g = goneAt.find(b);
      if (g != goneAt.end())
        for (int id : g->second) doDispose(id);  // queued Dispose() runs in the next block's DrainCommands
      Segment sg;
      sg.b0 = b;
      if (!segNodePool.empty()) {   // (storage of an earlier chunk's segment: warm, and no page faults of a fresh allocation)
        sg.nodes = std::move(segNodePool.back());
        segNodePool.pop_back();
      }
      sg.nodes.reserve(topo.size());
      sim.cur = &sg;
      sim.brel = b;
      sim.blockNumber = currentBlock + b + 1;
      // the first block of a steady chunk: nothing can have moved since the previous chunk's last segment (see Context::lastSegNodes)
      // -- its records are taken over, the traversal is skipped
      bool replayed = false;
      if (b == 0 && simReplay && lastSegStable && !lastSegNodes.empty() && lastSegEpoch == apiEpoch && lastSegGraphVersion == graphVersion &&
          !topoHasTimeNodes && lastSegNodes.size() == topo.size() && g == goneAt.end()) {
        bool same = true;
        for (const NodeSeg& ns : lastSegNodes) {   // every source still in the phase (and on the buffer) the records say
          if (ns.type != GA_NODE_BUFFER_SOURCE && ns.type != GA_NODE_CONSTANT_SOURCE && ns.type != GA_NODE_OSCILLATOR) continue;
          const NodeS& sn = *nodes[ns.id];
          const SrcSpan& sp = spanAt(sn, 0);
          if (sp.phase != ns.srcPhase || (ns.type == GA_NODE_BUFFER_SOURCE && ns.srcBuf != sn.bufId)) {
            same = false;
            break;
          }
        }
        if (same) {
          if (!sg.nodes.empty() || sg.nodes.capacity()) segNodePool.push_back(std::move(sg.nodes));
          sg.nodes = std::move(lastSegNodes);
          lastSegNodes.clear();
          for (NodeSeg& ns : sg.nodes) {
            switch (ns.type) {
              case GA_NODE_BUFFER_SOURCE:
                if (ns.srcPhase == SRC_PLAY) {
                  const SrcSpan& sp = spanAt(*nodes[ns.id], 0);
                  ns.srcPos = sp.pos + (0 - sp.b0) * kBlock;
                  ns.srcBlk = sp.blkIdx + (0 - sp.b0);
                }
                break;
              case GA_NODE_BIQUAD:   // (per-chunk flags the traversal would have set again: Sim::process)
                if (ns.bqDynamic && !ns.ins[0].silent) nodes[ns.id]->bqDynSeq = chunkSeq;
                break;
              case GA_NODE_STEREO_PANNER:
                if (ns.panDyn) nodes[ns.id]->panDynSeq = chunkSeq;
                break;
              case GA_NODE_DESTINATION:
                destOutCh = ns.outCh;
                break;
              default: break;
            }
          }
          sg.hash = lastSegHash;
          replayed = true;
          stats.sim_replays++;
        }
      }
      if (!replayed) {
        inRender = true;
        try {
          sim.evalNode(0);
        } catch (...) {
          inRender = false;
          topoVersion = 0;   // nodes may be left marked as processing: rebuild (and reset) everything next time
          throw;
        }
        inRender = false;
        for (int64_t x : extraBreaks)   // e.g. the block in which delayed audio reaches a DelayNode's output
          if (x > b && x < n) {
            auto it = std::lower_bound(breaks.begin(), breaks.end(), x);
            if (it == breaks.end() || *it != x) breaks.insert(it, x);
          }
        extraBreaks.clear();
        sg.hash = sim.hashSeg(sg);
      }
      lastSegStable = sg.hash == prevHash;   // (of the segment that turns out to be the chunk's last: a fixpoint of the traversal)
      int64_t nb;
      if (sg.hash != prevHash) {
        nb = b + 1;
      } else {
        auto it = std::upper_bound(breaks.begin(), breaks.end(), b);
        nb = it == breaks.end() ? n : *it;
      }
      sg.b1 = std::min(nb, n);
      prevHash = sg.hash;
      minDestCh = std::min(minDestCh, sg.nodes.back().outCh);
      b = sg.b1;
      segs.push_back(std::move(sg));
    }
    lastHash = prevHash;
    if (n < (int64_t)bt.size() - 1) {
      bt.resize(n + 1);
      // the chunk was cut short: the stream tables (window at the END of the chunk, pieces) are rebuilt for the blocks that run
      for (int id : r.streamIds) streamReplay(*nodes[id], n, bt, false);
    }
  }
  chunkMinDestCh = minDestCh;
  r.tmSim = nowMs();

}

// pass 3: per-chunk device resources (delay lines, slabs, zero page, bus)
void Context::chunkResources(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  // ---- DelayNode state: history [rings][maxDelay] (persistent) and the chunk's line [rings][maxDelay + frames] ----
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_DELAY) continue;
    nd.delayLoaded = false;
    int rings = std::max(nd.delayRings, 2);
    for (const Segment& sg : segs)
      for (const NodeSeg& ns : sg.nodes)
        if (ns.id == id) rings = std::max(rings, ns.ins[0].bufCh);   // EnsureChannelCount (:102-113): new rings start empty
    nd.delayRings = rings;
    const size_t maxD = (size_t)nd.maxDelaySamples;
    if (nd.delayHistRings < rings) {
      float* nh = (float*)dalloc(maxD * rings * sizeof(float));
      GA_HIP(hipMemsetAsync(nh, 0, maxD * rings * sizeof(float), stream));
      if (nd.delayHist) {
        GA_HIP(hipMemcpyAsync(nh, nd.delayHist, maxD * nd.delayHistRings * sizeof(float), hipMemcpyDeviceToDevice, stream));
        GA_HIP(hipStreamSynchronize(stream));
        dfree(nd.delayHist, maxD * nd.delayHistRings * sizeof(float));
      }
      nd.delayHist = nh;
      nd.delayHistRings = rings;
    }
    const int64_t cap = roundup(frames, 4096);
    if (nd.delayCap < cap || nd.delayLineRings < rings) {
      if (nd.delayLine) {
        GA_HIP(hipStreamSynchronize(stream));
        dfree(nd.delayLine, (maxD + (size_t)nd.delayCap) * nd.delayLineRings * sizeof(float));
      }
      nd.delayCap = std::max(nd.delayCap, cap);
      nd.delayLineRings = rings;
      nd.delayLine = (float*)dalloc((maxD + (size_t)nd.delayCap) * rings * sizeof(float));
    }
    nd.delayW.assign(rings, 0);
    nd.delayR.assign(rings, 0);
  }

  // ---- device resources for this chunk ----
  resetSlabs(*this, frames);
  if (zerosLen < frames) {
    if (zeros) {
      GA_HIP(hipStreamSynchronize(stream));
      dfree(zeros, (size_t)zerosLen * 4);
    }
    zerosLen = roundup(frames, 4096);
    zeros = (float*)dalloc((size_t)zerosLen * 4);
    GA_HIP(hipMemsetAsync(zeros, 0, (size_t)zerosLen * 4, stream));
  }
  if (busCapFrames < frames) {
    GA_HIP(hipStreamSynchronize(stream));
    for (float* p : busSlabs) dfree(p, (size_t)busCapFrames * 4);
    busSlabs.clear();
    busCapFrames = roundup(frames, 4096);
  }
  while ((int)busSlabs.size() < 32 && (int)busSlabs.size() < std::max(destOutCh, 2)) busSlabs.push_back((float*)dalloc((size_t)busCapFrames * 4));
  {
    int mx = 0;
    for (auto& sg : segs) mx = std::max(mx, sg.nodes.back().outCh);
    while ((int)busSlabs.size() < mx) busSlabs.push_back((float*)dalloc((size_t)busCapFrames * 4));
  }

}

// pass 4: AudioParam timelines -> device curves
void Context::chunkParamCurves(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  // ---- AudioParam curves (AudioParam.cs:93-166) for automated gain params: one launch for the whole chunk ----
  {
    // Which parameters of the reachable nodes carry a timeline?  The list stands while no API call, no drained command and no graph
    // edit happened (apiEpoch / graphVersion): a sweep over the parameter vectors of 28,672 nodes per chunk was 1.5 - 2 ms.
    if (curveListEpoch != apiEpoch || curveListGraphVersion != graphVersion || curveListTopoSize != topo.size()) {
      for (auto& e : curveList)   // (curves handed out for the previous list)
        if (e.first < (int)nodes.size() && nodes[e.first] && e.second < (int)nodes[e.first]->params.size()) nodes[e.first]->params[e.second].curve = nullptr;
      curveList.clear();
      for (int id : topo) {
        NodeS& nd = *nodes[id];
        for (auto& p : nd.params) p.curve = nullptr;
        if (nd.type != GA_NODE_GAIN && nd.type != GA_NODE_BIQUAD && nd.type != GA_NODE_CONSTANT_SOURCE && nd.type != GA_NODE_OSCILLATOR &&
            nd.type != GA_NODE_DELAY && nd.type != GA_NODE_STEREO_PANNER)
          continue;
        for (int pi = 0; pi < (int)nd.params.size(); pi++)
          if (!nd.params[pi].events.empty()) curveList.push_back({id, pi});
      }
      curveListEpoch = apiEpoch;
      curveListGraphVersion = graphVersion;
      curveListTopoSize = topo.size();
    }
    std::vector<ParamJob> pjobs;
    std::vector<ParamEvent> events;
    // identical timelines (same events, value and rate -- e.g. the same fade on every voice) share one curve: hash of the bytes,
    // verified against the job that owns the curve
    std::unordered_multimap<uint64_t, int> jobOf;
    auto sameTimeline = [&](const ParamJob& pj, const ParamS& p) {
      return pj.nev == (int)p.events.size() && pj.value == p.value && pj.arate == (p.arate ? 1 : 0) &&
             std::memcmp(&events[pj.ev0], p.events.data(), p.events.size() * sizeof(ParamEvent)) == 0;
    };
    for (auto& e : curveList) {
      ParamS& p = nodes[e.first]->params[e.second];
      p.curve = nullptr;
      uint64_t h = 1469598103934665603ull;
      const uint64_t* w = (const uint64_t*)p.events.data();
      for (size_t i = 0; i < p.events.size() * sizeof(ParamEvent) / 8; i++) h = (h ^ w[i]) * 1099511628211ull;
      uint32_t vb;
      std::memcpy(&vb, &p.value, 4);
      h = (h ^ vb ^ (p.arate ? 0x100000000ull : 0)) * 1099511628211ull;
      auto range = jobOf.equal_range(h);
      for (auto it = range.first; it != range.second && !p.curve; ++it)
        if (sameTimeline(pjobs[it->second], p)) p.curve = pjobs[it->second].out;
      if (p.curve) continue;
      p.curve = getSlab(*this);
      ParamJob pj;
      pj.out = p.curve;
      pj.ev0 = (int)events.size();
      pj.nev = (int)p.events.size();
      pj.value = p.value;
      pj.arate = p.arate ? 1 : 0;
      pj.b0 = 0;
      pj.nblocks = n;
      events.insert(events.end(), p.events.begin(), p.events.end());
      jobOf.emplace(h, (int)pjobs.size());
      pjobs.push_back(pj);
    }
    if (!pjobs.empty()) {
      size_t jo = ex.plan.putv(pjobs), eo = ex.plan.putv(events), bo = ex.plan.putv(bt);
      int nj = (int)pjobs.size();
      double dt = 1.0 / sampleRate;
      hipStream_t st = stream;
      int64_t nn = n;
      ex.plan.add(LK_OTHER, [=](uint8_t* base) {
        launch_param_curve(st, (const ParamJob*)(base + jo), nj, (const ParamEvent*)(base + eo), (const double*)(base + bo), dt, nn);
      });
    }
  }

}

// When everything the destination receives in this chunk is ONE fused group of formulation D convolvers (or a single one) with
// the bus's channel count, the group's inverse transforms write the bus themselves: the leader's output slabs ARE the bus
// rows (the caller's device rows or page-locked host rows when Context::render set busTarget), and the destination's mix -- a
// copy of one term -- disappears (Exec::resolveInSeg skips a forced target that already holds its only term).
void Context::aliasBusToLeader(ChunkRun& r) {
  Exec& ex = *r.ex;
  int leader = -1, nch = 0;
  for (const Segment& sg : r.segs) {
    if (sg.nodes.empty() || sg.nodes.back().id != 0 || sg.nodes.back().ins.empty()) return;   // (node 0 is the destination)
    const InSeg& is = sg.nodes.back().ins[0];
    for (const TermS& t : is.terms) {
      const NodeS& nd = *nodes[t.node];
      if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.convPath != 4 || nd.dLeader < 0 || t.out != 0 || t.ch != is.bufCh) return;
      if (leader < 0) {
        leader = nd.dLeader;
        nch = is.bufCh;
      } else if (leader != nd.dLeader || nch != is.bufCh) {
        return;
      }
    }
  }
  if (leader < 0 || nch < 1 || nch > (int)busSlabs.size()) return;
  const NodeS& ld = *nodes[leader];
  if (ld.outputs.empty() || ld.outputs[0].connectedInputs.size() != 1) return;
  const InRef& to = ld.outputs[0].connectedInputs[0];
  if (to.node != 0 || to.input != 0) return;
  for (int ch = 0; ch < nch; ch++) ex.setNodeOut(leader, ch, busTarget[ch] ? busTarget[ch] : busSlabs[ch]);
}

// pass 5: convolver formulations of new nodes, fusion groups, scratch arenas (sized before any recorded launch captures them)
void Context::chunkConvScratch(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  // resampler trajectories used in this chunk go into one device table
  for (auto& kv : resamplers) kv.second->devOffset = -1;

  // ---- convolver scratch planes are shared by all groups: size them for the largest group BEFORE any recorded
  //      launch captures their address ----
  if (topoHasConvolvers) {   // (a graph without convolvers -- tens of thousands of nodes of config 4 -- skips these sweeps)
    refOrderSensitivity(topo);
    assignConvPaths(topo, n);
    for (int id : topo) {
      NodeS& nd = *nodes[id];
      if (nd.type == GA_NODE_CONVOLVER) nd.refOrder = nd.refSens && nd.ir && (nd.convPath == 2 || nd.convPath == 3);
    }
    planCoarseFusion(topo, segs);
    for (int id : topo)
      if (nodes[id]->type == GA_NODE_CONVOLVER) nodes[id]->dGroupSize = 0;
    for (int id : topo) {
      const NodeS& nd = *nodes[id];
      if (nd.type == GA_NODE_CONVOLVER && nd.ir && nd.convPath == 4 && nd.dLeader >= 0) nodes[nd.dLeader]->dGroupSize++;
    }
    aliasBusToLeader(r);
  }
  bHistMax = 0;
  if (topoHasConvolvers) {
    size_t xMax = 0, yMax = 0;
    size_t bx = 0, by = 0;
    for (int id : topo) {
      NodeS& nd = *nodes[id];
      if (nd.type != GA_NODE_CONVOLVER || !nd.ir) continue;
      if (nd.convPath == 2 || nd.convPath == 3) {
        bx += nd.bInCh;
        by += nd.bSlots;
        bHistMax = std::max(bHistMax, nd.ir->P - 1);
      }
      for (auto& rr : nd.convRows) {
        ConvGroup& g = *rr.group;
        ensureGroupState(g);
        const int ty_ = (int)roundup(n, 64), tx_ = ty_ + g.P + 128;
        xMax = std::max(xMax, (size_t)kBins * tx_ * g.rp * sizeof(float));
        yMax = std::max(yMax, (size_t)kBins * ty_ * g.rp * sizeof(float));
      }
    }
    if (xMax) {
      ensure(planes[0], xMax);
      ensure(planes[1], xMax);
      ensure(planes[2], yMax);
      ensure(planes[3], yMax);
    }
    {  // formulation D: the stages of a chunk run one after the other on the stream and share the two arenas
      std::map<int, std::pair<size_t, size_t>> perDepth;   // depth -> (X frames, Y frames upper bound)
      const int64_t nT = (n * kBlock + kCoarseBlock - 1) / kCoarseBlock;
      for (int id : topo) {
        NodeS& nd = *nodes[id];
        if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.convPath != 4) continue;
        auto& pd = perDepth[nd.depth];
        pd.first += (size_t)nd.bInCh * (size_t)(nT + nd.ir->coarseP);   // (+ the window behind the chunk's last block: carried tails)
        // Y rows: one per slot unless fused; fused groups need (members / 32 + 1) x channels rows, never more than the slots
        pd.second += (size_t)nd.bSlots * (size_t)(nT + kCoarseMaxP);
      }
      size_t xf = 0, yf = 0;
      for (auto& kv : perDepth) {
        xf = std::max(xf, kv.second.first);
        yf = std::max(yf, kv.second.second);
      }
      if (xf) {
        ensure(coarseX, xf * kCoarseBins * sizeof(float2));
        ensure(coarseY, yf * kCoarseBins * sizeof(float2));
      }
      if (coarsePremix) {   // pre-mixed groups: [history | chunk] of the mixed signal per input channel of the group
        std::map<int, int> members;
        for (int id : topo) {
          NodeS& nd = *nodes[id];
          if (nd.type == GA_NODE_CONVOLVER && nd.ir && nd.convPath == 4) members[nd.dLeader >= 0 ? nd.dLeader : id]++;
        }
        std::map<int, size_t> pmDepth;
        for (auto& kv : members) {
          if (kv.second < 2) continue;
          const NodeS& ld = *nodes[kv.first];
          pmDepth[ld.depth] += (size_t)ld.bInCh * ((((size_t)(ld.dHistLen + n * kBlock) * sizeof(float)) + 255) & ~(size_t)255);
        }
        size_t pm = 0;
        for (auto& kv : pmDepth) pm = std::max(pm, kv.second);
        if (pm) ensure(coarseM, pm);
      }
    }
    bRowX = bRowY = 0;
    if (bx) {  // formulation B scratch: [row][bin][block]; x planes alternate between two pairs (flushPlaneHistories)
      const size_t txb = (size_t)roundup(bHistMax, 4) + roundup(n, 16) + 16, tyb = (size_t)roundup(n, 256);
      bPairWrite = bPairCur ^ 1;
      flushPlaneHistories(bPairWrite);
      // both pairs grow together (a render that continues reaches the other pair in its next chunk; growing it then would
      // put an allocation into the steady state), but only a pair without residents can be reallocated
      const size_t xb = bx * kBins * txb * sizeof(float);
      if (planesB[0].bytes < xb || planesBalt[0].bytes < xb) flushPlaneHistories(bPairCur);
      ensure(planesB[0], xb);
      ensure(planesB[1], xb);
      ensure(planesBalt[0], xb);
      ensure(planesBalt[1], xb);
      ensure(planesB[2], by * kBins * tyb * sizeof(float));
      ensure(planesB[3], by * kBins * tyb * sizeof(float));
    }
  }

  if (topoHasOscillators)
  for (int id : topo) {  // OscillatorNode._phase lives on the device (one double, zero at Start)
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_OSCILLATOR || nd.oscPhase) continue;
    nd.oscPhase = (double*)dalloc(64);
    GA_HIP(hipMemsetAsync(nd.oscPhase, 0, 64, stream));
  }
}

void Context::ensureBiquadState(NodeS& bn) {
    if (bn.bqDyn) return;
    const size_t per = (sizeof(BiquadDynState) + 31) & ~(size_t)31;
    const size_t blk = (size_t)1 << 20;
    if (bqBlocks.empty() || bqUsed + per > blk) {
      void* p = dalloc(blk);
      GA_HIP(hipMemsetAsync(p, 0, blk, stream));
      bqBlocks.push_back(p);
      bqUsed = 0;
    }
    bn.bqDyn = (BiquadDynState*)((char*)bqBlocks.back() + bqUsed);
    bn.bqState = (float*)((char*)bn.bqDyn + 24);
    bqUsed += per;
  }

// ---- pass 6, per node type -------------------------------------------------------------------------------------------------
// dense tables indexed by node id, validated by a per-(stage, segment) stamp: no hashing on the per-node path
struct DenseSeg {
  std::vector<uint32_t>& st; std::vector<const NodeSeg*>& v; uint32_t e;
  const NodeSeg* find(int id) const { return st[id] == e ? v[id] : nullptr; }
};
struct DenseInt {
  std::vector<uint32_t>& st; std::vector<int>& v; uint32_t e; int def;
  int get(int id) const { return st[id] == e ? v[id] : def; }
};
// one node of one segment being planned: what the per-type planners below share with Context::chunkPlanNodes
struct NodePlanCtx {
  ChunkRun& r; Exec& ex; size_t si; const Segment& sg; int64_t f0, nf, nb;
  const NodeSeg& ns; NodeS& nd; Views& ov;
  const DenseSeg& segNode; const DenseInt& absorbedBy; int levelBqHeads;
  int delayPhase = 0;   // DelayNode: 0 = the whole node, 1 = reader only, 2 = writer only (a loop cut at this node)
};

// ConstantSourceNode.Process (ConstantSourceNode.cs:76-141)
void Context::planConstantSource(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  if (ns.srcPhase != SRC_PLAY) return;
  ConstJob cj;
  cj.curve = ex.paramView((int)si, ns, 0);
  cj.out = ex.nodeOut(ns.id, 0);
  cj.value = nd.params[0].value;
  cj.pad_ = 0;
  cj.f0 = f0;
  cj.n = nf;
  cj.lo = nd.schedLo;
  cj.hi = nd.schedHi;
  ex.constJobs.push_back(cj);
  ov[0] = cj.out;
}

// OscillatorNode.Process (OscillatorNode.cs:91-196)
void Context::planOscillator(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  if (ns.srcPhase != SRC_PLAY) return;
  OscJob oj;
  oj.curve = ex.paramView((int)si, ns, 0);
  oj.out = ex.nodeOut(ns.id, 0);
  oj.phase = nd.oscPhase;
  oj.value = nd.params[0].value;
  oj.type = nd.oscType;
  oj.sample_rate = sampleRate;
  oj.pad_ = 0;
  oj.f0 = f0;
  oj.n = nf;
  oj.lo = nd.schedLo;
  oj.hi = nd.schedHi;
  ex.oscJobs.push_back(oj);
  ov[0] = oj.out;
}

// DelayNode.Process (DelayNode.cs:43-100): the segment's input appended to the rings, then a gather
void Context::planDelay(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  const int ch = ns.ins[0].bufCh;
  const int maxD = nd.maxDelaySamples;
  const size_t pitch = (size_t)maxD + (size_t)nd.delayCap;
  if (!nd.delayLoaded) {   // history of the previous chunks in front of every ring's line
    nd.delayLoaded = true;
    std::fill(nd.delayW.begin(), nd.delayW.end(), 0);
    std::fill(nd.delayR.begin(), nd.delayR.end(), 0);
    float* line = nd.delayLine;
    float* hist = nd.delayHist;
    const int rings = nd.delayHistRings;
    hipStream_t st = stream;
    ex.plan.add(LK_OTHER, [=](uint8_t*) {
      GA_HIP(hipMemcpy2DAsync(line, pitch * 4, hist, (size_t)maxD * 4, (size_t)maxD * 4, rings, hipMemcpyDeviceToDevice, st));
    });
  }
  // append this segment's input to the rings that are processed (a ring beyond the input's channel count does not
  // move, DelayNode.cs:62-94), then gather
  // The input is mixed STRAIGHT into the rings (the rings are the forced targets of the input's mix, like the destination's bus):
  // a second job that copies a mixed slab into the ring would sit in the same launch as the mix that produces the slab -- no
  // order between them (until round 3 a DelayNode with two connections, or behind a folded GainNode, read a half-written slab).
  SmallVec<float*, 4> ring((size_t)std::max(ch, 1), nullptr);
  // A DelayNode at which a feedback loop is cut (NodeS::delaySplit, Context::chunkTopology) is planned twice per segment: the READER
  // in front of everything (its gather only touches what the ring held before the chunk), the WRITER at the node's level -- possibly
  // a convolver depth later, i.e. after the readers of ALL segments: the reader counts the ring positions on its own (delayR).
  const bool reader = k.delayPhase != 2, writer = k.delayPhase != 1;
  for (int cch = 0; cch < ch; cch++)   // ring[c][f] = input sample of frame f
    ring[cch] = nd.delayLine + (size_t)cch * pitch + maxD + (k.delayPhase == 1 ? nd.delayR[cch] : nd.delayW[cch]) - f0;
  if (!writer) {
  } else if (!ns.ins[0].silent) {
    ex.resolveInput((int)si, ns, 0, true, ring.data());
  } else {
    for (int cch = 0; cch < ch; cch++) {   // zeros
      MixJob mj;
      mj.out = ring[cch];
      mj.term0 = (int)ex.terms.size();
      mj.nterms = 0;
      mj.f0 = f0;
      mj.n = nf;
      ex.noteAlign(ring[cch], f0);
      ex.mixJobs.push_back(mj);
    }
  }
  if (!reader) {
    for (int cch = 0; cch < ch; cch++) nd.delayW[cch] += nf;
    return;
  }
  const float* delayCurve = ex.paramView((int)si, ns, 0);
  for (int cch = 0; cch < ch; cch++) {
    float* base = ring[cch];
    DelayJob dj;
    dj.line = base;
    dj.curve = delayCurve;
    dj.out = ex.nodeOut(ns.id, cch);
    dj.value = nd.params[0].value;
    dj.sample_rate = sampleRate;
    dj.max_delay = maxD;
    dj.pad_ = 0;
    dj.f0 = f0;
    dj.n = nf;
    ex.delayJobs.push_back(dj);
    if (writer) nd.delayW[cch] += nf;
    else nd.delayR[cch] += nf;
    if (ns.delayAudible) ov[cch] = dj.out;   // a buffer still flagged silent is skipped by every consumer
  }
}

// StereoPannerNode.Process (StereoPannerNode.cs:36-153)
void Context::planStereoPanner(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  if (ns.ins[0].silent) return;   // cleared 2-channel output (:49-54)
  auto iv = ex.resolveInput((int)si, ns, 0, false, nullptr);
  if (ns.panDyn) {
    if (!nd.panDev) nd.panDev = (PanState*)dalloc(64);
    PanDynJob dj;
    dj.in_l = iv[0] ? iv[0] : zeros;
    dj.in_r = ns.panMode == 2 ? (iv[1] ? iv[1] : zeros) : nullptr;
    dj.out_l = ex.nodeOut(ns.id, 0);
    dj.out_r = ex.nodeOut(ns.id, 1);
    dj.curve = ex.paramView((int)si, ns, 0);
    dj.state = nd.panDev;
    dj.init_state = PanState{nd.panLast, nd.panGL, nd.panGR, 0.f};
    dj.value = nd.params[0].value;
    dj.pad_ = 0;
    dj.init = nd.panOnDevice ? 0 : 1;   // the host-tracked state is handed over once
    if (!nd.panOnDevice) deviceStateNodes.push_back(ns.id);
    nd.panOnDevice = true;
    dj.stereo = ns.panMode == 2 ? 1 : 0;
    dj.f0 = f0;
    dj.n = nf;
    ex.panDynJobs.push_back(dj);
    ov[0] = dj.out_l;
    ov[1] = dj.out_r;
    return;
  }
  PanJob pj;
  pj.in_l = iv[0] ? iv[0] : zeros;
  pj.in_r = ns.panMode == 2 ? (iv[1] ? iv[1] : zeros) : nullptr;
  pj.out_l = ex.nodeOut(ns.id, 0);
  pj.out_r = ex.nodeOut(ns.id, 1);
  pj.gain_l = ns.panGL;
  pj.gain_r = ns.panGR;
  pj.pan = ns.pan;
  pj.stereo = ns.panMode == 2 ? 1 : 0;
  pj.f0 = f0;
  pj.n = nf;
  ex.panJobs.push_back(pj);
  ov[0] = pj.out_l;
  ov[1] = pj.out_r;
}

// The per-sample table of a resampler trajectory on the device, up to (excluding) block `upto`: what extend() left in `pending` is
// appended (through the chunk's tables: a plan entry in front of the launches that read it), the buffer doubles when it is full.
bool Context::ensureResampleSamples(Exec& ex, Resampler& rs, int64_t upto) {
  const int64_t have = rs.devBlocks + (int64_t)rs.pending.size() / kBlock;
  if (upto > have) return false;                       // (blocks the trajectory was extended to before the table existed)
  if (upto <= rs.devBlocks || rs.pending.empty()) return upto <= rs.devBlocks;
  const int64_t need = have;
  if (need > rs.devCapBlocks) {
    const int64_t cap = std::max<int64_t>(4096, std::max(need, 2 * rs.devCapBlocks));
    ResampleSample* nw = (ResampleSample*)dalloc((size_t)cap * kBlock * sizeof(ResampleSample));
    if (rs.devSamples) {
      GA_HIP(hipStreamSynchronize(stream));   // (rare: the table doubles)
      GA_HIP(hipMemcpy(nw, rs.devSamples, (size_t)rs.devBlocks * kBlock * sizeof(ResampleSample), hipMemcpyDeviceToDevice));
      dfree(rs.devSamples, (size_t)rs.devCapBlocks * kBlock * sizeof(ResampleSample));
    }
    rs.devSamples = nw;
    rs.devCapBlocks = cap;
  }
  const size_t bytes = rs.pending.size() * sizeof(ResampleSample);
  const size_t off = ex.plan.put(rs.pending.data(), bytes);
  ResampleSample* dst = rs.devSamples + (size_t)rs.devBlocks * kBlock;
  hipStream_t st = stream;
  ex.plan.add(LK_OTHER, [=](uint8_t* base) { GA_HIP(hipMemcpyAsync(dst, base + off, bytes, hipMemcpyDeviceToDevice, st)); });
  rs.devBlocks = need;
  rs.pending.clear();
  return true;
}

// AudioBufferSourceNode.Process (AudioBufferSourceNode.cs:150-260): zero-copy windows, loop walks, resampler jobs, general replay
void Context::planBufferSource(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  const std::vector<int>& srcIds = k.r.srcIds; const std::vector<SrcPlanOut>& srcPlans = k.r.srcPlans;
  if (ns.srcPhase != SRC_PLAY) return;  // silent: ZERO views
  PlayBuf& pb = *buffers[ns.srcBuf];
  SrcGeom g = sourceGeom(*this, nd, pb);
  if (nd.gsr) {  // general replay: one host-made descriptor per block
    if (!nd.gsrUploaded) {
      nd.gsrDevOff = ex.plan.putv(nd.gsrBlocks);
      nd.gsrUploaded = true;
    }
    for (int ch = 0; ch < pb.channels; ch++) {
      GsrJob gj;
      gj.buf = pb.dev + (size_t)ch * pb.stride;
      gj.out = ex.nodeOut(ns.id, ch);
      gj.desc_off = nd.gsrDevOff + (uint64_t)ns.srcBlk * sizeof(GsrBlock);
      gj.b0 = sg.b0;
      gj.nblocks = nb;
      gj.loop_start = g.loopStartFrame;
      gj.loop_end = g.loopEndFrame;
      gj.loop = nd.loop ? 1 : 0;
      gj.pad_ = 0;
      ex.gsrJobs.push_back(gj);
      ov[ch] = gj.out;
    }
  } else if (g.effectiveRate == 1.0 && (ns.srcPos < 0 || (nd.loop ? g.loopEndFrame : ns.srcPos + nf) > pb.length)) {
    fail(GA_ERR_DEVICE, "internal: source window beyond the buffer");
  } else if (g.effectiveRate == 1.0 && !nd.loop) {
    // zero-copy: the node's output for these blocks IS the buffer (AudioBufferSourceNode.cs:186-222)
    for (int ch = 0; ch < pb.channels; ch++) ov[ch] = pb.dev + (size_t)ch * pb.stride + ns.srcPos - f0;
  } else if (g.effectiveRate == 1.0 && ns.srcPos < g.loopEndFrame && ns.srcPos + nf <= g.loopEndFrame) {
    // looping, but these blocks do not reach the loop end: still a plain window of the buffer (zero-copy)
    for (int ch = 0; ch < pb.channels; ch++) ov[ch] = pb.dev + (size_t)ch * pb.stride + ns.srcPos - f0;
  } else if (g.effectiveRate == 1.0) {
    for (int ch = 0; ch < pb.channels; ch++) {
      LoopJob lj;
      lj.buf = pb.dev + (size_t)ch * pb.stride;
      lj.out = ex.nodeOut(ns.id, ch);
      lj.pos0 = ns.srcPos;  // map() below handles positions beyond loopEnd
      lj.loop_start = g.loopStartFrame;
      lj.loop_end = g.loopEndFrame;
      lj.f0 = f0;
      lj.n = nf;
      ex.loopJobs.push_back(lj);
      ov[ch] = lj.out;
    }
  } else {
    Resampler& rs = resamplerFor(*this, g.effectiveRate);
    if (rs.devOffset < 0) {
      rs.devOffset = (int)ex.traj.size();
      ex.traj.insert(ex.traj.end(), rs.blocks.begin(), rs.blocks.end());
    }
    int64_t avail = g.durationEndFrame - nd.rsStartPos;
    // a partial block (input ran out) is its own one-block segment with a custom trajectory entry
    int traj0 = rs.devOffset + (int)ns.srcBlk;
    // (this source's plan by its index: a scan of the chunk's sources per source was 4096 x 4096 comparisons per chunk of config 4 --
    // 60 % of the planning time at 28,672 nodes)
    const int sk = ns.id < (int)k.r.srcIndex.size() ? k.r.srcIndex[ns.id] : -1;
    if (sk >= 0 && srcIds[sk] == ns.id && srcPlans[sk].partialBlock == sg.b0) {
      ResampleBlock rb = rs.blocks[ns.srcBlk];
      rb.produced = srcPlans[sk].partialProduced;
      traj0 = (int)ex.traj.size();
      ex.traj.push_back(rb);
    }
    {  // host-side bound of the device reads of this job: a wrong plan must be an error, not a GPU fault
      const bool partial = ex.traj[traj0].produced != kBlock;
      if (nd.rsStartPos < 0 || avail < 0 || nd.rsStartPos + avail > pb.length ||
          (!partial && rs.blocks[ns.srcBlk + nb].consumed > avail))
        fail(GA_ERR_DEVICE, "internal: resampler job reads beyond the source buffer");
    }
    // Full blocks of the shared trajectory: one lane per OUTPUT sample from the trajectory's per-sample table (resample_fast_kernel).
    // The table lives on the device and only grows; what extend() produced since the last upload rides in this chunk's tables.
    if (resampleFast && traj0 == rs.devOffset + (int)ns.srcBlk && rs.samplesOk && (int64_t)ns.srcBlk + nb <= (int64_t)rs.blocks.size() - 1 &&
        ensureResampleSamples(ex, rs, ns.srcBlk + nb)) {
      for (int ch = 0; ch < pb.channels; ch++) {
        ResampleFastJob fj;
        fj.buf = pb.dev + (size_t)ch * pb.stride;
        fj.out = ex.nodeOut(ns.id, ch);
        fj.samples = rs.devSamples + (size_t)ns.srcBlk * kBlock;
        fj.start_pos = nd.rsStartPos;
        fj.b0 = sg.b0;
        fj.nblocks = nb;
        ex.rsFastJobs.push_back(fj);
        ov[ch] = fj.out;
      }
      return;
    }
    for (int ch = 0; ch < pb.channels; ch++) {
      ResampleJob rj;
      rj.buf = pb.dev + (size_t)ch * pb.stride;
      rj.out = ex.nodeOut(ns.id, ch);
      rj.start_pos = nd.rsStartPos;
      rj.avail = avail;
      rj.traj0 = traj0;
      rj.rate = g.effectiveRate;
      rj.b0 = sg.b0;
      rj.nblocks = nb;
      ex.rsJobs.push_back(rj);
      ov[ch] = rj.out;
    }
  }
}

// AudioStreamSourceNodeBase.Process (AudioStreamSourceNodeBase.cs:132-301), replayed by the host (streamReplay)
void Context::planStreamSource(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  if (ns.outSilent) return;   // ProduceSilence / nothing rendered: cleared buffer
  if (!nd.stUploaded) {
    nd.stBlocksOff = ex.plan.putv(nd.stBlocks);
    nd.stPiecesOff = ex.plan.putv(nd.stPieces);
    nd.stSegsOff = ex.plan.putv(nd.stSegs);
    nd.stUploaded = true;
  }
  if (!nd.stWin[0]) {
    nd.stWin[0] = (float*)dalloc(32 * 4 * sizeof(float));
    nd.stWin[1] = (float*)dalloc(32 * 4 * sizeof(float));
    GA_HIP(hipMemsetAsync(nd.stWin[0], 0, 32 * 4 * sizeof(float), stream));
    GA_HIP(hipMemsetAsync(nd.stWin[1], 0, 32 * 4 * sizeof(float), stream));
  }
  for (int ch = 0; ch < ns.outCh && ch < 32; ch++) {
    StreamJob sj{};
    sj.out = ex.nodeOut(ns.id, ch);
    sj.win_in = nd.stWin[nd.stWinCur] + 4 * ch;
    sj.win_out = nd.stFed ? nd.stWin[nd.stWinCur ^ 1] + 4 * ch : nullptr;   // every job of the chunk writes the same end state
    sj.blocks_off = nd.stBlocksOff;
    sj.pieces_off = nd.stPiecesOff;
    sj.segs_off = nd.stSegsOff;
    sj.b0 = sg.b0;
    sj.nblocks = nb;
    for (int k = 0; k < 4; k++) {
      sj.wend[k] = nd.stWend[k];
      sj.wend_seg[k] = nd.stWendSeg[k];
    }
    sj.ch = ch;
    ex.streamJobs.push_back(sj);
    ov[ch] = sj.out;
  }
}

// GainNode.Process (GainNode.cs:36-80)
void Context::planGain(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  const float* gmod = nullptr;   // audio-rate modulation of gain: mixed to 1 channel (AudioParam.cs:68-70,123-135)
  if (!ns.pins.empty() && !ns.pins[0].silent) gmod = ex.resolveInSeg((int)si, ns.id, -1, ns.pins[0], false, nullptr)[0];
  auto iv = ex.resolveInput((int)si, ns, 0, false, nullptr);
  if (ns.ins[0].silent) return;  // cleared output (GainNode.cs:41-46)
  // a constant gain of exactly 1 (every GainNode's default: buses, splits and merges of effect chains) multiplies nothing:
  // x * 1.0f == x bit for bit, so the output IS the (mixed) input -- no launch, no pass over the samples
  const bool constant = !gmod && !nd.params[0].curve;
  const bool unity = constant && nd.params[0].value == 1.0f && gainPassThrough;
  // any other constant gain with ONE consumer connection: the consumer's mix multiplies (Exec::scaleOf) -- no pass of its own
  const bool fold = constant && !unity && gainFold && nd.outputs.size() == 1 && nd.outputs[0].connectedInputs.size() == 1;
  if (fold) ex.setScale((int)si, ns.id, nd.params[0].value);
  // a gain on a timeline (no modulation) with one consumer INPUT that mixes it channel by channel: the curve goes with the views
  // (Exec::curveOf).  Not in front of a down-mix (its kernel takes constants only), not for a node some consumer reads one block late.
  bool foldCurve = false;
  if (!gmod && nd.params[0].curve && gainFold && !nd.staleProducer && nd.outputs.size() == 1 && nd.outputs[0].connectedInputs.size() == 1) {
    const InRef& to = nd.outputs[0].connectedInputs[0];
    const NodeSeg* cs = to.input >= 0 ? k.segNode.find(to.node) : nullptr;   // (a consumer of this stage: same convolver depth)
    if (cs && to.input >= 0 && to.input < (int)cs->ins.size()) {
      const int dstCh = cs->ins[to.input].bufCh;
      foldCurve = !(ns.outCh > 1 && dstCh == 1);
    }
  }
  if (foldCurve) ex.setCurve((int)si, ns.id, nd.params[0].curve);
  for (int ch = 0; ch < ns.outCh; ch++) {
    if (!iv[ch]) continue;
    if (unity || fold || foldCurve) {
      ov[ch] = iv[ch];
      continue;
    }
    GainJob gj;
    gj.in = iv[ch];
    gj.out = ex.nodeOut(ns.id, ch);
    gj.curve = nd.params[0].curve;
    gj.mod = gmod;
    gj.vmin = nd.params[0].minv;
    gj.vmax = nd.params[0].maxv;
    gj.gain = nd.params[0].value;
    gj.f0 = f0;
    gj.n = nf;
    ex.gainJobs.push_back(gj);
    ov[ch] = gj.out;
  }
}

// BiQuadFilterNode.Process (BiQuadFilterNode.cs:96-143): automated parameters, fused constant-coefficient cascades, cascades split along time
void Context::planBiquad(NodePlanCtx& k) {
  Exec& ex = k.ex; const size_t si = k.si; const Segment& sg = k.sg; const NodeSeg& ns = k.ns; NodeS& nd = k.nd; Views& ov = k.ov;
  const int64_t f0 = k.f0, nf = k.nf, nb = k.nb; (void)si; (void)sg; (void)nb; (void)nd; (void)f0; (void)nf;
  const DenseSeg& segNode = k.segNode; const DenseInt& absorbedBy = k.absorbedBy; const int levelBqHeads = k.levelBqHeads;
  if (!ns.bqActive) {  // silent input: cleared output, state frozen (BiQuadFilterNode.cs:103-108)
    ex.resolveInput((int)si, ns, 0, false, nullptr);
    return;
  }
  if (ns.bqDynamic) {  // automated parameters: per-sample coefficient refresh on the device
    auto iv = ex.resolveInput((int)si, ns, 0, false, nullptr);
    ensureBiquadState(nd);
    BiquadDynJob dj{};
    for (int ch = 0; ch < ns.outCh && ch < 32; ch++) {
      dj.in[ch] = iv[ch];
      dj.out[ch] = ex.nodeOut(ns.id, ch);
      ov[ch] = dj.out[ch];
    }
    dj.fcurve = ex.paramView((int)si, ns, 0);
    dj.qcurve = ex.paramView((int)si, ns, 1);
    dj.gcurve = ex.paramView((int)si, ns, 2);
    dj.fval = nd.params[0].value;
    dj.qval = nd.params[1].value;
    dj.gval = nd.params[2].value;
    dj.channels = ns.outCh;
    dj.filter_type = nd.filterType;
    if (nd.coefOnDevice && nd.coefDirty) {   // the Type setter ran while the coefficient state lives on the device: hand the flag over
      dj.filter_type |= 0x100;
      nd.coefDirty = false;
    }
    dj.nyquist = sampleRate / 2.f;
    dj.sample_rate = (float)sampleRate;
    dj.state = nd.bqDyn;
    dj.b0 = sg.b0;
    dj.nblocks = nb;
    if (!nd.coefOnDevice) {  // hand the host-side coefficient state (constant-parameter runs) to the device once
      BiquadDynState init{};
      init.b0 = nd.b0; init.b1 = nd.b1; init.b2 = nd.b2; init.a1 = nd.a1; init.a2 = nd.a2;
      init.dirty = nd.coefDirty ? 1 : 0;
      GA_HIP(hipMemcpyAsync(nd.bqDyn, &init, 24, hipMemcpyHostToDevice, stream));
      GA_HIP(hipStreamSynchronize(stream));
      nd.coefOnDevice = true;
      deviceStateNodes.push_back(ns.id);
    }
    ex.bqDynJobs.push_back(dj);
    return;
  }
  if (absorbedBy.get(ns.id) >= 0) return;  // evaluated inside the cascade job of a downstream biquad
  // chain head ... this node: biquads connected output -> single input with equal channel counts
  SmallVec<const NodeSeg*, kMaxBiquadSections> chain{&ns};
  while (true) {
    const NodeSeg* h = chain.front();
    if (h->ins[0].terms.size() != 1) break;
    int up = h->ins[0].terms[0].node;
    if (absorbedBy.get(up) != h->id) break;
    chain.insert_front(segNode.find(up));
  }
  auto iv = ex.resolveInput((int)si, *chain.front(), 0, false, nullptr);
  for (const NodeSeg* cn : chain) {
    NodeS& cnd = *nodes[cn->id];
    ensureBiquadState(cnd);
  }
  // pieces along time (ga_kernels.hpp, BiquadScanJob): as many as keep every lane of the chip busy, each >= 1024 frames;
  // mode 1: only cascades whose float32 rounding noise is so small that a different rounding stays inside the budget
  int G = 1;
  float coefs[5 * kMaxBiquadSections];
  if (biquadTimeSplit && nf >= biquadSplitMinFrames) {
    const int64_t lanes = 64 * 1024, heads = std::max(levelBqHeads, 1);
    G = (int)std::max<int64_t>(1, std::min<int64_t>({(lanes + heads - 1) / heads, nf / 1024, 256}));
    int q = 0;
    for (const NodeSeg* cn : chain) {
      coefs[5 * q] = cn->b0; coefs[5 * q + 1] = cn->b1; coefs[5 * q + 2] = cn->b2; coefs[5 * q + 3] = cn->a1; coefs[5 * q + 4] = cn->a2;
      q++;
    }
    if (biquadTimeSplit == 1 && biquadDeviation(coefs, (int)chain.size()) > biquadSplitMaxDeviation) G = 1;
  }
  int64_t K = G > 1 ? ((nf + G - 1) / G + 3) / 4 * 4 : nf;
  if (G > 1) G = (int)((nf + K - 1) / K);
  if (G > 1 && ex.bqG == 0) {
    ex.bqG = G;
    ex.bqK = K;
  }
  if (G > 1 && (G != ex.bqG || K != ex.bqK)) G = 1;   // (one cut per level: the pieces of a level are expanded by one launch)
  const std::vector<float>* AK = G > 1 ? &biquadTransition(coefs, (int)chain.size(), K).M : nullptr;
  for (int ch = 0; ch < ns.outCh; ch++) {
    BiquadJob bj;
    bj.in = iv[ch] ? iv[ch] : zeros;
    bj.out = ex.nodeOut(ns.id, ch);
    bj.sec0 = (int)ex.bqSecs.size();
    bj.nsec = (int)chain.size();
    bj.f0 = f0;
    bj.n = nf;
    bj.state = nullptr;
    for (const NodeSeg* cn : chain) {
      BiquadSection sc;
      sc.b0 = cn->b0; sc.b1 = cn->b1; sc.b2 = cn->b2; sc.a1 = cn->a1; sc.a2 = cn->a2;
      sc.pad_ = 0.f;
      sc.state = nodes[cn->id]->bqState + 2 * ch;
      ex.bqSecs.push_back(sc);
    }
    ov[ch] = bj.out;
    if (G <= 1) {
      ex.bqJobs[bj.nsec].push_back(bj);
      continue;
    }
    stats.biquad_split_cascades++;
    float* scratch = bqSplitAlloc((size_t)(G - 1) * bj.nsec * 2);
    ex.bqMats[bj.nsec].push_back(AK);
    ex.bqScans[bj.nsec].push_back(BiquadScanJob{bj.in, bj.out, 0, scratch, bj.sec0, bj.nsec, f0, nf});
  }
}

// pass 6 (per convolver depth d): every segment, level by level -- node launches are batched per (level, type)
void Context::chunkPlanNodes(ChunkRun& r, int d) {
  Context& c_ = *this; (void)c_;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
    for (size_t si = 0; si < segs.size(); si++) {
      Segment& sg = segs[si];
      if (ex.outViews[si].empty()) {
        if (!viewsPool.empty()) {
          ex.outViews[si] = std::move(viewsPool.back());
          viewsPool.pop_back();
          for (Views& v : ex.outViews[si]) v.clear();
        }
        ex.outViews[si].resize(nodes.size());
      }
      const int64_t f0 = sg.b0 * kBlock, nf = (sg.b1 - sg.b0) * kBlock, nb = sg.b1 - sg.b0;
      // nodes of this stage ordered by level
      // (a stable counting sort: with tens of thousands of nodes a comparison sort that chases two node pointers per comparison
      // was a quarter of the host time of a chunk)
      std::vector<const NodeSeg*> todo;
      {
        std::vector<std::pair<int, const NodeSeg*>> mine;
        std::vector<int> count(maxLevel + 2, 0);
        for (const NodeSeg& ns : sg.nodes) {
          const NodeS& nd = *nodes[ns.id];
          if (nd.depth != d) continue;
          const int lv = std::min(std::max(nd.level, 0), maxLevel);
          mine.push_back({lv, &ns});
          count[lv + 1]++;
        }
        for (int lv = 0; lv <= maxLevel; lv++) count[lv + 1] += count[lv];
        todo.resize(mine.size());
        for (auto& m : mine) todo[count[m.first]++] = m.second;
      }
      // biquad cascade fusion: A is absorbed by B when B's only input term is A, A's only consumer is B and both run
      // (non-silent) with the same channel count; chains are capped at kMaxBiquadSections
      // (dense tables indexed by node id, validated by a per-(stage, segment) stamp: no hashing on the per-node path)
      if (fuseStamp.size() < nodes.size()) {
        fuseStamp.assign(nodes.size(), 0);
        fuseSeg.assign(nodes.size(), nullptr);
        fuseAbs.assign(nodes.size(), -1);
        fuseLen.assign(nodes.size(), 0);
      }
      const uint32_t stamp = ++fuseEpoch;
      DenseSeg segNode{fuseStamp, fuseSeg, stamp};
      for (const NodeSeg* nsp : todo) {
        fuseStamp[nsp->id] = stamp;
        fuseSeg[nsp->id] = nsp;
        fuseAbs[nsp->id] = -1;
        fuseLen[nsp->id] = 0;
      }
      DenseInt absorbedBy{fuseStamp, fuseAbs, stamp, -1}, chainLen{fuseStamp, fuseLen, stamp, 0};
      for (const NodeSeg* nsp : todo) {
        const NodeSeg& b_ = *nsp;
        if (nodes[b_.id]->type != GA_NODE_BIQUAD || !b_.bqActive || b_.bqDynamic) continue;
        fuseLen[b_.id] = 1;
        if (b_.ins[0].terms.size() != 1) continue;
        const TermS& t = b_.ins[0].terms[0];
        const NodeSeg* ia = segNode.find(t.node);
        if (!ia) continue;
        const NodeSeg& a_ = *ia;
        NodeS& an = *nodes[a_.id];
        if (an.type != GA_NODE_BIQUAD || !a_.bqActive || a_.bqDynamic || t.ch != b_.ins[0].bufCh || a_.outCh != b_.outCh) continue;
        if (an.outputs[0].connectedInputs.size() != 1) continue;
        int la_ = chainLen.get(a_.id) ? chainLen.get(a_.id) : 1;
        if (la_ >= kMaxBiquadSections) continue;
        fuseAbs[a_.id] = b_.id;
        fuseLen[b_.id] = la_ + 1;
      }
      int curLevel = -1, levelBqHeads = 0;
      if (d == 0 && topoHasCycles && cycleBlocks > 1) {   // the readers of the DelayNodes at which this chunk's loops are cut: sources
        for (const NodeSeg& ns : sg.nodes) {
          NodeS& nd = *nodes[ns.id];
          if (nd.type != GA_NODE_DELAY || !nd.delaySplit) continue;
          auto& ov = ex.outViews[si][ns.id];
          ov.assign(std::max(ns.outCh, 1), nullptr);
          NodePlanCtx k{r, ex, si, sg, f0, nf, nb, ns, nd, ov, segNode, absorbedBy, 0};
          k.delayPhase = 1;
          planDelay(k);
        }
        ex.flushLevel();
      }
      for (size_t ti = 0; ti < todo.size(); ti++) {
        const NodeSeg* nsp = todo[ti];
        if (ti + 4 < todo.size()) {   // (the sweep is bound by cache misses on the node records)
          const char* nx = (const char*)nodes[todo[ti + 4]->id].get();
          __builtin_prefetch(nx);
          __builtin_prefetch(nx + 64);
          __builtin_prefetch(nx + 128);
        }
        const NodeSeg& ns = *nsp;
        NodeS& nd = *nodes[ns.id];
        if (nd.level != curLevel) {
          ex.flushLevel();
          curLevel = nd.level;
          levelBqHeads = 0;   // constant-coefficient cascade outputs of this level (all levels' biquad launches are separate)
          for (size_t tj = ti; tj < todo.size() && nodes[todo[tj]->id]->level == curLevel; tj++) {
            const NodeSeg& o = *todo[tj];
            if (nodes[o.id]->type == GA_NODE_BIQUAD && o.bqActive && !o.bqDynamic && absorbedBy.get(o.id) < 0) levelBqHeads += std::max(o.outCh, 1);
          }
        }
        auto& ov = ex.outViews[si][ns.id];
        const bool cutDelay = nd.type == GA_NODE_DELAY && nd.delaySplit && topoHasCycles && cycleBlocks > 1;   // (its reader set the views)
        if (!cutDelay) ov.assign(nd.type == GA_NODE_CHANNEL_SPLITTER ? (int)nd.outputs.size() : std::max(ns.outCh, 1), nullptr);
        NodePlanCtx k{r, ex, si, sg, f0, nf, nb, ns, nd, ov, segNode, absorbedBy, levelBqHeads};
        if (cutDelay) k.delayPhase = 2;
        switch (nd.type) {
          case GA_NODE_CHANNEL_SPLITTER: {   // zero-copy: output o IS channel o of the mixed input
            if (!ns.outMask) break;
            auto iv = ex.resolveInput((int)si, ns, 0, false, nullptr);
            for (int o = 0; o < (int)nd.outputs.size(); o++)
              if ((ns.outMask >> o) & 1) ov[o] = iv[o];
            break;
          }
          case GA_NODE_CHANNEL_MERGER: {     // zero-copy: channel i IS channel 0 of input i
            for (int i = 0; i < (int)ns.ins.size(); i++) {
              if (!((ns.outMask >> i) & 1)) continue;
              auto iv = ex.resolveInput((int)si, ns, i, false, nullptr);
              ov[i] = iv.empty() ? nullptr : iv[0];
            }
            break;
          }
          case GA_NODE_CONSTANT_SOURCE: planConstantSource(k); break;
          case GA_NODE_OSCILLATOR: planOscillator(k); break;
          case GA_NODE_DELAY: planDelay(k); break;
          case GA_NODE_STEREO_PANNER: planStereoPanner(k); break;
          case GA_NODE_BUFFER_SOURCE: planBufferSource(k); break;
          case GA_NODE_STREAM_SOURCE: planStreamSource(k); break;
          case GA_NODE_GAIN: planGain(k); break;
          case GA_NODE_BIQUAD: planBiquad(k); break;
          case GA_NODE_CONVOLVER: {
            auto iv = ex.resolveInput((int)si, ns, 0, false, nullptr);
            if (!nd.ir) break;  // no IR: cleared output (ConvolverNode.cs:107-119)
            ex.convIn[ns.id][si] = iv;
            // formulation D: the outputs of a fused group are summed as spectra; the sum is the LEADER's output, the other
            // members hand their consumer a null (= contributes nothing) view (Context::planCoarseFusion)
            if (nd.convPath == 4 && nd.dLeader >= 0 && nd.dLeader != ns.id) break;
            // Nothing has reached this convolver since its delay line was created: the reference's partition sum is a sum of
            // exact zeros (PartitionedConvolver.cs:154-223), and consumers that compare values -- StereoPannerNode's `pan !=
            // _lastPan` (StereoPannerNode.cs:92-99), DelayNode's (int)(delayTime * sampleRate) -- see that.  The transform
            // formulations (C, D) leave ~1e-9 of circular rounding in front of an onset inside the same window, so the blocks
            // before the onset are served from the zero page instead of the output slab (fuzz session 42867).  The leader of a
            // fused group carries the other members' sum and keeps its slab.
#ifdef GA_EXPERIMENTS
            static const bool noZeroPage = getenv("GA_NO_ZERO_PAGE") != nullptr;   // (to show that the regression tests catch the defect)
#else
            constexpr bool noZeroPage = false;
#endif
            if (!noZeroPage && ns.outZero && !(nd.convPath == 4 && nd.dGroupSize > 1)) {
              for (int ch = 0; ch < ns.outCh; ch++) ov[ch] = zeros;
              break;
            }
            for (int ch = 0; ch < ns.outCh; ch++) ov[ch] = ex.nodeOut(ns.id, ch);
            break;
          }
          case GA_NODE_DESTINATION: {
            // the destination aliases its input buffer (AudioDestinationNode.cs:44-50): mix straight into the bus
            SmallVec<float*, 4> forced((size_t)std::max(ns.ins[0].bufCh, 1), nullptr);
            for (int ch = 0; ch < ns.ins[0].bufCh && ch < (int)busSlabs.size(); ch++) forced[ch] = busTarget[ch] ? busTarget[ch] : busSlabs[ch];
            ex.resolveInput((int)si, ns, 0, true, forced.data());
            break;
          }
          default: break;
        }
      }
      ex.flushLevel();
    }
}

// what the passes of Context::chunkPlanConvolvers share (one convolver depth of one chunk)
struct ConvGroupLess {   // ordered by (IR buffer, IR channel) so groups fed by the same inputs are adjacent
  bool operator()(const ConvGroup* a, const ConvGroup* b) const {
    if (a->ir.get() != b->ir.get()) return a->ir.get() < b->ir.get();
    if (a->irCh != b->irCh) return a->irCh < b->irCh;
    return a->depth < b->depth;
  }
};
struct ConvPlanCtx {
  std::map<ConvGroup*, std::vector<std::pair<int, int>>, ConvGroupLess> active;   // group -> (node, slot)
  std::vector<const float*> prevIns;
  int prevP = -1, prevRp = -1, prevRows = -1;
  std::vector<int> bNodes, dNodes;
  std::unordered_map<int, std::array<float*, 4>> tsTemps;   // true-stereo temp outputs per node
};

// formulation A: convolvers that share an impulse-response channel run as one group per (impulse response, channel) -- the
// banded-Toeplitz matrix-core kernel or the block-axis transforms over all their rows (PartitionedConvolver.cs:104-223)
void Context::planConvolversShared(ChunkRun& r, int d, ConvPlanCtx& k) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  (void)d; (void)topo;
  auto& active = k.active; auto& tsTemps = k.tsTemps; auto& prevIns = k.prevIns; int& prevP = k.prevP; int& prevRp = k.prevRp; int& prevRows = k.prevRows;
  for (auto& kv : active) {
    ConvGroup& g = *kv.first;
    const int P = g.P, hist = P - 1;
    const int nrows = (int)g.rows.size();
    std::vector<ConvRowIO> rio(nrows, ConvRowIO{nullptr, nullptr});
    for (auto& ns_ : kv.second) {
      NodeS& nd = *nodes[ns_.first];
      const int slot = ns_.second;
      const int idx = nd.convRows[slot].idx;
      // which input channel feeds this row: discrete -> slot ; true stereo -> L,L,R,R for h0,h1,h2,h3 (ConvolverNode.cs:127-151)
      const int inCh = nd.isTrueStereo ? (slot >> 1) : slot;
      const Exec::ConvInRow ci = ex.convIn[ns_.first];
      const float* stable = nullptr;
      bool same = true, first = true;
      for (size_t si = 0; si < segs.size(); si++) {
        const float* v = (ci[si].empty() || inCh >= (int)ci[si].size()) ? nullptr : ci[si][inCh];
        if (first) { stable = v; first = false; } else if (v != stable) same = false;
      }
      const float* in = stable;
      if (!same) {  // materialise: per segment copy / zero fill into a row slab
        float* slab = getSlab(*this);
        for (size_t si = 0; si < segs.size(); si++) {
          const float* v = (ci[si].empty() || inCh >= (int)ci[si].size()) ? nullptr : ci[si][inCh];
          MixJob mj;
          mj.out = slab;
          mj.term0 = (int)ex.terms.size();
          mj.nterms = v ? 1 : 0;
          mj.f0 = segs[si].b0 * kBlock;
          mj.n = (segs[si].b1 - segs[si].b0) * kBlock;
          if (v) {
            ex.terms.push_back(v);
            ex.noteAlign(v, mj.f0);
          }
          ex.mixJobs.push_back(mj);
        }
        in = slab;
      }
      float* out;
      if (nd.isTrueStereo) {
        out = getSlab(*this);  // temp1 / temp2, summed below (ConvolverNode.cs:137-143)
      } else {
        out = ex.nodeOut(ns_.first, slot);
      }
      rio[idx] = ConvRowIO{in, out};
      if (nd.isTrueStereo) {
        auto it = tsTemps.find(ns_.first);
        if (it == tsTemps.end()) it = tsTemps.emplace(ns_.first, std::array<float*, 4>{nullptr, nullptr, nullptr, nullptr}).first;
        it->second[slot] = out;
      }
    }
    ex.flushLevel();
    const int rp = g.rp;
    const int ty = (int)roundup(n, 64);
    const int tx = ty + P + 128;
    ConvPlanes pl{(float*)planes[0].p, (float*)planes[1].p, (float*)planes[2].p, (float*)planes[3].p, tx, ty, rp};
    size_t rioOff = ex.plan.putv(rio);
    hipStream_t st = stream;
    Twiddles tw{w128, w256};
    const int nn = (int)n;
    float* hR = g.histR;
    float* hI = g.histI;
    const bool hz = g.histZero;
    const float* hr = g.ir->hr + (size_t)g.irCh * kBins * P;
    const float* hi = g.ir->hi + (size_t)g.irCh * kBins * P;
    float* ovIn = g.overlap[g.ovCur];
    float* ovOut = g.overlap[g.ovCur ^ 1];
    g.ovCur ^= 1;
    g.histZero = false;
    // forward spectra depend only on the inputs: a group fed by exactly the same signals as the previous one (e.g. the
    // channels of one stereo IR behind mono voices) reuses the X rows that are still in the scratch planes
    std::vector<const float*> ins(nrows);
    for (int r = 0; r < nrows; r++) ins[r] = rio[r].in;
    const bool skipFwd = (prevP == P && prevRp == rp && prevRows == nrows && prevIns == ins);
    prevIns = ins;
    prevP = P;
    prevRp = rp;
    prevRows = nrows;
    ex.plan.add(LK_FFT, [=](uint8_t* base) {
      // frequency-domain delay line of the previous chunk(s) in front of this chunk's spectra
      if (hist > 0) {
        launch_plane_copy(st, pl.xr, tx, 0, hz ? nullptr : hR, hist, 0, hist, rp);
        launch_plane_copy(st, pl.xi, tx, 0, hz ? nullptr : hI, hist, 0, hist, rp);
      }
      // rows beyond this chunk that the banded MAC may touch for its (discarded) padded outputs
      int tail = std::min(tx - (hist + nn), 256);
      launch_plane_copy(st, pl.xr, tx, hist + nn, nullptr, 0, 0, tail, rp);
      launch_plane_copy(st, pl.xi, tx, hist + nn, nullptr, 0, 0, tail, rp);
      if (!skipFwd) launch_rfft_fwd(st, (const ConvRowIO*)(base + rioOff), nrows, nn, hist, pl, tw);
    });
    ex.plan.add(LK_MAC, [=](uint8_t*) { launch_spectral_mac_shared(st, pl, hr, hi, P, nn, nrows); });
    ex.plan.add(LK_FFT, [=](uint8_t* base) {
      launch_irfft_ola(st, (const ConvRowIO*)(base + rioOff), nrows, nn, pl, ovIn, ovOut, tw);
      if (hist > 0) {  // keep the last P-1 spectra for the next chunk (the FDL, PartitionedConvolver.cs:122-128)
        launch_plane_copy(st, hR, hist, 0, pl.xr, tx, nn, hist, rp);
        launch_plane_copy(st, hI, hist, 0, pl.xi, tx, nn, hist, rp);
      }
    });
    stats.mac_flops_total += 8.0 * P * kBins * (double)kv.second.size() * (double)n;
    // streaming-formulation bytes: per channel-instance per block FDL read + write + input, IR once per block per channel
    stats.mac_bytes_total += ((double)P * kBins * 8.0 + kBins * 8.0 + 512.0) * (double)kv.second.size() * (double)n +
                             (double)P * kBins * 8.0 * (double)n;
    stats.mac_launches += 1;
  }
}

// formulations B / C: nodes with an impulse response of their own (per-node planes; block-axis FFT segments or the direct sum)
void Context::planConvolversPrivate(ChunkRun& r, int d, ConvPlanCtx& k, bool refOrder) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  (void)d; (void)topo;
  const std::vector<int>& bNodes = k.bNodes; auto& tsTemps = k.tsTemps;
  const int hist = (int)roundup(bHistMax, 4);   // plane time origin, 16-byte aligned rows
  const int txb = hist + (int)roundup(n, 16) + 16, tyb = (int)roundup(n, 256);
  // rows of this depth start after the rows of the depths before it: a node's spectra stay intact for the next chunk
  const size_t rowX0 = bRowX, rowY0 = bRowY;
  ConvPlanesB plb{xPlane(bPairWrite, 0) + rowX0 * kBins * txb, xPlane(bPairWrite, 1) + rowX0 * kBins * txb,
                  (float*)planesB[2].p + rowY0 * kBins * tyb, (float*)planesB[3].p + rowY0 * kBins * tyb, txb, tyb};
  std::vector<ConvRowIO> xrows, yrows;
  std::vector<ConvSetB> sets;
  std::map<int, std::vector<ConvSetC>> setsC;   // by P: launches per distinct segment length
  struct SetTaps { IrSpectra* ir; int slot[16]; };
  std::map<int, std::vector<SetTaps>> setsCTaps;   // which taps spectra each column of a set needs (filled per FFT length)
  std::vector<HistJobB> restore;
  std::vector<const float*> ovIn;
  std::vector<float*> ovOut;
  double flops = 0;
  std::vector<const float*> chIn;   // (scratch vectors live outside the node loop: a thousand convolvers per chunk)
  std::vector<float*> slotOut;
  std::vector<int> cols;
  for (int id : bNodes) {
    NodeS& nd = *nodes[id];
    const int P = nd.ir->P, h = P - 1;
    const Exec::ConvInRow ci = ex.convIn[id];
    // chunk-long input pointer of every input channel (stable view, or a materialised copy)
    chIn.assign(nd.bInCh, nullptr);
    for (int c = 0; c < nd.bInCh; c++) {
      const float* stable = nullptr;
      bool same = true, first = true;
      for (size_t si = 0; si < segs.size(); si++) {
        const float* v = (ci[si].empty() || c >= (int)ci[si].size()) ? nullptr : ci[si][c];
        if (first) { stable = v; first = false; } else if (v != stable) same = false;
      }
      if (same) {
        chIn[c] = stable;
      } else {
        float* slab = getSlab(*this);
        for (size_t si = 0; si < segs.size(); si++) {
          const float* v = (ci[si].empty() || c >= (int)ci[si].size()) ? nullptr : ci[si][c];
          MixJob mj;
          mj.out = slab;
          mj.term0 = (int)ex.terms.size();
          mj.nterms = v ? 1 : 0;
          mj.f0 = segs[si].b0 * kBlock;
          mj.n = (segs[si].b1 - segs[si].b0) * kBlock;
          if (v) {
            ex.terms.push_back(v);
            ex.noteAlign(v, mj.f0);
          }
          ex.mixJobs.push_back(mj);
        }
        chIn[c] = slab;
      }
    }
    bool allSame = true;
    for (int c = 1; c < nd.bInCh; c++) allSame = allSame && (chIn[c] == chIn[0]);
    const size_t hstride = (size_t)kBins * std::max(h, 1);
    if (nd.bShared && !allSame) {
      // the channels start to differ: every channel inherits the (so far common) history of channel 0
      if (!nd.bHistZero && h > 0 && nd.bHistPlane < 0)   // (a plane-resident shared row is simply read by every channel)
        for (int c = 1; c < nd.bInCh; c++) {   // (plan entries: ordered with the chunk's launches)
          float *dr = nd.bHistR + c * hstride, *di = nd.bHistI + c * hstride;
          const float *sr = nd.bHistR, *sim = nd.bHistI;
          hipStream_t st = stream;
          ex.plan.add(LK_OTHER, [=](uint8_t*) {
            GA_HIP(hipMemcpyAsync(dr, sr, hstride * 4, hipMemcpyDeviceToDevice, st));
            GA_HIP(hipMemcpyAsync(di, sim, hstride * 4, hipMemcpyDeviceToDevice, st));
          });
        }
      nd.bShared = false;
    }
    const int nxr = nd.bShared ? 1 : nd.bInCh;
    const int x0 = (int)xrows.size();
    for (int c = 0; c < nxr; c++) {
      const int xi = x0 + c;
      xrows.push_back(ConvRowIO{chIn[c], nullptr});
      float* xr_row = plb.xr + (size_t)xi * kBins * txb;
      float* xi_row = plb.xi + (size_t)xi * kBins * txb;
      // [0, hist - h) zeros, [hist - h, hist) this channel's history, rows after the chunk zero (K padding reads them)
      if (hist - h > 0) {
        restore.push_back(HistJobB{xr_row, nullptr, txb, 0, hist - h, 0});
        restore.push_back(HistJobB{xi_row, nullptr, txb, 0, hist - h, 0});
      }
      if (h > 0) {
        const float *srcR = nullptr, *srcI = nullptr;
        int sstride = h;
        if (nd.bHistZero) {
        } else if (nd.bHistPlane >= 0) {   // the previous chunk's x planes (never the pair being written: flushed above)
          const size_t off = (size_t)(nd.bHistRow + (nd.bHistNx == 1 ? 0 : c)) * kBins * nd.bHistTxb + nd.bHistOff;
          srcR = xPlane(nd.bHistPlane, 0) + off;
          srcI = xPlane(nd.bHistPlane, 1) + off;
          sstride = nd.bHistTxb;
        } else {
          srcR = nd.bHistR + c * hstride;
          srcI = nd.bHistI + c * hstride;
        }
        restore.push_back(HistJobB{xr_row + (hist - h), srcR, txb, sstride, h, 0});
        restore.push_back(HistJobB{xi_row + (hist - h), srcI, txb, sstride, h, 0});
      }
      const int tailn = txb - (hist + (int)n);
      restore.push_back(HistJobB{xr_row + hist + (int)n, nullptr, txb, 0, tailn, 0});
      restore.push_back(HistJobB{xi_row + hist + (int)n, nullptr, txb, 0, tailn, 0});
    }
    // slots: discrete -> slot c reads input c, IR channel c ; true stereo -> (L,h0) (L,h1) (R,h2) (R,h3)
    slotOut.assign(nd.bSlots, nullptr);
    for (int slot = 0; slot < nd.bSlots; slot++) {
      if (nd.isTrueStereo) {
        float* tmp = getSlab(*this);
        slotOut[slot] = tmp;
        auto it = tsTemps.find(id);
        if (it == tsTemps.end()) it = tsTemps.emplace(id, std::array<float*, 4>{nullptr, nullptr, nullptr, nullptr}).first;
        it->second[slot] = tmp;
      } else {
        slotOut[slot] = ex.nodeOut(id, slot);
      }
    }
    // sets: columns grouped by the x-row they read, at most 16 per set, y rows consecutive per set
    for (int xc = 0; xc < nxr; xc++) {
      cols.clear();
      for (int slot = 0; slot < nd.bSlots; slot++) {
        int inc = nd.isTrueStereo ? (slot >> 1) : slot;
        if (nd.bShared || inc == xc) cols.push_back(slot);
      }
      for (size_t c0 = 0; c0 < cols.size(); c0 += 16) {
        ConvSetB st{};
        st.x = x0 + xc;
        st.y0 = (int)yrows.size();
        st.ncol = (int)std::min<size_t>(16, cols.size() - c0);
        st.P = P;
        ConvSetC sc{};
        sc.x = st.x;
        sc.y0 = st.y0;
        sc.ncol = st.ncol;
        sc.P = P;
        for (int j = 0; j < st.ncol; j++) {
          int slot = cols[c0 + j];
          st.hr[j] = nd.ir->hr + (size_t)slot * kBins * P;   // slot index == IR channel index in both modes
          st.hi[j] = nd.ir->hi + (size_t)slot * kBins * P;
          sc.hs[j] = nullptr;   // per FFT length, below
          yrows.push_back(ConvRowIO{nullptr, slotOut[slot]});
          ovIn.push_back(nd.bOverlap + ((size_t)slot * 2 + nd.bOvCur) * kBlock);
          ovOut.push_back(nd.bOverlap + ((size_t)slot * 2 + (nd.bOvCur ^ 1)) * kBlock);
        }
        if (refOrder) {
          sets.push_back(st);
          stats.ref_order_rows += st.ncol;
        } else if (nd.convPath == 3) {
          SetTaps tp{nd.ir.get(), {}};
          for (int j = 0; j < st.ncol; j++) tp.slot[j] = cols[c0 + j];
          setsC[P].push_back(sc);
          setsCTaps[P].push_back(tp);
        } else {
          sets.push_back(st);
        }
      }
    }
    nd.bOvCur ^= 1;
    nd.bHistZero = false;
    if (h > 0) {   // the history of the next chunk: the last h spectra of these rows
      if (nd.bHistPlane == bPairWrite) fail(GA_ERR_DEVICE, "internal: convolver history lives in the planes being written");
      nd.bHistPlane = bPairWrite;
      nd.bHistRow = (int)rowX0 + x0;
      nd.bHistNx = nxr;
      nd.bHistOff = hist + (int)n - h;
      nd.bHistTxb = txb;
      bResidents[bPairWrite].push_back(id);
    }
    flops += 8.0 * P * kBins * (double)nd.bSlots * (double)n;
  }
  bRowX += xrows.size();
  bRowY += yrows.size();
  bPairCur = bPairWrite;
  ex.flushLevel();
  size_t xo = ex.plan.putv(xrows), yo = ex.plan.putv(yrows), so = ex.plan.putv(sets), ro = ex.plan.putv(restore),
         oi = ex.plan.putv(ovIn), oo = ex.plan.putv(ovOut);
  const int nx = (int)xrows.size(), ny = (int)yrows.size(), ns_ = (int)sets.size(), nr = (int)restore.size();
  hipStream_t st = stream;
  Twiddles tw{w128, w256};
  const int nn = (int)n;
  const int maxn = std::max(hist, txb - hist - nn);
  const bool f64 = fft64 || refOrder;   // (formulation R: the reference's FftFlat precision around its own partition sum)
  ex.plan.add(LK_FFT, [=](uint8_t* base) {
    launch_hist_copy_b(st, (const HistJobB*)(base + ro), nr, std::max(maxn, 1));
    launch_rfft_fwd_b(st, (const ConvRowIO*)(base + xo), nx, nn, hist, plb, tw, f64);
  });
  if (ns_ > 0 && refOrder) ex.plan.add(LK_MAC, [=](uint8_t* base) { launch_refmac(st, (const ConvSetB*)(base + so), ns_, nn, hist, plb); });
  else if (ns_ > 0) ex.plan.add(LK_MAC, [=](uint8_t* base) { launch_spectral_mac_b(st, (const ConvSetB*)(base + so), ns_, nn, hist, plb); });
  for (auto& kv : setsC) {
    const int Pc = kv.first;
    static const char* r16env = expenv("GA_TCONV_RADIX16");   // A/B switches for measurements
    static const char* planenv = expenv("GA_TCONV_MIXED");
    const bool r16 = r16env ? atoi(r16env) != 0 : useRadix16;
    std::vector<TconvLaunch> tplan;
    if (debugTconvN2 > 0) {   // tests: a length no kernel exists for must come back as an error code, not abort the host
      const int Lc = std::max(1, debugTconvN2 - (Pc - 1));
      tplan.push_back(TconvLaunch{debugTconvN2, 0, (nn + Lc - 1) / Lc});
    } else if (r16 && !(planenv && atoi(planenv) == 0)) {
      tplan = tconvPlan(nn, Pc);
    } else {   // one FFT length for the whole chunk
      const int N2 = tapFftSize(Pc), Lc = N2 - (Pc - 1);
      tplan.push_back(TconvLaunch{N2, 0, (nn + Lc - 1) / Lc});
    }
    const std::vector<SetTaps>& taps = setsCTaps[Pc];
    for (const TconvLaunch& tl : tplan) {
      std::vector<ConvSetC> sv = kv.second;
      for (size_t i = 0; i < sv.size(); i++) {
        const float2* hsp = ensureTapSpectra(*taps[i].ir, tl.N2);
        for (int j = 0; j < sv[i].ncol; j++) sv[i].hs[j] = hsp + (size_t)taps[i].slot[j] * kBins * tl.N2;
      }
      size_t co = ex.plan.putv(sv);
      const int nc = (int)sv.size();
      const int N2 = tl.N2, tbase = tl.tbase, nseg = tl.nseg;
      const float2* twc = r16 ? twiddles16(N2) : twiddlesC(N2);
      ex.plan.add(LK_MAC, [=](uint8_t* base) {
        if (r16) launch_tconv16(st, (const ConvSetC*)(base + co), nc, nn, hist, plb, N2, twc, nseg, tbase);
        else launch_tconv(st, (const ConvSetC*)(base + co), nc, nn, hist, plb, N2, twc, nseg);
      });
    }
  }
  ex.plan.add(LK_FFT, [=](uint8_t* base) {
    launch_irfft_ola_b(st, (const ConvRowIO*)(base + yo), ny, nn, plb, (const float* const*)(base + oi), (float* const*)(base + oo), tw, f64);
  });
  stats.mac_flops_total += flops;
  // streaming-formulation bytes: per channel-instance per block FDL read + write + input; every distinct
  // IR channel is counted once per block however many nodes share it
  {
    std::map<std::pair<IrSpectra*, int>, int> distinct;
    double bytes = 0;
    for (int id : bNodes) {
      NodeS& nd = *nodes[id];
      const int P = nd.ir->P;
      bytes += ((double)P * kBins * 8.0 + kBins * 8.0 + 512.0) * nd.bSlots * (double)n;
      for (int sl = 0; sl < nd.bSlots; sl++) distinct[{nd.ir.get(), sl}] = P;
    }
    for (auto& kv : distinct) bytes += (double)kv.second * kBins * 8.0 * (double)n;
    stats.mac_bytes_total += bytes;
  }
  stats.mac_launches += 1;
}

// pass 7 (per convolver depth d): the convolvers whose inputs are complete, once per chunk over all blocks
void Context::chunkPlanConvolvers(ChunkRun& r, int d) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
    // ---- convolvers whose inputs are complete (depth d): once per chunk over all blocks ----
    // group -> (node, slot); ordered by (IR buffer, IR channel) so groups fed by the same inputs are adjacent
    ConvPlanCtx k;
    auto& active = k.active;
    std::vector<int>& bNodes = k.bNodes;  // formulation B / C nodes of this depth
    std::vector<int>& dNodes = k.dNodes;  // formulation D nodes of this depth
    for (int id : topo) {
      NodeS& nd = *nodes[id];
      if (nd.type != GA_NODE_CONVOLVER || !nd.ir || nd.depth != d) continue;
      if (!ex.convIn.has(id)) continue;
      if (nd.convPath == 4) {
        dNodes.push_back(id);
        continue;
      }
      if (nd.convPath >= 2) {
        bNodes.push_back(id);
        continue;
      }
      for (int slot = 0; slot < (int)nd.convRows.size(); slot++) active[nd.convRows[slot].group].push_back({id, slot});
    }
    if (!active.empty()) planConvolversShared(r, d, k);
    // ---- formulation D: coarse partitions, consumer sums fused in the frequency domain ----
    if (!dNodes.empty()) planCoarseStage(*this, ex, dNodes, n);
    // ---- formulations B / C: nodes with a private impulse response ----
    // (those that this chunk evaluates in the reference's own order -- formulation R -- in a pass of their own: double-precision
    // transforms and launch_refmac instead of the matrix-core / block-axis-FFT partition sums)
    if (!bNodes.empty()) {
      std::vector<int> plain, ref;
      for (int id : bNodes) (nodes[id]->refOrder ? ref : plain).push_back(id);
      if (!plain.empty()) {
        bNodes = plain;
        planConvolversPrivate(r, d, k, false);
      }
      if (!ref.empty()) {
        bNodes = ref;
        planConvolversPrivate(r, d, k, true);
      }
    }
    auto& tsTemps = k.tsTemps;
    // true stereo: outL = conv0(L) + conv2(R) ; outR = conv1(L) + conv3(R)  (ConvolverNode.cs:127-144)
    for (auto& kv : tsTemps) {
      hipStream_t st = stream;
      int64_t fr = frames;
      for (int o = 0; o < 2; o++) {
        float* out = ex.nodeOut(kv.first, o);
        float *a = kv.second[o], *b2 = kv.second[o + 2];
        if (a && b2) ex.plan.add(LK_OTHER, [=](uint8_t*) { launch_pair_sum(st, out, a, b2, fr); });
      }
    }
}

// feedback cycles, first chunk after an edit closed a loop: the reference's consumer finds the block the producer put out BEFORE the
// edit in the producer's output buffer.  That block is the tail of the producer's slab of the previous chunk, which nothing has
// overwritten yet when this chunk's first launch runs -- copied from there (only from memory the context knows to be alive: slabs
// and other producers' kept blocks; a zero-copy view of a sample buffer, which may have been released since, is not chased).
void Context::chunkStaleSeed(ChunkRun& r) {
  if (staleProducers.empty()) return;
  Exec& ex = *r.ex;
  std::vector<StaleJob> jobs;
  auto alive = [&](const float* p) {
    if (!p) return false;
    const size_t blockBytes = (size_t)slabFrames * sizeof(float) * std::max<size_t>(8, std::min<size_t>(1024, ((size_t)1 << 30) / std::max<size_t>((size_t)slabFrames * sizeof(float), 1)));
    for (void* b : slabBlocks)
      if ((const char*)p >= (const char*)b && (const char*)p + kBlock * sizeof(float) <= (const char*)b + blockBytes) return true;
    for (int id : staleProducers) {
      const NodeS& o = *nodes[id];
      if (o.staleBuf && p >= o.staleBuf && p + kBlock <= o.staleBuf + (size_t)o.staleRows * kBlock) return true;
    }
    return false;
  };
  for (int id : staleProducers) {
    NodeS& nd = *nodes[id];
    if (nd.staleBuf) continue;   // (a producer that already keeps its blocks)
    const int rows = nd.type == GA_NODE_CHANNEL_SPLITTER ? std::max<int>(1, (int)nd.outputs.size()) : 32;
    nd.staleRows = rows;
    nd.staleBuf = (float*)dalloc((size_t)rows * kBlock * sizeof(float));
    nd.staleNext = (float*)dalloc((size_t)rows * kBlock * sizeof(float));
    GA_HIP(hipMemsetAsync(nd.staleBuf, 0, (size_t)rows * kBlock * sizeof(float), stream));
    GA_HIP(hipMemsetAsync(nd.staleNext, 0, (size_t)rows * kBlock * sizeof(float), stream));
    if (lastViewSlabGen != slabGen || lastViewFrames < kBlock || id >= (int)lastViews.size()) continue;
    const Views& ov = lastViews[id];
    const float g = id < (int)lastViewScale.size() ? lastViewScale[id] : 1.f;
    for (int rw = 0; rw < rows && rw < (int)ov.size(); rw++) {
      const float* src = ov[rw] ? ov[rw] + (lastViewFrames - kBlock) : nullptr;
      if (alive(src)) jobs.push_back(StaleJob{nd.staleBuf + (size_t)rw * kBlock, src, g, 0});
    }
  }
  if (jobs.empty()) return;
  const size_t off = ex.plan.putv(jobs);
  const int nj = (int)jobs.size();
  hipStream_t st = stream;
  ex.plan.add(LK_OTHER, [=](uint8_t* base) { launch_stale_copy(st, (const StaleJob*)(base + off), nj); });
}

// pass 8b: feedback cycles -- what every stale producer put out in this (one-block) chunk is what the consumers that pull it while
// it is being processed will mix in the next block (TermS::stale).  Written to the OTHER copy: a pass-through node may hand on a
// view of another producer's current copy, and the jobs of one launch are not ordered.
void Context::chunkStaleCommit(ChunkRun& r) {
  if (staleProducers.empty()) return;
  Exec& ex = *r.ex;
  std::vector<StaleJob> jobs;
  const int si = (int)r.segs.size() - 1;
  for (int id : staleProducers) {
    NodeS& nd = *nodes[id];
    const int rows = nd.type == GA_NODE_CHANNEL_SPLITTER ? std::max<int>(1, (int)nd.outputs.size()) : 32;
    if (!nd.staleBuf || nd.staleRows < rows) {
      if (nd.staleBuf) {
        GA_HIP(hipStreamSynchronize(stream));
        dfree(nd.staleBuf, (size_t)nd.staleRows * kBlock * sizeof(float));
        dfree(nd.staleNext, (size_t)nd.staleRows * kBlock * sizeof(float));
      }
      nd.staleRows = rows;
      nd.staleBuf = (float*)dalloc((size_t)rows * kBlock * sizeof(float));
      nd.staleNext = (float*)dalloc((size_t)rows * kBlock * sizeof(float));
      GA_HIP(hipMemsetAsync(nd.staleBuf, 0, (size_t)rows * kBlock * sizeof(float), stream));
      GA_HIP(hipMemsetAsync(nd.staleNext, 0, (size_t)rows * kBlock * sizeof(float), stream));
    }
    const Views* ov = (si >= 0 && id < (int)ex.outViews[si].size()) ? &ex.outViews[si][id] : nullptr;
    const float g = si >= 0 ? ex.scaleOf(si, id) : 1.f;
    for (int rw = 0; rw < nd.staleRows; rw++) {
      const float* src = (ov && rw < (int)ov->size() && (*ov)[rw]) ? (*ov)[rw] + (r.n - 1) * kBlock : nullptr;   // (the chunk's LAST block)
      jobs.push_back(StaleJob{nd.staleNext + (size_t)rw * kBlock, src, g, 0});
    }
    std::swap(nd.staleBuf, nd.staleNext);
  }
  const size_t off = ex.plan.putv(jobs);
  const int nj = (int)jobs.size();
  hipStream_t st = stream;
  ex.plan.add(LK_OTHER, [=](uint8_t* base) { launch_stale_copy(st, (const StaleJob*)(base + off), nj); });
}

// pass 8: delay-line histories of the next chunk
void Context::chunkDelayCommit(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  std::vector<int>& topo = r.topo;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  // DelayNode: the last maxDelay samples every ring has seen become the history of the next chunk
  for (int id : topo) {
    NodeS& nd = *nodes[id];
    if (nd.type != GA_NODE_DELAY || !nd.delayLoaded) continue;
    const size_t maxD = (size_t)nd.maxDelaySamples, pitch = maxD + (size_t)nd.delayCap;
    for (int r = 0; r < nd.delayHistRings; r++) {
      if (nd.delayW[r] == 0) continue;
      float* dst = nd.delayHist + (size_t)r * maxD;
      const float* src = nd.delayLine + (size_t)r * pitch + nd.delayW[r];
      hipStream_t st = stream;
      ex.plan.add(LK_OTHER, [=](uint8_t*) { GA_HIP(hipMemcpyAsync(dst, src, maxD * sizeof(float), hipMemcpyDeviceToDevice, st)); });
    }
  }

}

// pass 9: upload the job tables, enqueue every recorded launch in order, profile events
void Context::chunkExecute(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  Exec& ex = *r.ex;
  r.tmPlan = nowMs();
  if (!pendingHandOver.empty()) {   // no pre-mix launch in this chunk took the previous chunk's hand-over along: copies in front
    std::vector<HandOver> hv;
    hv.swap(pendingHandOver);
    hipStream_t st = stream;
    ex.plan.launches.insert(ex.plan.launches.begin(), Plan::L{[hv, st](uint8_t*) {
      for (const HandOver& h : hv) GA_HIP(hipMemcpyAsync(h.dst_host, h.src, sizeof(float) * (size_t)h.n, hipMemcpyDeviceToHost, st));
    }, LK_OTHER, 0.0, 0.0});
  }
  // ---- upload tables, run ----
  ex.trajOffFinal = ex.plan.putv(ex.traj);
  size_t tbytes = ex.plan.host.size();
  const int slot = asyncMode ? (int)(chunkSeq & 1) : 0;   // async: the other buffer may still be waiting for its upload
  void*& thost = slot ? tablesHostB : tablesHost;
  size_t& thostBytes = slot ? tablesHostBBytes : tablesHostBytes;
  if (asyncMode && chunkDone[slot]) GA_HIP(hipEventSynchronize(chunkDone[slot]));   // chunk k - 2 is done: its staging is free
  if (thostBytes < tbytes) {
    if (thost) {
      GA_HIP(hipStreamSynchronize(stream));
      (void)hipHostFree(thost);
    }
    thostBytes = (tbytes + tbytes / 4 + 4096 + 15) & ~(size_t)15;
    GA_HIP(hipHostMalloc(&thost, thostBytes, hipHostMallocDefault));
  }
  ensure(tables, std::max(tablesHostBytes, tablesHostBBytes));
  std::memcpy(thost, ex.plan.host.data(), tbytes);
  if (tableUploadKernel) launch_table_upload(stream, tables.p, thost, tbytes);   // (staging and arena sizes are multiples of 16)
  else GA_HIP(hipMemcpyAsync(tables.p, thost, tbytes, hipMemcpyHostToDevice, stream));
  uint8_t* base = (uint8_t*)tables.p;

  std::vector<std::pair<hipEvent_t, hipEvent_t>> evs;
  std::vector<int> evKind;
  std::vector<double> evBytes;
  hipEvent_t evBegin = nullptr, evEnd = nullptr;
  const bool profile = profileNow;   // (this chunk is one of the sampled ones: option "profile_every")
  if (profile) {
    GA_HIP(hipEventCreate(&evBegin));
    GA_HIP(hipEventCreate(&evEnd));
    GA_HIP(hipEventRecord(evBegin, stream));
  }
  for (auto& l : ex.plan.launches) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profile) {
      GA_HIP(hipEventCreate(&e0));
      GA_HIP(hipEventCreate(&e1));
      GA_HIP(hipEventRecord(e0, stream));
    }
    l.fn(base);
    if (profile) {
      GA_HIP(hipEventRecord(e1, stream));
      evs.push_back({e0, e1});
      evKind.push_back(l.kind);
      evBytes.push_back(l.bytes);
    }
    if (l.kind == GA_STAGE_COARSE_SECTION) continue;   // (a section counts its own launches, per kernel)
    stats.kernel_launches++;
    if (l.kind >= 0 && l.kind < 16) {
      stats.stage_launches[l.kind]++;
      stats.stage_bytes[l.kind] += l.bytes;
      stats.stage_flops[l.kind] += l.flops;
    }
  }
  for (auto& x : extraProf) {   // pieces timed inside a launch (formulation D's overlapped section)
    evs.push_back({x.e0, x.e1});
    evKind.push_back(x.kind);
    evBytes.push_back(x.bytes);
  }
  extraProf.clear();
  if (profile) GA_HIP(hipEventRecord(evEnd, stream));
  GA_HIP(hipGetLastError());
  r.tmLaunch = nowMs();
  if (profileNow) pendingProf.push_back(ProfBatch{evBegin, evEnd, std::move(evs), std::move(evKind), std::move(evBytes)});
  if (asyncMode) {
    if (!chunkDone[slot]) GA_HIP(hipEventCreateWithFlags(&chunkDone[slot], hipEventDisableTiming));
    GA_HIP(hipEventRecord(chunkDone[slot], stream));
    harvestProfile(false);
  } else {
    GA_HIP(hipStreamSynchronize(stream));
    harvestProfile(true);
  }
  chunkSeq++;
  if (timing)
    fprintf(stderr, "[ga]   host detail: topo %.2f, sources %.2f, sim %.2f | resources %.2f, params %.2f, exec %.2f ms\n", r.tmTopo - r.tm0,
            r.tmSrc - r.tmTopo, r.tmSim - r.tmSrc, r.tmRes - r.tmSim, r.tmPre - r.tmRes, r.tmPlan - r.tmPre);
  if (timing)
    fprintf(stderr, "[ga] chunk %lld blocks: sim %.2f ms, plan %.2f ms, enqueue %.2f ms, wait %.2f ms\n", (long long)n, r.tmSim - r.tm0,
            r.tmPlan - r.tmSim, r.tmLaunch - r.tmPlan, nowMs() - r.tmLaunch);

}

// pass 10: commit the control state (source positions, Ended / Dispose bookkeeping, block clock) to the end of the chunk
void Context::chunkCommit(ChunkRun& r) {
  Context& c_ = *this; (void)c_;
  int& maxDepth = r.maxDepth; int& maxLevel = r.maxLevel; (void)maxDepth; (void)maxLevel;
  int64_t& n = r.n; (void)n;
  std::vector<double>& bt = r.bt; (void)bt;
  std::vector<int>& srcIds = r.srcIds; (void)srcIds;
  std::vector<SrcPlanOut>& srcPlans = r.srcPlans; (void)srcPlans;
  std::vector<Segment>& segs = r.segs; (void)segs;
  const int64_t frames = r.n * kBlock; (void)frames;
  int& bHistMax = r.bHistMax; (void)bHistMax;
  // ---- commit the control state to the end of the chunk ----
  for (size_t i = 0; i < srcIds.size(); i++) {
    NodeS& s = *nodes[srcIds[i]];
    SrcPlanOut& po = srcPlans[i];
    if (s.type != GA_NODE_BUFFER_SOURCE) {  // ConstantSourceNode / OscillatorNode: only the Ended + Dispose bookkeeping
      if (po.gone && po.goneAt <= n) {
        if (!s.endedRaised) endedQueue.push_back(srcIds[i]);
        s.endedRaised = true;
        if (po.goneAt == n) pending.push_back([this, id = srcIds[i]]() { doDispose(id); });
      }
      continue;
    }
    // recompute progress against the (possibly shortened) chunk
    if (s.spans.empty()) continue;
    int64_t firstPlay = -1;
    for (const SrcSpan& sp : s.spans)
      if ((sp.phase == SRC_PLAY || sp.phase == SRC_END) && firstPlay < 0) firstPlay = sp.b0;
    if (firstPlay < 0 || firstPlay >= n) continue;
    int64_t lastProcessed = n;
    if (po.gone && po.goneAt <= n) lastProcessed = po.goneAt;
    int64_t played = lastProcessed - firstPlay;
    PlayBuf* pb = s.bufId >= 0 ? buffers[s.bufId].get() : nullptr;
    bool rate1 = true;
    if (pb) rate1 = sourceGeom(*this, s, *pb).effectiveRate == 1.0;
    if (s.gsr) {
      if (!s.gsrBlocks.empty()) {  // the state at the start of block `played` (END blocks leave nothing to resume)
        const GsrBlock& e = s.gsrBlocks[std::min<size_t>((size_t)played, s.gsrBlocks.size() - 1)];
        s.playbackPosition = e.pp;
        for (int k = 0; k < 4; k++) s.gsrW[k] = e.w[k];
        s.gsrPos = e.pos;
        s.gsrReady = e.ready;
      }
    } else if (rate1) {
      s.playbackPosition += played * kBlock;
      if (s.loop && pb) {
        SrcGeom g = sourceGeom(*this, s, *pb);
        int64_t len = g.loopEndFrame - g.loopStartFrame;
        if (s.playbackPosition >= g.loopEndFrame && len > 0)
          s.playbackPosition = g.loopStartFrame + ((s.playbackPosition - g.loopEndFrame) % len);
      }
    } else {
      s.rsBlocks += played;
    }
    if (po.reachedEnd && po.endBlock < n) {
      s.stopTime = bt[po.endBlock + 1];
      s.hasStopped = true;
    }
    if (po.gone && po.goneAt <= n) {
      if (!s.endedRaised) endedQueue.push_back(srcIds[i]);
      s.endedRaised = true;
      if (po.goneAt == n) pending.push_back([this, id = srcIds[i]]() { doDispose(id); });  // runs in the next block's drain
    }
  }
  for (int id : r.streamIds) streamReplay(*nodes[id], n, bt, true);   // queue / resampler state at the end of the executed blocks
  currentBlock += n;
  currentTime = bt[n];
  stats.blocks_rendered = currentBlock;
  stats.chunks++;
  stats.segments += (int64_t)segs.size();
  chunkBlocksDone = n;
  chunkSegCh.clear();
  for (const Segment& sg : segs) chunkSegCh.push_back(SegCh{sg.b0, sg.b1, sg.nodes.back().outCh});
}

void Context::runChunkImpl(int64_t nblocks, float* const* /*unused*/) {
  ChunkRun r;
  r.n = nblocks;
  r.tm0 = nowMs();
  GA_HIP(hipSetDevice(device));
  if (disposed) fail(GA_ERR_DISPOSED, "context disposed");
  drain();  // AudioContextBase.cs:57
  if (!releasedPending.empty() || ++chunksSinceGc >= 64) collectGarbage();
  latched = true;
  profileNow = profile && (profileSeq++ % std::max(profileEvery, 1)) == 0;
  chunkTopology(r);
  chunkSimulate(r);
  chunkResources(r);
  r.tmRes = nowMs();
  bqSplitUsed = 0;   // (the blocks are reused chunk after chunk: every use is ordered on the stream behind the previous one)
  r.ex = std::make_unique<Exec>(*this, r.n, r.segs);
  r.ex->outViews.resize(r.segs.size());
  r.ex->plan.host.resize(16);  // reserved header
  chunkStaleSeed(r);
  chunkParamCurves(r);
  const double tmPar = nowMs();
  chunkConvScratch(r);
  r.tmPre = nowMs();
  double tmNodes = 0, tmConv = 0;
  for (int d = 0; d <= r.maxDepth; d++) {   // stages: convolver depth d
    const double a = nowMs();
    chunkPlanNodes(r, d);
    const double b = nowMs();
    chunkPlanConvolvers(r, d);
    tmNodes += b - a;
    tmConv += nowMs() - b;
  }
  chunkDelayCommit(r);
  chunkStaleCommit(r);
  chunkExecute(r);
  const double tmEx = nowMs();
  chunkCommit(r);
  // the last segment's output views stay for one chunk (Context::chunkStaleSeed); the other per-node tables go back to the pools
  if (!r.ex->outViews.empty() && !r.segs.empty()) {
    if (!lastViews.empty() && viewsPool.size() < 8) viewsPool.push_back(std::move(lastViews));
    lastViews = std::move(r.ex->outViews.back());
    r.ex->outViews.pop_back();
    const size_t sl = r.segs.size() - 1;
    if (sl < r.ex->outScale.size() && !r.ex->outScale[sl].empty()) lastViewScale = r.ex->outScale[sl];
    else lastViewScale.clear();
    lastViewFrames = r.n * kBlock;
    lastViewSlabGen = slabGen;
  }
  for (auto& ov : r.ex->outViews)
    if (!ov.empty() && viewsPool.size() < 8) viewsPool.push_back(std::move(ov));
  if (!r.segs.empty() && simReplay) {   // the last segment's records stay: the next chunk may take them over (Context::lastSegNodes)
    if (lastSegNodes.capacity() && segNodePool.size() < 8) {
      lastSegNodes.clear();
      segNodePool.push_back(std::move(lastSegNodes));
    }
    lastSegNodes = std::move(r.segs.back().nodes);
    lastSegHash = r.segs.back().hash;
    lastSegEpoch = apiEpoch;
    lastSegGraphVersion = graphVersion;
    r.segs.pop_back();
  }
  for (Segment& sg : r.segs) {
    sg.nodes.clear();
    if (segNodePool.size() < 8) segNodePool.push_back(std::move(sg.nodes));
  }
  if (timing) {
    const double tmCm = nowMs();
    r.ex.reset();
    fprintf(stderr, "[ga]   host detail: param curves %.3f, conv scratch %.3f, plan nodes %.3f, plan convolvers %.3f, commit %.3f, ~Exec %.3f ms\n",
            tmPar - r.tmRes, r.tmPre - tmPar, tmNodes, tmConv, tmCm - tmEx, nowMs() - tmCm);
  }
}

}  // namespace ga

// ga_sources.cpp -- source scheduling on the host: AudioBufferSourceNode / scheduled sources as per-chunk phase timelines, the resampler's
// position recurrence, general source replay, AudioStreamNodeBase replay (see ga_chunk_internal.hpp).
#include "ga_chunk_internal.hpp"

namespace ga {

SrcGeom sourceGeom(Context& c, NodeS& s, PlayBuf& b) {
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

Resampler& resamplerFor(Context& c, double rate) {
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

SrcPlanOut planSource(Context& c, NodeS& s, int64_t n, const std::vector<double>& bt) {
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
SrcPlanOut planScheduled(Context& c, NodeS& s, int64_t n, const std::vector<double>& bt) {
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

}  // namespace ga

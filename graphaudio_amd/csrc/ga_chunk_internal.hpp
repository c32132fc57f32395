// ga_chunk_internal.hpp -- what the translation units of the chunk engine share (ga_chunk.cpp: the passes of a chunk;
// ga_sources.cpp: source / stream timelines; ga_plan_nodes.cpp: per-node planning; ga_plan_conv.cpp: the convolver stages):
// the job-table builder, the control-plane simulation (Sim) and the per-chunk executor (Exec).  Not part of the library's interface.
#pragma once
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

// slabs: chunk-frame indexed float arrays handed to node outputs / mixed inputs for the duration of a chunk (ga_chunk.cpp)
float* getSlab(Context& c);
void resetSlabs(Context& c, int64_t frames);

// host phase times per chunk on stderr (GA_TIMING=1: measurements only)
extern const bool gaTiming;
double nowMs();

// ======================================================================================================
// source scheduling (AudioBufferSourceNode.Process control flow, AudioBufferSourceNode.cs:131-389) as a per-chunk
// timeline of phases.  bt[i] = block start times, bt[i+1] = t1 of block i (AudioContextBase.cs:78-79).
// ======================================================================================================
struct SrcGeom {
  int64_t loopStartFrame, loopEndFrame, durationEndFrame;
  double effectiveRate;
};

SrcGeom sourceGeom(Context& c, NodeS& s, PlayBuf& b);
Resampler& resamplerFor(Context& c, double rate);

struct SrcPlanOut {
  int64_t playedBlocks = 0;  // PLAY + END blocks inside the chunk (advance of the node's state)
  bool reachedEnd = false;   // an END block with stopTime NaN was reached (stopTime := t1)
  int64_t endBlock = -1;
  bool gone = false;         // Ended raised + Dispose queued inside the chunk
  int64_t goneAt = -1;       // first block at which the node is disconnected
  int64_t partialBlock = -1; // resampler: block with fewer than 128 outputs
  int partialProduced = 0;
};

SrcPlanOut planSource(Context& c, NodeS& s, int64_t n, const std::vector<double>& bt);
SrcPlanOut planScheduled(Context& c, NodeS& s, int64_t n, const std::vector<double>& bt);

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
        // (a stale term shows the producer's PREVIOUS block, and so does `o.zero` while the producer is being processed: a loop of
        // convolvers that nothing has reached yet carries exact zeros round and round -- fuzz session 64064: a DelayNode behind
        // such a loop must not be taken to hold audio)
        allZero = allZero && o.zero;
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
    ns.level = (int16_t)std::min(n_.level, 32767);
    ns.depth = (int16_t)std::min(n_.depth, 32767);
    ns.fan1 = !n_.outputs.empty() && n_.outputs[0].connectedInputs.size() == 1;
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
    // Twin channels: a channel whose term list (views, gains, curves, in this order) equals channel 0's gets the same sums -- mono
    // material in a stereo input.  It shares channel 0's view; where the caller wants its own row (forcedSlabs: the destination
    // bus, a delay ring) channel 0's job writes its sums to both rows (MixJob::out2).
    int twinOf0 = 0;   // channels 1 .. twinOf0 equal channel 0
    if (c.twinChannels && dstCh > 1 && !lists[0].empty()) {
      for (int ch = 1; ch < dstCh; ch++) {
        const auto &a = lists[0], &b = lists[ch];
        bool eq = a.size() == b.size();
        for (size_t j = 0; eq && j < a.size(); j++) eq = a[j].p == b[j].p && a[j].g == b[j].g && a[j].c == b[j].c;
        if (!eq) break;
        twinOf0 = ch;
      }
      if (twinOf0 > 1 && forcedSlabs) twinOf0 = 1;   // (one extra row per job)
    }
    int job0 = -1;     // channel 0's MixJob, if it got one
    for (int ch = 0; ch < dstCh; ch++) {
      auto& l = lists[ch];
      if (ch > 0 && ch <= twinOf0) {
        if (!forcedSlabs) {
          views[ch] = views[0];
          c.stats.twin_rows++;
          continue;
        }
        float* row = forcedSlabs[ch];
        if (row && job0 >= 0 && row != mixJobs[job0].out) {
          mixJobs[job0].out2 = row;
          noteAlign(row, f0);
          views[ch] = row;
          c.stats.twin_rows++;
          continue;
        }
      }
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
      if (ch == 0) job0 = (int)mixJobs.size();
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
    // buses of many terms go to the wide kernel while there are few of them (ga_kernels.hip, mix_wide_kernel)
    std::vector<MixJob> wideJobs;
    {
      size_t nw = 0;
      for (auto& j : mixJobs) nw += j.nterms >= kMixWideMinTerms;
      if (nw > 0 && nw <= 64) {
        std::vector<MixJob> rest;
        rest.reserve(mixJobs.size() - nw);
        for (auto& j : mixJobs) (j.nterms >= kMixWideMinTerms ? wideJobs : rest).push_back(j);
        mixJobs.swap(rest);
      }
    }
    if (!wideJobs.empty()) {
      size_t off = plan.putv(wideJobs);
      int nj = (int)wideJobs.size();
      int64_t mx = 0;
      double mixBytes = 0;
      for (auto& j : wideJobs) {
        mx = std::max(mx, j.n);
        mixBytes += 4.0 * (double)(j.nterms + 1) * (double)j.n;
      }
      hipStream_t st = c.stream;
      plan.add(LK_MIX, [=](uint8_t* base) {
        launch_mix_wide(st, (const MixJob*)(base + off), nj, (const float* const*)(base + termsOff), mx, scaled ? (const float*)(base + gainsOff) : nullptr,
                        curved ? (const float* const*)(base + curvesOff) : nullptr);
      }, mixBytes);
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
        for (const Grp& g : grps) launch_biquad_lanes(st, g.pa, g.n * (G - 1), (const BiquadSection*)(base + soff), g.k, true);
        for (const Grp& g : grps) launch_biquad_scan(st, (const BiquadScanJob*)(base + g.off), g.n, G, (const BiquadSection*)(base + soff), base, g.k);
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

}  // namespace ga

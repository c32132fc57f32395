// ga_chunk.cpp -- the passes of one render chunk (see ga_engine.hpp, ga_chunk_internal.hpp): topology, control-plane simulation,
// per-chunk resources, execution and commit.  Node planning: ga_plan_nodes.cpp; convolver stages: ga_plan_conv.cpp; source timelines: ga_sources.cpp.
#include "ga_chunk_internal.hpp"

namespace ga {

// ======================================================================================================
// slabs: chunk-frame indexed float arrays handed to node outputs / mixed inputs for the duration of a chunk
// ======================================================================================================
float* getSlab(Context& c) {
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

void resetSlabs(Context& c, int64_t frames) {
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
// runChunk
// ======================================================================================================
const bool gaTiming = getenv("GA_TIMING") != nullptr;   // measurements only: host phase times per chunk on stderr

double nowMs() {
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
    const double t0 = gaTiming ? nowMs() : 0.0;
    runChunkImpl(n, bus);
    if (gaTiming) fprintf(stderr, "[ga]   chunk total on the host (incl. destructors): %.3f ms\n", nowMs() - t0);
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
  staleLeavers.clear();
  for (auto& np : nodes) {
    np->prevReachable = np->reachable;
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
  for (auto& np : nodes)
    if (np->prevReachable && !np->reachable && !np->disposed) staleLeavers.push_back(np->id);
  }
  // Feedback: the loop closes through the block a producer put out LAST.  Unless every loop can be cut at a DelayNode (above: chunks
  // of `cycleBlocks` blocks) nothing can be batched along time -- the chunk is one block, the reference's own granularity (a 10 s
  // render = 3,750 chunks: launch bound, ~0.1 - 0.3 ms each)
  if (topoHasCycles) r.n = std::min<int64_t>(r.n, cycleBlocks);
  if (topoStatsVersion != graphVersion || topoStatsSize != topo.size()) {   // (cached with the order: a sweep over 28,672 node records is 0.5 ms)
    topoMaxDepth = topoMaxLevel = 0;
    topoHasTimeNodes = topoHasConvolvers = topoHasOscillators = topoHasStreams = false;
    for (int id : topo) {
      const NodeS& nd = *nodes[id];
      topoMaxDepth = std::max(topoMaxDepth, nd.depth);
      topoMaxLevel = std::max(topoMaxLevel, nd.level);
      if (nd.type == GA_NODE_DELAY || nd.type == GA_NODE_STREAM_SOURCE) topoHasTimeNodes = true;
      if (nd.type == GA_NODE_STREAM_SOURCE) topoHasStreams = true;
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
          !topoHasStreams && lastSegNodes.size() == topo.size() && g == goneAt.end()) {
        bool same = true;
        for (const NodeSeg& ns : lastSegNodes) {   // every source still in the phase (and on the buffer) the records say
          if (ns.type == GA_NODE_DELAY) {
            // a DelayNode's control state is its output flag, which is sticky once raised (DelayNode.cs:96-97): a node that is audible
            // with a constant delay time behaves like any other node; until then its ring model has to be walked block by block
            const NodeS& dn = *nodes[ns.id];
            if (!ns.delayAudible || !dn.delayAudible || !dn.params[0].events.empty() || !dn.params[0].modulation.empty()) {
              same = false;
              break;
            }
            continue;
          }
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
  if (gaTiming)
    fprintf(stderr, "[ga]   host detail: topo %.2f, sources %.2f, sim %.2f | resources %.2f, params %.2f, exec %.2f ms\n", r.tmTopo - r.tm0,
            r.tmSrc - r.tmTopo, r.tmSim - r.tmSrc, r.tmRes - r.tmSim, r.tmPre - r.tmRes, r.tmPlan - r.tmPre);
  if (gaTiming)
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
    if (sl < r.ex->outCurve.size() && !r.ex->outCurve[sl].empty()) lastViewCurve = r.ex->outCurve[sl];
    else lastViewCurve.clear();
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
  if (gaTiming) {
    const double tmCm = nowMs();
    r.ex.reset();
    fprintf(stderr, "[ga]   host detail: param curves %.3f, conv scratch %.3f, plan nodes %.3f, plan convolvers %.3f, commit %.3f, ~Exec %.3f ms\n",
            tmPar - r.tmRes, r.tmPre - tmPar, tmNodes, tmConv, tmCm - tmEx, nowMs() - tmCm);
  }
}

}  // namespace ga

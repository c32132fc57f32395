// ga_plan_nodes.cpp -- per-node planning of a chunk: parameter curves, sources, gains, biquads, panners, delays, oscillators; the kept blocks
// of feedback loops; delay-line commit (see ga_chunk_internal.hpp).
#include "ga_chunk_internal.hpp"

namespace ga {

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
      // The table doubles.  No wait: the copy is a plan entry in front of the launches that read the new table, and the old one is freed
      // when the stream is next known to be idle -- jobs planned earlier in this chunk (and the chunk in flight) still read it.
      // (A synchronisation here stalled a pipelined render for a whole chunk at every doubling: config 4, steps 3 and 7 of the bench.)
      ResampleSample* old = rs.devSamples;
      const size_t cb = (size_t)rs.devBlocks * kBlock * sizeof(ResampleSample);
      hipStream_t stc = stream;
      ex.plan.add(LK_OTHER, [=](uint8_t*) { GA_HIP(hipMemcpyAsync(nw, old, cb, hipMemcpyDeviceToDevice, stc)); });
      retired.push_back({old, (size_t)rs.devCapBlocks * kBlock * sizeof(ResampleSample)});
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
    nd.bqTwinN = 1;   // (every channel is walked on its own from here on)
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
  int64_t K = G > 1 ? ((nf + G - 1) / G + 31) / 32 * 32 : nf;   // (whole cache lines per piece: biquad1_kernel reads 128 bytes per round)
  if (G > 1) G = (int)((nf + K - 1) / K);
  if (G > 1 && ex.bqG == 0) {
    ex.bqG = G;
    ex.bqK = K;
  }
  if (G > 1 && (G != ex.bqG || K != ex.bqK)) G = 1;   // (one cut per level: the pieces of a level are expanded by one launch)
  const std::vector<float>* AK = G > 1 ? &biquadTransition(coefs, (int)chain.size(), K).M : nullptr;
  // Twin channels: a mono signal in a stereo node arrives as the SAME view on every channel (AudioNodeInput.cs:182-244 copies the
  // mono mix to all of them), and while every channel's state has been the same so far (NodeS::bqTwinN) the reference computes
  // the same numbers once per channel (BiQuadFilterNode.cs:117-146).  One job then stands for all of them: every channel's output
  // view is the one row, the end state goes to every channel's slot.
  int twins = 1;
  if (twinChannels && ns.outCh > 1) {
    bool same = true;
    for (int ch = 1; ch < ns.outCh; ch++) same = same && iv[ch] == iv[0];
    for (const NodeSeg* cn : chain) same = same && nodes[cn->id]->bqTwinN >= ns.outCh;
    if (same) twins = ns.outCh;
  }
  for (const NodeSeg* cn : chain) nodes[cn->id]->bqTwinN = twins;   // (channels beyond outCh are not advanced: no longer the same)
  if (twins > 1) stats.twin_rows += twins - 1;
  for (int ch = 0; ch < ns.outCh; ch++) {
    if (twins > 1 && ch > 0) {
      ov[ch] = ov[0];
      continue;
    }
    BiquadJob bj;
    bj.in = iv[ch] ? iv[ch] : zeros;
    bj.out = ex.nodeOut(ns.id, ch);
    bj.sec0 = (int)ex.bqSecs.size();
    bj.nsec = (int)chain.size();
    bj.f0 = f0;
    bj.n = nf;
    bj.state = nullptr;
    bj.twins = twins;
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
    ex.bqScans[bj.nsec].push_back(BiquadScanJob{bj.in, bj.out, 0, scratch, bj.sec0, bj.nsec, f0, nf, twins, 0});
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
          if (ns.depth != d) continue;
          const int lv = std::min(std::max((int)ns.level, 0), maxLevel);
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
        if (b_.type != GA_NODE_BIQUAD || !b_.bqActive || b_.bqDynamic) continue;
        fuseLen[b_.id] = 1;
        if (b_.ins[0].terms.size() != 1) continue;
        const TermS& t = b_.ins[0].terms[0];
        const NodeSeg* ia = segNode.find(t.node);
        if (!ia) continue;
        const NodeSeg& a_ = *ia;
        if (a_.type != GA_NODE_BIQUAD || !a_.bqActive || a_.bqDynamic || t.ch != b_.ins[0].bufCh || a_.outCh != b_.outCh) continue;
        if (!a_.fan1) continue;
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
        if (ns.level != curLevel) {
          ex.flushLevel();
          curLevel = ns.level;
          levelBqHeads = 0;   // constant-coefficient cascade outputs of this level (all levels' biquad launches are separate)
          for (size_t tj = ti; tj < todo.size() && todo[tj]->level == curLevel; tj++) {
            const NodeSeg& o = *todo[tj];
            if (o.type == GA_NODE_BIQUAD && o.bqActive && !o.bqDynamic && absorbedBy.get(o.id) < 0) levelBqHeads += std::max(o.outCh, 1);
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

// feedback cycles, first chunk after an edit closed a loop: the reference's consumer finds the block the producer put out BEFORE the
// edit in the producer's output buffer.  That block is the tail of the producer's slab of the previous chunk, which nothing has
// overwritten yet when this chunk's first launch runs -- copied from there (only from memory the context knows to be alive: slabs
// and other producers' kept blocks; a zero-copy view of a sample buffer, which may have been released since, is not chased).
void Context::chunkStaleSeed(ChunkRun& r) {
  if (staleProducers.empty() && staleLeavers.empty()) return;
  Exec& ex = *r.ex;
  std::vector<StaleJob> jobs;
  auto alive = [&](const float* p) {
    if (!p) return false;
    const size_t blockBytes = (size_t)slabFrames * sizeof(float) * std::max<size_t>(8, std::min<size_t>(1024, ((size_t)1 << 30) / std::max<size_t>((size_t)slabFrames * sizeof(float), 1)));
    for (void* b : slabBlocks)
      if ((const char*)p >= (const char*)b && (const char*)p + kBlock * sizeof(float) <= (const char*)b + blockBytes) return true;
    for (const auto& np : nodes) {
      const NodeS& o = *np;
      if (o.staleBuf && p >= o.staleBuf && p + kBlock <= o.staleBuf + (size_t)o.staleRows * kBlock) return true;
      // (the OTHER copy: what a pass-through node behind a stale edge showed in the previous chunk -- the producer's kept block of that
      // chunk -- sits there since the swap at that chunk's end, untouched until this chunk's commit runs; fuzz session 70427)
      if (o.staleNext && p >= o.staleNext && p + kBlock <= o.staleNext + (size_t)o.staleRows * kBlock) return true;
    }
    return false;
  };
  const bool haveLast = lastViewSlabGen == slabGen && lastViewFrames >= kBlock;
  // the block node `id` put out LAST (the previous chunk's last block, through the gain / gain curve a folded GainNode stood for) -> its staleBuf
  auto keepLastBlock = [&](NodeS& nd) {
    const int id = nd.id;
    const int rows = nd.type == GA_NODE_CHANNEL_SPLITTER ? std::max<int>(1, (int)nd.outputs.size()) : 32;
    if (!nd.staleBuf) {
      nd.staleRows = rows;
      nd.staleBuf = (float*)dalloc((size_t)rows * kBlock * sizeof(float));
      nd.staleNext = (float*)dalloc((size_t)rows * kBlock * sizeof(float));
      GA_HIP(hipMemsetAsync(nd.staleNext, 0, (size_t)rows * kBlock * sizeof(float), stream));
    }
    const bool haveViews = haveLast && id < (int)lastViews.size();
    const float g = (haveViews && id < (int)lastViewScale.size()) ? lastViewScale[id] : 1.f;
    const float* cv = (haveViews && id < (int)lastViewCurve.size() && lastViewCurve[id]) ? lastViewCurve[id] + (lastViewFrames - kBlock) : nullptr;
    for (int rw = 0; rw < nd.staleRows; rw++) {
      const float* src = (haveViews && rw < (int)lastViews[id].size() && lastViews[id][rw]) ? lastViews[id][rw] + (lastViewFrames - kBlock) : nullptr;
      StaleJob sj{nd.staleBuf + (size_t)rw * kBlock, alive(src) ? src : nullptr, g, 0};
      sj.curve = (sj.src && cv && alive(cv)) ? cv : nullptr;
      jobs.push_back(sj);   // (rows without a view: zeros)
#ifdef GA_EXPERIMENTS
      if (expenv("GA_DEBUG_STALE") && rw < 2)
        fprintf(stderr, "[stale] chunk %llu node %d type %d row %d: views %d (of %zu nodes) view %p alive %d scale %g curve %p\n", (unsigned long long)chunkSeq, id,
                nd.type, rw, haveViews ? (int)lastViews[id].size() : -1, lastViews.size(), (const void*)src, (int)alive(src), g, (const void*)sj.curve);
#endif
    }
  };
  // A node's output buffer holds the block it put out last for as long as the node lives (AudioNode.cs:153-160 hands THAT out when the
  // node is pulled from inside its own evaluation).  Three ways a node comes to be a stale producer in this chunk:
  //   it was one in the chunk before                      its kept block is current (chunkStaleCommit)
  //   it was evaluated in the chunk before, in no loop
  //     or in a loop that was entered somewhere else      its last block is in the previous chunk's views (fuzz sessions 60001, 61173)
  //   it was out of the graph (unplugged) for a while     the block it had when it LEFT: kept at that moment (staleLeavers, below)
  for (int id : staleProducers) {
    NodeS& nd = *nodes[id];
    if (nd.staleBuf && nd.staleSeq + 1 == chunkSeq) continue;
    if (nd.prevReachable || !nd.staleBuf) keepLastBlock(nd);
  }
  // nodes an edit has just taken out of the graph: what their output buffers hold stays there until they are evaluated again
  for (int id : staleLeavers) {
    NodeS& nd = *nodes[id];
    if (nd.staleBuf && nd.staleSeq + 1 == chunkSeq) continue;   // (a stale producer of the previous chunk: kept already)
    if (!haveLast || id >= (int)lastViews.size()) continue;
    bool any = false;
    for (const float* v : lastViews[id]) any = any || v != nullptr;
    if (any || nd.staleBuf) keepLastBlock(nd);   // (silent and never kept: nothing to remember -- a later seed reads zeros)
  }
  staleLeavers.clear();
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
    nd.staleSeq = chunkSeq;
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

}  // namespace ga

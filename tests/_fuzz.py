"""Random graph generator shared by the fuzz tests: builds the SAME random graph on any context."""
import numpy as np

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ChannelCountMode, ChannelInterpretation, ChannelMergerNode,
                            ChannelSplitterNode, ConstantSourceNode, ConvolverNode, DelayNode, FilterType, GainNode, OscillatorNode,
                            OscillatorType, PlayableAudioBuffer, StereoPannerNode)

SR = 48000


def build_random_graph(ctx, seed, frames, keep=None, handles=None):
    """Sources -> random chains (gain / biquad / convolver) -> optional shared bus nodes -> destination."""
    rng = np.random.default_rng(seed)
    rng2 = np.random.default_rng(seed + 7777)   # later additions draw from their own stream: old seeds keep their graphs
    rng3 = np.random.default_rng(seed + 31337)  # round 3: unity gains (handed on without a kernel) -- seeds >= 20000 only
    unity = seed >= 20000
    dest_ch = int(rng.choice([1, 2, 2, 4]))
    ctx.Destination.SetChannelCount(dest_ch)
    if rng.random() < 0.3:
        ctx.Destination.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
    ir_len = int(rng.integers(50, 1500))
    shared_ir = PlayableAudioBuffer.FromChannelArrays(
        [(rng.standard_normal(ir_len) * 0.05).astype(np.float32) for _ in range(int(rng.choice([1, 2])))], SR)
    buses = []
    for _ in range(int(rng.integers(0, 3))):
        g = GainNode(ctx)
        g.Gain.Value = float(rng.uniform(0.3, 1.0))
        if unity and rng3.random() < 0.4:
            g.Gain.Value = 1.0
        if rng.random() < 0.5:
            g.Inputs[0].SetChannelCount(int(rng.choice([1, 2])))
        if rng.random() < 0.3:
            g.Inputs[0].SetChannelCountMode(ChannelCountMode(int(rng.integers(0, 3))))
        g.Connect(ctx.Destination)
        buses.append(g)
    nvoices = int(rng.integers(2, 10))
    earlier_nodes = []
    voice_chains = []
    rng4 = np.random.default_rng(seed + 4242)   # seeds >= 30000: signals of one chain into PARAMETERS of a later one, scheduled sources into chains
    wild = seed >= 30000
    for v in range(nvoices):
        nch = int(rng.choice([1, 1, 2]))
        src_sr = int(rng.choice([SR, SR, 44100]))
        length = int(rng.integers(128 * 2, frames + 600))
        data = [(rng.standard_normal(length) * 0.25).astype(np.float32) for _ in range(nch)]
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromChannelArrays(data, src_sr)
        if src_sr == SR and rng.random() < 0.3:
            s.Loop = True
            if rng.random() < 0.5:
                s.LoopStart = float(rng.integers(0, length // 3)) / SR
                s.LoopEnd = float(rng.integers(length // 2, length)) / SR
        if src_sr == SR and rng.random() < 0.2:
            s.PlaybackRate.Value = float(rng.choice([0.5, 1.25]))
            s.Loop = False
        # looping while resampling, odd rates, and a playbackRate timeline (k-rate, sampled at block starts)
        if rng2.random() < 0.2:
            s.Loop = True
            if rng2.random() < 0.6:
                s.LoopStart = float(rng2.integers(0, length // 3)) / src_sr
                s.LoopEnd = float(rng2.integers(length // 2, length)) / src_sr
        if rng2.random() < 0.15:
            s.PlaybackRate.Value = float(rng2.choice([0.75, 2.0, 3.7, 1.0]))
        if rng2.random() < 0.15:
            s.PlaybackRate.SetValueAtTime(float(rng2.uniform(0.5, 2.0)), float(rng2.uniform(0, frames / SR * 0.5)))
            if rng2.random() < 0.6:
                s.PlaybackRate.LinearRampToValueAtTime(float(rng2.uniform(0.5, 2.0)), float(rng2.uniform(frames / SR * 0.5, frames / SR)))
        node = s
        chain_of_voice = [s]
        if handles is not None:
            handles.setdefault("sources", []).append(s)
        for _ in range(int(rng.integers(0, 4))):
            kind = rng.choice(["gain", "gain_auto", "biquad", "biquad", "conv_shared", "conv_private"])
            if kind == "gain":
                n = GainNode(ctx)
                n.Gain.Value = float(rng.uniform(0.2, 1.2))
                if unity and rng3.random() < 0.5:
                    n.Gain.Value = 1.0
            elif kind == "gain_auto":
                n = GainNode(ctx)
                n.Gain.SetValueAtTime(float(rng.uniform(0, 1)), 0.0)
                n.Gain.LinearRampToValueAtTime(float(rng.uniform(0, 1)), float(rng.uniform(0.005, frames / SR)))
                if rng.random() < 0.5:
                    n.Gain.SetTargetAtTime(float(rng.uniform(0, 1)), float(rng.uniform(0.0, frames / SR)), float(rng.uniform(0.001, 0.05)))
            elif kind == "biquad":
                n = BiQuadFilterNode(ctx)
                n.Type = FilterType(int(rng.integers(0, 8)))
                n.Frequency.Value = float(rng.uniform(80, 12000))
                n.Q.Value = float(rng.uniform(0.3, 3.0))
                n.Gain.Value = float(rng.uniform(-9, 9))
            elif kind == "conv_shared":
                n = ConvolverNode(ctx)
                n.Buffer = shared_ir
            else:
                n = ConvolverNode(ctx)
                c = int(rng.choice([1, 2, 4]))
                taps = int(rng.integers(10, 700))
                n.Normalize = bool(rng.random() < 0.7)
                n.EnableTrueStereo = bool(rng.random() < 0.7)
                ir = [(rng.standard_normal(taps) * 0.1).astype(np.float32) for _ in range(c)]
                if rng2.random() < 0.25:   # more than 64 partitions: the block-axis FFT formulation (1024 / 2048 points)
                    taps = int(rng2.integers(128 * 64 + 1, 128 * 300))
                    env = np.exp(-6.0 * np.arange(taps) / taps)
                    ir = [(rng2.standard_normal(taps) * 0.02 * env).astype(np.float32) for _ in range(c)]
                n.Buffer = PlayableAudioBuffer.FromChannelArrays(ir, SR)
            if rng.random() < 0.25 and not isinstance(n, ConvolverNode):
                n.Inputs[0].SetChannelCount(int(rng.choice([1, 2, 3])))
                n.Inputs[0].SetChannelCountMode(ChannelCountMode(int(rng.integers(0, 3))))
            node.Connect(n)
            node = n
            chain_of_voice.append(n)
            if handles is not None:
                handles.setdefault(type(n).__name__, []).append(n)
        # the remaining pure-Core nodes (drawn from rng2: the graphs of old seeds keep their shape and gain a tail)
        extra = float(rng2.random())
        if seed >= 40000 and extra >= 0.30 and rng4.random() < 0.3:
            extra = 0.25   # more splitter -> merger tails
        if extra < 0.12:
            n = StereoPannerNode(ctx)
            if rng2.random() < 0.5:
                n.Pan.Value = float(rng2.uniform(-1, 1))
            else:
                n.Pan.SetValueAtTime(float(rng2.uniform(-1, 1)), 0.0)
                n.Pan.LinearRampToValueAtTime(float(rng2.uniform(-1, 1)), float(rng2.uniform(0.01, frames / SR)))
            node.Connect(n)
            node = n
            if handles is not None:
                handles.setdefault("StereoPannerNode", []).append(n)
        elif extra < 0.22:
            n = DelayNode(ctx, float(rng2.choice([0.01, 0.05, 1.0])))
            if rng2.random() < 0.7:
                n.DelayTime.Value = float(rng2.uniform(0, 0.01))
            else:
                n.DelayTime.SetValueAtTime(float(rng2.uniform(0, 0.01)), 0.0)
                n.DelayTime.LinearRampToValueAtTime(float(rng2.uniform(0, 0.01)), float(rng2.uniform(0.01, frames / SR)))
            node.Connect(n)
            node = n
            if handles is not None:
                handles.setdefault("DelayNode", []).append(n)
        elif extra < 0.30:
            sp = ChannelSplitterNode(ctx, int(rng2.integers(1, 4)))
            mg = ChannelMergerNode(ctx, int(rng2.integers(1, 4)))
            node.Connect(sp)
            for o in range(sp._output_count):
                if rng2.random() < 0.8:
                    sp.Connect(mg, o, int(rng2.integers(0, mg._input_count)))
            node = mg
        if node is not chain_of_voice[-1]:
            chain_of_voice.append(node)   # the panner / delay / merger appended above
        # seeds >= 20000: a second connection into a node of this chain from a node of an EARLIER voice (no cycle: voices are
        # ordered) -- inputs that have to be mixed in front of biquads, convolvers, delays, panners, mergers, not only in front of gains
        if unity and earlier_nodes and len(chain_of_voice) > 1 and rng3.random() < 0.5:
            frm = earlier_nodes[int(rng3.integers(0, len(earlier_nodes)))]
            to = chain_of_voice[1 + int(rng3.integers(0, len(chain_of_voice) - 1))]
            if not isinstance(frm, (ChannelSplitterNode,)) and not isinstance(to, ChannelMergerNode):
                frm.Connect(to)
            elif seed >= 40000:   # splitter outputs / merger inputs by index
                oi = int(rng3.integers(0, frm._output_count)) if isinstance(frm, ChannelSplitterNode) else 0
                ii = int(rng3.integers(0, to._input_count)) if isinstance(to, ChannelMergerNode) else 0
                frm.Connect(to, oi, ii)
        # seeds >= 50000 (round 4): FEEDBACK -- a later node of the chain feeds an earlier one through a gain below 1.  The reference
        # renders such loops with an implicit one-block delay on the edge that closes them (Nodes/AudioNode.cs:153-156), the device
        # path one block per chunk since round 4.  (Drawn from a generator of their own: older seeds keep their graphs.)
        if seed >= 50000:
            rng5 = np.random.default_rng(seed * 31 + v)
            inner = [n for n in chain_of_voice[1:] if not isinstance(n, (ChannelSplitterNode, ChannelMergerNode))]
            if len(inner) >= 2 and rng5.random() < 0.6:
                i0 = int(rng5.integers(0, len(inner) - 1))
                i1 = int(rng5.integers(i0 + 1, len(inner)))
                fb = GainNode(ctx)
                fb.Gain.Value = float(rng5.uniform(0.05, 0.35))
                inner[i1].Connect(fb)
                if rng5.random() < 0.25 and isinstance(inner[i0], GainNode):
                    fb.Connect(inner[i0].Gain)      # the loop closes through a parameter
                else:
                    fb.Connect(inner[i0])
                # (not handed to the session's edit list: a feedback gain edited up to 1.2 makes the loop blow up)
            elif len(inner) == 1 and isinstance(inner[0], (GainNode, DelayNode, BiQuadFilterNode)) and rng5.random() < 0.4:
                fb = GainNode(ctx)              # node -> gain -> node
                fb.Gain.Value = float(rng5.uniform(0.05, 0.35))
                inner[0].Connect(fb)
                fb.Connect(inner[0])
        earlier_nodes.extend(chain_of_voice)
        voice_chains.append(chain_of_voice)
        target = buses[int(rng.integers(0, len(buses)))] if buses and rng.random() < 0.6 else ctx.Destination
        live = keep is None or v in keep  # (minimiser hook: unconnected voices are never pulled)
        if live:
            node.Connect(target)
        if rng.random() < 0.15 and buses and live:
            node.Connect(buses[0])  # fan-out
        when = float(rng.choice([0.0, 0.0, rng.uniform(0, frames / SR * 0.7)]))
        if rng.random() < 0.25:
            s.Start(when, float(rng.uniform(0, 0.01)), float(rng.uniform(0.005, frames / SR)))
        else:
            s.Start(when)
        if rng.random() < 0.2:
            s.Stop(float(rng.uniform(when, frames / SR)))
    # oscillator / constant-source voices (scheduled sources with sample-accurate start and stop)
    for _ in range(int(rng2.integers(0, 3))):
        if rng2.random() < 0.6:
            o = OscillatorNode(ctx)
            o.Type = OscillatorType(int(rng2.integers(0, 4)))
            if rng2.random() < 0.6:
                o.Frequency.Value = float(rng2.uniform(20, 8000))
            else:
                o.Frequency.SetValueAtTime(float(rng2.uniform(50, 2000)), 0.0)
                o.Frequency.LinearRampToValueAtTime(float(rng2.uniform(50, 4000)), float(rng2.uniform(0.01, frames / SR)))
        else:
            o = ConstantSourceNode(ctx)
            o.Offset.Value = float(rng2.uniform(-0.5, 0.5))
            if rng2.random() < 0.5:
                o.Offset.LinearRampToValueAtTime(float(rng2.uniform(-0.5, 0.5)), float(rng2.uniform(0.01, frames / SR)))
        g = GainNode(ctx)
        g.Gain.Value = float(rng2.uniform(0.05, 0.3))
        o.Connect(g)
        g.Connect(buses[0] if buses and rng2.random() < 0.5 else ctx.Destination)
        if wild and voice_chains and rng4.random() < 0.6:   # the oscillator / constant also feeds a node inside a voice chain
            chain = voice_chains[int(rng4.integers(0, len(voice_chains)))]
            if len(chain) > 1:
                to = chain[1 + int(rng4.integers(0, len(chain) - 1))]
                if not isinstance(to, ChannelMergerNode):
                    g.Connect(to)
        when = float(rng2.uniform(0, frames / SR * 0.5))
        if rng2.random() < 0.4:
            o.Start(when, 0.0, float(rng2.uniform(0.001, frames / SR * 0.6)))
        else:
            o.Start(when)
            if rng2.random() < 0.5:
                o.Stop(float(rng2.uniform(when, frames / SR)))
        if handles is not None:
            handles.setdefault("scheduled", []).append(o)
            handles.setdefault("GainNode", []).append(g)
    # seeds >= 30000: audio-rate modulation of a parameter of voice j by a signal of voice i < j (through a depth gain)
    if wild:
        for j in range(1, len(voice_chains)):
            if rng4.random() >= 0.35:
                continue
            targets = []
            for n in voice_chains[j][1:]:
                if isinstance(n, GainNode): targets.append((n.Gain, 0.3))
                elif isinstance(n, BiQuadFilterNode): targets.append((n.Frequency, 300.0))
                elif isinstance(n, DelayNode): targets.append((n.DelayTime, 0.003))
                elif isinstance(n, StereoPannerNode): targets.append((n.Pan, 0.5))
            i = int(rng4.integers(0, j))
            srcs = [n for n in voice_chains[i] if not isinstance(n, ChannelSplitterNode)]
            if not targets or not srcs:
                continue
            prm, depth = targets[int(rng4.integers(0, len(targets)))]
            frm = srcs[int(rng4.integers(0, len(srcs)))]
            dg = GainNode(ctx)
            dg.Gain.Value = float(depth * rng4.uniform(0.3, 2.0))
            frm.Connect(dg)
            dg.Connect(prm)
            if handles is not None:
                handles.setdefault("mod_gains", []).append(dg)
    if handles is not None:
        handles.update(buses=buses, shared_ir=shared_ir)
    return dest_ch


def run_random_session(ctx, seed, frames=128 * 48, max_piece=128 * 9, keep=None):
    """Render a random graph in random pieces and EDIT it between the pieces (parameter writes and automation, stop,
    new voices, dispose, rewiring, impulse-response swaps, audio-rate modulation, channel settings).  The same seed
    replays the same session on any context.  Returns (output, log of (piece, action, exception type or None))."""
    h = {}
    ch = build_random_graph(ctx, seed, frames, keep=keep, handles=h)
    rng = np.random.default_rng(seed ^ 0x5EED)
    out = np.zeros((ch, frames), np.float32)
    log = []
    gains = h.get("GainNode", []) + h["buses"]
    biquads = h.get("BiQuadFilterNode", [])
    convs = h.get("ConvolverNode", [])
    sources = h.get("sources", [])
    everything = lambda: gains + biquads + convs + sources + h.get("StereoPannerNode", []) + h.get("DelayNode", []) + h.get("scheduled", [])
    dead = set()

    def pick(lst):
        lst = [x for x in lst if id(x) not in dead]
        return lst[int(rng.integers(0, len(lst)))] if lst else None

    def new_voice(now):
        s = AudioBufferSourceNode(ctx)
        n = int(rng.integers(200, 3000))
        s.Buffer = PlayableAudioBuffer.FromChannelArrays(
            [(rng.standard_normal(n) * 0.2).astype(np.float32) for _ in range(int(rng.choice([1, 2])))], SR)
        if rng.random() < 0.5:
            s.Loop = True
        tail = s
        if rng.random() < 0.5:
            g = GainNode(ctx)
            g.Gain.Value = float(rng.uniform(0.1, 1.0))
            s.Connect(g)
            gains.append(g)
            tail = g
        tgt = pick(h["buses"]) if rng.random() < 0.5 else None
        tail.Connect(tgt if tgt is not None else ctx.Destination)
        s.Start(now + float(rng.choice([0.0, rng.uniform(0, 0.02)])))
        sources.append(s)

    detail = []

    def nid(x):
        return getattr(x, '_id', None)

    def act(now):
        detail.clear()
        kind = str(rng.choice(["gain_value", "gain_sched", "gain_cancel", "bq_value", "bq_type", "bq_ramp", "stop", "voice",
                               "dispose", "rewire", "ir_swap", "modulate", "dest_ch", "interp", "loop_toggle", "rate_value",
                               "rate_sched", "pan", "delay", "osc"]))
        if kind == "gain_value":
            g = pick(gains)
            if g: g.Gain.Value = float(rng.uniform(0, 1.2))
        elif kind == "gain_sched":
            g = pick(gains)
            if g:
                t = now + float(rng.uniform(0, 0.02))
                m = int(rng.integers(0, 4))
                if m == 0: g.Gain.SetValueAtTime(float(rng.uniform(0, 1)), t)
                elif m == 1: g.Gain.LinearRampToValueAtTime(float(rng.uniform(0, 1)), t + 0.01)
                elif m == 2:
                    g.Gain.SetValueAtTime(float(rng.uniform(0.1, 1)), t)
                    g.Gain.ExponentialRampToValueAtTime(float(rng.uniform(0.05, 1)), t + float(rng.uniform(0.003, 0.03)))
                else: g.Gain.SetTargetAtTime(float(rng.uniform(0, 1)), t, float(rng.uniform(0.001, 0.02)))
        elif kind == "gain_cancel":
            g = pick(gains)
            if g: g.Gain.CancelScheduledValues(now + float(rng.uniform(0, 0.02)))
        elif kind == "bq_value":
            b = pick(biquads)
            if b:
                b.Frequency.Value = float(rng.uniform(300, 12000))
                if rng.random() < 0.5: b.Q.Value = float(rng.uniform(0.3, 3))
        elif kind == "bq_type":
            b = pick(biquads)
            if b: b.Type = FilterType(int(rng.integers(0, 8)))
        elif kind == "bq_ramp":
            b = pick(biquads)
            if b:
                b.Frequency.SetValueAtTime(float(rng.uniform(300, 8000)), now)
                b.Frequency.LinearRampToValueAtTime(float(rng.uniform(300, 8000)), now + float(rng.uniform(0.005, 0.03)))
        elif kind == "stop":
            s = pick(sources)
            if s: s.Stop(now + float(rng.choice([0.0, rng.uniform(0, 0.02)])))
        elif kind == "voice":
            new_voice(now)
        elif kind == "dispose":
            n = pick(everything())
            if n and not any(n is b for b in h["buses"]):
                n.Dispose()
                dead.add(id(n))
        elif kind == "rewire":
            n = pick(gains + biquads + convs)
            if n and not any(n is b for b in h["buses"]):
                n.Disconnect()
                tgt = pick(h["buses"]) if rng.random() < 0.5 else None
                n.Connect(tgt if tgt is not None else ctx.Destination)
        elif kind == "ir_swap":
            c = pick(convs)
            if c:
                if rng.random() < 0.4:
                    c.Buffer = h["shared_ir"]
                else:
                    taps = int(rng.integers(10, 900))
                    c.Buffer = PlayableAudioBuffer.FromChannelArrays(
                        [(rng.standard_normal(taps) * 0.1).astype(np.float32) for _ in range(int(rng.choice([1, 2])))], SR)
        elif kind == "modulate":
            # an audio-rate signal into a parameter: GainNode.gain, or a parameter of a panner / delay / biquad / scheduled source
            cands = [(g, g.Gain, 0.3) for g in gains if id(g) not in dead]
            cands += [(n, n.Pan, 0.5) for n in h.get("StereoPannerNode", []) if id(n) not in dead]
            cands += [(n, n.DelayTime, 0.003) for n in h.get("DelayNode", []) if id(n) not in dead]
            cands += [(n, n.Frequency, 300.0) for n in biquads if id(n) not in dead]
            cands += [(n, n.Frequency if isinstance(n, OscillatorNode) else n.Offset, 50.0 if isinstance(n, OscillatorNode) else 0.2)
                      for n in h.get("scheduled", []) if id(n) not in dead]
            if cands:
                tgt_node, prm, depth = cands[int(rng.integers(0, len(cands)))]
                detail.append((type(tgt_node).__name__, nid(tgt_node), depth))
                lfo = AudioBufferSourceNode(ctx)
                lfo.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(700) * depth).astype(np.float32), SR)
                lfo.Loop = True
                lfo.Connect(prm)
                lfo.Start(now)
                sources.append(lfo)
        elif kind == "dest_ch":
            ctx.Destination.SetChannelCount(int(rng.integers(ch, 5)))
        elif kind == "interp":
            n = pick(gains + biquads)
            if n: n.Inputs[0].SetChannelInterpretation(ChannelInterpretation(int(rng.integers(0, 2))))
        elif kind == "loop_toggle":
            s = pick(sources)
            if s and s.Buffer is not None:
                s.Loop = not s.Loop
                detail.append((nid(s), s.Loop, s.Buffer.SampleRate, s.Buffer.Length))
        elif kind == "pan":
            pn = pick(h.get("StereoPannerNode", []))
            if pn:
                if rng.random() < 0.5:
                    pn.Pan.Value = float(rng.uniform(-1, 1))
                else:
                    pn.Pan.SetValueAtTime(float(rng.uniform(-1, 1)), now)
                    pn.Pan.LinearRampToValueAtTime(float(rng.uniform(-1, 1)), now + float(rng.uniform(0.005, 0.03)))
        elif kind == "delay":
            dn = pick(h.get("DelayNode", []))
            if dn: dn.DelayTime.Value = float(rng.uniform(0, 0.01))
        elif kind == "osc":
            o = pick(h.get("scheduled", []))
            if o is not None:
                detail.append((type(o).__name__, nid(o)))
                if isinstance(o, OscillatorNode):
                    if rng.random() < 0.5: o.Type = OscillatorType(int(rng.integers(0, 4)))
                    else: o.Frequency.Value = float(rng.uniform(20, 8000))
                if rng.random() < 0.3: o.Stop(now + float(rng.uniform(0, 0.02)))
        elif kind == "rate_value":
            s = pick(sources)
            if s:
                s.PlaybackRate.Value = float(rng.choice([0.5, 1.0, 1.25, 2.5]))
                detail.append((nid(s), s.PlaybackRate.Value, s.Loop, s.Buffer.SampleRate if s.Buffer else None))
        elif kind == "rate_sched":
            s = pick(sources)
            if s:
                s.PlaybackRate.SetValueAtTime(float(rng.uniform(0.5, 2.0)), now + float(rng.uniform(0, 0.01)))
                if rng.random() < 0.5:
                    s.PlaybackRate.LinearRampToValueAtTime(float(rng.uniform(0.5, 2.0)), now + float(rng.uniform(0.01, 0.04)))
        return kind

    # nodes taken out of the graph for a while (not reachable from the destination = not processed, state frozen) and put back
    # later; decided by a generator of its own so that the sessions of earlier sweeps keep their action sequences
    rng3 = np.random.default_rng(seed ^ 0x0F0F)
    parked = []

    def unplug_or_replug():
        if parked and rng3.random() < 0.5:
            n = parked.pop(int(rng3.integers(0, len(parked))))
            if id(n) in dead:
                return "replug_dead"
            buses = [b for b in h["buses"] if id(b) not in dead and b is not n]
            tgt = buses[int(rng3.integers(0, len(buses)))] if buses and rng3.random() < 0.5 else None
            n.Connect(tgt if tgt is not None else ctx.Destination)
            return "replug:" + type(n).__name__
        cands = [x for x in gains + biquads + convs + h.get("StereoPannerNode", []) + h.get("DelayNode", [])
                 if id(x) not in dead and not any(x is q for q in parked)]
        if not cands:
            return "unplug_none"
        n = cands[int(rng3.integers(0, len(cands)))]
        n.Disconnect()
        parked.append(n)
        return "unplug:" + type(n).__name__

    global last_pieces, details
    last_pieces = []
    details = []
    pos = 0
    piece = 0
    while pos < frames:
        n = int(min(frames - pos, rng.integers(1, max_piece)))
        ctx.Render(out, n, pos)
        pos += n
        piece += 1
        last_pieces.append(pos)
        if rng3.random() < 0.2:
            try:
                log.append((piece, unplug_or_replug(), None))
            except Exception as e:
                log.append((piece, "?", type(e).__name__))
        for _ in range(int(rng.integers(0, 4))):
            try:
                k = act(ctx.CurrentTime)
                log.append((piece, k, None))
                details.append((piece, k, list(detail), ctx.CurrentTime))
            except Exception as e:  # the other implementation must raise the same exception type at the same point
                log.append((piece, "?", type(e).__name__))
    return out, log

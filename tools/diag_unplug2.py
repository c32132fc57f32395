import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graphaudio_amd import *
from tests import _graphs as G
SR=48000
ctx=OfflineAudioContext(SR)
ctx.Destination.SetChannelCount(2)
s=AudioBufferSourceNode(ctx); v=G.voice(70,128*40); s.Buffer=PlayableAudioBuffer.FromMonoArray(v,SR)
c=GainNode(ctx); c.Gain.Value=0.5
s.Connect(c); c.Connect(ctx.Destination); s.Start()
out=np.zeros((2,128*40),np.float32)
ctx.Render(out,128*5,0); c.Disconnect(); ctx.Render(out,128*3,128*5); c.Connect(ctx.Destination); ctx.Render(out,128*32,128*8)
for b in (4,5,7,8,9):
    blk=out[0,b*128:(b+1)*128]
    best=[(k, float(np.abs(blk-0.5*v[k*128:(k+1)*128]).max())) for k in range(39)]
    k,e=min(best,key=lambda x:x[1])
    print("block",b,"rms",G.rms(blk),"best matching source block",k,"maxdiff",e)

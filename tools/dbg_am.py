import sys
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np
import composable_sdr_amd as cs, oracle_lib as O
from synth import synth_cf32
M=1; frames=[30000,10960,20485]; nf=sum(frames)
x=synth_cf32(M*nf,M,seed=4242)
a=cs.Chain(channels=1,demod="none",max_frames=30000); b=cs.Chain(channels=1,demod="am",max_frames=30000)
oa=O.Chain(1,demod="none"); ob=O.Chain(1,demod="am")
pos=0; ga=[];gb=[];wa=[];wb=[]
for f in frames:
    xa=x[pos:pos+f]; ga.append(a.process(xa)); gb.append(b.process(xa)); wa.append(oa.process(xa)); wb.append(ob.process(xa)); pos+=f
ga,gb,wa,wb=[np.concatenate(v,axis=-1).ravel() for v in (ga,gb,wa,wb)]
print("deno err",np.abs(ga-wa).max(),"am err",np.abs(gb-wb).max(), "at", np.argmax(np.abs(gb-wb)))
am_on_gpu_deno=O.AmpDem().demodulate_block(ga)
print("oracle AM on gpu deno vs gpu am",np.abs(am_on_gpu_deno-gb).max(),"vs oracle am",np.abs(am_on_gpu_deno-wb).max())
e=np.abs(gb-wb); print("err by region", [float(e[i:i+5000].max()) for i in range(0,nf,5000)])
print("|x| range", np.abs(wa).min(), np.abs(wa).max())

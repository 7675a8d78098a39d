import os, sys
os.environ["CSDR_AGC_L"]="256"; os.environ["CSDR_AGC_W"]="512"
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np
import composable_sdr_amd as cs
from composable_sdr_amd import _lib
from test_gpu_parity import _bursty
M,nf=20,4096
x=_bursty(M,nf,1234+M,"bursts")
def run(demod,flags):
    ch=cs.Chain(channels=M,demod=demod,kf=0.3,agc=8.0,max_frames=nf,flags=_lib.FLAG_QUIET|flags)
    y=ch.process(x); ch.close(); return y
ya=run("fm",0); ya2=run("fm",0); yb=run("fm",_lib.FLAG_AGC_SEQUENTIAL); yb2=run("fm",_lib.FLAG_AGC_SEQUENTIAL)
na=run("none",0); nb=run("none",_lib.FLAG_AGC_SEQUENTIAL)
print("spec fm deterministic", np.array_equal(ya.view(np.uint32),ya2.view(np.uint32)), "seq fm deterministic", np.array_equal(yb.view(np.uint32),yb2.view(np.uint32)), "none equal", np.array_equal(na.view(np.uint32), nb.view(np.uint32)))
def fm_pipe(z):
    p=cs.fmDemodulator(0.3); r=p._start(); y=p._process(r,z); p._done(r); return np.asarray(y)
for c in range(3):
    f=fm_pipe(na[c])
    print("ch",c,"pipe vs spec mism",int((f.view(np.uint32)!=ya[c].view(np.uint32)).sum()),"pipe vs seq mism",int((f.view(np.uint32)!=yb[c].view(np.uint32)).sum()))
f32=np.float32
def fma(a,b,c): return f32(np.float64(a)*np.float64(b)+np.float64(c))
def atan2_rn(y,x):
    ax,ay=abs(x),abs(y); mx,mn=max(ax,ay),min(ax,ay)
    a=f32(mn*f32(f32(1)/mx)) if mx>0 else f32(0)
    z=f32(a*a); p=f32(2.456645248e-03)
    for c in [-1.440101303e-02,3.978060186e-02,-7.234797627e-02,1.049891263e-01,-1.416121870e-01,1.998590529e-01,-3.333259821e-01,9.999998808e-01]:
        p=fma(p,z,f32(c))
    r=f32(p*a)
    if ay>ax: r=f32(f32(1.57079632679489662)-r)
    if np.signbit(x): r=f32(f32(3.14159265358979324)-r)
    return f32(np.copysign(r,y))
ref=f32(1.0/(2*np.pi*0.3))
c=1
f=fm_pipe(na[c]); idx=np.flatnonzero(f.view(np.uint32)!=ya[c].view(np.uint32))[:12]
for t in idx:
    rp=na[c][t-1]; r=na[c][t]
    re=f32(f32(rp.real*r.real)+f32(rp.imag*r.imag)); im=f32(f32(rp.real*r.imag)-f32(rp.imag*r.real))
    e=f32(atan2_rn(im,re)*ref)
    e64=np.arctan2(np.float64(rp.real)*r.imag-np.float64(rp.imag)*r.real, np.float64(rp.real)*r.real+np.float64(rp.imag)*r.imag)/(2*np.pi*0.3)
    print(t,"rp",rp,"r",r,"spec",ya[c][t],"seq",f[t],"emu",e,"f64",e64)

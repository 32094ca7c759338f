import sys, os, ctypes as C, json, time
import numpy as np
ROOT=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import search_tools as st
from search_runner import aligned, same
from turingcodec_amd import decisions, workload
W,H=1920,1080
S=1
planes, stride = st.clip_planes(W, H, 7, 8)
planes=[aligned(p) for p in planes]
pad=96
pus, first, cx, cy = workload.picture_pus(W,H,3,1.0)
par = st.medium_params(W,H,8,32)
rate=(45000,98000)
dev=C.CDLL(os.path.join(ROOT,'turingcodec_amd','libhavoc_mi355x.so'), mode=C.RTLD_GLOBAL)
vp, ip = C.c_void_p, C.c_ssize_t
dev.havoc_mi355x_create.argtypes=[C.POINTER(vp), C.c_int, vp]
dev.havoc_mi355x_malloc.argtypes=[vp, C.POINTER(vp), C.c_size_t]
dev.havoc_mi355x_h2d.argtypes=[vp,vp,vp,C.c_size_t]
dev.havoc_mi355x_interp_planes.argtypes=[vp,C.c_int,C.c_int,vp,ip,vp,ip,C.c_int,C.c_int,C.c_int,C.c_int]
dev.havoc_mi355x_sync.argtypes=[vp]
ctx=vp(); assert dev.havoc_mi355x_create(C.byref(ctx),0,vp(-1 & 0xFFFFFFFFFFFFFFFF))==0
n=planes[0].size; pe=(n+63)&~63
dpic=vp(); assert dev.havoc_mi355x_malloc(ctx,C.byref(dpic),3*pe*S+256)==0
for k,p in enumerate(planes): assert dev.havoc_mi355x_h2d(ctx,dpic.value+k*pe*S,p.ctypes.data,n*S)==0
dphase=vp(); assert dev.havoc_mi355x_malloc(ctx,C.byref(dphase),32*pe*S+256)==0
for r in (0,1):
    base=dphase.value+r*16*pe*S
    assert dev.havoc_mi355x_h2d(ctx,base,planes[1+r].ctypes.data,n*S)==0
    assert dev.havoc_mi355x_interp_planes(ctx,S,8,base,pe,dpic.value+(1+r)*pe*S,stride,12,4,W+2*pad-24,H+2*pad-8)==0
dev.havoc_mi355x_sync(ctx)
origin=pad*stride+pad
def run(bi):
    t0=time.perf_counter()
    r = decisions.picture_uni(ctx,S,par,dpic.value,origin,stride,dpic.value,(pe+origin,2*pe+origin),stride,pad,dphase.value,pe,(origin,16*pe+origin),pus,first,cx,cy,rate,on_device=True,bi=bi)
    return r, time.perf_counter()-t0
(a,fa,_),t = run(False)
print('uni', t)
for k in range(12):
    (b,fb,_,bb),t = run(True)
    bad = same(b,a)
    print('bi run',k,round(t,4),'uni mismatches',len(bad), bad[:6], [ (int(pus[i//2]['x0']),int(pus[i//2]['y0']),int(pus[i//2]['w']),int(pus[i//2]['h'])) for i in bad[:4]])
    if k: print('   bi vs previous bi run', len(same(bb, prev, ["mv","mvd","mvp_flag","calls","cost_subpel"])))
    prev=bb
(a2,fa2,_),t = run(False)
print('uni again', t, len(same(a2,a)))

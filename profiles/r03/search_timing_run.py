import sys, os, ctypes as C, json, time
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import search_tools as st
from search_runner import aligned
from turingcodec_amd import decisions, workload
W,H=1920,1080
S=1
planes, stride = st.clip_planes(W, H, 7, 8)
planes=[aligned(p) for p in planes]
pad=96
pus, first, cx, cy = workload.picture_pus(W,H,3,1.0)
par = st.medium_params(W,H,8,32)
rate=(45000,98000)
dev=C.CDLL(os.path.join(ROOT,'turingcodec_amd',os.environ.get('LIBDEV','libhavoc_mi355x.so')), mode=C.RTLD_GLOBAL)
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
for attempt in range(3):
    t0=time.perf_counter()
    got, field, stats = decisions.picture_uni(ctx,S,par,dpic.value,origin,stride,dpic.value,(pe+origin,2*pe+origin),stride,pad,dphase.value,pe,(origin,16*pe+origin),pus,first,cx,cy,rate,on_device=True)
    t=time.perf_counter()-t0
ticks=got['replays'].astype(np.int64)
area=np.repeat(pus['w']*pus['h'],2)
out={'seconds':t,'sum_us':float(ticks.sum()/100.0),'mean_us':float(ticks.mean()/100.0),'calls_mean':float(got['calls'].mean())}
for lo,hi in ((0,64),(65,256),(257,1024),(1025,4096)):
    m=(area>=lo)&(area<=hi)
    if m.any(): out[f'area_{lo}_{hi}']={'n':int(m.sum()),'mean_us':float(ticks[m].mean()/100.0),'calls':float(got['calls'][m].mean())}
# per-CTU chain sums (list 0): critical path estimate
chain=[ticks[2*first[c]:2*first[c+1]:2].sum() for c in range(cx*cy)]
out['ctu_chain_mean_us']=float(np.mean(chain)/100.0); out['ctu_chain_max_us']=float(np.max(chain)/100.0)
print(json.dumps(out))

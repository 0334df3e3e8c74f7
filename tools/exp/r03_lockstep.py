import time, torch, sys
sys.path.insert(0,'/root/repo')
from ug_stereomatcher_amd import _lib, synth
W,H=4928,3264
dev=torch.device("cuda:0")
pairs=[]
for j in range(4):
    L,R,_,_=synth.make_pair(W,H,synth.BASE_SEED+j); pairs.append((torch.from_numpy(L).to(dev),torch.from_numpy(R).to(dev)))
outs=[torch.empty((3,H,W),dtype=torch.float32,device=dev) for _ in range(4)]
torch.cuda.synchronize()
with _lib.Context(levels=14,slots=4) as c:
    lib,h=c.lib,c.handle
    def sub(s): 
        dL,dR=pairs[s]; c.check(lib.ugsm_submit_full(h,s,dL.data_ptr(),dR.data_ptr(),W,H,W*3,outs[s].data_ptr()))
    def staggered(n):
        for i in range(n):
            s=i%4
            if i>=4: c.check(lib.ugsm_wait(h,s))
            sub(s)
        c.check(lib.ugsm_wait_all(h))
    def lockstep(n):
        for i in range(n//4):
            for s in range(4): sub(s)
            c.check(lib.ugsm_wait_all(h))
    staggered(16)
    for name,f in (("staggered",staggered),("lockstep",lockstep),("staggered",staggered),("lockstep",lockstep)):
        t0=time.perf_counter(); f(128); print(name, round(128/(time.perf_counter()-t0),2),"pairs/s")

import torch
dev=torch.device("cuda",0)
for n in (1<<24, 1<<26, 1<<28):
    a=torch.zeros(n,dtype=torch.float32,device=dev); b=torch.empty_like(a)
    for name,fn in (("add",lambda: torch.add(a,1.0,out=b)),("copy_",lambda: b.copy_(a))):
        fn(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        reps=max(4,(1<<30)//n)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        print(n*4>>20,"MiB",name, round(reps*2*n*4/(e0.elapsed_time(e1)*1e-3)/1e9),"GB/s r+w")

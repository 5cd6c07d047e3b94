import time, torch, sys
dev=torch.device("cuda",0)
torch.zeros(1,device=dev); torch.cuda.synchronize()
def run(sizes, label):
    held=[]; ts=[]
    for s in sizes:
        torch.cuda.synchronize(); t0=time.perf_counter()
        held.append(torch.empty(s<<30,dtype=torch.int8,device=dev))
        torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    print(label, " ".join("%.1f"%t for t in ts), "total %.0f ms"%sum(ts), flush=True)
    return held
mode=sys.argv[1]
X=torch.empty(13<<30,dtype=torch.int8,device=dev)   # the matrix
if mode=="a": h=run([4]*30,"30 x 4 GiB:")
if mode=="b": h=run([4,12]*7+[4],"4/12 alternating:")
if mode=="c": h=run([16]*7,"7 x 16 GiB:")
if mode=="d":
    tmp=[torch.empty(8<<30,dtype=torch.int8,device=dev) for _ in range(4)]; del tmp; torch.cuda.empty_cache()   # freed memory first (like a generator's temporaries)
    h=run([4]*30,"after freeing 32 GiB: 30 x 4 GiB:")

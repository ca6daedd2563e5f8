"""Weight-gradient kernel (g2s_conv2d_wgrad) on the trained nets' layers, HIP-graph-replay timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gan2shape_amd
from gan2shape_amd.op.conv import _wgrad
torch.cuda.set_stream(torch.cuda.Stream())
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (4 * n) * 1e3
L = [(1,3,32,128,4,2,1),(1,32,64,64,4,2,1),(1,64,128,32,4,2,1),(1,128,256,16,4,2,1),(1,256,256,4,3,1,1),(1,128,128,8,3,1,1),(1,64,64,16,3,1,1),(1,32,32,32,3,1,1),(1,32,32,128,3,1,1),(1,32,32,128,5,1,2),
     (9,3,32,128,4,2,1),(9,32,64,64,4,2,1),(9,64,128,32,4,2,1),(9,128,256,16,4,2,1),(9,256,512,8,4,2,1),(9,512,512,4,4,1,0),
     (8,32,64,64,3,2,1),(8,64,64,32,3,1,1),(8,128,128,16,3,1,1),(8,256,256,8,3,1,1),(8,512,512,4,3,1,1),(8,512,1024,4,4,1,0)]
tot = 0
for B,cin,cout,h,k,s,p in L:
    x = torch.randn(B,cin,h,h,device="cuda"); oh=(h+2*p-k)//s+1
    gy = torch.randn(B,cout,oh,oh,device="cuda")
    t = timeit(lambda: _wgrad(gy,x,k,s,p)); tot += t
    print(f"B={B} {cin}->{cout} {h}^2 k{k}s{s}: {t:6.1f} us", flush=True)
print("total", tot)

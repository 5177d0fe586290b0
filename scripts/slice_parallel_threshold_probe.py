import sys; sys.path.insert(0,'/root/repo')
import torch, laser_amd
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/6)
    ts.sort(); return ts[2]
for (M,N,K) in [(2048,2048,2048),(2048,2048,8192),(1024,2048,4096),(3072,3072,4096),(1536,1536,6144),(2048,1024,2048),(4096,4096,4096),(1280,1280,2560),(2048,2048,16384)]:
    A=(torch.rand((M,K),device='cuda')-0.5)*0.2; B=(torch.rand((K,N),device='cuda')-0.5)*0.2; C=torch.zeros((M,N),device='cuda')
    r=[]
    for on in (0, 100000):
        laser_amd.set_slice_parallel(on)
        r.append(bench(lambda: laser_amd.matmul(A,B,1,0,C)))
    fl=2.0*M*N*K
    print(f"{M}x{N}x{K} tiles64={((M+63)//64)*((N+63)//64)}: sequential {r[0]:.4f} ms ({fl/r[0]/1e9:.1f} TF)  slice-parallel {r[1]:.4f} ms ({fl/r[1]/1e9:.1f} TF)", flush=True)

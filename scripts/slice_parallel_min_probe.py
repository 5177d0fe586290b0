import sys; sys.path.insert(0,'/root/repo')
import torch, laser_amd
def bench(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/8)
    ts.sort(); return ts[2]
laser_amd.set_slice_parallel(2); laser_amd.set_slice_parallel(100000)   # tuning override: any tile count, from 2 slices
for (M,N,K) in [(512,512,1024),(768,768,1536),(1024,1024,1024),(1024,1024,1536),(1024,1024,2048),(1280,1280,1280),(512,512,2048),(256,256,1024),(1536,1536,1536),(128,128,1024),(1024,512,1024)]:
    A=(torch.rand((M,K),device='cuda')-0.5)*0.2; B=(torch.rand((K,N),device='cuda')-0.5)*0.2; C=torch.zeros((M,N),device='cuda')
    r=[]
    for on in (0, 1):
        laser_amd.set_slice_parallel(on)
        r.append(bench(lambda: laser_amd.matmul(A,B,1,0,C)))
    fl=2.0*M*N*K
    print(f"{M}x{N}x{K} tiles64={((M+63)//64)*((N+63)//64)} slices={(K+511)//512}: sequential {r[0]:.4f} ms ({fl/r[0]/1e9:.1f} TF)  slice-parallel {r[1]:.4f} ms ({fl/r[1]/1e9:.1f} TF)", flush=True)

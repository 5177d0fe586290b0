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
for (M,N,K) in [(8192,16,8192),(8192,32,8192),(8192,64,8192),(8192,128,8192),(65536,64,1024),(16,8192,8192),(64,8192,8192),(32768,32,2048)]:
    A=(torch.rand((M,K),device='cuda')-0.5)*0.2; B=(torch.rand((K,N),device='cuda')-0.5)*0.2; C=torch.zeros((M,N),device='cuda')
    ms=bench(lambda: laser_amd.matmul(A,B,1,0,C))
    byts=4.0*(M*K+K*N+M*N)
    print(f"{M}x{N}x{K}: {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.1f} TF  {byts/ms/1e9:.2f} TB/s  cfg {laser_amd.f32_configs()[laser_amd.last_f32_config()]}", flush=True)

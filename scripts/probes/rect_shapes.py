import json, sys
sys.path.insert(0, "/root/repo")
import torch, laser_amd
from scripts.bench_configs import ev_time
names = {0: "compiler"}
for (M, N, K) in [(8192, 1024, 8192), (512, 16384, 4096), (16384, 512, 1024), (2048, 2048, 8192), (1024, 1024, 8192), (4096, 1024, 1024), (640, 640, 4096), (3000, 3000, 3000), (5000, 5000, 5000)]:
    A = (torch.rand((M, K), device="cuda") - 0.5) * 0.2; B = (torch.rand((K, N), device="cuda") - 0.5) * 0.2; C = torch.zeros((M, N), device="cuda")
    rec = {"shape": [M, N, K]}
    for mode in (0, 1):
        laser_amd.set_float_mode(mode)
        for asm in (1, 0):
            laser_amd.set_f32_asm(asm)
            ms, _ = ev_time(lambda: laser_amd.matmul(A, B, 1, 0, C), iters=5)
            rec[("laser" if mode == 0 else "fast") + ("_asm" if asm else "_off")] = [round(2.0 * M * N * K / ms / 1e9, 1), laser_amd.last_f32_asm()]
    print(json.dumps(rec), flush=True)
laser_amd.set_float_mode(0); laser_amd.set_f32_asm(1)

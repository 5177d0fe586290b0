"""The reference's conv bench shape 200 times through one form of the small-channel direct kernels (argv[1] = option conv_direct):
the command the rocprofv3 passes of scripts/gpu_conv_direct_pmc.sh wrap."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import laser_amd
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
g = torch.Generator(device="cuda").manual_seed(1)
ishape, kshape, pad, st = (batch, 3, 224, 224), (20, 3, 3, 3), (0, 0), (1, 1)
x = torch.rand(ishape, generator=g, device="cuda"); w = torch.rand(kshape, generator=g, device="cuda")
oshape = laser_amd.conv2d_out_shape(ishape, kshape, pad, st); o = torch.zeros(oshape, device="cuda")
laser_amd.set_option("conv_direct", v)
for _ in range(200):
    laser_amd.conv2d_im2col(o, oshape, x, ishape, w, kshape, pad, st, None)
torch.cuda.synchronize()

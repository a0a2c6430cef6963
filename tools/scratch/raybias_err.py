import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import stnerf_oracle as O
from stnerf_amd import ops, synthetic as syn
torch.manual_seed(3)
for use_time in (False, True):
    rs = np.random.RandomState(11)
    sd = syn.spacenet_state("net", rs, use_time)
    n = 4000
    dirs = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    times = torch.where(torch.rand(n) < 0.5, torch.floor(torch.rand(n) * 30), torch.rand(n) * 30) + 1
    net = ops.pack_spacenet(sd, "net")
    def want(dt):
        enc = [O.positional_encoding(dirs.to(dt), 4)]
        if use_time: enc.append(O.positional_encoding(times.to(dt).reshape(n, 1), 10))
        e = torch.relu(torch.cat(enc, -1))
        W, b = sd["net.rgb_net.1.weight"].to(dt), sd["net.rgb_net.1.bias"].to(dt)
        return b + e @ W[:, 256:].T, e
    w64, e64 = want(torch.float64); w32, e32 = want(torch.float32)
    got = ops.rgb_ray_bias(net, dirs.cuda(), times.cuda() if use_time else None).cpu().double()
    f = lambda a: (float((a - w64).abs().max()), float(((a - w64) ** 2).mean().sqrt()))
    print("use_time", use_time, "raybias err vs fp64: GPU max %.3e rms %.3e | CPU f32 max %.3e rms %.3e | scale %.3f" % (*f(got), *f(w32.double()), float(w64.abs().max())))
    # PE accuracy on the GPU
    pe_d = ops.encode(dirs.cuda(), 4).cpu()
    print("  PE4(dir) err: GPU max %.3e rms %.3e | CPU f32 max %.3e rms %.3e" % (float((pe_d.double() - O.positional_encoding(dirs.double(), 4)).abs().max()), float(((pe_d.double() - O.positional_encoding(dirs.double(), 4))**2).mean().sqrt()), float((O.positional_encoding(dirs, 4).double() - O.positional_encoding(dirs.double(), 4)).abs().max()), float(((O.positional_encoding(dirs, 4).double() - O.positional_encoding(dirs.double(), 4))**2).mean().sqrt())))
    if use_time:
        t1 = times.reshape(n, 1)
        pe_t = ops.encode(t1.cuda(), 10).cpu()
        r64 = O.positional_encoding(t1.double(), 10)
        print("  PE10(t) err: GPU max %.3e rms %.3e | CPU f32 max %.3e rms %.3e" % (float((pe_t.double() - r64).abs().max()), float(((pe_t.double() - r64)**2).mean().sqrt()), float((O.positional_encoding(t1, 10).double() - r64).abs().max()), float(((O.positional_encoding(t1, 10).double() - r64)**2).mean().sqrt())))

import sys, time, torch
sys.path.insert(0, '.')
from reinforcement_learning_rendezvous_amd import _native
if len(sys.argv) > 1: _native.LIB_PATH = sys.argv[1]
from reinforcement_learning_rendezvous_amd.batch import RendezvousBatch
n = 65536
for variant in ("split", "fused"):
    env = RendezvousBatch(n, device="cuda:0", storage="f32", seed=0, variant=variant)
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [(torch.rand((n, 6), device="cuda:0", generator=gen) * 2 - 1).contiguous() for _ in range(16)]
    env.reset()
    for t in range(64): env.step(acts[t % 16])
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(512): env.step(acts[t % 16])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(sys.argv[1:] , variant, "us/launch", e0.elapsed_time(e1) * 1e3 / (8 * 512))

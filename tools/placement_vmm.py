#!/usr/bin/env python3
"""Diagnostic: the fused step kernel at 4.2 M envs with the WORKSPACE mapped from separately created physical blocks (HIP virtual-memory
API, tools/vmm_alloc.hip) in different block sizes and orders, against the plain allocation — does physical scatter select the fast
mode of profiles/r02_large_n_placement.txt?"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from reinforcement_learning_rendezvous_amd import batch as B

MiB = 1 << 20
vmm = C.CDLL(os.path.join(ROOT, "tools", "libvmm_alloc.so"))
vmm.vmm_alloc.restype = C.c_void_p
vmm.vmm_alloc.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
vmm.vmm_free.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
acts = [(torch.rand((n, 6), device=dev, generator=g) * 2 - 1).contiguous() for _ in range(2)]


class Raw:
    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def data_ptr(self):
        return self.ptr


plain_alloc = B.RendezvousBatch._alloc


def run(label, block=None, order=0):
    held = {}

    def alloc(self, name, shape, dtype):
        if name != "workspace" or block is None:
            return plain_alloc(self, name, shape, dtype)
        gran = C.c_size_t(0)
        p = vmm.vmm_alloc(shape[0], block, order, 0, C.byref(gran))
        if not p:
            raise RuntimeError("vmm_alloc failed")
        held["ws"] = (p, shape[0], gran.value)
        return Raw(p, shape[0])
    B.RendezvousBatch._alloc = alloc
    env = B.RendezvousBatch(n, device="cuda:0", storage="f32", seed=0)
    env.reset()
    for t in range(24):
        env.step(acts[t % 2])
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(16):
            env.step(acts[t % 2])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 16)
    st = env.get_stats()
    env.close()
    del env
    if held:
        p, nb, gran = held["ws"]
        rc = vmm.vmm_free(p, nb, block, 0)
        label += f" (granularity {gran // 1024} KiB, free rc {rc})"
    torch.cuda.empty_cache()
    print(f"{label:78s}: {best:7.1f} us per launch, {293 * n / (best * 1e-6) / 8e12:.3f} of 8 TB/s, episodes {st['episodes']}", flush=True)


GiB = 1024 * MiB
for rep in range(8):
    run("plain allocation (torch caching allocator -> hipMalloc)")
    run("workspace as ONE block", 4 * GiB, 0)

"""Does a hipMemsetAsync node captured in a hipGraph execute on replay on this stack, and does a multi-block torch reduction
(at::native::reduce_kernel: a temporary semaphore buffer cleared with cudaMemsetAsync) stay correct across replays?"""
import ctypes, sys
import torch
dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
x = torch.ones(1 << 20, device=dev)
buf = torch.full((1024,), 7, dtype=torch.int32, device=dev)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    s = torch.cuda.current_stream().cuda_stream
    rc = hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(4096), ctypes.c_void_p(s))
torch.cuda.synchronize()
print("memset rc", rc, "after capture (not executed yet):", int(buf.sum()))
g.replay(); torch.cuda.synchronize()
print("after replay 1:", int(buf.sum()), "(0 = the memset node executed)")
buf.fill_(5); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("after refill + replay 2:", int(buf.sum()))
# multi-block reductions inside a graph, with other small allocations around them that reuse the semaphore memory
xs = [torch.rand(1 << 21, device=dev) for _ in range(4)]
want = [float(t.double().mean()) for t in xs]
out = torch.zeros(4, device=dev)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    vals = []
    for t in xs:
        junk = torch.full((128,), 3.0e38, device=dev) * 1.0   # small temporaries full of non-zero bits, freed at once
        del junk
        vals.append(t.mean())
        junk2 = torch.full((64,), -1, dtype=torch.int32, device=dev) + 0
        del junk2
    out.copy_(torch.stack(vals))
bad = 0
for it in range(200):
    f = 1.0 + 0.01 * ((it * 7) % 13)
    for t in xs:
        t.mul_(f)  # the inputs CHANGE between replays: a reduction whose final write is skipped shows as a stale value
    want = [float(t.double().mean()) for t in xs]
    g2.replay()
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    for t in xs:
        t.div_(f)
    if any(abs(o[k] - want[k]) > 1e-4 * abs(want[k]) for k in range(4)):
        bad += 1
        if bad <= 5:
            print("replay %d WRONG" % it, o, want)
print("reductions wrong in %d of 200 replays" % bad)
for k in range(4):
    buf.fill_(5 + k); torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    print("memset replay", k + 3, "first words", buf[:4].cpu().tolist(), "distinct", len(set(buf.cpu().tolist())))

import time, torch, numpy as np
n = 37_748_736 // 4
a = np.random.rand(n).astype(np.float32)
d = torch.empty(n, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
ta = torch.from_numpy(a)
print("H2D pageable 37.7MB: %.2f ms" % t(lambda: d.copy_(ta)))
out = torch.empty(n)
print("D2H pageable: %.2f ms" % t(lambda: out.copy_(d)))
pa = ta.pin_memory(); po = torch.empty(n).pin_memory()
print("H2D pinned: %.2f ms" % t(lambda: d.copy_(pa, non_blocking=True)))
print("D2H pinned: %.2f ms" % t(lambda: po.copy_(d, non_blocking=True)))
t0 = time.perf_counter(); out.copy_(po); print("host memcpy pinned->pageable: %.2f ms" % ((time.perf_counter()-t0)*1e3))
t0 = time.perf_counter(); b = np.empty(n, np.float32); b[:] = 0; print("np.empty+touch: %.2f ms" % ((time.perf_counter()-t0)*1e3))
rt = torch.cuda.cudart()
x = torch.empty(n)
t0 = time.perf_counter(); r = rt.cudaHostRegister(x.data_ptr(), n*4, 0); t1 = time.perf_counter(); print("hostRegister: %.2f ms rc=%s" % ((t1-t0)*1e3, r))
print("D2H registered: %.2f ms" % t(lambda: x.copy_(d, non_blocking=True)))
t0 = time.perf_counter(); rt.cudaHostUnregister(x.data_ptr()); print("hostUnregister: %.2f ms" % ((time.perf_counter()-t0)*1e3))

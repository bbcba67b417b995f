import torch, time
dev = torch.device('cuda', 0)
n = 5_400_000_000 // 8
x = torch.rand(n, device=dev, dtype=torch.float64)
y = torch.empty_like(x)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: y.copy_(x)); print("copy 5.4GB: %.3f ms  -> %.2f TB/s (r+w)" % (dt*1e3, 2*n*8/dt/1e12))
dt = t(lambda: x.sum()); print("sum  5.4GB: %.3f ms  -> %.2f TB/s (read)" % (dt*1e3, n*8/dt/1e12))
dt = t(lambda: y.zero_()); print("zero 5.4GB: %.3f ms  -> %.2f TB/s (write)" % (dt*1e3, n*8/dt/1e12))

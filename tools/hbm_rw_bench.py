"""Read / write / copy bandwidth of the box as plain torch kernels see it (fill = write only, sum = read only, copy = 1 : 1)."""
import time
import torch
n = 1 << 28          # 1 GiB of fp32
x = torch.empty(n, device='cuda'); y = torch.empty(n, device='cuda')
def bench(name, f, nbytes, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f'{name:28s} {dt*1e6:8.1f} us  {nbytes/dt/1e12:5.2f} TB/s')
bench('fill (write 1 GiB)', lambda: x.fill_(1.5), 4 * n)
bench('zero_ (memset 1 GiB)', lambda: x.zero_(), 4 * n)
bench('sum (read 1 GiB)', lambda: x.sum(), 4 * n)
bench('copy (read 1 + write 1 GiB)', lambda: y.copy_(x), 8 * n)
bench('mul 2 -> 1 (read 2 + write 1)', lambda: torch.mul(x, y, out=y), 12 * n)
xs = x[: n // 4]; o = torch.empty(n, device='cuda')
bench('repeat 1 -> 4 (read 1/4 + write 1)', lambda: o.view(4, -1).copy_(xs.view(1, -1).expand(4, -1)), 5 * n)

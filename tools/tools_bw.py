import torch, time
M = 16 * 321 * 101
for shape in ((M, 256), (M, 64)):
    y = torch.empty(*shape, device='cuda'); x = torch.randn(*shape, device='cuda')
    for name, f in (('fill', lambda: y.fill_(1.0)), ('copy', lambda: y.copy_(x)), ('add', lambda: torch.add(x, 1.0, out=y))):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): f()
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        nbytes = y.numel() * 4 * (1 if name == 'fill' else 2)
        print(shape, name, f'{dt*1e6:.1f} us', f'{nbytes/dt/1e9:.0f} GB/s')

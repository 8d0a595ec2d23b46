import torch, time
x = torch.empty(120, 512, 256, 256, device='cuda')
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n * 1e3
ms = t(lambda: x.fill_(1.0)); print('fill 16.1 GB: %.2f ms = %.2f TB/s' % (ms, x.numel() * 4 / ms / 1e9))
ms = t(lambda: x.zero_()); print('zero 16.1 GB: %.2f ms = %.2f TB/s' % (ms, x.numel() * 4 / ms / 1e9))
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x)); print('copy 16.1 GB: %.2f ms = %.2f TB/s (r+w)' % (ms, 2 * x.numel() * 4 / ms / 1e9))
ms = t(lambda: x.sum()); print('sum (read) 16.1 GB: %.2f ms = %.2f TB/s' % (ms, x.numel() * 4 / ms / 1e9))

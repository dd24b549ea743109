import torch, sys
sys.path.insert(0, '/root/repo')
n, h = 64000, 256
dt = torch.randn(n, h, device='cuda').half(); x = torch.randn(n, h, device='cuda').half()
def wg(per):
    chunks = n // per
    rows = chunks * per
    part = torch.bmm(dt[:rows].view(chunks, per, h).transpose(1, 2), x[:rows].view(chunks, per, h))
    return part.float().sum(0)
def wg32(per):
    chunks = n // per
    part = torch.empty(chunks, h, h, device='cuda', dtype=torch.float32)
    torch.bmm(dt.view(chunks, per, h).transpose(1, 2), x.view(chunks, per, h), out=None)
for per in (250, 500, 1000, 2000, 4000, 8000):
    for _ in range(3): wg(per)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): wg(per)
    e.record(); torch.cuda.synchronize()
    print(per, round(a.elapsed_time(e) / 20 * 1e3, 1), 'us')
# single tall matmul for reference
for _ in range(3): torch.matmul(dt.t(), x)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): torch.matmul(dt.t(), x)
e.record(); torch.cuda.synchronize()
print('tall', round(a.elapsed_time(e) / 20 * 1e3, 1), 'us')

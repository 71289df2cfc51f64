import torch
x = torch.empty(256*1024*1024, dtype=torch.float32, device='cuda')  # 1 GiB
y = torch.empty_like(x)
for n in (x.numel(), x.numel()//4, x.numel()//16):
    a, b = x[:n], y[:n]
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f'copy {n*4/2**20:.0f} MiB: {t*1e3:.1f} us, read+write {2*n*4/t/1e9:.2f} TB/s')

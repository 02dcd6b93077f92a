import torch, time
dev = torch.device("cuda", 0)
n = 96 * 3840 * 2160 * 3 // 2
a = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev); b = torch.empty_like(a)
for view, name in ((torch.uint8, "u8"), (torch.int32, "i32"), (torch.int64, "i64")):
    x, y = a.view(view), b.view(view)
    y.copy_(x); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): y.copy_(x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(name, "copy of %.2f GB: %.3f ms -> %.2f TB/s (read + write)" % (n / 1e9, ms, 2 * n / ms / 1e9))

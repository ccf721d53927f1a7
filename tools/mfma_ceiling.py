"""What dense f16 matrix rate does the vendor GEMM (hipBLASLt through torch.matmul) sustain on this box?  A reference point
for the fringe kernels' roofline fraction: the 2.5 PFLOP/s peak assumes 2.4 GHz, the chip holds ~1.7-1.8 GHz under matrix load."""
import torch, time
dev = torch.device('cuda', 0)
for dt in (torch.float16, torch.bfloat16):
    for n in (4096, 8192, 16384):
        a = torch.randn(n, n, device=dev, dtype=dt)
        b = torch.randn(n, n, device=dev, dtype=dt)
        for _ in range(3):
            c = a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(3, int(2e14 / (2.0 * n ** 3)))
        e0.record()
        for _ in range(reps):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('%s %5d^3: %.3f ms  %.0f TFLOP/s  (%.2f of 2.5 PFLOP/s)' % (str(dt)[6:], n, ms, 2.0 * n ** 3 / ms / 1e9, 2.0 * n ** 3 / ms / 1e9 / 2500))

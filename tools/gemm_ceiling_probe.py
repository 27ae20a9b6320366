"""What a plain library GEMM reaches on this box (hipBLASLt through torch.matmul): the practical 16-bit matrix-core ceiling next to the
2.5 PFLOP/s datasheet figure the roofline entries are priced against.   python tools/gemm_ceiling_probe.py"""
import json
import time

import torch

dev = torch.device("cuda")
for dt in (torch.float16, torch.bfloat16):
    for n, k in ((8192, 8192), (16384, 4608), (8192, 4608)):
        a = torch.randn(n, k, device=dev, dtype=dt)
        b = torch.randn(k, n if n == 8192 else 512, device=dev, dtype=dt)
        for _ in range(5):
            c = a @ b
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 30
        for _ in range(R):
            c = a @ b
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / R
        print(json.dumps({"dtype": str(dt), "m": n, "k": k, "n": b.shape[1], "ms": round(dtm * 1e3, 4),
                          "tflops": round(2.0 * n * k * b.shape[1] / dtm / 1e12, 1)}), flush=True)

import torch, time
dev = "cuda"
B = 524288
for (M, K) in [(64, 10), (64, 64), (4, 64)]:
    W = torch.randn(M, K, device=dev); b = torch.randn(M, device=dev)
    Xf = torch.randn(K, B, device=dev); Xb = Xf.t().contiguous()
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    print(f"M={M} K={K}: feature-major addmm {t(lambda: torch.addmm(b[:, None], W, Xf)):8.1f} us | batch-major linear {t(lambda: torch.nn.functional.linear(Xb, W, b)):8.1f} us"
          f" | fm matmul+add {t(lambda: W @ Xf + b[:, None]):8.1f} us | bf16 batch-major {t(lambda: torch.nn.functional.linear(Xb.bfloat16(), W.bfloat16(), b.bfloat16())):8.1f} us")

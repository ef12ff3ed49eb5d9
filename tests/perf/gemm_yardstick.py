"""What the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the DiT-L products at 1 024 cells (16 384 tokens), bf16
operands, as a yardstick for bgemm256_kernel / bgemm8_kernel (not used by the product)."""
import torch, time
dev = torch.device("cuda:0")
T = 16384
shapes = [("qkv fwd", T, 3072, 1024), ("proj fwd", T, 1024, 1024), ("w1 fwd", T, 2736, 1024), ("c_proj fwd", T, 1024, 2736),
          ("dgrad qkv", T, 1024, 3072), ("dgrad c_proj", T, 2736, 1024), ("wgrad qkv", 3072, 1024, T), ("wgrad w1", 2736, 1024, T)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{name:14s} M={M:6d} N={N:5d} K={K:6d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s (bf16 in, bf16 out, A B^T)")

import sys, time, torch
sys.path.insert(0, "/root/repo")
from scldm_amd.stochastic_layers import NegativeBinomial
from scldm_amd.datamodule import dense_to_csr
N, G = 8192, 17002
mu = torch.rand(N, G, device="cuda") * 0.5
theta = torch.rand(N, G, device="cuda") + 0.5
nb = NegativeBinomial(mu, theta)
for _ in range(2): x = nb.sample()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): x = nb.sample()
torch.cuda.synchronize(); print("NB sample (8192 x 17002):", (time.perf_counter() - t0) / 3 * 1e3, "ms")
for _ in range(2): dense_to_csr(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): ip, idx, dat = dense_to_csr(x)
torch.cuda.synchronize(); print("dense_to_csr:", (time.perf_counter() - t0) / 3 * 1e3, "ms, nnz frac", dat.numel() / x.numel())
t0 = time.perf_counter(); xc = x.cpu(); print("dense .cpu():", (time.perf_counter() - t0) * 1e3, "ms")
t0 = time.perf_counter(); a, b, c = ip.cpu(), idx.cpu(), dat.cpu(); print("csr arrays .cpu():", (time.perf_counter() - t0) * 1e3, "ms")

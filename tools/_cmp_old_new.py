import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_train as T
class MP:
    def setenv(self, *a, **k): pass
    def delenv(self, *a, **k): pass
for n in (48, 50, 5):
    torch.manual_seed(0)
    e = T._bf16_step_vs_oracle(n, fused=True, monkeypatch=MP())
    worst = sorted(e.items(), key=lambda kv: -kv[1])[:3]
    print(os.environ.get("SCLDM_LIB", "tree")[-12:], n, ["%s %.4e" % (k[-22:], v) for k, v in worst])

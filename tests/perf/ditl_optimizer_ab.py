import sys, time; sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
for name in ("replogle_train_ditl_b256", "replogle_train_ditl_b1024"):
    for opt in ("native", "torch"):
        tl = dict(bench.TRAIN_WORKLOADS[name])
        dt, _ = bench.time_training(tl, "bf16", dev, 6, 3, False, 1, optimizer=opt)
        print(name, opt, f"{1e3 * dt / 6:.2f} ms/step")
        torch.cuda.empty_cache()

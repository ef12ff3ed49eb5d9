import sys, time, cProfile, pstats; sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
tw = dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"])
for opt in ("native", "torch"):
    dt, _ = bench.time_training(tw, "bf16", dev, 20, 5, False, 1, optimizer=opt)
    print(opt, "eager ms/step", 1e3 * dt / 20)
pr = cProfile.Profile(); pr.enable()
dt, _ = bench.time_training(tw, "bf16", dev, 20, 5, False, 1)
pr.disable()
print("profiled ms/step", 1e3 * dt / 20)
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)

"""bench.train_end_to_end alone (the LDM step from tokenised counts with clipping and EMA): graph ms, eager ms, AdamW-stage us."""
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch, bench
r = bench.train_end_to_end(dict(bench.TRAIN_WORKLOADS["replogle_train_b1024"]), "bf16", torch.device("cuda:0"))
print(round(r["ms_per_step_graph"],3), round(r["ms_per_step_eager_one_c_call"],3), round(1e3*r["stage_ms"]["adamw_and_ema"],1))

"""Generic bf16 training route on other shapes than DiT-L (512 / 768 / 2048 wide): this round's kernels (LDS-DMA GEMM, matrix-core
attention, fused elementwise steps, batched weight gradients beside the chain) against the round-2 set of knobs, in child processes."""
import os, sys, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
OLD = dict(SCLDM_BGEMM8="0", SCLDM_DGRAD_WT="0", SCLDM_MLP_MERGE="0", SCLDM_DHID16="0", SCLDM_Y16="0", SCLDM_GRAD16="0", SCLDM_ATTN_MFMA="0",
           SCLDM_FUSE_RES="0", SCLDM_FUSE_GATE="0", SCLDM_ADA_STACKED="0", SCLDM_BATCH_SIDE="0", SCLDM_EPI_LDS="0", SCLDM_MIN_TILES256="224")
if len(sys.argv) > 1 and sys.argv[1] == "run":
    import torch, bench
    n_embed, n_layer, n_head, B = (int(a) for a in sys.argv[2:6])
    wl = dict(vocab={"cell_line": 4, "gene": 2024}, strategy="joint", B=B, shape=dict(n_embed=n_embed, n_layer=n_layer, n_head=n_head))
    dt, loss = bench.time_training(wl, "bf16", torch.device("cuda:0"), 6, 2, False, 1)
    print(f"{n_embed} wide x {n_layer} layers, {n_head} heads, {B} cells [{sys.argv[6]}]: {1e3 * dt / 6:8.2f} ms/step  loss {loss:.4f}")
else:
    for shape in ((512, 12, 8, 512), (512, 12, 8, 2048), (768, 12, 12, 1024), (2048, 8, 32, 512), (1024, 24, 16, 128)):
        for tag, env in (("round 3", {}), ("round-2 knobs", OLD), ("round 3", {}), ("round-2 knobs", OLD)):
            subprocess.run([sys.executable, __file__, "run", *map(str, shape), tag], env=dict(os.environ, **env))

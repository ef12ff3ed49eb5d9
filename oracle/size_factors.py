"""CPU restatement of LatentDiffusion._sample_log_size_factors.  TEST INFRASTRUCTURE ONLY.

Follows src/scldm/models.py:473-597: no statistics / no condition -> zeros (:493-497); joint strategy with a joint key
present in both maps -> per-cell key "i_j" of the component labels looked up in joint_idx_2_classes, then mean/std of that
joint class (:501-551); otherwise one condition key - size_factor_condition_key if usable, else the alphabetically first
key common to the condition and both maps (:553-582) - and a per-cell lookup (:584-596).  Cells without statistics stay 0.
The per-cell draw Normal(mean, std).sample() is restated as mean + std * eps with the standard-normal eps injected.
Pinned against outputs of the reference method itself (tests/golden/size_factors.npz; make_golden.gen_size_factors
executes the method from the reference file) for the lookup logic; the draw is distributional.
"""
from __future__ import annotations

import numpy as np


def sample_log_size_factors(vocab_encoder, condition_strategy: str, condition: dict | None, batch_size: int,
                            eps: np.ndarray | None = None) -> np.ndarray:
    mu_map = getattr(vocab_encoder, "mu_size_factor", None)
    sd_map = getattr(vocab_encoder, "sd_size_factor", None)
    out = np.zeros(batch_size, dtype=np.float32)
    eps = np.zeros(batch_size, dtype=np.float32) if eps is None else eps
    if condition is None or mu_map is None or sd_map is None:
        return out
    joint_idx_2_classes = getattr(vocab_encoder, "joint_idx_2_classes", None)
    joint_key = getattr(vocab_encoder, "joint_key", None)
    use_joint = (condition_strategy == "joint" and joint_idx_2_classes is not None and joint_key is not None
                 and joint_key in mu_map and joint_key in sd_map)
    if use_joint:
        components = getattr(vocab_encoder, "joint_components", None)
        keys = [k for k in components if k in condition] if components is not None else list(condition.keys())
        if any(len(condition[k]) != batch_size for k in keys):
            return out
        for i in range(batch_size):
            key = "_".join(str(int(condition[k][i])) for k in keys)
            if key not in joint_idx_2_classes:
                continue
            cls = joint_idx_2_classes[key]
            mean, std = mu_map[joint_key].get(cls), sd_map[joint_key].get(cls)
            if mean is None or std is None:
                continue
            out[i] = np.float32(mean) + np.float32(std) * eps[i]
        return out
    sel = getattr(vocab_encoder, "size_factor_condition_key", None)
    if not (sel and sel in condition and sel in mu_map and sel in sd_map):
        inter = sorted(set(condition.keys()) & set(mu_map.keys()) & set(sd_map.keys()))
        if not inter:
            return out
        sel = inter[0]
    labels = condition[sel]
    if len(labels) != batch_size:
        raise ValueError(f"Condition '{sel}' length ({len(labels)}) must match batch size ({batch_size})")
    for i in range(batch_size):
        mean, std = mu_map[sel].get(int(labels[i])), sd_map[sel].get(int(labels[i]))
        if mean is None or std is None:
            continue
        out[i] = np.float32(mean) + np.float32(std) * eps[i]
    return out

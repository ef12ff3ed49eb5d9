"""The stand-alone GEMM harness as a parity test: tests/perf/gemm_probe runs bgemm8_kernel (LDS-DMA staging, both operand
orientations, every epilogue) against the register-staged bgemm256_kernel on the same operands and counts the elements that
differ - same MFMA, same k order, exact transposition: the count must be ZERO.  Shapes: ragged tiles in m and n, k tails, one and
two stages, the DiT-L widths with their 2 732 / 2 736 columns."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "perf", "gemm_probe")


def _probe(*args):
    if not os.path.exists(PROBE):    # (build.sh builds it; a box without the binary compiles it here)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I", "include",
                        "tests/perf/gemm_probe.hip", "-o", PROBE], cwd=ROOT, check=True)
    r = subprocess.run([PROBE, *map(str, args)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def _check(out, min_checked):
    counts = re.findall(r"^\s*(.*?)\s+[\d.]+ us .*?(\d+) elements differ", out, re.M)
    assert len(counts) >= min_checked, out[-3000:]
    bad = [(name, int(n)) for name, n in counts if int(n) != 0]
    assert not bad, bad
    return len(counts)


def test_k_contiguous_products_are_bit_identical_to_the_register_staged_kernel():
    # (N % 8 == 0 or N % 8 == 4: the shapes the host sends to the vector / LDS epilogues)
    out = _probe(4000, 1000, 520, 300, 264, 64, 520, 512, 128, 2100, 2732, 1024, 777, 1024, 2736)
    n = _check(out, 5 * 7)
    print(f"[parity] bgemm8_kernel<KC, KC> vs bgemm256_kernel: {n} kernel x shape comparisons, every element identical")


def test_m_contiguous_products_are_bit_identical_and_row_sums_agree():
    out = _probe("mc", 1000, 520, 4000, 300, 264, 64, 2732, 1024, 2080, 520, 512, 136)
    n = _check(out, 4 * 2)
    sums = re.findall(r"row sums: max \|diff\| ([\d.e+-]+) against max \|value\| ([\d.e+-]+)", out)
    assert len(sums) >= 8 and all(float(d) <= 1e-5 * max(1.0, float(v)) for d, v in sums), sums
    print(f"[parity] bgemm8_kernel<MC, MC> vs bgemm256_kernel: {n} comparisons identical, {len(sums)} row-sum vectors within 1e-5")


def test_small_tile_lds_dma_kernel_is_bit_identical_to_the_register_staged_one():
    """bgemm4_kernel (128 x 128 tiles, two workgroups per CU, results through LDS or element-wise) against bgemm_kernel<KC, KC>."""
    out = _probe("small", 1000, 520, 328, 4096, 1024, 1024, 300, 264, 64, 2100, 2732, 1096, 777, 1024, 2736)
    n = _check(out, 5 * 4)
    print(f"[parity] bgemm4_kernel vs bgemm_kernel: {n} kernel x shape comparisons, every element identical")

"""The library's kernels must not read scratch memory they have not written: with CURV_DEBUG_POISON=1 every scratch buffer is
filled with NaN bit patterns each time it is handed to the library (and no descriptor table is assumed to survive in it), so an
unwritten gap shows up as a non-finite factor instead of depending on what an earlier call left there.  Round 6 found one that
way: the gathered border arrays of the shifted-correlation path are padded to a multiple of four floats and the LDS-DMA kernel,
which zeroes only ONE operand side behind a row's end, read the padding (tests/test_syrk_gpu.py::test_conv_factors[...case27]
failed once in ~25 runs of the suite, whenever the workspace held fp64 data whose bits are a NaN in fp32)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_factor_build_and_estimator_chain_with_poisoned_scratch(gpu):
    """The environment variable is read when `curvature_amd.ops` is imported, hence a subprocess: every shifted-correlation
    geometry, the unfolded / pre-tiled / packed-pair paths, and the KFAC / EFB / INF chain on LeNet-5."""
    env = dict(os.environ, CURV_DEBUG_POISON="1")
    cases = " or ".join(f"case{i}]" for i in (3, 7, 12, 14, 18, 20, 21, 22, 23, 24, 25, 26, 27, 30, 38, 39, 43))
    for args in (["tests/test_syrk_gpu.py", "-k", cases], ["tests/test_kfac_api_gpu.py", "tests/test_efb_inf_gpu.py", "-k", "not lowrank_path_of_the_library"]):
        proc = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + args,
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert proc.returncode == 0, proc.stdout[-3000:]

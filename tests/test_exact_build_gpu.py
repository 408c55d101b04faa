"""The exact-arithmetic BUILD of the library (-DRN_EXACT_OCC_EXP: the library's own exponential
sequence in the occupancy and the softmax instead of v_exp_f32 on the rounded product, DESIGN.md
section 6) stays under test: built into a temporary path, named by its rn_version(), and run
through the saturated golden of the reference's mrf/mrf_np.py (tests/test_saturated_golden.py)
and the plane-sweep parity tests in a child process that loads THAT build (RAYNET_HIP_LIB)."""
import ctypes
import os
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.gpu
def test_exact_exponential_build_passes_the_saturated_golden(tmp_path):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU (no CPU fallback exists)"
    from raynet_amd import _lib
    out = str(tmp_path / "libraynet_hip_exact.so")
    _lib.build(force=True, extra_flags=["-DRN_EXACT_OCC_EXP"], out=out)
    lib = ctypes.CDLL(out)
    lib.rn_version.restype = ctypes.c_char_p
    version = lib.rn_version().decode()
    assert " RN_EXACT_OCC_EXP" in version and "-DRN_EXACT_OCC_EXP" in version, version
    env = dict(os.environ, RAYNET_HIP_LIB=out)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu",
                        os.path.join(REPO, "tests", "test_saturated_golden.py"),
                        os.path.join(REPO, "tests", "test_hip_parity_gpu.py"), "-k",
                        "resident_kernels or similarities or bp_backend or fused_k1_k2 or resident_scene_path"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout

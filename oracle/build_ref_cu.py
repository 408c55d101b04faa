#!/usr/bin/env python3
"""Builds the reference's OWN device code for gfx950 -- TEST INFRASTRUCTURE, never the product.

The reference compiles its kernels at run time: cuda_implementations/raynet_fp.py:43-53 joins six
.cu files, appends the fused kernels as a string literal, fills the $placeholders with
string.Template.substitute (:230-248) and hands the text to nvcc through PyCUDA.  This script does
the same with the toolchain of this image:

  * reads the six .cu files and the kernel literals of raynet_fp.py, similarities.py and
    mvcnn_with_ray_marching_and_voxels_mapping.py WHERE THEY LIE under /root/reference (the
    literals through `ast`: the modules import PyCUDA and cannot be imported),
  * substitutes the placeholders exactly as the reference does (str() of the same values),
  * compiles the text UNCHANGED -- no macros, no shim header, no stand-in for anything --
        hipcc -x hip -include hip/hip_runtime.h --offload-arch=gfx950 --cuda-device-only
    (hip_runtime.h is what `hipcc file.hip` includes implicitly; PyCUDA's nvcc does the same with
    cuda_runtime.h),
  * writes ONLY the code objects and a manifest into oracle/_ref/ (git-ignored; travels to the GPU
    box with the snapshot).  The substituted text lives in a temp dir and is deleted.

Every shape is built twice:
  raynet_ref_<shape>_fma.co    default floating-point contraction (what nvcc --fmad=true and hipcc
                               do by default: `out += m * v` becomes one fused multiply-add) --
                               what a PyCUDA run of the reference computes;
  raynet_ref_<shape>_nofma.co  -ffp-contract=off: the convention of the CPU oracle, of the
                               reference's NumPy `project` and of its Cython traversal (gcc on
                               x86-64 fuses nothing).

tests/ref_cu.py loads them with hipModuleLoad and launches the reference's kernels with the
reference's launch shape (one thread per ray).  Without /root/reference (the GPU box) this script
keeps the prebuilt files and exits 0.
"""
import ast
import json
import os
import shutil
import subprocess
import sys
import tempfile
from string import Template

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF = os.environ.get("RAYNET_REFERENCE", "/root/reference")
CUDIR = os.path.join(REF, "raynet", "cuda_implementations")

CU_FILES = ["ray_tracing.cu", "utils.cu", "planes_voxels_mapping.cu",
            "feature_similarities.cu", "sampling_schemes.cu", "mrf_bp.cu"]      # raynet_fp.py:43-50
LITERAL_SOURCES = ["raynet_fp.py", "similarities.py", "mvcnn_with_ray_marching_and_voxels_mapping.py"]

# the three shapes of tests/golden/gen_cu_crosscheck.py (their inputs are committed fixtures), the
# mock Restrepo plumbing case and BASELINE's configs 2 / 4
SHAPES = {
    "small": dict(M=48, D=16, N=3, F=8, H=24, W=32, padding=5, bbox=[-1, -1, -1, 1, 1, 1],
                  grid=[16, 16, 16]),
    "wide": dict(M=96, D=64, N=5, F=32, H=30, W=40, padding=11, bbox=[-1, -1, -1, 1, 1, 1],
                 grid=[32, 32, 32]),
    "aniso": dict(M=64, D=32, N=4, F=16, H=20, W=28, padding=11,
                  bbox=[-1.5, -1, -0.5, 1.5, 1, 0.75], grid=[24, 16, 10]),
    # BASELINE.json configs[0]: the mock Restrepo scene's box -- its -0.7 is NOT a float32 value, so the
    # decimal text the reference substitutes ("-0.7", a double literal) and the float32 the caller
    # holds differ in the 9th digit (tests/test_reference_kernels.py: what that moves)
    "config1": dict(M=96, D=16, N=2, F=32, H=36, W=64, padding=11, bbox=[-5, -5, -0.7, 5, 5, 1.5],
                    grid=[32, 32, 32]),
    "config2": dict(M=384, D=64, N=5, F=32, H=480, W=640, padding=11, bbox=[-1, -1, -1, 1, 1, 1],
                    grid=[128, 128, 128]),
    "config4": dict(M=768, D=128, N=9, F=32, H=480, W=640, padding=11, bbox=[-1, -1, -1, 1, 1, 1],
                    grid=[256, 256, 256]),
}


def kernel_literals(pyfile):
    """The string literals a reference module appends to the .cu text: Template(cu_source_code +
    \"\"\"...\"\"\").  Identical literals (raynet_fp.py builds the same template twice) once."""
    with open(os.path.join(CUDIR, pyfile)) as fh:
        tree = ast.parse(fh.read())
    out = []
    for node in ast.walk(tree):
        if (isinstance(node, ast.Call) and getattr(node.func, "id", "") == "Template" and node.args
                and isinstance(node.args[0], ast.BinOp) and isinstance(node.args[0].op, ast.Add)
                and isinstance(node.args[0].right, ast.Constant)
                and isinstance(node.args[0].right.value, str)):
            if node.args[0].right.value not in out:
                out.append(node.args[0].right.value)
    return out


def template_text():
    src = ""
    for f in CU_FILES:                                   # utils.py:26-37 parse_cu_files_to_string
        with open(os.path.join(CUDIR, f)) as fh:
            src += fh.read()
    for py in LITERAL_SOURCES:
        for lit in kernel_literals(py):
            src += lit
    return src


def substituted(shape):
    bbox = np.asarray(shape["bbox"], np.float32)         # the reference passes float32 arrays;
    grid = np.asarray(shape["grid"], np.int32)           # Template.substitute str()s each value
    return Template(template_text()).substitute(         # raynet_fp.py:230-248
        max_voxels=shape["M"], depth_planes=shape["D"], n_views=shape["N"], padding=shape["padding"],
        features_dimensions=shape["F"], width=shape["W"], height=shape["H"],
        grid_x=grid[0], grid_y=grid[1], grid_z=grid[2],
        bbox_min_x=bbox[0], bbox_min_y=bbox[1], bbox_min_z=bbox[2],
        bbox_max_x=bbox[3], bbox_max_y=bbox[4], bbox_max_z=bbox[5],
        sampling_scheme="sample_in_bbox")


def kernel_symbols(co):
    """{plain kernel name: mangled symbol} from the code object's kernel descriptors."""
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    txt = subprocess.check_output([readelf, "-s", "-W", co], text=True)
    out = {}
    for line in txt.splitlines():
        sym = line.split()[-1] if line.split() else ""
        if sym.endswith(".kd") and sym.startswith("_Z"):
            mangled = sym[:-3]
            digits = ""
            i = 2
            while mangled[i].isdigit():
                digits += mangled[i]
                i += 1
            out[mangled[i:i + int(digits)]] = mangled
    return out


def main():
    if not os.path.isdir(CUDIR):
        print("build_ref_cu: %s not present (not the build container) - keeping prebuilt files"
              % CUDIR, file=sys.stderr)
        return 0
    os.makedirs(OUT, exist_ok=True)
    manifest_path = os.path.join(OUT, "raynet_ref_cu.json")
    newest_src = max(os.path.getmtime(os.path.join(CUDIR, f)) for f in CU_FILES + LITERAL_SOURCES)
    newest_src = max(newest_src, os.path.getmtime(os.path.abspath(__file__)))
    if "--force" not in sys.argv and os.path.exists(manifest_path) and \
            os.path.getmtime(manifest_path) >= newest_src:
        return 0
    scratch = tempfile.mkdtemp(prefix="raynet_ref_cu_")
    manifest = {"flags": {}, "shapes": {}, "kernels": {}}
    try:
        for name, shape in SHAPES.items():
            cu = os.path.join(scratch, "ref_%s.cu" % name)
            with open(cu, "w") as fh:
                fh.write(substituted(shape))
            for variant, extra in (("fma", []), ("nofma", ["-ffp-contract=off"])):
                co = os.path.join(OUT, "raynet_ref_%s_%s.co" % (name, variant))
                cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-x", "hip", "-include", "hip/hip_runtime.h",
                       "--offload-arch=gfx950", "--cuda-device-only", "--no-gpu-bundle-output",
                       "-O3", "-w"] + extra + [cu, "-o", co]
                subprocess.check_call(cmd)
                manifest["flags"][variant] = " ".join(cmd[1:-3])
                manifest["kernels"] = kernel_symbols(co)
            manifest["shapes"][name] = shape
        with open(manifest_path, "w") as fh:
            json.dump(manifest, fh, indent=1, sort_keys=True)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    print("built %d code objects in %s: kernels %s" % (2 * len(SHAPES), OUT,
                                                       ", ".join(sorted(manifest["kernels"]))))
    return 0


if __name__ == "__main__":
    sys.exit(main())

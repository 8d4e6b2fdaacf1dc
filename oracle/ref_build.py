"""oracle/ref_build.py -- TEST INFRASTRUCTURE: builds ``oracle/_ref/shacira_ref_ops.so``, the reference's OWN hash-grid
operators (reference wisp/csrc/ops/hashgrid_interpolate_cuda.cu, hashgrid_interpolate2d_cuda.cu,
hashgrid_interpolate.cpp, hashgrid_interpolate.h), compiled for gfx950 from the sources where they lie under
/root/reference. Nothing under ``shacira_amd/`` imports this or its output; tests use it as a second checker beside
``oracle/hashgrid_oracle.c`` and ``tests/golden/make_ref_kernel_vectors.py`` uses it to produce the committed vectors
that pin the C restatement.

Recipe (what a maintainer porting the reference's extension to ROCm would do, minus its setup.py):
  1. the two ``.cu`` files are run through torch's bundled hipify (``torch.utils.hipify`` -- the translation torch's own
     extension builder applies to every CUDA extension on a ROCm install: cuda* -> hip* API names, ``<<<>>>`` ->
     ``hipLaunchKernelGGL``, ``at::cuda`` -> the masquerading ``at::hip`` names). The translated text exists only in a
     scratch directory under ``oracle/_ref/`` for the duration of the build and is deleted afterwards; no reference source
     text stays in the tree, nothing is written to /root/reference (hipify's in-place mode is NOT used).
  2. ``hipcc -O3`` (the reference's setup.py:72,87 passes ``-O3`` and nothing else to both compilers) on the two
     translated files, on ``hashgrid_interpolate.cpp`` read directly from /root/reference with ``-DWITH_CUDA`` (its
     setup.py:85 define), and on ``oracle/ref_binding.cpp`` (pybind11 exports under the reference's operator names).
  3. ``-include oracle/ref_compat.h``: ONE overload that torch dropped after the reference's torch 1.12
     (``detail::scalar_type(DeprecatedTypeProperties)``, needed by ``AT_DISPATCH_*(feats.type(), ...)``); see that file.
No reference build system is run, no header / library / tool / generated file is stood in for: ATen, c10, pybind11 and
hipify are the image's own.

The GPU box has no /root/reference: there this script does nothing and the prebuilt ``.so`` (git-ignored, not
gpurun-ignored) is used as it travelled.
"""
import glob
import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_OPS = "/root/reference/wisp/csrc/ops"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "shacira_ref_ops.so")
CU = ["hashgrid_interpolate_cuda.cu", "hashgrid_interpolate2d_cuda.cu"]
ARCH = os.environ.get("SHACIRA_ARCH", "gfx950")


def _stamp_inputs():
    files = [os.path.join(REF_OPS, f) for f in CU + ["hashgrid_interpolate.cpp", "hashgrid_interpolate.h"]]
    files += [os.path.join(HERE, f) for f in ("ref_binding.cpp", "ref_compat.h", "ref_build.py")]
    return files


def up_to_date() -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    return all(os.path.getmtime(f) <= t for f in _stamp_inputs())


def build(force: bool = False) -> str:
    """Returns 'built', 'up-to-date', or 'skipped: <why>'."""
    if not os.path.isdir(REF_OPS):
        return "skipped: /root/reference absent (prebuilt oracle/_ref is used if it travelled)"
    if not force and up_to_date():
        return "up-to-date"
    import torch  # noqa: F401  (the build needs torch's headers and hipify)
    from torch.utils import cpp_extension
    from torch.utils.hipify import hipify_python

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = os.path.join(OUT_DIR, ".build_tmp")
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    try:
        for f in CU:  # scratch copies for the translator only (deleted below)
            shutil.copyfile(os.path.join(REF_OPS, f), os.path.join(tmp, f))
        res = hipify_python.hipify(project_directory=tmp, output_directory=tmp, includes=[os.path.join(tmp, "*")],
                                   extra_files=[os.path.join(tmp, f) for f in CU], show_detailed=False,
                                   is_pytorch_extension=True, hipify_extra_files_only=True)
        hip_src = [res[os.path.join(tmp, f)].hipified_path for f in CU]
        inc = [f"-I{p}" for p in cpp_extension.include_paths(device_type="cuda")]
        inc += [f"-I{sysconfig.get_paths()['include']}", f"-I{REF_OPS}"]
        defs = ["-DWITH_CUDA", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
                f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
        common = [hipcc, "-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-w", "-include",
                  os.path.join(HERE, "ref_compat.h")] + defs + inc
        objs = []
        units = [(s, "-x", "hip") for s in hip_src]
        units += [(os.path.join(REF_OPS, "hashgrid_interpolate.cpp"), "-x", "c++"),
                  (os.path.join(HERE, "ref_binding.cpp"), "-x", "c++")]
        for i, (src, x, lang) in enumerate(units):
            obj = os.path.join(tmp, f"u{i}.o")
            subprocess.check_call(common + [x, lang, "-c", src, "-o", obj])
            objs.append(obj)
        libs = [f"-L{p}" for p in cpp_extension.library_paths(device_type="cuda")]
        rpath = [f"-Wl,-rpath,{p}" for p in cpp_extension.library_paths(device_type="cuda")]
        link = [hipcc, "-shared", f"--offload-arch={ARCH}", "-o", OUT] + objs + libs + rpath + \
               ["-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python", "-lamdhip64"]
        subprocess.check_call(link)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert not glob.glob(os.path.join(OUT_DIR, "*.cu")) and not glob.glob(os.path.join(OUT_DIR, "*.hip"))
    return "built"


def load():
    """Imports the built module (needs torch imported first: the .so links libtorch). Raises if it is absent."""
    import importlib.util

    import torch  # noqa: F401
    if not os.path.exists(OUT):
        raise FileNotFoundError(f"{OUT} not built (run python oracle/ref_build.py where /root/reference exists)")
    spec = importlib.util.spec_from_file_location("shacira_ref_ops", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

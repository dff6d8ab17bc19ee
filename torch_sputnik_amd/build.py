"""In-tree build of the two native pieces (gfx950 only):

  lib/libsputnik_hip.so        HIP kernels behind the C ABI (include/sputnik_hip.h)
  lib/libtorch_sputnik_ops.so  TORCH_LIBRARY registration on top of it (g++, no device code)

Run as a script, ``python torch_sputnik_amd/build.py`` (NOT ``-m``: importing
the package loads the libraries this script is about to build);
``__graft_entry__.build()`` loads this file by path and calls ``build_all()``.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")
KERNEL_LIB = os.path.join(LIB, "libsputnik_hip.so")
OPS_LIB = os.path.join(LIB, "libtorch_sputnik_ops.so")


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_kernels(jobs=6):
    _run(["make", "-C", CSRC, f"-j{jobs}", "all"])
    return KERNEL_LIB


def build_torch_ops():
    import torch
    from torch.utils import cpp_extension

    src = os.path.join(CSRC, "torch_binding.cpp")
    header = os.path.join(HERE, "..", "include", "sputnik_hip.h")
    if _newer(OPS_LIB, [src, header, KERNEL_LIB]):
        return OPS_LIB
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    for inc in cpp_extension.include_paths("cuda"):
        cmd += ["-isystem", inc]
    cmd += [src, "-o", OPS_LIB,
            f"-L{LIB}", "-lsputnik_hip",
            f"-L{torch_lib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch",
            "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{torch_lib}"]
    _run(cmd)
    return OPS_LIB


def lint_kernels():
    """Post-build ISA lint of the hand-counted kernels (tools/isa_lint.py): the
    build fails if the compiler broke an assumption the counted waits rely on."""
    import importlib.util
    path = os.path.join(HERE, "..", "tools", "isa_lint.py")
    spec = importlib.util.spec_from_file_location("_isa_lint", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.lint(KERNEL_LIB, verbose=False) != 0:
        raise SystemExit("isa_lint: libsputnik_hip.so violates the hand-counted vmcnt scheme")
    print("isa_lint: ok", flush=True)


def build_all():
    os.makedirs(LIB, exist_ok=True)
    stamp = os.path.getmtime(KERNEL_LIB) if os.path.exists(KERNEL_LIB) else None
    build_kernels()
    if stamp is None or os.path.getmtime(KERNEL_LIB) != stamp:
        lint_kernels()   # (only when the library was relinked)
    build_torch_ops()


if __name__ == "__main__":
    build_all()
    print("built:", KERNEL_LIB, OPS_LIB)
    sys.exit(0)

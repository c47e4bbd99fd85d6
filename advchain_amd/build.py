"""Builds ``advchain_amd/csrc/libadvchain_hip.so`` (gfx950 only) with hipcc, in-tree.

    python -m advchain_amd.build [--force] [--verbose]

The library is a plain C-ABI shared object (``include/advchain_hip.h``).  It is linked against the
HIP runtime that PyTorch-ROCm itself loads (``torch/lib/libamdhip64.so``, SONAME libamdhip64.so.7)
so that stream handles obtained from ``torch.cuda.current_stream()`` are valid inside it.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libadvchain_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def torch_lib_dir():
    import torch
    return os.path.join(os.path.dirname(torch.__file__), "lib")


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = torch_lib_dir()
    cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "-I" + os.path.join(HERE, "..", "include"), "-I" + CSRC]
    cmd += sources()
    cmd += ["-o", LIB_PATH, "-no-hip-rt", "-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed building %s" % LIB_PATH)
    if verbose:
        print(res.stdout)
    return LIB_PATH


if __name__ == "__main__":
    p = build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print("built", p, os.path.getsize(p), "bytes")

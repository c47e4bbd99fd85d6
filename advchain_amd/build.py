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


# Per-source compiler flags (measured, DESIGN.md lesson 42).  clang's SLP vectoriser packs pairs of independent fp32 operations
# into v_pk_fma / v_pk_mul / v_pk_add: no faster per flop on gfx950 (tools/microbench/valubench: 4.85 clk against 2 x 2.6),
# but the operands must sit in adjacent register pairs -- v_mov / v_pk_mov shuffles and, in the marching adjoint, 228 instead
# of 152 VGPRs.  Off where the A/B says so (3D marching adjoint -5..-16 %, 2D gather form -3..-8 %, Gaussian passes -7 %);
# the forward samplers and the scatters keep it (2D squaring forward +49 % without).
PER_FILE_FLAGS = {
    "adjoint_march.hip": ["-fno-slp-vectorize"],
    "adjoint_gather.hip": ["-fno-slp-vectorize"],
    "adjoint_fused2d.hip": ["-fno-slp-vectorize"],      # (the same code generation as the per-squaring gather form: bit-identical results)
    "fields.hip": ["-fno-slp-vectorize"],
}
if os.environ.get("ADVCHAIN_BUILD_SLP_EVERYWHERE"):     # A/B knob: the compiler default for every source
    PER_FILE_FLAGS = {}


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


def _compile_one(args):
    hipcc, src, obj, flags, verbose = args
    cmd = [hipcc, "-c"] + flags + [src, "-o", obj]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return src, res.returncode, res.stdout


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 (one object per source, in parallel, rebuilt only when the source or a header
    is newer) and link them into one shared library.  Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tl = torch_lib_dir()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
             "-I" + os.path.join(HERE, "..", "include"), "-I" + CSRC]
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    newest_header = max([os.path.getmtime(h) for h in headers] + [os.path.getmtime(os.path.abspath(__file__))])
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            jobs.append((hipcc, src, obj, flags + PER_FILE_FLAGS.get(os.path.basename(src), []), verbose))
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as pool:
        for src, rc, log in pool.map(_compile_one, jobs):
            if rc != 0:
                sys.stderr.write(log)
                raise RuntimeError("hipcc failed compiling %s" % src)
            if verbose:
                print(log)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + \
          ["-o", LIB_PATH, "-no-hip-rt", "-L" + tl, "-lamdhip64", "-Wl,-rpath," + tl]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("hipcc failed linking %s" % LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    p = build_library(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print("built", p, os.path.getsize(p), "bytes")

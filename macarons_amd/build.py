"""Build libmacarons_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m macarons_amd.build [--force] [--verbose]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import glob
import hashlib
import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmacarons_hip.so")
STAMP = LIB_PATH + ".stamp"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-ffp-contract=fast",
         "-Wall", "-Wno-unused-function"] + os.environ.get("MCR_HIPCC_FLAGS", "").split()


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for p in sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(glob.glob(os.path.join(CSRC, "*.inc"))):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def hipcc_path():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def per_file_flags(src):
    """A source may carry a line `// MCR_HIPCC_FLAGS: <flags>`; those flags are appended for that file only
    (e.g. knn.hip needs -ffp-contract=off: bit-exact parity forbids FMA contraction of the distance)."""
    extra = []
    with open(src) as f:
        for line in f:
            if line.startswith("// MCR_HIPCC_FLAGS:"):
                extra += line.split(":", 1)[1].split()
    return extra


def _clean_unbundled(lib_path, stamp):
    """hipcc leaves per-object unbundled images (`<lib>.N.hipv4-...`, `<lib>.N.host-...`) next to its output; they are never
    loaded, but they would travel to the GPU box with every snapshot."""
    for tmp in glob.glob(lib_path + ".*"):
        if tmp != stamp:
            os.remove(tmp)


def build(force=False, verbose=False):
    dig = _digest()
    _clean_unbundled(LIB_PATH, STAMP)               # also when nothing is rebuilt (left by an interrupted or manual link)
    if not force and os.path.exists(LIB_PATH) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB_PATH
    objs = []
    obj_dir = os.path.join(PKG_DIR, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    procs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        cmd = [hipcc_path()] + [f for f in FLAGS if f != "-shared"] + per_file_flags(src) + ["-c", src, "-I", CSRC, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    cmd = [hipcc_path(), "-shared", "-fPIC", f"--offload-arch={ARCH}"] + objs + ["-o", LIB_PATH]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode()}")
    _clean_unbundled(LIB_PATH, STAMP)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB_PATH


TORCH_LIB_PATH = os.path.join(PKG_DIR, "libmacarons_torch.so")
TORCH_SRC = os.path.join(PKG_DIR, "csrc_torch", "macarons_torch.cpp")


def build_torch_ops(force=False, verbose=False):
    """libmacarons_torch.so: the C++ TORCH_LIBRARY extension (torch.ops.macarons.*) over the C ABI; host code only, compiled with
    hipcc against the installed torch's headers and linked to libmacarons_hip.so next to it."""
    import torch
    tdir = os.path.dirname(torch.__file__)
    h = hashlib.sha256()
    for p in (TORCH_SRC, os.path.join(os.path.dirname(PKG_DIR), "include", "macarons_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(torch.__version__.encode())
    stamp = TORCH_LIB_PATH + ".stamp"
    _clean_unbundled(TORCH_LIB_PATH, stamp)
    if not force and os.path.exists(TORCH_LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == h.hexdigest():
        return TORCH_LIB_PATH
    cmd = [hipcc_path(), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-w",
           "-I", os.path.join(tdir, "include"), "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
           "-I", "/opt/rocm/include", "-I", os.path.join(os.path.dirname(PKG_DIR), "include"), TORCH_SRC, "-o", TORCH_LIB_PATH,
           "-L", os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-L", PKG_DIR, "-lmacarons_hip",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(tdir, "lib")]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {TORCH_SRC}:\n{r.stdout.decode()}")
    _clean_unbundled(TORCH_LIB_PATH, stamp)
    with open(stamp, "w") as f:
        f.write(h.hexdigest())
    return TORCH_LIB_PATH


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv)
    print(p)
    print(build_torch_ops(force="--force" in sys.argv, verbose="--verbose" in sys.argv))

"""Dev helper: build an experimental copy of libmacarons_hip.so with extra -D flags on chosen sources.
    python tools/build_variant.py NAME "local_pct3.hip,local_pct4.hip" -DL3_PF_OVERRIDE=4 ...
    (a source given as +path/to/file.hip is an EXTRA translation unit, e.g. +tools/experiments/local_pct8.hip)
-> tools/_libs/libmacarons_hip_NAME.so  (select with MCR_DEV_LIB=NAME in tools/time_local_pct_ab.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from macarons_amd import build as B
B.build()
name, srcs, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
extra = [os.path.join(ROOT, x[1:]) for x in srcs if x.startswith("+")]
srcs = [x for x in srcs if not x.startswith("+")]
out = os.path.join(ROOT, "tools", "_libs"); os.makedirs(out, exist_ok=True)
objs = []
for src in B.sources() + extra:
    base = os.path.basename(src)
    obj = os.path.join(B.PKG_DIR, "_obj", base + ".o")
    if base in srcs or src in extra:
        obj = os.path.join(out, f"{name}_{base}.o")
        cmd = [B.hipcc_path()] + [f for f in B.FLAGS if f != "-shared"] + B.per_file_flags(src) + flags + ["-c", src, "-I", B.CSRC, "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode: sys.exit(r.stdout.decode())
    objs.append(obj)
lib = os.path.join(out, f"libmacarons_hip_{name}.so")
r = subprocess.run([B.hipcc_path(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}"] + objs + ["-o", lib], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
if r.returncode: sys.exit(r.stdout.decode())
print(lib)

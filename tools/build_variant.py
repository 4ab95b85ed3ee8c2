#!/usr/bin/env python
"""Build an experimental variant of the product library: tools/build_variant.py NAME [-DMACRO ...] -> cerberus_amd/lib/libvilo_gpu_NAME.so
(objects under cerberus_amd/lib/obj_NAME). The variants travel to the GPU box with the snapshot and are A/B-timed there with
tools/ab.sh (VILO_GPU_LIB selects the library api.py loads). Never what the product loads by default."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, extra = sys.argv[1], sys.argv[2:]
csrc = os.path.join(ROOT, "cerberus_amd", "csrc")
lib = os.path.join(ROOT, "cerberus_amd", "lib")
objdir = os.path.join(lib, "obj_" + name)
os.makedirs(objdir, exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-I", os.path.join(ROOT, "include"), "-Wno-unused-result"] + extra
srcs = sorted(glob.glob(os.path.join(csrc, "*.hip")))
hdrs = glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
stamp = os.path.join(objdir, "flags.txt")
same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
jobs, objs = [], []
for s in srcs:
    o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
    objs.append(o)
    if not same_flags or not os.path.exists(o) or any(os.path.getmtime(x) > os.path.getmtime(o) for x in [s] + hdrs):
        jobs.append(subprocess.Popen([hipcc] + flags + ["-c", s, "-o", o]))
if any(j.wait() != 0 for j in jobs):
    sys.exit("hipcc failed")
open(stamp, "w").write(" ".join(flags))
out = os.path.join(lib, "libvilo_gpu_%s.so" % name)
subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + objs)
print(out)

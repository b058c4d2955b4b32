"""Builds libdartk.so (the HIP/gfx950 C-ABI library) in-tree with hipcc. No torch involved."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdartk.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the association path must round exactly like the CPU restatement (no silent FMAs);
# the accumulation loops ask for FMAs explicitly (__builtin_fmaf).
FLAGS = os.environ.get("DARTK_EXTRA_FLAGS", "").split() + ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(HERE, "..", "include")]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


FLAGS_FILE = os.path.join(HERE, ".libdartk.flags")      # the flags the existing library was built with


def needs_build():
    if not os.path.exists(LIB):
        return True
    try:
        if open(FLAGS_FILE).read() != " ".join(FLAGS):
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "dartk.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC, *FLAGS, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s\n" % src)
    if failed:
        raise RuntimeError("libdartk build failed")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(FLAGS_FILE, "w") as f:
        f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

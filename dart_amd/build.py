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


COUNT_LIB = os.path.join(HERE, "libdartk_count.so")      # diagnostic build: counts the trilinear lookups performed (bench.py roofline)
VARIANTS = [(LIB, [], ".o"), (COUNT_LIB, ["-DDARTK_COUNT_LOOKUPS"], ".count.o")]


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(COUNT_LIB):
        return True
    try:
        if open(FLAGS_FILE).read() != " ".join(FLAGS):
            return True
    except OSError:
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(COUNT_LIB))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [os.path.join(HERE, "..", "include", "dartk.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    jobs = []
    for lib, extra, suffix in VARIANTS:
        for src in sources():
            jobs.append((lib, src, src[:-4] + suffix, [HIPCC, *FLAGS, *extra, "-c", src, "-o", src[:-4] + suffix]))
    failed = False
    width = max(1, min(len(jobs), int(os.environ.get("MAX_JOBS", "0")) or (os.cpu_count() or 8)))
    running = []
    pending = list(jobs)
    while pending or running:
        while pending and len(running) < width:
            j = pending.pop(0)
            running.append((j, subprocess.Popen(j[3], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        (j, p) = running.pop(0)
        out, _ = p.communicate()
        if out and verbose:
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s\n" % j[1])
    if failed:
        raise RuntimeError("libdartk build failed")
    for lib, extra, suffix in VARIANTS:
        objs = [j[2] for j in jobs if j[0] == lib]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    with open(FLAGS_FILE, "w") as f:
        f.write(" ".join(FLAGS))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

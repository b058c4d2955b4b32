"""Builds a diagnostic variant of the library next to the product one: python tools/build_variant.py stamps -DDARTK_STAMPS
-> dart_amd/libdartk_stamps.so (objects *.stamps.o). Load it with dart_amd._abi.LIB_PATH = <that file> before the first call."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dart_amd import build as B

name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(B.HERE, "libdartk_%s.so" % name)
procs = []
for src in B.sources():
    obj = src[:-4] + ".%s.o" % name
    procs.append((obj, subprocess.Popen([B.HIPCC, *B.FLAGS, *extra, "-c", src, "-o", obj])))
objs = []
for obj, p in procs:
    if p.wait() != 0:
        raise SystemExit("hipcc failed for %s" % obj)
    objs.append(obj)
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print(out)

"""Build tuning variants of libprt_hip.so into prt_amd/lib/var/ (diagnostic).  usage: build_variants.py name:-DA=1,-DB=2 ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prt_amd import _build as B
out = os.path.join(B.LIB_DIR, "var")
os.makedirs(out, exist_ok=True)
for f in os.listdir(out):
    os.remove(os.path.join(out, f))
procs = []
for spec in sys.argv[1:]:
    name, _, defs = spec.partition(":")
    cmd = [B.hipcc()] + B.FLAGS + [f'-DPRT_SOURCE_SHA16="{B.source_sha16()}+{name}"'] + [d for d in defs.split(",") if d] + [os.path.join(B.CSRC, s) for s in B.SOURCES] + ["-o", os.path.join(out, f"libprt_hip_{name}.so")]
    procs.append((name, subprocess.Popen(cmd)))
    if len(procs) % 4 == 0:
        for n, p in procs[-4:]:
            p.wait()
for n, p in procs:
    if p.wait() != 0:
        print("FAILED", n)
print("built", [n for n, _ in procs])

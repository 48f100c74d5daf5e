#!/usr/bin/env python3
"""Developer tool: tools/kbench.py once per library variant under mpcholonavigation_amd/variants/
(and the product library), each in its own process."""
import glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [("product", os.path.join(root, "mpcholonavigation_amd", "libsmpc.so"))]
libs += [(os.path.basename(p)[8:-3], p) for p in sorted(glob.glob(os.path.join(root, "mpcholonavigation_amd", "variants", "libsmpc_*.so")))]
for name, path in libs:
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "kbench.py"), name], env=dict(os.environ, SMPC_LIB=path),
                       capture_output=True, text=True)
    lines = [l for l in (r.stdout + r.stderr).splitlines() if "kbench" in l or "Error" in l or "error" in l]
    print("\n".join(lines) if lines else f"[{name}] no output, rc={r.returncode}\n{(r.stdout + r.stderr)[-1500:]}", flush=True)

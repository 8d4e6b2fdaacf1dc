#!/usr/bin/env python3
"""times the backward's kernels on S1 / D through rocprof-less event timing of the whole operator (variants via SHACIRA_HIP_LIB)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess
for v in sys.argv[1:]:
    env = dict(os.environ)
    if v != "default":
        env["SHACIRA_HIP_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "shacira_amd/lib/variants", v + ".so")
    print("==", v, flush=True)
    subprocess.run(["bash", "tools/prof.sh", "fp_" + v, "tools/r3_ab.py", "S1", "bwd_fork=0"], env=env)

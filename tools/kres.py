#!/usr/bin/env python3
"""Kernel resources of a built library: tools/kres.py <lib.so> [name filter] -> vgprs / sgprs / scratch / LDS per kernel."""
import glob, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp(prefix="kres_")
try:
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for co in glob.glob(os.path.join(tmp, "lib.so.*gfx950*")):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        # one metadata block per kernel: the YAML list items of amdhsa.kernels start at `- .agpr_count:` (keys are alphabetical, so
        # the kernel's own `.name: _Z...` sits in the middle of its block, behind the `.name` entries of its arguments)
        for block in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
            mname = re.search(r"\n\s+\.name:\s+(_Z\S+)", block)
            if not mname or flt not in mname.group(1):
                continue
            g = lambda k: (re.search(rf"\n\s+\.{k}:\s+(\d+)", block) or [None, "-"])[1]
            agpr = re.match(r"\s*(\d+)", block)
            dem = subprocess.run(["c++filt", mname.group(1)], capture_output=True, text=True).stdout.strip()
            print(f"vgpr {g('vgpr_count'):>4} agpr {agpr.group(1) if agpr else '-':>3} sgpr {g('sgpr_count'):>4} scratch "
                  f"{g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6} spill {g('vgpr_spill_count'):>3}  {dem[:150]}")
finally:
    shutil.rmtree(tmp, ignore_errors=True)

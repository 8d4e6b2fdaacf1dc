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
        for m in re.finditer(r"\.name:\s+(\S+)(.*?)(?=\.name:\s+_Z|\Z)", notes, re.S):
            if flt not in m.group(1):
                continue
            b = m.group(2)
            g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", b) or [None, "-"])[1]
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            print(f"vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>3} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6} spill {g('vgpr_spill_count'):>3}  {dem[:150]}")
finally:
    shutil.rmtree(tmp, ignore_errors=True)

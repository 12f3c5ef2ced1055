"""Registers / spills / scratch of the kernels in one object file of the library (auto_round_amd/lib/obj/<name>.o): the gfx950 code
object is cut out of the host object's .hip_fatbin section and its metadata notes printed.   python tools/kernel_resources.py ar_attn_exact"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
o = os.path.join(ROOT, "auto_round_amd", "lib", "obj", name + ".o")
subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "--dump-section", ".hip_fatbin=/tmp/_kr.fatbin", o], check=True)
b = open("/tmp/_kr.fatbin", "rb").read()
open("/tmp/_kr.co", "wb").write(b[b.find(b"\x7fELF"):])
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/_kr.co"], capture_output=True, text=True).stdout
cur = {}
for line in out.splitlines():
    line = line.strip().lstrip("- ")
    for k in (".name:", ".vgpr_count:", ".agpr_count:", ".sgpr_count:", ".vgpr_spill_count:", ".private_segment_fixed_size:", ".group_segment_fixed_size:"):
        if line.startswith(k):
            cur[k] = line.split()[-1]
    if line.startswith(".vgpr_spill_count:"):
        dem = subprocess.run(["c++filt", cur.get(".name:", "?")], capture_output=True, text=True).stdout.strip()
        print(f"{dem[:70]:70s} vgpr {cur.get('.vgpr_count:')} agpr {cur.get('.agpr_count:')} spill {cur.get('.vgpr_spill_count:')} scratch {cur.get('.private_segment_fixed_size:')}")

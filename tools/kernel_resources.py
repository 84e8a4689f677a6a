"""Register / spill / LDS figures of the gfx950 kernels in one object of the library build:
   python tools/kernel_resources.py kissabc.jl_amd/csrc/build/ais_inst_1.o [name-substring ...]
(unbundles the device code object and reads its AMDGPU metadata notes)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1])
pats = sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    cp = os.path.join(d, "o.o")
    os.symlink(obj, cp)
    subprocess.check_call([f"{LLVM}/llvm-objdump", "--offloading", cp], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL, cwd=d)
    out = [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f][0]
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
for blk in txt.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name:
        continue
    nm = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    if pats and not all(p in nm for p in pats):
        continue
    g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]   # noqa: E731
    agpr = blk.split("\n")[0].strip()
    print(f"{nm[:110]:110s} vgpr {g('vgpr_count'):>3} agpr {agpr:>3} sgpr {g('sgpr_count'):>3} "
          f"vspill {g('vgpr_spill_count'):>3} sspill {g('sgpr_spill_count'):>3} lds {g('group_segment_fixed_size'):>6} "
          f"scratch {g('private_segment_fixed_size'):>5}")

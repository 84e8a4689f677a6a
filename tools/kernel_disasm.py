"""Static instruction statistics of gfx950 kernels, from an object of the library build or from a
code object of the run-time compilation cache (kissabc.jl_amd/lib/rtc_cache/kabc_*.co):

   python tools/kernel_disasm.py <file.o | file.co | file.so> [name-substring ...] [--dump DIR]

Per kernel: registers / spills / LDS (metadata notes) and the static count of VALU instructions by
class (f64 add / mul / fma, v_mad_u64_u32, other VALU), SALU, LDS, VMEM, s_waitcnt, s_nop.  Static
counts are not executed counts (loops), but two builds of one kernel compare well on them.
--dump DIR writes each selected kernel's disassembly to DIR/<n>.s."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(path, d):
    raw = open(path, "rb").read()
    if raw[:8] == b"KABCRTC1":  # the library's cache file: names, then the code object
        import struct
        off = 8
        (n,) = struct.unpack_from("<I", raw, off)
        off += 4
        for _ in range(n):
            (ln,) = struct.unpack_from("<I", raw, off)
            off += 4 + ln
        (cs,) = struct.unpack_from("<Q", raw, off)
        off += 8
        out = os.path.join(d, "rtc.co")
        open(out, "wb").write(raw[off:off + cs])
        return out
    cp = os.path.join(d, "o.o")
    os.symlink(os.path.abspath(path), cp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, cwd=d)
    outs = [os.path.join(d, f) for f in os.listdir(d) if "amdgcn" in f]
    return outs[0] if outs else cp


def classify(op):
    if op.startswith("v_"):
        if op in ("v_add_f64", "v_add_f64_e32", "v_add_f64_e64"):
            return "f64add"
        if op.startswith("v_mul_f64"):
            return "f64mul"
        if op.startswith(("v_fma_f64", "v_fmac_f64")):
            return "f64fma"
        if op.startswith("v_mad_u64_u32"):
            return "mad64"
        if op.startswith(("v_cmp", "v_cmpx")) and "f64" in op:
            return "f64cmp"
        if "f64" in op:
            return "f64other"
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    args = sys.argv[1:]
    dump = None
    if "--dump" in args:
        i = args.index("--dump")
        dump = args[i + 1]
        del args[i:i + 2]
        os.makedirs(dump, exist_ok=True)
    path, pats = args[0], args[1:]
    with tempfile.TemporaryDirectory() as d:
        co = code_object(path, d)
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True,
                             text=True).stdout
    meta = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: (re.search(rf"\.{k}:\s+(\d+)", blk) or [None, "?"])[1]   # noqa: E731
        meta[name.group(1)] = (g("vgpr_count"), blk.split("\n")[0].strip(), g("sgpr_count"), g("vgpr_spill_count"),
                               g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"))
    n = 0
    for sec in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis):
        m = re.match(r"[0-9a-f]+ <([^>]+)>:", sec)
        if not m or m.group(1) not in meta:
            continue
        sym = m.group(1)
        nm = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        if pats and not all(p in nm for p in pats):
            continue
        cnt = {}
        for line in sec.split("\n")[1:]:
            t = line.strip().split()
            if not t or t[0].endswith(":"):
                continue
            c = classify(t[0])
            cnt[c] = cnt.get(c, 0) + 1
        v, a, s, vs, ss, lds, scr = meta[sym]
        valu = sum(cnt.get(k, 0) for k in ("f64add", "f64mul", "f64fma", "mad64", "f64cmp", "f64other", "valu"))
        print(f"{nm[:100]}\n    vgpr {v} agpr {a} sgpr {s} vspill {vs} sspill {ss} lds {lds} scratch {scr}\n"
              f"    VALU {valu} (f64 add {cnt.get('f64add', 0)} mul {cnt.get('f64mul', 0)} fma {cnt.get('f64fma', 0)} "
              f"cmp {cnt.get('f64cmp', 0)} other-f64 {cnt.get('f64other', 0)} mad_u64 {cnt.get('mad64', 0)} "
              f"rest {cnt.get('valu', 0)})  SALU {cnt.get('salu', 0)}  LDS {cnt.get('lds', 0)}  VMEM {cnt.get('vmem', 0)}  "
              f"waitcnt {cnt.get('waitcnt', 0)}  nop {cnt.get('nop', 0)}")
        if dump:
            open(os.path.join(dump, f"{n}.s"), "w").write(f"; {nm}\n{sec}\n")
        n += 1


if __name__ == "__main__":
    main()

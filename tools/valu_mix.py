"""Mix-weighted VALU issue time of one kernel of the shipped library:
   python tools/valu_mix.py kissabc.jl_amd/csrc/build/ais_insthi_2.o "ais_half_kernel<8, 2, 0, 1>" > profiles/r04_valu_mix.json
Every static VALU instruction of the kernel is priced with this chip's MEASURED issue time for its
opcode (profiles/r01m_valu_rate_8waves.json: ns per wave-instruction per SIMD with 8 waves sharing
the SIMD, tools/valu_rate.hip); opcodes that were not measured take their class's figure (f64 and
64-bit integer: the v_add_f64 time; everything else: v_add_f32's).  bench.py multiplies the PMC count
of executed VALU instructions by the average."""
import json
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_disasm import LLVM, code_object  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rates = json.load(open(os.path.join(ROOT, "profiles", "r01m_valu_rate_8waves.json")))["rates"]
ns = {k: v["ns_per_wave_instr_per_simd"] for k, v in rates.items() if k.startswith("v_")}
ns["v_cndmask_b32"] = ns["v_cndmask_b32_e64_vcc"]   # (the bare entry is a dependent-chain probe)
F64, F32 = ns["v_add_f64"], ns["v_add_f32"]


def price(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in ns:
        return ns[base], True
    if base.startswith("v_fmac_f64"):
        return ns["v_fma_f64"], True
    if base.startswith(("v_cmp", "v_cmpx")):
        return (ns["v_cmp_lt_f64"] if "64" in base else ns["v_cmp_lt_u32_vcc"]), False
    if "f64" in base or "u64" in base or "i64" in base or "b64" in base:
        return F64, False
    return F32, False


path, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as d:
    co = code_object(path, d)
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout
out = None
for sec in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:)", dis):
    m = re.match(r"[0-9a-f]+ <([^>]+)>:", sec)
    if not m:
        continue
    nm = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    if pat not in nm or "(.kd)" in nm or nm.endswith(".kd"):
        continue
    cnt, tot_ns, n, measured = {}, 0.0, 0, 0
    for line in sec.split("\n")[1:]:
        t = line.strip().split()
        if not t or not t[0].startswith("v_"):
            continue
        p, known = price(t[0])
        tot_ns += p
        n += 1
        measured += known
        key = re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])
        cnt[key] = cnt.get(key, 0) + 1
    if n == 0:
        continue
    top = dict(sorted(cnt.items(), key=lambda kv: -kv[1])[:14])
    out = {"kernel": nm, "static_valu_instructions": n, "priced_by_own_measurement": measured,
           "avg_issue_ns": tot_ns / n, "avg_issue_cycles": tot_ns / n * 2.4,
           "rates_source": "profiles/r01m_valu_rate_8waves.json (tools/valu_rate.hip, 8 waves per SIMD)",
           "top_opcodes": top,
           "note": "static mix of the whole kernel (producer and consumer loops + prologue); cycles at 2.4 GHz"}
print(json.dumps(out, indent=1))

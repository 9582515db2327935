#!/usr/bin/env python3
"""Build-time check of the generated gfx950 code around the hand-written node loads.

rt_kernels.hip issues the seven 16-byte loads of a traversal step in inline assembly (LDS reads
in one arm of a branch, global loads in the other, into the same registers) and waits for them in
separate assembly statements.  Between the two the destination registers hold data still in
flight, which the compiler does not know: a register copy at the control-flow merge, a spill, or
any other instruction it placed there that touches them would read stale values without any
diagnostic.  This script compiles the file to assembly with the library's own flags and fails if

  * the two arms load into different registers (a merge would then need copies), or
  * any compiler-generated instruction between a group of loads and the wait that completes it
    mentions a register of a load still in flight (a spill of one is such an instruction).
The one exception: code of the second arm that runs before that arm's own loads (its address
arithmetic).  It executes under the complementary execution mask, i.e. in other lanes than the
first arm's loads write, so using those registers as temporaries there is harmless; the checker
recognises the arm by the instruction that complements the mask and by the merge that ends it.

  python3 tools/check_isa.py [extra compiler flags ...]        (no GPU needed)
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
SRC = ROOT / "raytracer_amd" / "csrc" / "rt_kernels.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-fno-slp-vectorize"]
LOAD = re.compile(r"^\s*(global_load_dwordx4|ds_read_b128)\s+v\[(\d+):(\d+)\]")
WAIT = re.compile(r"^\s*s_waitcnt vmcnt\((\d+)\) lgkmcnt\((\d+)\)")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text):
    out = set()
    for a, b, c in VREG.findall(text):
        out.update(range(int(a), int(b) + 1) if a else [int(c)])
    return out


def check_function(name, lines):
    """Returns a list of complaints for one kernel's instruction listing."""
    bad, in_asm = [], False
    group, arm, pending = [], [], []   # destination ranges of the open group, of the arm being read, registers in flight
    other_lanes = False                # in the second arm, before its loads: other lanes than the loads in flight write
    for no, line in lines:
        if ";;#ASMSTART" in line:
            in_asm, arm = True, []
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            if arm:
                if group and arm != group:
                    bad.append(f"{name}:{no}: the two arms load into different registers: {group} / {arm}")
                group = arm
                pending = [set(range(a, b + 1)) for a, b in group]
                other_lanes = False
            continue
        code = line.split(";")[0]
        if not in_asm and pending and re.match(r"^\s*(s_andn2_saveexec_b64|s_or_saveexec_b64|s_xor_b64\s+exec)", code):
            other_lanes = True
        if not in_asm and re.match(r"^\s*s_or_b64\s+exec,\s*exec", code):
            other_lanes = False
        if in_asm:
            m = LOAD.match(code)
            if m:
                arm.append((int(m.group(2)), int(m.group(3))))
                continue
            m = WAIT.match(code)
            if m and pending:
                left = min(int(m.group(1)), int(m.group(2)))   # loads of either kind that may still be outstanding
                for i in range(len(pending) - left):
                    pending[i] = set()
                if left == 0:
                    group, pending = [], []
            continue
        if not pending or other_lanes or not code.strip() or code.strip().endswith(":") or code.lstrip().startswith("."):
            continue
        inflight = set().union(*pending)
        hit = vregs(code) & inflight
        if hit:
            bad.append(f"{name}:{no}: `{code.strip()}` touches v{sorted(hit)} while its load is in flight")
    if pending:
        bad.append(f"{name}: a group of loads is never waited for")
    return bad


def main():
    with tempfile.TemporaryDirectory() as td:
        asm = Path(td) / "rt_kernels.s"
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *sys.argv[1:], "-S", "--cuda-device-only", "-x", "hip", str(SRC), "-o", str(asm)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        text = asm.read_text().splitlines()
    bad, functions, cur, with_loads = [], {}, None, set()
    for no, line in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            functions[cur] = []
        elif line.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            functions[cur].append((no, line))
            if LOAD.match(line):
                with_loads.add(cur)
    for name in sorted(with_loads):
        bad += check_function(name, functions[name])
    if bad:
        print("\n".join(bad))
        sys.exit(f"check_isa: {len(bad)} problem(s) in {len(with_loads)} kernel(s) with hand-written loads")
    print(f"check_isa: {len(with_loads)} kernels with hand-written loads: both arms load into the same registers and no "
          "compiler-generated instruction touches a register whose load is in flight")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Static check of the built library for the gfx950 store-data hazard the compiler does not cover.

Hardware rule (measured, scripts/micro/store_x4_hazard.hip -> profiles/r03_store_x4_hazard.txt): after a buffer store of more
than 64 bits of data, a VALU write of the store's DATA registers needs two wait states; an SGPR `soffset` on the store buys one
of them.  hipcc's hazard recogniser (LLVM GCNHazardRecognizer::createsVALUHazard) inserts the two wait states when soffset is an
immediate and NOTHING when it is a register -- correct for parts that need one wait state, one short on this one.  The symptom
is lane-tied garbage in the last four lanes of every sixteen (csrc/pw_wgrad.h, round 2).

This script disassembles every gfx950 code object of libssdseg_hip.so and reports each
    buffer_store_dwordx3/x4 | buffer_store_format_xyz(w)   vdata, vaddr, srsrc, s<N> ...
whose NEXT issue slot (fewer than one wait state: an intervening instruction or `s_nop` counts) holds a VALU instruction that
writes one of the store's data registers.  Exit code 1 when anything is found.  Also run by tests/test_cpu_cabi_and_host.py.
usage: python scripts/check_store_hazard.py [path/to/libssdseg_hip.so]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

STORE = re.compile(r"^\s*(buffer_store_dwordx[34]|buffer_store_format_xyzw?|buffer_store_format_d16_xyzw?)\s+(v\[(\d+):(\d+)\]|a\[(\d+):(\d+)\]),\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VDST = re.compile(r"^\s*(v_\S+)\s+(v(\d+)|v\[(\d+):(\d+)\])\b")


def code_objects(lib: str, workdir: str):
    local = os.path.join(workdir, "lib.so")
    shutil.copy(lib, local)
    subprocess.run([OBJDUMP, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if "amdgcn" in f and os.path.getsize(os.path.join(workdir, f)) > 0)


def instructions(co: str):
    """-> [(kernel symbol, mnemonic line)] in program order"""
    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    sym = "?"
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            sym = m.group(1)
            continue
        text = line.split("//")[0].strip()
        if text and not text.endswith(":"):
            yield sym, text


def scan(lib: str):
    findings, stores = [], 0
    with tempfile.TemporaryDirectory() as wd:
        for co in code_objects(lib, wd):
            ins = list(instructions(co))
            for i, (sym, text) in enumerate(ins):
                m = STORE.match(text)
                if not m:
                    continue
                stores += 1
                soffset = m.group(8).rstrip(",")
                if not re.fullmatch(r"s\d+|vcc_lo|vcc_hi|m0|ttmp\d+", soffset):
                    continue            # immediate soffset: the compiler keeps the two wait states itself
                if m.group(3) is None:
                    continue            # data in accumulation registers: no VALU instruction writes those
                lo, hi = int(m.group(3)), int(m.group(4))
                if i + 1 >= len(ins) or ins[i + 1][0] != sym:
                    continue
                nxt = ins[i + 1][1]
                d = VDST.match(nxt)
                if not d or nxt.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
                    continue
                dlo = int(d.group(3)) if d.group(3) is not None else int(d.group(4))
                dhi = dlo if d.group(3) is not None else int(d.group(5))
                if dlo <= hi and dhi >= lo:
                    findings.append((sym, text, nxt))
    return stores, findings


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "multi-task-learning-object-detection-semantic-segmentation_amd", "ssdseglib", "libssdseg_hip.so")
    stores, findings = scan(lib)
    print(f"{lib}: {stores} buffer stores of more than 64 bits, {len(findings)} with an SGPR soffset and a VALU write of their data in the next slot")
    for sym, st, nxt in findings:
        print(f"  {sym}\n      {st}\n      {nxt}")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())

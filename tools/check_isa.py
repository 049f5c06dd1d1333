#!/usr/bin/env python3
"""Holds a built library's gfx950 code to a rule the compiler does not know.

On gfx950 the 64-bit shifts (v_lshlrev_b64, v_lshrrev_b64, v_ashrrev_i64) read their 32-bit shift amount as a
register pair.  When the amount sits in the last vector register of the wave's allocation, the pair's second half
lies outside it, and every wave but the first on its SIMD then gets the shift's result for an amount of 0
(tools/shift_hazard_probe.hip measures it: a quarter of the results wrong on waves 4..15 of a 1024-thread block,
none with the amount one register lower, none for 32-bit shifts, v_mad_u64_u32 or v_lshl_add_u64).  The register
allocator knows no such rule, so whether a build is hit depends on scheduling.  This script disassembles every
code object of the library and fails if any 64-bit shift takes its amount from a vector register that is the last
one of any kernel's allocation in that code object (allocations are rounded up to 8 registers).

usage: check_isa.py [libmirt.so ...]      (default: raytracing_amd/libmirt.so)
"""
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
SHIFTS = ("v_lshlrev_b64", "v_lshrrev_b64", "v_ashrrev_i64")


def code_objects(lib, tmp):
    fat = tmp / "fatbin.bin"
    subprocess.check_call([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)])
    data = fat.read_bytes()
    out, pos, k = [], 0, 0
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size:
                f = tmp / f"co_{k}.elf"
                f.write_bytes(data[i + off:i + off + size])
                out.append(f)
                k += 1
        pos = i + 1


def check(lib):
    bad = []
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d)
        for co in code_objects(lib, tmp):
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True).stdout
            allocs = sorted({(int(v) + 7) // 8 * 8 for v in re.findall(r"\.vgpr_count:\s+(\d+)", notes)})
            last = {a - 1 for a in allocs if a}
            dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True).stdout
            func, shifts = "?", 0
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    func = m.group(1)
                    continue
                t = line.strip()
                if not t.startswith(SHIFTS):
                    continue
                ops = [o.strip() for o in re.split(r",(?![^\[]*\])", t.split("//")[0].split(None, 1)[1])]
                m = re.match(r"^v(\d+)$", ops[1])
                if m:
                    shifts += 1
                    if int(m.group(1)) in last:
                        bad.append(f"{lib.name}: {func}: {t.split('//')[0].strip()}  (kernel allocations in this code object: {allocs})")
            print(f"{lib}: code object with allocations {allocs}: {shifts} 64-bit shift(s) with the amount in a vector register")
    return bad


if __name__ == "__main__":
    libs = [Path(a) for a in sys.argv[1:]] or [ROOT / "raytracing_amd" / "libmirt.so"]
    problems = [p for lib in libs for p in check(lib)]
    for p in problems:
        print("HAZARD:", p)
    sys.exit(1 if problems else 0)

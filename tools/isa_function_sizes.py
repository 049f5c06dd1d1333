"""Instruction counts per function of a built library's gfx950 code objects (all / vector), for comparing two builds:
python tools/isa_function_sizes.py LIB [NAME_PART] [--dump DIR]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from check_isa import LLVM, code_objects  # noqa: E402


def main():
    lib = Path(sys.argv[1])
    part = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    dump = Path(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else None
    with tempfile.TemporaryDirectory() as d:
        for k, co in enumerate(code_objects(lib, Path(d))):
            dis = subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--mcpu=gfx950", str(co)], capture_output=True, text=True).stdout
            if dump:
                dump.mkdir(parents=True, exist_ok=True)
                (dump / f"co_{k}.s").write_text(dis)
            cur, counts = None, {}
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                    counts[cur] = [0, 0]
                elif cur and re.match(r"^\s+[a-z]+_", line):
                    counts[cur][0] += 1
                    counts[cur][1] += bool(re.match(r"^\s+v_", line))
            for name, (n, v) in counts.items():
                if part in name and n:
                    print(f"{n:7d} {v:7d}  {name[:110]}")


main()

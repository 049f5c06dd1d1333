# usage: build_sched_variant.sh NAME STRATEGY FUNCTION_SUBSTRING...
# Builds variants/NAME/libmirt.so with the machine-scheduler strategy STRATEGY (e.g. max-ilp) applied only to the
# device functions of kernels.hip whose mangled name contains one of the substrings ("ALL": every function), through
# the per-function attribute "amdgpu-sched-strategy"; everything else is compiled as the Makefile does.
# Steps: device IR -> attribute edit -> llc -> lld -> offload bundle -> host object with that bundle -> link.
set -e
name=$1; strategy=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/raytracing_amd/csrc
out=$root/variants/$name; mkdir -p "$out"
tmp=$(mktemp -d)
llvm=/opt/rocm/lib/llvm/bin
flags="$MIRT_EXTRA_DEFS -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize"
cd "$src"
/opt/rocm/bin/hipcc $flags --cuda-device-only -emit-llvm -S -o $tmp/dev.ll kernels.hip 2>/dev/null
python3 - "$tmp/dev.ll" "$strategy" "$@" <<'PY'
import re, sys
path, strategy, picks = sys.argv[1], sys.argv[2], sys.argv[3:]
text = open(path).read()
groups = dict(re.findall(r'^attributes (#\d+) = \{(.*)\}$', text, re.M))
new_group = {}
count = 0
def retag(m):
    global count
    head, fname, rest, group = m.group(1), m.group(2), m.group(3), m.group(4)
    if not any(p == "ALL" or p in fname for p in picks):
        return m.group(0)
    count += 1
    if group not in new_group:
        new_group[group] = "#%d" % (900 + len(new_group))
    print("  ", strategy, "->", fname, file=sys.stderr)
    return "%s@%s(%s %s {" % (head, fname, rest, new_group[group])
text = re.sub(r'^(define [^@\n]*)@(\w+)\((.*) (#\d+)(?: comdat)? \{$', retag, text, flags=re.M)
for old, new in new_group.items():
    text += '\nattributes %s = {%s "amdgpu-sched-strategy"="%s" }\n' % (new, groups[old], strategy)
open(path, "w").write(text)
assert count, "no function matched"
PY
$llvm/llc -O3 -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 -filetype=obj -o $tmp/dev.o $tmp/dev.ll
$llvm/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $tmp/dev.out $tmp/dev.o
$llvm/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$tmp/dev.out -output=$tmp/dev.hipfb
/opt/rocm/bin/hipcc $flags --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $tmp/dev.hipfb -c -o $tmp/kernels.o kernels.hip 2>/dev/null
/opt/rocm/bin/hipcc $flags -c -o $tmp/bvh_gpu.o bvh_gpu.hip 2>/dev/null
/opt/rocm/bin/hipcc $flags -c -o $tmp/mirt_api.o mirt_api.cpp 2>/dev/null
/opt/rocm/bin/hipcc $flags -c -o $tmp/bvh_build.o bvh_build.cpp 2>/dev/null
/opt/rocm/bin/hipcc $flags -shared -Wl,-soname,libmirt.so -o "$out/libmirt.so" $tmp/kernels.o $tmp/bvh_gpu.o $tmp/mirt_api.o $tmp/bvh_build.o 2>/dev/null
rm -rf $tmp
ls -la "$out/libmirt.so"

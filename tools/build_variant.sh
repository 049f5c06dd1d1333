# usage: build_variant.sh NAME [COMMIT]  -- builds libmirt.so of COMMIT (default: the working tree) into variants/NAME/
set -e
name=$1; commit=$2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/variants/$name"
if [ -z "$commit" ]; then
  make -s -C "$root/raytracing_amd/csrc" OUT="$root/variants/$name/libmirt.so"
else
  tmp=$(mktemp -d)
  git -C "$root" archive "$commit" raytracing_amd/csrc include | tar -x -C "$tmp"
  (cd "$tmp/raytracing_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -shared -Wl,-soname,libmirt.so -o "$root/variants/$name/libmirt.so" $(ls *.hip *.cpp))
  rm -rf "$tmp"
fi

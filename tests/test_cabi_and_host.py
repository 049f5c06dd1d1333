"""CPU-side checks of the product: the C ABI library loads and exports what include/mirt.h declares, the ctypes
mirrors match the C structs, and the host mirror lowers the example scenes the way the reference's containers
would visit them.  No kernel is launched here."""
import ctypes as C
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import raytracing_amd as R
import oracle_lib as O

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "mirt.h"


def declared_functions():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mirt_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = R.mirt()
    names = declared_functions()
    assert set(names) == set(R.MIRT_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.mirt_version() >= 100


def test_no_wide_shift_reads_its_amount_from_the_last_register():
    """gfx950: a 64-bit shift whose amount sits in the last vector register of a wave's allocation shifts by 0 on
    most waves (tools/shift_hazard_probe.hip); the built library's code is disassembled and held to that rule."""
    if not Path("/opt/rocm/lib/llvm/bin/llvm-objdump").exists():
        pytest.skip("no disassembler")
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "check_isa.py"), str(R.LIBMIRT)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "code object" in out.stdout


def test_the_shift_checker_finds_the_pattern_where_it_is(tmp_path):
    """The reproducer of the gfx950 behaviour (tools/shift_hazard_probe.hip) puts shift amounts in the last register
    of its kernels' allocations on purpose: the checker the library's build relies on must say so."""
    hipcc = Path("/opt/rocm/bin/hipcc")
    if not hipcc.exists():
        pytest.skip("no hipcc")
    exe = tmp_path / "shift_hazard_probe"
    subprocess.run([str(hipcc), "-O2", "--offload-arch=gfx950", "-o", str(exe), str(ROOT / "tools" / "shift_hazard_probe.hip")],
                   check=True, capture_output=True)
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "check_isa.py"), str(exe)], capture_output=True, text=True)
    assert out.returncode == 1, out.stdout[-2000:]
    assert "HAZARD" in out.stdout and "v119" in out.stdout and "v127" in out.stdout


def test_ctypes_structs_match_the_header(tmp_path):
    structs = {"mirt_prim_header": R.PrimHeader, "mirt_sphere": R.Sphere, "mirt_moving_sphere": R.MovingSphere,
               "mirt_rect": R.Rect, "mirt_box": R.Box, "mirt_chain_op": R.ChainOp, "mirt_chain": R.Chain,
               "mirt_medium": R.Medium, "mirt_material": R.Material, "mirt_texture": R.Texture,
               "mirt_perlin": R.Perlin, "mirt_image": R.Image, "mirt_light": R.Light,
               "mirt_light_group": R.LightGroup, "mirt_scene_desc": R.SceneDesc, "mirt_camera": R.Camera,
               "mirt_render_params": R.RenderParams, "mirt_stats": R.Stats, "mirt_ray": R.Ray}
    src = tmp_path / "sizes.c"
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in structs)
    src.write_text(f'#include <stdio.h>\n#include "{HEADER}"\nint main(void) {{\n{body}\n  return 0;\n}}\n')
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)], check=True)  # header is plain C
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n, t in structs.items():
        assert int(sizes[n]) == C.sizeof(t), n


def test_context_create_fails_loudly_without_a_gpu():
    import os
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    rc = R.mirt().mirt_context_create(0, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert R.mirt().mirt_last_error(None)      # a message, not a silent CPU fallback
    with pytest.raises(R.MirtError):
        R.ExampleScene("cornel_box")


class Lowered:
    def __init__(self, name, seed=1):
        self.L = R.examples()
        self.L.rt_lowered_open.restype = C.c_void_p
        self.L.rt_lowered_open.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        self.L.rt_lowered_close.argtypes = [C.c_void_p]
        self.L.rt_lowered_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        self.L.rt_lowered_prims.restype = C.c_uint32
        self.L.rt_lowered_prims.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        self.h = self.L.rt_lowered_open(name.encode(), seed, str(R.ASSET_DIR).encode())
        assert self.h, self.L.rt_example_last_error()
        c = (C.c_uint32 * 16)()
        self.L.rt_lowered_counts(self.h, c)
        keys = ["spheres", "moving_spheres", "rects", "boxes", "chains", "chain_ops", "media", "materials", "textures",
                "perlins", "images", "lights", "light_groups", "lights_mode", "objects"]
        self.counts = dict(zip(keys, (int(x) for x in c)))
        n = self.L.rt_lowered_prims(self.h, None, 0)
        rows = np.zeros((n, 6), np.uint32)
        self.L.rt_lowered_prims(self.h, rows.ctypes.data, n)
        self.prims = rows  # tag, type_index, object_index, order, chain, material kind

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rt_lowered_close(self.h)


def test_lowering_spheres_matches_oracle_scene():
    lo = Lowered("spheres")
    o = O.Scene("spheres")
    c = lo.counts
    assert c["spheres"] + c["moving_spheres"] == o.object_count() == c["objects"]
    assert c["rects"] == c["boxes"] == c["media"] == 0 and c["chains"] == 1 and c["lights_mode"] == 0
    # world.rs:75-94 visits column by column: order follows (type_index, object_index)
    p = lo.prims[np.argsort(lo.prims[:, 3])]
    key = p[:, 1].astype(np.int64) * 100000 + p[:, 2]
    assert np.all(np.diff(key) > 0)
    # the five concrete sphere types of examples/spheres.rs get five columns
    assert len(set(p[:, 1])) == 5
    # the ground sphere is object (0,0) and the first visited
    assert tuple(p[0, 1:4]) == (0, 0, 0)


def test_lowering_bvh_order_is_the_oracle_tree_leaf_order():
    lo = Lowered("spheres_bvh")
    o = O.Scene("spheres_bvh")
    n = o.object_count()
    ti = np.zeros(n, np.uint32)
    oi = np.zeros(n, np.uint32)
    lib = O.lib()
    lib.oracle_scene_bvh_leaf_order.restype = C.c_uint32
    lib.oracle_scene_bvh_leaf_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    assert lib.oracle_scene_bvh_leaf_order(o.h, ti.ctypes.data, oi.ctypes.data, n) == n
    p = lo.prims[np.argsort(lo.prims[:, 3])]
    assert np.array_equal(p[:, 1], ti) and np.array_equal(p[:, 2], oi)


def test_lowering_cornell_box():
    lo = Lowered("cornel_box")
    c = lo.counts
    assert (c["spheres"], c["moving_spheres"], c["rects"], c["boxes"]) == (1, 0, 6, 1)
    assert c["objects"] == 8 and c["media"] == 0
    # FlipFace(light) and Translate(Rotate(box)) are the two non-empty chains
    assert c["chains"] == 3 and c["chain_ops"] == 3
    # lights: a World of an XZRect and a Sphere -> two groups of one (cornel_box.rs:12-26)
    assert (c["lights_mode"], c["lights"], c["light_groups"]) == (R.C.c_int(2).value, 2, 2)
    # 6 concrete types -> type_index 0..5 ; the two YZ rects and the two Lambertian XZ rects share columns
    assert sorted(set(lo.prims[:, 1])) == list(range(6))


def test_lowering_final_scene():
    lo = Lowered("final_scene")
    c = lo.counts
    assert c["boxes"] == 400 and c["rects"] == 1 and c["moving_spheres"] == 1
    # instanced cloud + glass, metal, earth, noise + the two media boundaries (stored, but not surfaces)
    assert c["spheres"] == 1000 + 4 + 2
    assert np.sum(lo.prims[:, 0] == 0) == 1004
    assert c["media"] == 2 and c["perlins"] == 1 and c["images"] == 1
    assert c["objects"] == 10 and c["lights_mode"] == 1 and c["lights"] == 1
    p = lo.prims
    # the 400 boxes belong to top-level object (0,0); the 1000 spheres to the Translate(Rotate(World)) object
    boxes = p[p[:, 0] == 3]
    assert np.all(boxes[:, 1] == 0) and np.all(boxes[:, 2] == 0)
    cloud = p[(p[:, 0] == 0) & (p[:, 4] != 0)]
    assert len(cloud) == 1000 and len(set(cloud[:, 1])) == 1 and len(set(cloud[:, 4])) == 1
    # visiting order is unique
    assert len(set(p[:, 3])) == len(p)


def test_sample_range_sharding_partitions_the_samples():
    for spp in (1, 5, 1024, 1000):
        for world in (1, 2, 3, 8):
            got = [R.shard_samples(spp, world, r) for r in range(world)]
            assert sum(c for _, c in got) == spp
            pos = 0
            for first, count in got:
                assert first == pos
                pos += count
            assert max(c for _, c in got) - min(c for _, c in got) <= 1

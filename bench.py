#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on final_scene at 1920x1080, 1024 spp, depth 5.

One "step" is one full render of the frame: every rank (one process per GPU) traces its
share of the samples of every pixel into an f32 sum image resident in HBM, the sums are
reduced to rank 0 over RCCL (only when N > 1) and rank 0 encodes the RGB24 image
(renderer.rs:232-259).  By default the `--spp` samples of every pixel are split over the ranks (scaling
"strong": an N-GPU step is the same frame as the one-GPU step); `--scaling weak` has every GPU trace `--spp`
samples per pixel instead, so an N-GPU step is the frame at N x spp samples per pixel.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `roofline` prices the render kernel (one launch traces a whole
step: camera paths, BVH walk, shading) from HIP-event durations taken inside the timed steps, against the
roof that binds the scene: vector-ALU work when the CUs serve nodes and primitive records from their LDS copies
of the scene, HBM bytes when those come from global memory.  `cpu_baseline` times the CPU oracle on a bounded
sample of the same scene.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
# RCCL shares buffers between the ranks of a node through dmabuf handles, the only kind the host driver here
# supports; the legacy IPC mode fails with "hipIpcGetMemHandle: invalid argument".  (Set before HIP loads.)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, spec
VALU_PEAK_TFLOPS = 157.3    # MI355X FP32 vector, spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="final_scene")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=0, help="0 = the example's own max depth")
    ap.add_argument("--pool", type=int, default=0, help="paths in flight per GPU (0 = library default)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N GPUs: strong = --spp split over the GPUs, weak = --spp samples per pixel on every GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--save", default="", help="write the last frame as PPM (rank 0)")
    ap.add_argument("--in-library", action="store_true",
                    help="one process: the frame is shared out over --gpus devices inside the library "
                         "(mirt_context_create_multi) instead of one process per GPU with an RCCL reduce")
    ap.add_argument("--row-bands", type=int, default=0, help="--in-library: bands of rows (0 = sample ranges only)")
    return ap.parse_args()


def kernel_source_hash():
    """What the committed counter runs are stamped with: the render kernel's sources as they stand."""
    import hashlib
    h = hashlib.sha256()
    for name in ("kernels.hip", "scene_device.h"):
        h.update((ROOT / "raytracing_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def counter_run(path):
    """A committed counter summary and whether it was measured on the kernel sources of this tree: per-segment
    figures of another build are still scaled to the launch, but the line says so (and stderr warns)."""
    m = json.loads(path.read_text())
    stamp = m.get("kernel_source_hash")
    m["_stale"] = stamp != kernel_source_hash()
    if m["_stale"]:
        print(f"bench.py: {path.name} was measured on kernel sources {stamp}, this tree has {kernel_source_hash()}: "
              "its per-segment figures describe another build", file=sys.stderr)
    return m


def measured_traffic(scene, segments_per_launch):
    """HBM bytes per launch from the committed counter run of this scene (FETCH_SIZE / WRITE_SIZE collected in
    separate rocprofv3 --pmc passes, per segment), scaled to this launch, and where it came from; (None, None)
    when no such run is on file."""
    for name in ("hbm_traffic.json", "hbm_traffic_" + scene.replace(":", "_") + ".json"):
        f = ROOT / "profiles" / name
        if f.exists():
            m = counter_run(f)
            if m.get("scene", "final_scene") == scene:
                return m["bytes_per_segment"] * segments_per_launch, {
                    "file": "profiles/" + name, "bytes_per_segment": m["bytes_per_segment"], "sq": m.get("sq"),
                    "kernel_source_hash": m.get("kernel_source_hash"), "commit": m.get("commit"), "stale": m["_stale"]}
    return None, None


def measured_ceiling(shape):
    """What the chip's memory system delivered to an access shape of the render kernel in a bare copy / gather kernel
    (tools/fetch_calibrate.hip, committed run): information beside the spec peak the roofline is priced against."""
    f = ROOT / "profiles" / "hbm_ceilings.json"
    if not f.exists():
        return None
    m = json.loads(f.read_text()).get(shape)
    return {"gb_per_s": m["gb_per_s"], "shape": m["shape"], "source": "profiles/hbm_ceilings.json"} if m else None


def valu_issue(scene, segments, kernel_seconds):
    """Share of the chip's vector-instruction issue slots the launch used: VALU instructions per segment from the
    committed SQ counter run of this scene x this launch's segments x the measured issue cost of a wave64
    instruction, over SIMDs x shader clock x the launch's duration.  None when no counter run is on file."""
    f = ROOT / "profiles" / "valu_issue.json"
    if not f.exists():
        return None
    m = counter_run(f)
    if m.get("scene") != scene:
        return None
    insts = m["valu_instructions_per_segment"] * segments
    slots = m["simds"] * m["shader_clock_ghz"] * 1e9 * kernel_seconds / m["cycles_per_wave64_instruction"]
    return {"frac": insts / slots, "valu_instructions_per_segment": m["valu_instructions_per_segment"],
            "cycles_per_wave64_instruction": m["cycles_per_wave64_instruction"],
            "shader_clock_ghz": m["shader_clock_ghz"], "active_lanes": m.get("active_lanes"),
            "wait_share": m.get("wait_share"), "source": m.get("source"),
            "kernel_source_hash": m.get("kernel_source_hash"), "commit": m.get("commit"), "stale": m["_stale"]}


def host_threads():
    """Hardware threads this process may really use: the CPUs it is allowed on, cut down to the container's CPU
    quota where there is one (a GPU box hands a job a share of the machine; its CPU count shows the whole
    machine)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    share = os.environ.get("MIRT_CPU_SHARE") or os.environ.get("OMP_NUM_THREADS")
    if share and share.isdigit() and int(share) > 0:
        n = min(n, int(share))
    return max(1, n)


def cpu_baseline(scene, width, height, depth, large=False):
    """Times the CPU oracle (the restatement of the reference's renderer: strip threads over a linear scan of the
    World) on a bounded sample, one thread per hardware thread this job may use.  `large`: a synthetic scene of
    more than 100 000 spheres, where the reference's containers (every object against every ray, world.rs:75-94,
    bvh.rs:92-145 with its always-true box test) would take hours per frame: the oracle's objects, materials and
    integrator behind a flattened BVH whose box test culls (oracle.hpp CulledBVH; kind "port-bvh")."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    cores = host_threads()
    kind, how = "port", "world.rs linear scan"
    if large:
        scene = "synthetic_culled:" + scene.split(":")[1]
        kind, how = "port-bvh", "the oracle's objects and integrator behind a culling BVH, not the reference's exhaustive containers"
    o = O.Scene(scene)
    # the same framing at a quarter of the resolution per axis; spp sized to about 15 s of CPU work
    w, h = max(2, width // 4), max(2, height // 4)
    t0 = time.perf_counter()
    o.render(w, h, 2, depth=depth, aspect=width / height, threads=cores)
    probe = time.perf_counter() - t0
    spp = int(min(1024, max(8, 15.0 / max(probe / 2, 1e-6))))
    t0 = time.perf_counter()
    _, _, segments = o.render(w, h, spp, depth=depth, aspect=width / height, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": w * h * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": kind,
            "host_cpu_count": os.cpu_count(),
            "sample": f"{scene} {w}x{h}, {spp} spp, depth {depth}: same scene and camera "
                      f"({how}), {cores} threads = every hardware thread this job may use "
                      f"(the box shows {os.cpu_count()}), {dt:.1f} s",
            "segments_per_sample": segments / (w * h * spp)}


def report(a, R, scene, world, rank0_count, first, total_spp, depth, aspect, elapsed, kernel_ms, kernel_launches,
           iterations, pool_used, golden, probe, sharding, rgb_image):
    """The JSON line of rank 0.  `probe(spp)` renders rank 0's share of a few samples per pixel with
    MIRT_RENDER_COUNT_WORK (outside the timed region) and returns the stats."""
    count = rank0_count
    samples_per_step = a.width * a.height * total_spp
    value = samples_per_step * a.steps / elapsed / 1e6
    # work per segment, from an instrumented pass outside the timed region (rank 0's share, few spp)
    probe_spp = max(1, min(count, 32))
    ws = probe(probe_spp)
    probe_samples = a.width * a.height * probe_spp
    s = ws["segments"] / probe_samples               # segments per sample
    n = ws["nodes_visited"] / max(1, ws["segments"])   # BVH nodes per segment
    p = ws["prims_tested"] / max(1, ws["segments"])    # primitive tests per segment
    roofline = None
    models = None
    if kernel_launches[R.MIRT_KERNEL_RENDER]:
        launches = kernel_launches[R.MIRT_KERNEL_RENDER]
        ker_s = kernel_ms[R.MIRT_KERNEL_RENDER] / 1e3
        my_segments = s * a.width * a.height * count * a.steps
        # Per segment, two models of the work (both from the counts s, n, p measured above):
        #   bytes  192 B of path state (ray, hit, throughput, queue entry, written once and read once per
        #          stage) + a 64-B node per tree step + a 32-B primitive record per primitive test
        #   flops  two slab tests per tree step (44), ~26 per primitive test, ~160 to shade a hit, three
        #          Philox4x32-10 blocks of ~100 integer operations
        bytes_per_segment = 192 + 64 * n + 32 * p
        flops_per_segment = 44 * n + 26 * p + 160 + 3 * 100
        gbs = my_segments * bytes_per_segment / ker_s / 1e9
        tf = my_segments * flops_per_segment / ker_s / 1e12
        traffic, traffic_source = measured_traffic(a.scene, my_segments / launches)
        cyc = ws["phase_cycles"]
        names = ["start_fetch", "tree_steps", "primitive_tests", "shade", "hand_in", "total"]
        common = {"kernel": "k_render", "avg_launch_ms": kernel_ms[R.MIRT_KERNEL_RENDER] / launches,
                  "launches": launches, "segments_per_launch": my_segments / launches,
                  "segments_per_sample": s, "nodes_per_segment": n, "prims_per_segment": p,
                  "scene_in_lds": ws["scene_in_lds"],
                  "lanes_at_work": {"tree_steps": ws["nodes_visited"] / max(1, 64 * ws["node_wave_steps"]),
                                    "primitive_tests": ws["prims_tested"] / max(1, 64 * ws["prim_wave_steps"])},
                  "wave_time_share": {k: cyc[i] / max(1, cyc[5]) for i, k in enumerate(names[:5])}}
        if ws["scene_in_lds"]:
            # Nodes and primitive records never leave the CU: the vector ALUs bound the launch.  HBM sees the
            # path records only; `traffic` is what the counters measured of it.
            roofline = {"bound": "valu", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / VALU_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                        "flops_per_segment": flops_per_segment, **common}
            roofline["issue"] = valu_issue(a.scene, my_segments / launches, ker_s / launches)
            if traffic is not None:
                hbm = traffic / (ker_s / launches) / 1e9
                roofline["hbm_measured"] = {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": hbm / HBM_PEAK_GBS, "copy_kernel": measured_ceiling("stream16")}
        else:
            # The fraction of the HBM roof is priced from what the counters saw reach HBM: the L2 and the
            # Infinity Cache serve most of the node and record lines many rays share, so the byte model (every
            # node and record read counted, wherever it is served from) is not an HBM figure; it stays under
            # `work_models`.  Without a counter run of this scene on file no fraction is claimed.
            hbm = traffic / (ker_s / launches) / 1e9 if traffic is not None else None
            roofline = {"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm / HBM_PEAK_GBS if hbm is not None else None, "traffic": traffic,
                        "traffic_source": traffic_source,
                        "priced_from": "FETCH_SIZE + WRITE_SIZE of the committed counter run, scaled by segments"
                        if traffic is not None else "nothing: no counter run of this scene on file",
                        "byte_model_bytes_per_segment": bytes_per_segment,
                        "gather_kernel": measured_ceiling("gather64"), **common}
            # what the same counter run's SQ passes say of the vector ALUs: the walk issues its tree steps and
            # primitive tests with partly filled waves, and that, not HBM, is the nearest roof of this launch
            sq = (traffic_source or {}).get("sq")
            if sq:
                insts = sq["valu_instructions_per_segment"] * my_segments / launches
                slots = 1024 * 2.072e9 * (ker_s / launches) / 2.5
                roofline["issue"] = {"frac": insts / slots, "cycles_per_wave64_instruction": 2.5, "shader_clock_ghz": 2.072, **sq}
        # both models side by side, whichever roof binds (informational)
        models = {"bytes_per_segment": bytes_per_segment, "algorithmic_gb_per_s": gbs,
                  "flops_per_segment": flops_per_segment, "algorithmic_tflop_per_s": tf,
                  "note": "the byte model counts node and record reads wherever they are served from"}
    size = "1080p" if (a.width, a.height) == (1920, 1080) else f"{a.width}x{a.height}"
    result = {
        "metric": f"Msamples/s on {a.scene} {size}", "value": value, "unit": "Msamples/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.scene} {a.width}x{a.height}, {total_spp} spp, depth {depth}",
                   "sharding": sharding, "path_pool": pool_used or ws["path_pool"], "seed": a.seed,
                   "rounds_per_step": iterations / max(1, a.steps)},
        "roofline": roofline, "work_models": models,
    }
    # (rank 0 alone, after the timed region; with N > 1 the other ranks wait at the barrier below)
    if a.no_cpu_baseline:
        result["cpu_baseline"] = None
    else:
        large = a.scene.startswith("synthetic") and scene.primitive_count() > 100000
        result["cpu_baseline"] = cpu_baseline(a.scene, a.width, a.height, depth, large=large)
    result["golden"] = golden
    if a.save:  # the frame of the last timed step
        with open(a.save, "wb") as f:
            f.write(f"P6\n{a.width} {a.height}\n255\n".encode())
            f.write(rgb_image().tobytes())
    return result

def main_in_library(a):
    """`--in-library`: one process, the frame shared out over `--gpus` devices inside the library
    (mirt_context_create_multi: a scene copy, a workspace and a host thread per device, the other devices' rows
    copied to the first one over the GPUs' links and added there, resolve on the first device).  A step is one
    mirt_render into host buffers -- what a drop-in Renderer::render does -- so the 6 MB RGB24 download is inside
    the timed region.  MIRT_BENCH_REHEARSE=1 lists device 0 N times (a one-GPU box)."""
    import numpy as np
    import torch
    import raytracing_amd as R
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        print("bench.py: --in-library is a single process; do not launch it under torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible", file=sys.stderr)
        sys.exit(2)
    n = a.gpus
    rehearse = os.environ.get("MIRT_BENCH_REHEARSE") == "1"
    if not rehearse and torch.cuda.device_count() < n:
        print(f"bench.py: --gpus {n} but {torch.cuda.device_count()} device(s) visible", file=sys.stderr)
        sys.exit(2)
    devices = [0] * n if rehearse else list(range(n))
    total_spp = a.spp * n if a.scaling == "weak" else a.spp
    single = R.ExampleScene(a.scene, scene_seed=1, device=0)      # camera, depth, and the instrumented pass
    depth = a.depth or single.depth
    aspect = a.width / a.height
    cam, bg = single.camera(aspect)
    shared = R.SharedScene(a.scene, devices, row_bands=a.row_bands)
    flags = 0 if a.no_kernel_timing else R.MIRT_RENDER_TIME_KERNELS
    for _ in range(a.warmup):
        shared.render(cam, bg, a.width, a.height, total_spp, depth, seed=a.seed, path_pool=a.pool, want_sums=False)
    torch.cuda.synchronize()
    kernel_ms = [0.0] * R.MIRT_KERNEL_COUNT
    kernel_launches = [0] * R.MIRT_KERNEL_COUNT
    iterations = 0
    pool_used = 0
    rgb = sums = None
    t0 = time.perf_counter()
    for k in range(a.steps):
        rgb, sums, st = shared.render(cam, bg, a.width, a.height, total_spp, depth, seed=a.seed, flags=flags,
                                      path_pool=a.pool, want_sums=(k == a.steps - 1))
        iterations += st["iterations"]
        pool_used = st["path_pool"]
        for j in range(R.MIRT_KERNEL_COUNT):
            kernel_ms[j] += st["kernel_ms"][j]           # the slowest device's launch
            kernel_launches[j] += 1 if st["kernel_launches"][j] else 0
    elapsed = time.perf_counter() - t0
    golden = None
    fixture = ROOT / "tests" / "golden" / f"{a.scene}_blocks.npz"
    if a.steps > 0 and fixture.exists():
        from raytracing_amd.imageops import compare_with_block_fixture
        fx = np.load(fixture)
        if (int(fx["width"]), int(fx["height"]), int(fx["depth"])) == (a.width, a.height, depth):
            golden = compare_with_block_fixture(sums, total_spp, fx)
            golden["fixture"] = f"tests/golden/{fixture.name} (oracle, {int(fx['spp'])} spp per pixel)"
    first, count = R.shard_samples(total_spp, n if a.row_bands <= 1 else max(1, n // a.row_bands), 0)

    def probe(spp):
        _, _, st = single.render(a.width, a.height, spp, depth=depth, aspect=aspect, seed=a.seed, first_sample=first,
                                 path_pool=a.pool, flags=R.MIRT_RENDER_COUNT_WORK)
        return st
    sharding = (f"inside the library: one context over {n} device(s) {devices}, row bands {a.row_bands or 1} x sample "
                "ranges, peer copies + add on the first device; one mirt_render per step, host buffers")
    result = report(a, R, single, n, count, first, total_spp, depth, aspect, elapsed, kernel_ms, kernel_launches,
                    iterations, pool_used, golden, probe, sharding, lambda: rgb)
    if a.row_bands > 1 and result.get("roofline"):
        result["roofline"]["note"] = "row bands: a device renders part of the rows; per-launch segments are an estimate"
    print(json.dumps(result), flush=True)


def main():
    a = parse()
    if a.in_library:
        return main_in_library(a)
    import torch
    import torch.distributed as dist
    import raytracing_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible", file=sys.stderr)
        sys.exit(2)
    # MIRT_BENCH_REHEARSE=1: every rank on GPU 0 over gloo, to walk through the N > 1 code on a one-GPU box
    rehearse = world > 1 and os.environ.get("MIRT_BENCH_REHEARSE") == "1"
    device_index = 0 if rehearse else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene = R.ExampleScene(a.scene, scene_seed=1, device=device_index)
    depth = a.depth or scene.depth
    aspect = a.width / a.height
    # sample-range sharding: rank r renders samples [first, first + count) of every pixel
    total_spp = a.spp * world if a.scaling == "weak" else a.spp
    first, count = R.shard_samples(total_spp, world, rank)

    sums = torch.zeros((a.height, a.width, 3), dtype=torch.float32, device=dev)
    rgb = torch.zeros((a.height, a.width, 3), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    flags = 0 if a.no_kernel_timing else R.MIRT_RENDER_TIME_KERNELS

    def step(step_flags):
        sums.zero_()
        st = None
        if count > 0:
            st = scene.render_accumulate(sums.data_ptr(), a.width, a.height, count, depth=depth, aspect=aspect,
                                         seed=a.seed, first_sample=first, path_pool=a.pool, flags=step_flags,
                                         stream=stream)
        if world > 1:
            if rehearse:  # gloo has no reduce for device tensors
                dist.all_reduce(sums, op=dist.ReduceOp.SUM)
            else:
                dist.reduce(sums, dst=0, op=dist.ReduceOp.SUM)
        if rank == 0:
            scene.resolve(sums.data_ptr(), a.width, a.height, total_spp, rgb.data_ptr(), stream=stream)
        return st

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(0)
    fence()
    t0 = time.perf_counter()
    kernel_ms = [0.0] * R.MIRT_KERNEL_COUNT
    kernel_launches = [0] * R.MIRT_KERNEL_COUNT
    iterations = 0
    pool_used = 0
    for _ in range(a.steps):
        st = step(flags)
        if st:
            iterations += st["iterations"]
            pool_used = st["path_pool"]
            for k in range(R.MIRT_KERNEL_COUNT):
                kernel_ms[k] += st["kernel_ms"][k]
                kernel_launches[k] += st["kernel_launches"][k]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The frame of the last timed step (on rank 0, after the reduce: the whole frame) as an image: block means
    # against the oracle's frame of the same size (tests/golden/<scene>_blocks.npz, when the workload is the one the
    # fixture was made for).  Taken now, before the instrumented pass below adds its samples to the buffer.
    golden = None
    fixture = ROOT / "tests" / "golden" / f"{a.scene}_blocks.npz"
    if rank == 0 and a.steps > 0 and fixture.exists():
        import numpy as np
        from raytracing_amd.imageops import compare_with_block_fixture
        fx = np.load(fixture)
        if (int(fx["width"]), int(fx["height"]), int(fx["depth"])) == (a.width, a.height, depth):
            golden = compare_with_block_fixture(sums.cpu().numpy(), total_spp, fx)
            golden["fixture"] = f"tests/golden/{fixture.name} (oracle, {int(fx['spp'])} spp per pixel)"

    result = None
    if rank == 0:
        def probe(spp):
            st = scene.render_accumulate(sums.data_ptr(), a.width, a.height, spp, depth=depth, aspect=aspect,
                                         seed=a.seed, first_sample=first, path_pool=a.pool,
                                         flags=R.MIRT_RENDER_COUNT_WORK, stream=stream)
            torch.cuda.synchronize()
            return st
        sharding = (f"sample ranges over {world} GPU(s), one process each, RCCL sum-reduce of the f32 image"
                    if world > 1 else "single GPU")
        result = report(a, R, scene, world, count, first, total_spp, depth, aspect, elapsed, kernel_ms, kernel_launches,
                        iterations, pool_used, golden, probe, sharding, lambda: rgb.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

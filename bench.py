#!/usr/bin/env python3
"""Headline benchmark: Msamples/s on final_scene at 1920x1080, 1024 spp, depth 5.

One "step" is one full render of the frame: every rank (one process per GPU) traces its
share of the samples of every pixel into an f32 sum image resident in HBM, the sums are
reduced to rank 0 over RCCL (only when N > 1) and rank 0 encodes the RGB24 image
(renderer.rs:232-259).  By default every GPU traces `--spp` samples per pixel, so an N-GPU step is the
frame at N x spp samples per pixel (scaling "weak": the work per GPU is fixed and no rank waits for another
before the one reduce); `--scaling strong` splits the same `--spp` over the ranks instead.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints one JSON line.  `roofline` prices the render kernel (one launch traces a whole
step: camera paths, BVH walk, shading) from HIP-event durations taken inside the timed steps;
`cpu_baseline` times the CPU oracle on a bounded sample of the same scene.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

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
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N GPUs: weak = --spp samples per pixel on every GPU, strong = --spp split over the GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--save", default="", help="write the last frame as PPM (rank 0)")
    return ap.parse_args()


def measured_traffic(segments_per_launch):
    """HBM bytes per launch from the committed counter run (FETCH_SIZE / WRITE_SIZE collected in separate
    rocprofv3 --pmc passes over this workload, per segment), scaled to this launch; None without the file."""
    f = ROOT / "profiles" / "hbm_traffic.json"
    if not f.exists():
        return None
    m = json.loads(f.read_text())
    return m["bytes_per_segment"] * segments_per_launch


def cpu_baseline(scene, width, height, depth):
    """Times the CPU oracle (the restatement of the reference's renderer) on a bounded sample."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as O
    cores = min(16, os.cpu_count() or 1)
    o = O.Scene(scene)
    # about 12 M samples of the same framing (10-20 s of CPU work): a quarter of the resolution per axis, 96 spp
    w, h, spp = max(2, width // 4), max(2, height // 4), 96
    t0 = time.perf_counter()
    _, _, segments = o.render(w, h, spp, depth=depth, aspect=width / height, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": w * h * spp / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{scene} {w}x{h}, {spp} spp, depth {depth}: same scene, camera and algorithm "
                      f"(world.rs linear scan), {dt:.1f} s",
            "segments_per_sample": segments / (w * h * spp)}


def main():
    a = parse()
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
    for _ in range(a.steps):
        st = step(flags)
        if st:
            iterations += st["iterations"]
            for k in range(R.MIRT_KERNEL_COUNT):
                kernel_ms[k] += st["kernel_ms"][k]
                kernel_launches[k] += st["kernel_launches"][k]
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    result = None
    if rank == 0:
        samples_per_step = a.width * a.height * total_spp
        value = samples_per_step * a.steps / elapsed / 1e6
        # work per segment, from an instrumented pass outside the timed region (rank 0's share, few spp)
        probe_spp = max(1, min(count, 4))
        ws = scene.render_accumulate(sums.data_ptr(), a.width, a.height, probe_spp, depth=depth, aspect=aspect,
                                     seed=a.seed, first_sample=first, path_pool=a.pool,
                                     flags=R.MIRT_RENDER_COUNT_WORK, stream=stream)
        torch.cuda.synchronize()
        probe_samples = a.width * a.height * probe_spp
        s = ws["segments"] / probe_samples               # segments per sample
        n = ws["nodes_visited"] / max(1, ws["segments"])   # BVH nodes per segment
        p = ws["prims_tested"] / max(1, ws["segments"])    # primitive tests per segment
        roofline = None
        valu = None
        if kernel_launches[R.MIRT_KERNEL_RENDER]:
            # One launch of k_render traces a whole step.  Algorithmic bytes per segment: 192 B of path state
            # (ray, hit, throughput, queue entry, written once and read once per stage) + a 64-B node per tree
            # step + a 32-B primitive record per primitive test.
            bytes_per_segment = 192 + 64 * n + 32 * p
            my_segments = s * a.width * a.height * count * a.steps
            ker_s = kernel_ms[R.MIRT_KERNEL_RENDER] / 1e3
            achieved = my_segments * bytes_per_segment / ker_s / 1e9
            cyc = ws["phase_cycles"]
            names = ["start_fetch", "tree_steps", "primitive_tests", "shade", "hand_in", "total"]
            roofline = {"kernel": "k_render", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": measured_traffic(my_segments / kernel_launches[R.MIRT_KERNEL_RENDER]),
                        "avg_launch_ms": kernel_ms[R.MIRT_KERNEL_RENDER] / kernel_launches[R.MIRT_KERNEL_RENDER],
                        "launches": kernel_launches[R.MIRT_KERNEL_RENDER],
                        "bytes_per_segment": bytes_per_segment,
                        "segments_per_launch": my_segments / kernel_launches[R.MIRT_KERNEL_RENDER],
                        "segments_per_sample": s, "nodes_per_segment": n, "prims_per_segment": p,
                        "lanes_at_work": {"tree_steps": ws["nodes_visited"] / max(1, 64 * ws["node_wave_steps"]),
                                          "primitive_tests": ws["prims_tested"] / max(1, 64 * ws["prim_wave_steps"])},
                        "wave_time_share": {k: cyc[i] / max(1, cyc[5]) for i, k in enumerate(names[:5])}}
            # two slab tests per tree step (44 flops), ~26 per primitive test, ~160 to shade a hit,
            # three Philox4x32-10 blocks of ~100 integer operations
            flops_per_segment = 44 * n + 26 * p + 160 + 3 * 100
            tf = my_segments * flops_per_segment / ker_s / 1e12
            valu = {"kernel": "k_render", "bound": "valu", "achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / VALU_PEAK_TFLOPS, "flops_per_segment": flops_per_segment}
        result = {
            "metric": "Msamples/s on final_scene 1080p", "value": value, "unit": "Msamples/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.scene} {a.width}x{a.height}, {total_spp} spp, depth {depth}",
                       "sharding": f"sample ranges over {world} GPU(s), RCCL sum-reduce of the f32 image" if world > 1
                       else "single GPU", "path_pool": ws["path_pool"], "seed": a.seed,
                       "rounds_per_step": iterations / max(1, a.steps)},
            "roofline": roofline, "roofline_valu": valu,
        }
        if not a.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(a.scene, a.width, a.height, depth)
        else:
            result["cpu_baseline"] = None
        if a.save:  # the frame of the last timed step
            img = rgb.cpu().numpy()
            with open(a.save, "wb") as f:
                f.write(f"P6\n{a.width} {a.height}\n255\n".encode())
                f.write(img.tobytes())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

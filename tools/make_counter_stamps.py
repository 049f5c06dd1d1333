#!/usr/bin/env python3
"""Turns the raw outputs of tools/gpu_r3_profile.sh TAG (gpurun_out/TAG) into what bench.py reads and what is kept:
profiles/valu_issue.json, profiles/hbm_traffic.json, profiles/hbm_traffic_synthetic_bvh_1581.json -- each stamped
with the hash of the kernel sources and the commit they were measured on -- and copies of the run's summaries as
profiles/TAG_*.   usage: make_counter_stamps.py TAG [lines]   (run in the tree the GPU run was sent from, before any
further change to raytracing_amd/csrc/kernels.hip or scene_device.h)"""
import csv
import io
import json
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import kernel_source_hash  # noqa: E402

tag = sys.argv[1]
O = ROOT / "gpurun_out" / tag
P = ROOT / "profiles"
if len(sys.argv) > 2 and sys.argv[2] == "lines":  # after tools/gpu_r3_lines.sh: only the bench lines and kernel stats are new
    for f in sorted(O.iterdir()):
        if "bench" in f.name and f.suffix == ".json" or f.name.endswith("kernel_stats.csv"):
            shutil.copy(f, P / f"{tag}_stamped_{f.name}")
            print("profiles/" + f"{tag}_stamped_{f.name}")
    sys.exit(0)
commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "raytracing_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
stamp = {"kernel_source_hash": kernel_source_hash(), "commit": commit + ("+uncommitted changes to raytracing_amd/csrc" if dirty else "")}


def counters(path):
    out = {}
    for line in path.read_text().splitlines():
        m = re.match(r"\s+(\w+)\s+([0-9.e+]+)$", line)
        if m:
            out[m.group(1)] = float(m.group(2))
    return out


def segments_of(path):
    return int(re.search(r"segments (\d+)", path.read_text()).group(1))


def rows(path):
    out = {}
    for r in csv.reader(io.StringIO(path.read_text())):
        if len(r) > 16:
            out[r[15]] = float(r[16])
    return out


# SQ counters of the final_scene kernel (32-spp frame) -> issue-slot model
c = counters(O / "pmc_sq_counters.txt")
seg32 = segments_of(O / "phase_32.log")
valu = {
    "scene": "final_scene",
    "workload": "final_scene 1920x1080, 32 spp, depth 5 (one k_render launch)",
    "segments": seg32,
    "SQ_INSTS_VALU": c["SQ_INSTS_VALU"],
    "valu_instructions_per_segment": c["SQ_INSTS_VALU"] / seg32,
    "SQ_WAVE_CYCLES_quad_cycles": c["SQ_WAVE_CYCLES"],
    "SQ_WAVES": c["SQ_WAVES"],
    "simds": 1024,
    "shader_clock_ghz": 2.072,
    "cycles_per_wave64_instruction": 2.5,
    "active_lanes": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
    "wait_share": {"SQ_WAIT_ANY": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "SQ_WAIT_INST_ANY": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]},
    "source": f"SQ counters: profiles/{tag}_pmc_sq_counters.txt (rocprofv3 --pmc, two passes, tools/gpu_r3_profile.sh {tag}); active lanes = "
              "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU; issue cost: profiles/r2_valu_issue_probe.jsonl (v_add/v_fma/min/max mix at 4 waves "
              "per SIMD: 2.44-2.55 cycles per wave64 instruction per SIMD; v_rcp_f32 8.1, v_mad_u64_u32 4.7, so the figure understates the "
              "slots this kernel's divisions, square roots and Philox rounds take)",
    **stamp,
}
(P / "valu_issue.json").write_text(json.dumps(valu, indent=1))

# HBM-side traffic of the benchmark's own frame
r = rows(O / "pmc_fetch_write_rows_1024spp.csv")
seg1024 = segments_of(O / "phase_1024.log")
fetch, write = r["FETCH_SIZE"] * 1024.0, r["WRITE_SIZE"] * 1024.0
hbm = {
    "scene": "final_scene",
    "workload": "final_scene 1920x1080, 1024 spp, depth 5, pool sized by the library (32 Mi): the benchmark's own frame, one k_render launch",
    "segments": seg1024,
    "segments_source": "MIRT_RENDER_COUNT_WORK run of the same frame (tools/phase_probe.py final_scene 1024)",
    "FETCH_SIZE_KB": r["FETCH_SIZE"], "WRITE_SIZE_KB": r["WRITE_SIZE"],
    "collected": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (tools/gpu_r3_profile.sh {tag}); raw rows in {tag}_pmc_fetch_write_rows_1024spp.csv",
    "correction": "FETCH_SIZE doubled: the kernel's fetches are streams of consecutive 16-byte quads, one per lane, from the four arrays of a "
                  "record buffer, the shape tools/fetch_calibrate.hip's k_stream16 shows reported at half (r2_e_fetch_size_calibration_rows.csv: "
                  "1 572 876 KB of 3 221 225 472 B); WRITE_SIZE as reported (exact in the same calibration)",
    "bytes_per_segment": (2.0 * fetch + write) / seg1024,
    "fetch_bytes_per_segment_raw": fetch / seg1024,
    "write_bytes_per_segment": write / seg1024,
    **stamp,
}
(P / "hbm_traffic.json").write_text(json.dumps(hbm, indent=1))

# the walk from global memory on config 5's scene
r = rows(O / "big10m_pmc_fetch_write_rows.csv")
segbig = segments_of(O / "big10m_count.log")
big_bench = json.loads((O / "big10m_bench.json").read_text())
old = json.loads((P / "hbm_traffic_synthetic_bvh_1581.json").read_text())
fetch, write = r["FETCH_SIZE"] * 1024.0, r["WRITE_SIZE"] * 1024.0
cb = counters(O / "big10m_pmc_sq_counters.txt")
big = {
    "scene": "synthetic_bvh:1581",
    "workload": "synthetic_bvh:1581 (10 004 571 spheres) 1920x1080, 64 spp, depth 10, default pool (one k_render launch)",
    "segments": segbig,
    "segments_source": "MIRT_RENDER_COUNT_WORK run of the same frame (tools/big_bench_probe.py 1581 64 count)",
    "FETCH_SIZE_KB": r["FETCH_SIZE"], "WRITE_SIZE_KB": r["WRITE_SIZE"],
    "collected": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (tools/gpu_r3_profile.sh {tag}); raw rows in {tag}_big10m_pmc_fetch_write_rows.csv",
    "correction": old["correction"],
    "bytes_per_segment": (fetch + write) / segbig,
    "fetch_bytes_per_segment": fetch / segbig,
    "write_bytes_per_segment": write / segbig,
    "kernel_ms": big_bench["roofline"]["avg_launch_ms"],
    "sq": {"valu_instructions_per_segment": cb["SQ_INSTS_VALU"] / segbig,
           "active_lanes": cb["SQ_THREAD_CYCLES_VALU"] / cb["SQ_ACTIVE_INST_VALU"],
           "SQ_WAIT_ANY_share": cb["SQ_WAIT_ANY"] / cb["SQ_WAVE_CYCLES"], "SQ_WAIT_INST_ANY_share": cb["SQ_WAIT_INST_ANY"] / cb["SQ_WAVE_CYCLES"],
           "source": f"profiles/{tag}_big10m_pmc_sq_counters.txt"},
    "before": {"build": old.get("build"), "bytes_per_segment": old["bytes_per_segment"], "kernel_ms": old.get("kernel_ms")},
    **stamp,
}
(P / "hbm_traffic_synthetic_bvh_1581.json").write_text(json.dumps(big, indent=1))

for f in sorted(O.iterdir()):
    if f.suffix in (".json", ".csv", ".txt") or f.name.startswith("phase_") or f.name == "big10m_count.log":
        shutil.copy(f, P / f"{tag}_{f.name.replace('.log', '.txt')}")
print(json.dumps({"valu_per_segment": valu["valu_instructions_per_segment"], "active_lanes": valu["active_lanes"],
                  "hbm_bytes_per_segment": hbm["bytes_per_segment"], "big_bytes_per_segment": big["bytes_per_segment"], **stamp}, indent=1))

#!/usr/bin/env python3
"""Turns the csv output of profiles/run_rocprof.sh (rocprofv3 --kernel-trace --stats and the separate --pmc passes) into
  <out>/<tag>_kernel_stats.csv        the --stats table as rocprofv3 wrote it (kernels of this repo + top torch ones)
  <out>/<tag>_rocprof_summary.md      kernel-trace average of the dominant kernel + the counter table, per launch
  <out>/pmc_traffic_<pipeline>_<C>x<N>.json   what bench.py reports as roofline.traffic -- stamped with the sha256 of the
                                      kernel sources it was measured on (bench.py refuses it for any other sources)
Usage (on the GPU box, after run_rocprof.sh):  python profiles/pmc_summary.py <prof_dir> <out_dir> <tag> [C N]
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE counts wide coalesced reads at half size (x2)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KERNEL = "k_fused"
STEPS = 6            # --steps of the profiled command: the last STEPS launches are its timed region


def counters(path):
    """{counter: mean value per launch of KERNEL} (the counter files hold one row per dispatch and counter)."""
    acc = {}
    if not os.path.exists(path):
        return {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    # the timed region = the last STEPS launches (before them: clock-ramp and warm-up passes)
    return {k: sum(v[-STEPS:]) / len(v[-STEPS:]) for k, v in acc.items()}


def main(prof, out, tag, C=4096, N=36000):
    os.makedirs(out, exist_ok=True)
    stats_rows, kern = [], None
    with open(os.path.join(prof, "trace", "r02_kernel_stats.csv"), newline="") as f:
        for row in csv.DictReader(f):
            stats_rows.append(row)
            if KERNEL in row["Name"]:
                kern = row
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(stats_rows[0].keys()))
        w.writeheader()
        for row in stats_rows[:12]:
            w.writerow(row)
    # per-launch durations in launch order: the first launches after idle run on a ramping clock (bench.py RAMP_STEPS)
    durs = []
    with open(os.path.join(prof, "trace", "r02_kernel_trace.csv"), newline="") as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"]:
                durs.append((int(row["Start_Timestamp"]), (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
    durs = [d for _, d in sorted(durs)]
    timed = durs[-STEPS:]
    c = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        c.update(counters(os.path.join(prof, sub, "r02_counter_collection.csv")))
    avg_ms = float(kern["AverageNs"]) / 1e6
    fetch_kb, write_kb = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
    traffic = (fetch_kb * 1024 * 2 + write_kb * 1024) if fetch_kb is not None and write_kb is not None else None
    algo = 9.0 * C * N
    wgs = (C + 15) // 16
    issue = {}
    if "SQ_INSTS_VALU" in c:
        issue["valu_insts_per_workgroup_sample"] = round(c["SQ_INSTS_VALU"] / wgs / N, 1)
        issue["valu_insts_per_channel_sample"] = round(c["SQ_INSTS_VALU"] / C / N, 2)
    if "SQ_INSTS_SALU" in c:
        issue["salu_insts_per_workgroup_sample"] = round(c["SQ_INSTS_SALU"] / wgs / N, 1)
    if "SQ_INSTS_LDS" in c:
        issue["lds_insts_per_workgroup_sample"] = round(c["SQ_INSTS_LDS"] / wgs / N, 1)
    if "GRBM_GUI_ACTIVE" in c:
        issue["cycles_per_sample"] = round(c["GRBM_GUI_ACTIVE"] / 8 / N, 1)      # 8 XCDs each count the launch's clocks
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        issue["valu_pipes_busy_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 3)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        issue["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
    issue["note"] = ("per launch of k_fused, mean over the timed launches of separate rocprofv3 --pmc passes of `bench.py --steps 6 "
                     "--warmup 1`; SQ_ACTIVE_INST_VALU x4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8)")
    import bench
    d = {"kernel": KERNEL, "workload": "%dx%d" % (C, N), "kernel_source_sha256": bench.kernel_source_hash(),
         "kernel_trace_avg_ms": round(avg_ms, 4), "kernel_trace_calls": int(kern["Calls"]),
         "kernel_trace_timed_region_avg_ms": round(sum(timed) / len(timed), 4), "kernel_trace_launch_ms": [round(d, 4) for d in durs],
         "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "traffic_bytes_per_launch": traffic,
         "algorithmic_bytes_per_launch": algo,
         "correction": "FETCH_SIZE KiB*1024*2 (gfx950 half-count of wide coalesced reads) + WRITE_SIZE KiB*1024",
         "source": "profiles/%s_rocprof_summary.md" % tag, "issue": issue, "raw_counters_per_launch": c}
    with open(os.path.join(out, "pmc_traffic_fused_%dx%d.json" % (C, N)), "w") as f:
        json.dump(d, f, indent=1)
    with open(os.path.join(out, tag + "_rocprof_summary.md"), "w") as f:
        f.write("# %s: rocprofv3 summary of `python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-check --no-host-path --no-large-batch --no-config5 --no-time-major --no-chain --channels %d`\n\n" % (tag, C))
        f.write("kernel sources sha256 `%s`\n\n" % d["kernel_source_sha256"])
        f.write("## --kernel-trace --stats (top rows; full table in %s_kernel_stats.csv)\n\n| kernel | calls | avg ms | %% |\n|---|---|---|---|\n" % tag)
        for row in stats_rows[:6]:
            f.write("| `%s` | %s | %.4f | %s |\n" % (row["Name"][:100], row["Calls"], float(row["AverageNs"]) / 1e6, row["Percentage"]))
        f.write("\n%s, --stats row: %.4f ms average over ALL %s launches -> 9 B x %d x %d = %.3f GB / launch -> %.1f GB/s = %.2f %% of 8 TB/s\n\n"
                % (KERNEL, avg_ms, kern["Calls"], C, N, algo / 1e9, algo / avg_ms / 1e6, algo / avg_ms / 1e6 / 80.0))
        f.write("launch by launch (ms): %s\n\n**timed region (last %d launches, what bench.py's HIP events cover): %.4f ms average** -> %.1f GB/s = "
                "%.2f %% of 8 TB/s; the launches before it are bench.py's clock-ramp and warm-up passes\n\n"
                % (", ".join("%.3f" % d for d in durs), len(timed), sum(timed) / len(timed), algo / (sum(timed) / len(timed)) / 1e6,
                   algo / (sum(timed) / len(timed)) / 1e6 / 80.0))
        f.write("## counters per launch (separate --pmc passes)\n\n| counter | per launch |\n|---|---|\n")
        for k in sorted(c):
            f.write("| %s | %.4g |\n" % (k, c[k]))
        if traffic:
            f.write("\nHBM traffic = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 = %.4f GB per launch = %.3f x the algorithmic %.4f GB\n"
                    % (traffic / 1e9, traffic / algo, algo / 1e9))
        f.write("\n## derived\n\n")
        for k, v in issue.items():
            f.write("- %s: %s\n" % (k, v))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], a[3], *(int(x) for x in a[4:6]))

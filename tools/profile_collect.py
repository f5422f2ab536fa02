#!/usr/bin/env python3
"""Collect what tools/profile_round.sh measured: gpurun_out/pmc_traffic.json (per kernel and per whole step, stamped with the
kernel-source hash bench.py checks) and the SQ breakdowns.  Usage: python3 tools/profile_collect.py TAG workload..."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha, nothing else)

STEP_KERNELS = {"driving": ["drv_step_kernel"], "robocup": ["rc_step_kernel"],
                "driving_partial": ["drv_step_partial_kernel", "drv_partial_obs_deferred_kernel"],
                "robocup_partial": ["rc_step_partial_kernel", "rc_partial_obs_deferred_kernel", "rc_partial_finalize_kernel"],
                "hbm": ["arr_pad_cols_kernel", "obs_unpack_peers_rows_kernel"]}   # tools/hbm_kernels_run.py: the two HBM-bound kernels, separate programs of one run


def counters(pattern):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag, workloads = sys.argv[1], sys.argv[2:]
    path = "gpurun_out/pmc_traffic.json"
    out = {}
    if os.path.exists(os.path.join(ROOT, "profiles", "pmc_traffic.json")):
        old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if old.get("kernel_source_sha16") == bench.kernel_source_sha():
            out = old  # same kernels: keep the workloads not re-measured in this call
    out["note"] = ("bytes per launch, mean over the launches of whole episodes at 4096 envs; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in "
                   "separate passes, FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE, both in KB; "
                   "step_bytes_all_kernels adds the Partial paths' deferred / finalize launches; tools/profile_round.sh")
    out["kernel_source_sha16"] = bench.kernel_source_sha()
    out["tag"] = tag
    sq_path = "gpurun_out/sq_counters.json"
    sq = {}
    if os.path.exists(os.path.join(ROOT, "profiles", "sq_counters.json")):
        old_sq = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))
        if old_sq.get("kernel_source_sha16") == bench.kernel_source_sha():
            sq = old_sq
    sq["note"] = ("per launch of the workload's step kernel, mean over the launches of whole episodes at 4096 envs; rocprofv3 --pmc in two passes "
                  "(tools/profile_round.sh).  valu_busy = SQ_INSTS_VALU x 4 / (1024 SIMDs x launch duration x 2.4 GHz), the launch duration being the "
                  "AverageNs of the --kernel-trace --stats pass of the same round; wave_time_shares = SQ_ACTIVE_INST_* / SQ_WAIT_* over SQ_WAVE_CYCLES")
    sq["kernel_source_sha16"] = bench.kernel_source_sha()
    sq["tag"] = tag
    for w in workloads:
        f = counters("gpurun_out/%s_prof_%s_f/*/*counter_collection.csv" % (tag, w))
        wr = counters("gpurun_out/%s_prof_%s_w/*/*counter_collection.csv" % (tag, w))
        total = 0.0
        per = {}
        for k in STEP_KERNELS[w]:
            if k not in f or k not in wr:
                continue
            fv, wv = f[k]["FETCH_SIZE"], wr[k]["WRITE_SIZE"]
            fb, wb = sum(fv) / len(fv) * 1024 * 2, sum(wv) / len(wv) * 1024
            per[k] = {"fetch_bytes_x2": fb, "write_bytes": wb, "launches": len(fv), "workload": w}
            total += fb + wb
        if w == "hbm":   # two unrelated kernels: each its own total; their algorithmic bytes and kernel-stats durations beside the traffic
            try:
                alg = json.loads(open("gpurun_out/%s_bench_hbm_under_rocprof.json" % tag).read().strip().splitlines()[-1])
                rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open("gpurun_out/%s_kernel_stats_hbm.csv" % tag))}
            except (OSError, ValueError, IndexError):
                alg, rows = {}, {}
            for k, d in per.items():
                d["alg_bytes"] = alg.get(k, {}).get("alg_bytes")
                d["average_ns"] = rows.get(k)
                if d["alg_bytes"] and d["average_ns"]:
                    d["achieved_GBps"] = d["alg_bytes"] / d["average_ns"]
                    d["frac_of_8TBps"] = d["achieved_GBps"] / 8000.0
                    d["traffic_over_alg"] = (d["fetch_bytes_x2"] + d["write_bytes"]) / d["alg_bytes"]
                total = d["fetch_bytes_x2"] + d["write_bytes"]
                d["step_bytes_all_kernels"] = total
                out[k + "_bytes_per_launch"] = total
                out[k + "_detail"] = d
            continue
        for k, d in per.items():
            d["step_bytes_all_kernels"] = total
            out[k + "_bytes_per_launch"] = d["fetch_bytes_x2"] + d["write_bytes"]
            out[k + "_detail"] = d
        lines = []
        for p in ("s1", "s2"):
            c = counters("gpurun_out/%s_prof_%s_%s/*/*counter_collection.csv" % (tag, w, p))
            k = STEP_KERNELS[w][0]
            for name in sorted(c.get(k, {})):
                v = c[k][name]
                lines.append("%-22s n=%d mean=%.5g  first50=%.5g last50=%.5g" % (name, len(v), sum(v) / len(v), sum(v[:50]) / 50, sum(v[-50:]) / 50))
        if lines:
            m = {ln.split()[0]: float(ln.split("mean=")[1].split()[0]) for ln in lines}
            wc = m.get("SQ_WAVE_CYCLES")
            if wc:
                lines.append("")
                lines.append("shares of a wave's lifetime (quad-cycles of SQ_WAVE_CYCLES): executing an instruction %.1f %% (VALU %.1f %%, scalar %.1f %%, LDS %.1f %%), "
                             "parked in s_waitcnt / barrier (SQ_WAIT_ANY) %.1f %%, issue stalls (SQ_WAIT_INST_ANY) %.1f %%"
                             % tuple(100 * m.get(x, 0) / wc for x in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")))
                if "SQ_WAVES" in m and m["SQ_WAVES"]:
                    lines.append("per wave and step: VALU %.0f, SALU %.0f, LDS %.0f, branches %.0f, VMEM rd %.0f / wr %.0f instructions"
                                 % tuple(m.get(x, 0) / m["SQ_WAVES"] for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")))
                try:  # VALU issue-slot utilisation over the launch: a wave64 VALU instruction occupies its SIMD for >= 4 cycles
                    rows = list(csv.DictReader(open("gpurun_out/%s_kernel_stats_%s.csv" % (tag, w))))
                    avg_ns = [float(r["AverageNs"]) for r in rows if r["Name"].split("(")[0] == STEP_KERNELS[w][0]][0]
                    ws = m["SQ_WAVES"] if m.get("SQ_WAVES") else None
                    sq[STEP_KERNELS[w][0]] = {
                        "workload": w, "launch_us": avg_ns / 1e3, "valu_busy": m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * avg_ns * 2.4),
                        "wave_time_shares": {"executing": m.get("SQ_ACTIVE_INST_ANY", 0) / wc, "valu": m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                                             "scalar": m.get("SQ_ACTIVE_INST_SCA", 0) / wc, "lds": m.get("SQ_ACTIVE_INST_LDS", 0) / wc,
                                             "waitcnt": m.get("SQ_WAIT_ANY", 0) / wc, "issue_stall": m.get("SQ_WAIT_INST_ANY", 0) / wc},
                        "insts_per_wave": (None if ws is None else {k_: m.get(c_, 0) / ws for k_, c_ in (
                            ("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS"), ("branch", "SQ_INSTS_BRANCH"),
                            ("vmem_rd", "SQ_INSTS_VMEM_RD"), ("vmem_wr", "SQ_INSTS_VMEM_WR"))})}
                    lines.append("SIMD VALU busy >= %.1f %% of the launch (SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x %.1f us x 2.4 GHz); fp64 instructions hold the "
                                 "SIMD longer than 4 cycles, so this is a lower bound)" % (100 * m.get("SQ_INSTS_VALU", 0) * 4 / (1024 * avg_ns * 2.4), avg_ns / 1e3))
                except (OSError, IndexError, KeyError, ValueError):
                    pass
            open("gpurun_out/%s_sq_breakdown_%s.txt" % (tag, w), "w").write(
                "%s, 4096 envs, one whole episode (kernel sources %s); rocprofv3 --pmc, per launch\n" % (STEP_KERNELS[w][0], bench.kernel_source_sha()) + "\n".join(lines) + "\n")
    json.dump(out, open(path, "w"), indent=1)
    json.dump(sq, open(sq_path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k.endswith("_bytes_per_launch") or k == "kernel_source_sha16"}, indent=1))


if __name__ == "__main__":
    main()

"""Rewrite profiles/pmc_traffic.json from the files tools/copy_profiles.sh put under profiles/ for one round:

    python tools/update_pmc_traffic.py r03

For every kernel bench.py prices a roofline on it stores, per shape, the rocprofv3 --kernel-trace --stats AverageNs and
the PMC bytes per launch (2 * FETCH_SIZE + WRITE_SIZE, KB -> bytes: gfx950 FETCH_SIZE counts a 16-B/lane streaming read
at one half, MI355X_MICROARCH.md "HBM"), each with the file it came from.  bench.py reads this file; it never measures
PMC itself (counter passes need their own rocprofv3 runs)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = sys.argv[1] if len(sys.argv) > 1 else "r03"
D = os.path.join(ROOT, "profiles")


def stats(fn):
    out = {}
    path = os.path.join(D, fn)
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            out[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
    return out


def pmc(fn):
    out = {}
    path = os.path.join(D, fn)
    if not os.path.exists(path):
        return out
    with open(path) as f:
        for r in csv.DictReader(f):
            out.setdefault(r["kernel"], {})[r["counter"]] = float(r["avg_value"])
    return out


def find(d, sub):
    for k, v in d.items():
        if sub in k:
            return k, v
    return None, None


def entry(kernel_sub, stats_fn, pmc_fn, note=None):
    e = {}
    k, v = find(stats(stats_fn), kernel_sub)
    if k:
        e["rocprof_avg_ns"] = round(v[0], 1)
        e["rocprof_calls"] = v[1]
        e["rocprof_kernel"] = k[:100]
        e["source"] = f"profiles/{stats_fn}"
    k, c = find(pmc(pmc_fn), kernel_sub)
    if k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["fetch_size_kb"] = c["FETCH_SIZE"]
        e["write_size_kb"] = c["WRITE_SIZE"]
        e["traffic_bytes"] = int(round((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1000))
        e["source"] = (e.get("source", "") + " + " if e.get("source") else "") + f"profiles/{pmc_fn}"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
            e["mfma_busy"] = {"SQ_VALU_MFMA_BUSY_CYCLES": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                              "GRBM_GUI_ACTIVE_sum8xcd": c["GRBM_GUI_ACTIVE"], "simds": 1024,
                              "util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 3)}
    if note:
        e["note"] = note
    return e


kernels = {
    "k_level0_fwd": {"B20_N500": entry("k_level0_fwd", f"{P}_dd_kernel_stats.csv", f"{P}_dd_step_pmc_summary.csv",
                                       "in-step launches (bench.py --steps 50; PMC passes with --no-graph)")},
    "k_level0_bwd": {"B20_N500": entry("k_level0_bwd", f"{P}_dd_kernel_stats.csv", f"{P}_dd_step_pmc_summary.csv",
                                       "in-step launches (bench.py --steps 50; PMC passes with --no-graph)")},
    "k_aggregate_packed": {
        "B20_N500_C40": entry("k_aggregate<false, 3", f"{P}_dd_probe_only_kernel_stats.csv",
                              f"{P}_dd_probe_pmc_summary.csv", "bench.py --probe-only"),
        "B256_N1024_C40": entry("k_aggregate_wide<3>", f"{P}_er_probe_kernel_stats.csv",
                                f"{P}_er_probe_pmc_summary.csv", "bench.py --workload er --probe-only"),
    },
    "k_aggregate_wide_dma": {
        "B256_N1024_K256": entry("k_aggregate_wide_dma", f"{P}_er_probe_kernel_stats.csv",
                                 f"{P}_er_probe_pmc_summary.csv", "bench.py --workload er --probe-only"),
    },
}
doc = {"_comment": "written by tools/update_pmc_traffic.py from profiles/%s_* (rocprofv3 --kernel-trace --stats AverageNs; "
                   "PMC bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB; separate --pmc passes).  Read by bench.py: these "
                   "are figures of an earlier run of the same build, not of the run that prints them." % P,
       "round": P, "kernels": kernels}
with open(os.path.join(D, "pmc_traffic.json"), "w") as f:
    json.dump(doc, f, indent=1)
print(json.dumps(doc, indent=1))

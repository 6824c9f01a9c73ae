"""Turns the counter runs of tools/pmc_collect.sh into profiles/<tag>_pmc.json (read by bench.py) and a table in markdown.

    python tools/pmc_summarise.py gpurun_out/pmc profiles/r02_pmc.json profiles/r02_pmc_summary.md

The workload renders a first frame of one sample per pixel (after which the library regroups the tree by sampled rays) and
then the measured frame; launches dispatched before the end of the first frame (its k_finalize) are left out.
Per configuration and kernel (k_traverse, k_shade), summed over the launches of the measured frame:
  valu_busy        SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)   share of cycles a SIMD issues a vector instruction
  l2_hit_rate      TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  hbm_bytes        (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the counters are in KiB, and on gfx950 FETCH_SIZE tallies the 128-byte
                   requests of wide coalesced reads at 64 bytes (MI355X_MICROARCH.md, HBM section); separate passes
  hbm_gbps         hbm_bytes / the kernel's summed duration in the trace run
  valu_lane_slots_per_segment   SQ_INSTS_VALU * 64 / path segments
The json also names the sha256 of the kernel sources (rt_hip.hip, rt_core.h) of the tree the summary is made in -- run it in the
tree whose library was measured: bench.py marks the counters stale when its own sources differ.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict


def kernel_of(name):
    if "k_traverse" in name:
        return "k_traverse"
    if "k_shade" in name:
        return "k_shade"
    return None


def measured_rows(f):
    """The rows of a rocprofv3 csv whose dispatch comes after the first k_finalize launch (all rows if there is only one frame)."""
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    ends = sorted({int(r["Dispatch_Id"]) for r in rows if "k_finalize" in r["Kernel_Name"]})
    cut = ends[0] if len(ends) > 1 else -1
    return [r for r in rows if int(r["Dispatch_Id"]) > cut]


def read_counters(d):
    """{kernel: {counter: sum}} of a --pmc run's counter_collection csv."""
    out = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in measured_rows(f):
            k = kernel_of(row["Kernel_Name"])
            if k:
                out[k][row["Counter_Name"]] += float(row["Counter_Value"])
    return out


def read_durations(d):
    """{kernel: (launches, total ns)} of the trace run's kernel_trace csv."""
    out = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in measured_rows(f):
            k = kernel_of(row["Kernel_Name"])
            if k:
                out[k][0] += 1
                out[k][1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return out


def main(root, out_json, out_md):
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256()
    for rel in ("raytracing_amd/csrc/rt_hip.hip", "raytracing_amd/csrc/rt_core.h"):  # (bench.py compares this with the sources it runs on)
        with open(os.path.join(here, rel), "rb") as fh:
            sha.update(fh.read())
    doc = {"kernel_source_sha256": sha.hexdigest(),
           "source": "rocprofv3 --pmc, one run per counter group (SQ_*, TCC_HIT/MISS, FETCH_SIZE, WRITE_SIZE) plus a --kernel-trace run, on "
                     "tools/prof_workload.py <config> <spp> (RT_RENDER_NO_TAIL | RT_RENDER_ONE_STREAM: one k_traverse and one k_shade launch per bounce; "
                     "the launches of its first frame of 1 spp, after which the tree is regrouped by sampled rays, are left out)",
           "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (KiB counters; gfx950 FETCH_SIZE counts half of wide coalesced reads)",
           "configs": {}}
    lines = ["| config | kernel | launches | ms total | VALU busy | VALU lane slots / segment | L2 hit rate | HBM B / segment (algorithmic) | HBM GB/s | wait share |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    algo = {"k_traverse": 32, "k_shade": 103}
    for cfg in sorted(os.listdir(root)):
        wl = os.path.join(root, cfg, "workload.json")
        if not os.path.exists(wl):
            continue
        with open(wl) as fh:
            work = json.load(fh)
        seg = work["segments"]
        sq, tcc = read_counters(os.path.join(root, cfg, "sq")), read_counters(os.path.join(root, cfg, "tcc"))
        fe, wr = read_counters(os.path.join(root, cfg, "fetch")), read_counters(os.path.join(root, cfg, "write"))
        dur = read_durations(os.path.join(root, cfg, "stats"))
        rec = {"workload": work, "kernels": {}}
        fused = "k_traverse" not in sq   # a handful of leaves: k_shade walks and shades, and no hit record is written or read
        for k in ("k_traverse", "k_shade"):
            if k not in sq:
                continue
            if fused and k == "k_shade":
                # ray in (48 B; the first bounce makes its own camera rays) and ray out (48 B) for every segment but a path's last, which writes the sample's value (16 B)
                paths = work["width"] * work["height"] * work["spp"]
                algo_k = round((96.0 * max(seg - paths, 0) + 16.0 * min(paths, seg)) / seg, 1)
            else:
                algo_k = algo[k]
            s = sq[k]
            launches, ns = dur.get(k, [0, 0.0])
            hbm = (2.0 * fe[k].get("FETCH_SIZE", 0.0) + wr[k].get("WRITE_SIZE", 0.0)) * 1024.0
            hit, miss = tcc[k].get("TCC_HIT_sum", 0.0), tcc[k].get("TCC_MISS_sum", 0.0)
            r = {"launches": launches, "ms_total": round(ns / 1e6, 3),
                 "valu_busy": round(s["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (s["GRBM_GUI_ACTIVE"] / 8.0), 4) if s.get("GRBM_GUI_ACTIVE") else None,
                 "valu_lane_slots_per_segment": round(s["SQ_INSTS_VALU"] * 64.0 / seg, 1),
                 "salu_per_valu": round(s["SQ_INSTS_SALU"] / s["SQ_INSTS_VALU"], 3) if s.get("SQ_INSTS_VALU") else None,
                 "wait_share": round(s["SQ_WAIT_ANY"] / s["SQ_WAVE_CYCLES"], 4) if s.get("SQ_WAVE_CYCLES") else None,
                 "l2_hit_rate": round(hit / (hit + miss), 4) if hit + miss else None,
                 "hbm_bytes": hbm, "hbm_bytes_per_segment": round(hbm / seg, 2), "algorithmic_bytes_per_segment": algo_k, "walks_and_shades": bool(fused and k == "k_shade"),
                 "hbm_gbps": round(hbm / ns, 1) if ns else None,
                 "raw": {"sq": dict(s), "tcc": dict(tcc[k]), "FETCH_SIZE_KiB": fe[k].get("FETCH_SIZE"), "WRITE_SIZE_KiB": wr[k].get("WRITE_SIZE")}}
            rec["kernels"][k] = r
            tag = " (walk + shade)" if fused and k == "k_shade" else ""
            lines.append(f"| {cfg} ({work['width']}x{work['height']}, {work['spp']} spp, {seg / 1e6:.1f} M segments) | `{k}`{tag} | {launches} | {r['ms_total']} | "
                         f"{r['valu_busy']} | {r['valu_lane_slots_per_segment']} | {r['l2_hit_rate']} | {r['hbm_bytes_per_segment']} ({algo_k}) | {r['hbm_gbps']} | {r['wait_share']} |")
        doc["configs"][cfg] = rec
    with open(out_json, "w") as fh:
        json.dump(doc, fh, indent=1)
    with open(out_md, "w") as fh:
        fh.write("# Counters of the two wavefront kernels on the four GPU configurations (1x MI355X)\n\n" + doc["source"] + ".\n" + doc["correction"] +
                 ".\nVALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8); L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum); "
                 "HBM GB/s over the kernel's summed duration in the trace run; wait share = SQ_WAIT_ANY / SQ_WAVE_CYCLES.\n\n" + "\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(*sys.argv[1:4])

"""Summarise gpurun_out/prof_<tag> (tools/collect_profiles.sh) into profiles/: kernel stats CSV, a PMC summary per kernel
(HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE in KiB units as MI355X_MICROARCH.md prescribes for gfx950; SQ cycle
shares) and profiles/pmc_traffic.json, which bench.py reads to fill roofline.traffic.
usage: python tools/summarize_profiles.py <tag> <round-prefix>"""
import collections, csv, glob, json, os, shutil, sys

tag, prefix = sys.argv[1], sys.argv[2]
root = os.path.join("gpurun_out", f"prof_{tag}")
os.makedirs("profiles", exist_ok=True)
st = glob.glob(os.path.join(root, "stats", "*", "*_kernel_stats.csv"))
if st:
    shutil.copy(st[0], os.path.join("profiles", f"{prefix}_bench_kernel_stats.csv"))


def short(name):
    for key in ("inner_light3_kernel", "inner_light2_kernel", "inner_light_kernel", "bvh_trace_kernel", "flow_kernel", "sdf_kernel", "shape_shade_kernel", "point_prep_kernel",
                "cube_lookup_fwd_kernel", "shade_dirs_kernel", "shade_reduce_kernel", "compact_mask_kernel", "composite_fwd_kernel",
                "march_uniform_kernel", "vm_gather_kernel", "cube_filter_kernel"):
        if key in name:
            return key
    return None


def per_kernel(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


fetch, write, sq = per_kernel("fetch"), per_kernel("write"), per_kernel("sq")
summary, traffic = {}, {}
for k in sorted(set(fetch) | set(write) | set(sq)):
    e = {}
    if k in fetch and "FETCH_SIZE" in fetch[k]:
        v = fetch[k]["FETCH_SIZE"]
        e["FETCH_SIZE_KiB_avg"] = sum(v) / len(v)
        e["launches"] = len(v)
    if k in write and "WRITE_SIZE" in write[k]:
        v = write[k]["WRITE_SIZE"]
        e["WRITE_SIZE_KiB_avg"] = sum(v) / len(v)
    if "FETCH_SIZE_KiB_avg" in e and "WRITE_SIZE_KiB_avg" in e:
        # gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled (upper bound for narrow accesses)
        e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE_KiB_avg"] + e["WRITE_SIZE_KiB_avg"]) * 1024.0
        traffic[k] = dict(hbm_bytes_per_launch=e["hbm_bytes_per_launch"], fetch_kib=e["FETCH_SIZE_KiB_avg"],
                          write_kib=e["WRITE_SIZE_KiB_avg"], launches=e["launches"],
                          note="2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of `bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train`")
    if k in sq:
        c = {n: sum(v) / len(v) for n, v in sq[k].items()}
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            e["sq_share_of_wave_cycles"] = {n: round(c[n] / wc, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if n in c}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                e["mfma_busy_cycles_over_wave_quadcycles_x4"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * wc), 4)
        e["sq_raw_avg"] = {n: c[n] for n in sorted(c)}
    summary[k] = e
json.dump(summary, open(os.path.join("profiles", f"{prefix}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
json.dump(traffic, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "sq_raw_avg"} for k, v in summary.items()}, indent=1))

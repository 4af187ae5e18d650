"""Summarise a rocprofv3 --kernel-trace of tools/exp_cosched.py: for the three stage kernels, how much of each launch's duration another
stage kernel (on another stream) was running.  usage: python tools/trace_overlap.py <kernel_trace.csv> <out.json>"""
import csv, json, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    cls = "bvh" if "bvh_trace_kernel" in n else "inner" if "inner_light3_kernel" in n else "flow" if n.startswith("void flow_kernel") or "flow_kernel<" in n else None
    if cls is None:
        continue
    wg, grid = int(r["Workgroup_Size_X"]), int(r["Grid_Size_X"])
    blocks = grid // wg
    sig = {"bvh": f"bvh {blocks // 256}/CU", "inner": f"inner {'1 team' if wg == 256 else '2 teams'}", "flow": f"flow {wg // 64} waves"}[cls]
    rows.append(dict(cls=cls, sig=sig, s=int(r["Start_Timestamp"]), e=int(r["End_Timestamp"]), q=r["Queue_Id"]))
rows.sort(key=lambda x: x["s"])
groups = collections.defaultdict(list)
for i, a in enumerate(rows):
    ov = collections.Counter()
    for b in rows:
        if b is a or b["cls"] == a["cls"] or b["e"] <= a["s"] or b["s"] >= a["e"]:
            continue
        ov[b["sig"]] += min(a["e"], b["e"]) - max(a["s"], b["s"])
    dur = a["e"] - a["s"]
    partner = " + ".join(sorted(k for k, v in ov.items() if v > 0.05 * dur)) or "alone"
    tot = min(dur, sum(ov.values()))
    groups[(a["sig"], partner)].append((dur / 1e6, tot / dur))
out = []
for (sig, partner), v in sorted(groups.items()):
    if len(v) < 2:
        continue
    out.append(dict(kernel=sig, beside=partner, launches=len(v), ms_avg=round(sum(d for d, _ in v) / len(v), 3),
                    overlapped_fraction_avg=round(sum(f for _, f in v) / len(v), 3)))
    print(f"{sig:16s} beside {partner:34s}: {len(v):3d} launches, {out[-1]['ms_avg']:7.3f} ms each, {100 * out[-1]['overlapped_fraction_avg']:5.1f} % of the launch overlapped")
json.dump(dict(how="rocprofv3 --kernel-trace -- python3 tools/exp_cosched.py 65536 (REPS=3): start / end timestamps of the three stage kernels; a launch is 'beside' "
                   "the other-class launches that cover > 5 % of it; overlapped fraction = share of its duration another class's kernel was running on the device",
               groups=out), open(sys.argv[2], "w"), indent=1)

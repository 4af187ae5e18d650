#!/bin/bash
# Run ON THE GPU BOX: fabric bytes and wait share of bvh_trace_kernel without a bound and under the conservative 256^3 occupancy bound
# (tools/exp_bvh_tmax.py; verdict r5 item 2 asks for: rays retired by the pre-test, fabric bytes per traced ray, SQ_WAIT_ANY share).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export BVH_TMAX_ONLY=${1:-grid_256}
out=gpurun_out/prof_bvh_pretest
rm -rf $out; mkdir -p $out
BVH_TMAX_SAVE=/tmp/bvh_tmax_saved.pt python3 tools/exp_bvh_tmax.py build_variants/lib_bvhtmax.so 32768 > $out/unprofiled.log 2>&1
CMD="python3 tools/exp_bvh_tmax_prof.py build_variants/lib_bvhtmax.so /tmp/bvh_tmax_saved.pt"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $CMD > $out/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $CMD > $out/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/sq -- $CMD > $out/sq.log 2>&1
echo "sq done"
python3 - <<'PY'
import csv, glob, json, os
root = "gpurun_out/prof_bvh_pretest"
def rows(sub):
    out = {}
    for f in glob.glob(os.path.join(root, sub, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "bvh_trace_kernel" in r["Kernel_Name"]:
                out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}
f, w, s = rows("fetch"), rows("write"), rows("sq")
res = json.load(open("gpurun_out/bvh_tmax.json"))
traced = res["rays_traced"]
def arm(sl):
    m = lambda xs: sum(xs[sl]) / len(xs[sl])
    fetch_b, write_b = 2.0 * m(f["FETCH_SIZE"]) * 1024, m(w["WRITE_SIZE"]) * 1024     # gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md)
    return dict(fabric_bytes_per_traced_ray=(fetch_b + write_b) / traced, fetch_bytes_per_traced_ray=fetch_b / traced,
                sq_wait_any_share=m(s["SQ_WAIT_ANY"]) / m(s["SQ_WAVE_CYCLES"]), valu_active_share=m(s["SQ_ACTIVE_INST_VALU"]) / m(s["SQ_WAVE_CYCLES"]),
                valu_instructions_per_traced_ray=m(s["SQ_INSTS_VALU"]) * 64 / traced if "SQ_INSTS_VALU" in s else None)
n = len(f["FETCH_SIZE"])
print("bvh launches per pass:", n)
out = {"launch_order": "0..6 = no bound; 7..13 = the bounded variant (first launch of each arm dropped)", "no_bound": arm(slice(1, 7)), os.environ["BVH_TMAX_ONLY"]: arm(slice(8, 14))}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/bvh_pretest_pmc.json", "w"), indent=1)
PY

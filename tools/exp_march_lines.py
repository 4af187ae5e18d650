"""Dev (round 6, north_star "LDS staging of the line factors"): the full-frame ray-march of BASELINE configs[1] under a variant build of the
library.  With -DSDF_ABLATE_LINES the two line taps of every chunk are not fetched at all (results garbage): the time saved is the UPPER
bound of what serving the lines from LDS instead of L2 could return.   python tools/exp_march_lines.py <lib.so>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tensoflow_amd.lib as L

L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch

import bench

r = bench.march_probe(torch.device("cuda:0"), 3)
res = dict(lib=os.path.basename(sys.argv[1]), frame_ms=r["frame_ms"], sdf_alpha_ms_per_frame=r["sdf_alpha_ms_per_frame"],
           sdf_kernel_avg_launch_ms=r["sdf_kernel_avg_launch_ms"], live_samples_per_frame=r["live_samples_per_frame"],
           sdf_alpha_samples_per_s=r["sdf_alpha_samples_per_s"])
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/march_lines.jsonl", "a") as f:
    f.write(json.dumps(res) + "\n")

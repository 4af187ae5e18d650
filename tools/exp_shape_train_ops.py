"""Dev: which torch ops (and from where) make up the launches of a shape-stage training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
bench.shape_train_probe(dev, 2)          # warm
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    bench.shape_train_probe(dev, 1)
ka = prof.key_averages(group_by_stack_n=4)
rows = sorted(ka, key=lambda e: -e.count)
seen = 0
for e in rows:
    if not e.key.startswith("aten::"):
        continue
    st = [s for s in e.stack if "tensoflow_amd" in s or "bench.py" in s]
    print(f"{e.count:6d} {e.key:28s} {st[0] if st else ''}")
    seen += 1
    if seen > 45:
        break

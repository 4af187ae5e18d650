"""Dev: where the host syncs (aten::item / nonzero) of a shape-stage training step come from (python stacks)."""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
bench.shape_train_probe(dev, 2)
sites = collections.Counter()
def hook(name, orig):
    def f(*a, **k):
        st = [fr for fr in traceback.extract_stack()[:-1] if "tensoflow_amd" in fr.filename or fr.filename.endswith("bench.py")]
        if st:
            fr = st[-1]
            sites[(name, os.path.basename(fr.filename), fr.lineno, fr.line.strip()[:90])] += 1
        return orig(*a, **k)
    return f
T = torch.Tensor
for n in ("item", "__bool__", "__int__", "__float__", "nonzero", "tolist", "cpu", "__index__"):
    setattr(T, n, hook(n, getattr(T, n)))
_nz = torch.nonzero
torch.nonzero = hook("torch.nonzero", _nz)
bench.shape_train_probe(dev, 1)
for k, v in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(v, k)

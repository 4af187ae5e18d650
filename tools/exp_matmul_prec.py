import torch
torch.manual_seed(0)
a = torch.randn(900, 256, device="cuda"); b = torch.randn(256, 256, device="cuda")
ref = (a.double() @ b.double())
for name, fn in [("matmul", lambda: a @ b), ("linear", lambda: torch.nn.functional.linear(a, b.t().contiguous())), ("t-matmul", lambda: (a.t().contiguous().t() @ b))]:
    out = fn()
    print(name, float((out.double() - ref).abs().max() / ref.abs().max()))
print("flags", torch.backends.cuda.matmul.allow_tf32, torch.get_float32_matmul_precision(), torch.backends.cuda.preferred_blas_library())
g = torch.randn(900, 256, device="cuda")
dw = g.t() @ a
print("dW", float((dw.double() - g.double().t() @ a.double()).abs().max() / dw.abs().max()))

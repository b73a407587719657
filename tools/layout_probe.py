"""Do the prompt pass's library products run faster with the weight stored [K', N] (NN) instead of [N, K'] (TN)?"""
import torch
dev = torch.device("cuda:0")
OD = torch.float32
def t_us(fn, n=12, warm=3):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(i); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
rows = 960
for name, N, K in (("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008)):
    for mult in (3, 2, 1):
        K3 = K * mult
        x = torch.randn(rows, K3, device=dev).half()
        xT = x.t().contiguous()
        ws = [(torch.randn(N, K3, device=dev) / 64).half() for _ in range(3)]
        wTs = [w.t().contiguous() for w in ws]
        r = {}
        r["TN  x[r,K] . w[N,K]^T (current)"] = t_us(lambda i: torch.mm(x, ws[i % 3].t(), out_dtype=OD))
        r["NN  x[r,K] . wT[K,N]"] = t_us(lambda i: torch.mm(x, wTs[i % 3], out_dtype=OD))
        r["TN' xT[K,r]^T . w[N,K]^T"] = t_us(lambda i: torch.mm(xT.t(), ws[i % 3].t(), out_dtype=OD))
        r["NT  xT[K,r]^T . wT[K,N]"] = t_us(lambda i: torch.mm(xT.t(), wTs[i % 3], out_dtype=OD))
        r["out^T  w[N,K] . x[r,K]^T"] = t_us(lambda i: torch.mm(ws[i % 3], x.t(), out_dtype=OD))
        fl = 2.0 * rows * N * K3
        print(f"{name:8s} N={N:6d} K'={K3:6d} (x{mult}): " + "; ".join(f"{k}: {v:7.1f} us = {fl / v / 1e9:5.0f} TF" for k, v in r.items()), flush=True)
        del ws, wTs

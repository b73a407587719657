import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libmembench.so"))
lib.membench.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
out = torch.zeros(4, device=dev)
for (N, K) in [(4096, 4096), (22016, 4096), (4096, 11008)]:
    ncopy = max(2, int(800e6 / (N * K * 2)) + 1)
    ws = [torch.randint(0, 30000, (N, K), device=dev, dtype=torch.int16) for _ in range(ncopy)]
    for waves in (4, 8, 16):
        for pat in (0, 1, 2):
            st = torch.cuda.current_stream().cuda_stream
            def run(n):
                for i in range(n):
                    lib.membench(pat, waves, ws[i % ncopy].data_ptr(), out.data_ptr(), N, K, st)
            run(8); torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); run(16); e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / 16)
            print(f"N={N} K={K} waves={waves} pattern={pat}: {best*1e3:7.1f} us  {N*K*2/best/1e6:7.0f} GB/s", flush=True)

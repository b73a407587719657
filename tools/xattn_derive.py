"""Appends the derived figures to a cross-attention counter file written by tools/collect_profiles.sh
(profiles/rNN_xattn_pmc_n*.txt): duration, HBM traffic vs algorithmic bytes, dense-equivalent MFMA fraction, executed
matrix-pipe busy fraction.  python tools/xattn_derive.py <file> <N objects> [L=256]"""
import re
import sys

path, N = sys.argv[1], int(sys.argv[2])
L = int(sys.argv[3]) if len(sys.argv) > 3 else 256
txt = [ln for ln in open(path).read().splitlines() if not ln.startswith("# --- derived") and not ln.startswith("# derived:")]
val, dur = {}, {}
for ln in txt:
    m = re.match(r"(\w+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", ln)
    if m:
        val[m.group(1)] = float(m.group(3))
    m = re.match(r"void (cross_attn_\w+)<.*?,(\d+),([\d.]+),([\d.]+),", ln)
    if m:
        dur[m.group(1)] = float(m.group(4))
P = N * N
t = dur["cross_attn_dma_kernel"]
alg = 2 * P * 33 * 768 * 2                                  # Q in + context out, bf16
rd, wr = 2 * val["FETCH_SIZE"] * 1024, val["WRITE_SIZE"] * 1024   # gfx950: FETCH_SIZE counts 2 KB units
flop = 4 * P * 33 * L * 768
out = txt + [
    "# --- derived (kernel-trace durations of the same passes; SQ_* records are per shader engine = 8 CUs) ---",
    f"# derived: cross_attn_dma_kernel {t:.1f} us per launch"
    + (f" (first-generation cross_attn_mfma_kernel on the same inputs: {dur['cross_attn_mfma_kernel']:.1f} us)"
       if "cross_attn_mfma_kernel" in dur else ""),
    f"# derived: HBM traffic: read 2*FETCH_SIZE = {rd / 1e6:.1f} MB, written {wr / 1e6:.1f} MB; algorithmic Q in + context out = "
    f"{alg / 1e6:.1f} MB -> ratio {(rd + wr) / alg:.3f}; {(rd + wr) / t / 1e6:.2f} TB/s = {(rd + wr) / t / 1e6 / 8:.3f} of the 8 TB/s peak",
    f"# derived: dense-equivalent contraction 4*P*33*L*768 = {flop / 1e9:.1f} GFLOP -> {flop / t / 1e6:.0f} TFLOP/s = "
    f"{flop / t / 1e6 / 2500:.3f} of the 2.5 PFLOP/s dense bf16 MFMA peak",
    f"# derived: executed matrix work: SQ_VALU_MFMA_BUSY_CYCLES / (32 SIMDs * GRBM_GUI_ACTIVE) = "
    f"{100 * val['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * val['GRBM_GUI_ACTIVE']):.1f} % busy (key tiles nobody attends to are skipped - "
    "exact - so the executed MFMA work is a fraction of the dense-equivalent)",
    f"# derived: VALU : MFMA instructions = {val['SQ_INSTS_VALU'] / val['SQ_INSTS_MFMA']:.0f} : 1; VMEM instructions per record {val['SQ_INSTS_VMEM']:.0f}",
]
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out[-5:]))

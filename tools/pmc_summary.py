"""HBM traffic of the weight-streaming GEMM from two rocprofv3 PMC passes over
`python tools/bench_kernels.py skinny` (FETCH_SIZE and WRITE_SIZE collected separately):

    python tools/pmc_summary.py <fetch.db> <write.db> [out.json]

Bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half of a wide coalesced
streaming read (MI355X_MICROARCH.md, HBM section).  The benchmark runs the five Llama-2-7B shapes in a fixed
order, each as one uninterrupted run of launches, so runs of consecutive dispatches identify the shape."""
import json
import sqlite3
import sys

SHAPES = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008),
          ("lm_head", 32000, 4096)]


PREFIX = ["skinny_gemm"]                                   # kernel-name prefix of the runs (argv[5] overrides)


def runs(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select dispatch_id, name, counter_value from pmc_events where counter_name = ? "
                            "order by dispatch_id", (counter,)))
    out, cur_run = [], []
    for _, name, val in rows:
        if name.startswith("void " + PREFIX[0]) or name.startswith(PREFIX[0]):
            cur_run.append(val)
        elif cur_run:
            out.append(cur_run)
            cur_run = []
    if cur_run:
        out.append(cur_run)
    out = [r for r in out if len(r) >= 100]                 # the timed runs (224 launches each), not warm-ups
    assert len(out) == len(SHAPES), [len(r) for r in out]
    return [sum(r) / len(r) for r in out], rows[0][1] if rows else ""


def main(fetch_db, write_db, out=None, esz=2):
    """esz: bytes per weight element (2: the 16-bit kernel, `bench_kernels.py skinny`; 4: the fp32 kernel, `skinny32`)."""
    esz = int(esz)
    f, _ = runs(fetch_db, "FETCH_SIZE")
    w, _ = runs(write_db, "WRITE_SIZE")
    per, tot_a, tot_h = {}, 0.0, 0.0
    for (name, N, K), fk, wk in zip(SHAPES, f, w):
        alg = N * K * esz
        hbm = 2 * fk * 1024 + wk * 1024
        per[name] = dict(algorithmic_bytes=alg, fetch_size_kb=round(fk, 1), write_size_kb=round(wk, 1),
                         hbm_bytes=int(hbm), ratio=round(hbm / alg, 3))
        mult = 1 if name == "lm_head" else 32
        tot_a += alg * mult
        tot_h += hbm * mult
    res = dict(kernel=("batch_gemm_kernel<EF16, 2, 1, 1, PAIR> (psg_split_gemm_w16, psg_batch_gemm.hip; bench_kernels.py skinnysplit)"
                       if PREFIX[0] != "skinny_gemm" else
                       "skinny_gemm_dma_kernel<8, 1, 3, 2>" if esz == 2 else "skinny_gemm_f32_kernel (psg_gemm_f32.hip)"),
               method="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/bench_kernels.py "
                      + ("skinnysplit" if PREFIX[0] != "skinny_gemm" else "skinny" if esz == 2 else "skinny32") + " (M=20); hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 FETCH_SIZE correction per "
                      "MI355X_MICROARCH.md); summarised by tools/pmc_summary.py",
               per_shape=per, launches_per_decode_step=129, hbm_bytes_per_launch=int(tot_h / 129),
               algorithmic_bytes_per_launch=int(tot_a / 129), ratio=round(tot_h / tot_a, 3))
    # hashes of the kernel's sources at collection time: bench.py reports `traffic` only while they are unchanged
    import hashlib
    import os
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = (["openpsg_amd/csrc/psg_batch_gemm.hip"] if PREFIX[0] != "skinny_gemm" else
             ["openpsg_amd/csrc/psg_gemm.hip"] if esz == 2 else ["openpsg_amd/csrc/psg_gemm_f32.hip"]) + \
        ["openpsg_amd/csrc/psg_common.h"]
    res["kernel_sources"] = {f: hashlib.sha256(open(os.path.join(repo, f), "rb").read()).hexdigest()[:16] for f in files}
    text = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(text + "\n")
    return text


if __name__ == "__main__":
    if len(sys.argv) > 5:
        PREFIX[0] = sys.argv[5]
    print(main(*sys.argv[1:5]))

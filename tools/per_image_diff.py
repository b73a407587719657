"""Difference of two rocprofv3 kernel traces (rocpd sqlite) that differ by `extra` timed steps -> per-step kernel
table (calls and microseconds per step).  python tools/per_image_diff.py a.db b.db extra [out.csv]"""
import sqlite3
import sys


def totals(db):
    cur = sqlite3.connect(db).cursor()
    return {r[0]: (r[1], r[2]) for r in cur.execute("select name, count(*), sum(duration) from kernels group by name")}


def main(a, b, extra, out=None):
    ta, tb = totals(a), totals(b)
    rows = []
    for name, (cb, db_) in tb.items():
        ca, da = ta.get(name, (0, 0))
        if cb == ca:
            continue
        rows.append((name, (cb - ca) / extra, (db_ - da) / extra / 1e3))
    rows.sort(key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    lines = ["kernel,calls_per_step,us_per_step,avg_us,pct"]
    for name, c, us in rows:
        lines.append(f"{name.replace(',', ';')[:110]},{c:.2f},{us:.1f},{us / c if c else 0:.2f},{100 * us / tot:.2f}")
    lines.append(f"TOTAL,{sum(r[1] for r in rows):.1f},{tot:.1f},,100")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    return text


if __name__ == "__main__":
    print(main(sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else None))

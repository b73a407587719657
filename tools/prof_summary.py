"""Turns a rocprofv3 (rocpd sqlite) kernel trace into the per-kernel summary kept under profiles/."""
import sqlite3
import sys


def main(db, out=None, top=40):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc"))
    total = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,pct"]
    for r in rows[:top]:
        name = r[0].replace(",", ";")[:110]
        lines.append(f"{name},{r[1]},{r[2] / 1e3:.1f},{r[3] / 1e3:.2f},{r[4] / 1e3:.2f},{r[5] / 1e3:.2f},"
                     f"{100.0 * r[2] / total:.2f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    return text


if __name__ == "__main__":
    print(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None))

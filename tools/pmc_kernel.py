"""Average PMC counter values per dispatch of the kernels whose name contains a substring:
    python tools/pmc_kernel.py <substring> <rocprofv3 results.db> [...more dbs]"""
import sqlite3
import sys


def main(sub, dbs):
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select counter_name, count(*), avg(counter_value) from pmc_events where name like ? "
                           "group by counter_name", (f"%{sub}%",))
        for name, n, avg in rows:
            print(f"{name:36s} n={n:5d} avg={avg:16.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])

"""Summarises a rocprofv3 --kernel-trace results db: python tools/prof_kernels.py <dir> [x <substring for per-dispatch list>]"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, max(vgpr_count), max(lds_size), max(scratch_size) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
for r in rows:
    print(f"{r[0][:70]:70s} n={r[1]:5d} total={r[2]:9.2f} ms ({100 * r[2] / tot:5.1f} %) avg={r[3]:9.1f} us vgpr={r[4]} lds={r[5]} scratch={r[6]}")
if len(sys.argv) > 3:
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    print([round((e - s) / 1e3) for n, s, e in rows if sys.argv[3] in n][:64])

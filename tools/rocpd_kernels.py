import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
rows=c.execute("select name, grid_x, count(*), avg(end-start) from kernels group by name, grid_x order by min(start)").fetchall()
for r in rows: print(f"{r[2]:5d} {r[3]/1e3:8.1f}us grid={r[1]:9d} {r[0][:120]}")

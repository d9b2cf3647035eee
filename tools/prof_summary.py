"""Per-kernel summary (calls, total/avg/min us, share) from a rocprofv3 rocpd results database."""
import sqlite3, sys
db, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"Name,Calls,TotalDurationUs,AverageUs,MinUs,Percentage,UsPerStep(steps={steps:g})")
for r in rows:
    print(f"\"{r[0][:110]}\",{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]:.2f},{100*r[2]/tot:.2f},{r[2]/steps:.1f}")
print(f"\"TOTAL\",{sum(r[1] for r in rows)},{tot:.1f},,,100,{tot/steps:.1f}")

"""Print HIP API call statistics from a rocprofv3 --hip-trace results database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
rg = [t for t in tabs if t.startswith('rocpd_region')][0]
st = [t for t in tabs if t.startswith('rocpd_string')][0]
rows = list(cur.execute(f"select s.string, count(*), sum(r.end-r.start), avg(r.end-r.start) from {rg} r join {st} s on r.name_id = s.id group by s.string order by 3 desc limit 18"))
for r in rows:
    print(f"{r[0][:44]:44s} n={r[1]:7d} total={r[2]/1e6:9.2f}ms avg={r[3]/1e3:8.2f}us")

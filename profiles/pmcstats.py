"""Average PMC counter values per kernel from a rocprofv3 --pmc results database."""
import sqlite3, sys, json
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
pe, pi, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
# sum over counter instances per dispatch, then average over dispatches
q = f"""select s.kernel_name, p.name, d.id, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id
join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name, d.id"""
acc = {}
for name, pmc, did, v in cur.execute(q):
    acc.setdefault((name, pmc), []).append(v)
out = {}
for (name, pmc), vs in acc.items():
    if any(k in name for k in sys.argv[2:]):
        out.setdefault(name.split('(')[0][:40], {})[pmc] = {'avg': sum(vs) / len(vs), 'n': len(vs)}
print(json.dumps(out, indent=1))

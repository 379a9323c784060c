"""Summarise a rocprofv3 --kernel-trace results database (sqlite) into a markdown table.
usage: python profiles/summarize.py <results.db> <out.md> "<title>" "<command>" """
import sqlite3
import sys


def main(dbp, out, title, cmd, extra=''):
    db = sqlite3.connect(dbp)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = list(cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    with open(out, 'w') as f:
        f.write(f"# {title}\n\nCommand: `{cmd}`\n\n{extra}\n\nTotal kernel time {tot / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches.\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows[:36]:
            f.write(f"| `{r[0][:90]}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | {100 * r[2] / tot:.1f} |\n")
    for r in rows[:22]:
        print(f"{r[0][:60]:60s} n={r[1]:5d} total={r[2]/1e6:8.3f}ms avg={r[3]/1e3:8.2f}us {100*r[2]/tot:5.1f}%")
    print('total ms', tot / 1e6)


if __name__ == '__main__':
    main(*sys.argv[1:])

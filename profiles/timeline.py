"""Analyse one bench step from a rocprofv3 --kernel-trace database: union busy time, overlap, gaps."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = list(cur.execute(f"select d.start, d.end, s.kernel_name, d.stream_id, d.queue_id from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
# find step boundaries: k_embed_fwd marks the start of a forward
starts = [i for i, r in enumerate(rows) if 'k_embed_fwd' in r[2]]
print('steps found', len(starts))
# the last complete TRAINING step: an interval between two forward starts that contains a backward kernel (the bench's
# forward-only inference passes and secondary workloads, if any ran, are skipped)
cands = [(a, b) for a, b in zip(starts[:-1], starts[1:]) if any('k_edge_bwd' in r[2] for r in rows[a:b])]
i0, i1 = cands[-2] if len(cands) > 1 else cands[-1]
step = rows[i0:i1]
t0, t1 = step[0][0], max(r[1] for r in step)
print(f'step wall {(t1-t0)/1e3:.1f} us, kernels {len(step)}, sum durations {sum(r[1]-r[0] for r in step)/1e3:.1f} us')
# union
iv = sorted((r[0], r[1]) for r in step); busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
print(f'union busy {busy/1e3:.1f} us, idle {(t1-t0-busy)/1e3:.1f} us')
qs = {}
for r in step: qs.setdefault(r[4], []).append(r)
for q, rs in qs.items(): print('queue', q, 'kernels', len(rs), 'sum', sum(r[1]-r[0] for r in rs)/1e3)
# per-kernel: gap to previous end (on the global timeline)
prev_end = step[0][0]; gaps = []
for r in step:
    gaps.append((r[0] - prev_end, r[2][:40])); prev_end = max(prev_end, r[1])
gaps.sort(reverse=True)
print('largest gaps (us):', [(round(g/1e3, 1), n) for g, n in gaps[:12]])
pos = [g for g, n in gaps if g > 0]
print(f'positive gaps: n={len(pos)} total={sum(pos)/1e3:.1f} us, mean={sum(pos)/max(1,len(pos))/1e3:.2f} us')
if len(sys.argv) > 2:
    for r in step[:int(sys.argv[2])]: print(f'{(r[0]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f} q{r[4]} {r[2][:50]}')
if len(sys.argv) > 3:
    pat = sys.argv[3]
    print(pat, [round((r[1]-r[0])/1e3, 1) for r in step if pat in r[2]])

"""The per-workload tables of DESIGN.md section 7 from the committed evidence of one tag:
  profiles/<TAG>_kernels_<W>.md        rocprofv3 --kernel-trace --stats table of the workload alone (30 steps: 5 warm-up + 20 timed + capture)
  profiles/<TAG>_issue_floor.json      SQ-counter issue floors (profiles/issue_floor.py)
  profiles/<TAG>_traffic.json          FETCH_SIZE / WRITE_SIZE per launch (profiles/merge_pmc.py)
  profiles/<TAG>_pmc_mfma.json         MfmaUtil
usage: python profiles/design_tables.py TAG W[,W..]   ->  markdown on stdout"""
import json
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wls = sys.argv[1], sys.argv[2].split(',')
STEPS = 30.0
FAM = ('k_edge_attn_fwd', 'k_edge_fwd', 'k_edge_bwd', 'k_attn_fwd', 'k_attn_bwd_kvds', 'k_attn_bwd_qds', 'k_attn_bwd_gather',
       'k_attn_bwd80_gather', 'k_attn_bwd', 'k_rowres80', 'k_rowres', 'k_rowchain', 'k_linear', 'k_atb_reduce', 'k_atb', 'k_node_gather',
       'k_keypoint_bwd', 'k_keypoint', 'k_head_u_bwd', 'k_head_u', 'k_kabsch_fwd', 'k_kabsch_bwd', 'k_scalar_loss', 'k_embed',
       'k_reduce_segments', 'k_seg', 'k_qmean')


def fam_of(name):
    for f in FAM:
        if f in name:
            return f
    return None


def jload(p):
    try:
        return json.load(open(os.path.join(root, 'profiles', p)))
    except Exception:
        return {}


floors, traffic, mfma = jload(f'{tag}_issue_floor.json'), jload(f'{tag}_traffic.json'), jload(f'{tag}_pmc_mfma.json')
for w in wls:
    rows = {}
    total = 0.0
    for line in open(os.path.join(root, 'profiles', f'{tag}_kernels_{w}.md')):
        m = re.match(r'\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', line)
        if not m:
            continue
        ms = float(m.group(3))
        if '__amd_rocclr' in m.group(1) or m.group(1).startswith('_ZN2at'):
            continue
        total += ms
        f = fam_of(m.group(1)) or m.group(1)[:24]
        r = rows.setdefault(f, [0, 0.0])
        r[0] += int(m.group(2))
        r[1] += ms
    print(f'**{w}** (`profiles/{tag}_kernels_{w}.md`: {total / STEPS * 1e3:.0f} us of library kernels per step)\n')
    print('| kernel family | launches / step | us / launch | us / step | share | issue floor: VALU / LDS / MFMA us (floor / launch) | waves waiting | MfmaUtil | HBM bytes per launch (2 FETCH + WRITE) |')
    print('|---|---|---|---|---|---|---|---|---|')
    fl, tr, mf = floors.get(w, {}), traffic.get(w, {}), mfma.get(w, {})
    for f, (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        if ms / total < 0.008:
            continue
        e = fl.get(f)
        fcol = f"{e['valu_us']:.1f} / {e['lds_us']:.1f} / {e['mfma_us']:.1f} ({e['frac_of_floor']:.2f})" if e else ''
        wcol = f"{100 * e['waves_waiting_frac']:.0f} %" if e and e.get('waves_waiting_frac') is not None else ''
        t = tr.get(f)
        tcol = f"{(2 * t['FETCH_SIZE_KB'] + t['WRITE_SIZE_KB']) / 1024:.1f} MB" if t else ''
        mu = [v['MfmaUtil'] for k, v in mf.items() if f in k and fam_of(k) == f and v.get('MfmaUtil') is not None]
        mcol = f"{max(mu):.1f} %" if mu else ''
        print(f"| `{f}` | {n / STEPS:.1f} | {ms / n * 1e3:.1f} | {ms / STEPS * 1e3:.0f} | {100 * ms / total:.1f} % | {fcol} | {wcol} | {mcol} | {tcol} |")
    print()

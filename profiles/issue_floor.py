"""Issue floors per kernel family from the SQ counter passes (profiles/measure_r05.sh sq:<W>): the time a launch would take if
its busiest issue pipe never waited.
  valu_us  = SQ_ACTIVE_INST_VALU (quad-cycles a SIMD's vector ALU is busy, summed over all SIMDs) x 4 / (1024 SIMDs x f)
  lds_us   = SQ_ACTIVE_INST_LDS  x 4 / (1024 x f)         (the issue side of the LDS pipe)
  mfma_us  = MfmaUtil (%) x the launch's duration          (matrix pipe busy time, from the MfmaUtil pass)
  floor_us = max of the three; frac_of_floor = floor_us / duration (1 = the launch runs at its busiest pipe's rate)
f = 2.4 GHz (MI355X peak engine clock; the counters are in quad-cycles of the SIMD clock).  Durations: the rocprofv3
--kernel-trace table of the same workload (profiles/<tag>_kernels_<W>.md), dispatch-weighted over a family's instances.
usage: python profiles/issue_floor.py SQ_TAG KERNELS_TAG[,..] W[,W..]  ->  profiles/<SQ_TAG>_issue_floor.json"""
import datetime, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sq_tag, ktags, wls = sys.argv[1], sys.argv[2].split(','), sys.argv[3].split(',')
src = os.path.join(root, 'gpurun_out')
FAM = ('k_edge_attn_fwd', 'k_edge_fwd', 'k_edge_bwd', 'k_attn_fwd', 'k_attn_bwd_kvds', 'k_attn_bwd_qds', 'k_attn_bwd_gather', 'k_attn_bwd',
       'k_rowres', 'k_rowchain', 'k_linear', 'k_atb_reduce', 'k_atb')
F_GHZ, SIMDS = 2.4, 1024.0


def fam_of(name):
    for f in FAM:
        if f in name:
            return f
    return None


def jload(p):
    txt = open(p).read()
    return json.loads(txt[txt.index('{'):])


out = {'_comment': __doc__, '_collected': datetime.datetime.now().isoformat(timespec='seconds')}
for w in wls:
    p1, p2 = jload(os.path.join(src, f'{sq_tag}_sq_{w}_pass1.json')), jload(os.path.join(src, f'{sq_tag}_sq_{w}_pass2.json'))
    mu = {}
    for cand in (os.path.join(src, f'{sq_tag}_pmc_{w}_MfmaUtil.json'),):
        if os.path.exists(cand):
            mu = jload(cand)
    # durations per mangled-name prefix from the kernel table(s)
    dur = {}
    for kt in ktags:
        md = os.path.join(root, 'profiles', f'{kt}_kernels_{w}.md')
        if not os.path.exists(md):
            md = os.path.join(src, f'{kt}_kernels_{w}.md')
        if not os.path.exists(md):
            continue
        for line in open(md):
            m = re.match(r'\| `([^`]+)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', line)
            if m:
                dur[m.group(1)[:40]] = (int(m.group(2)), float(m.group(4)))
        break
    acc = {}
    for name, v in p1.items():
        f = fam_of(name)
        if not f or 'SQ_ACTIVE_INST_VALU' not in v:
            continue
        n = v['SQ_ACTIVE_INST_VALU']['n']
        d_us = dur.get(name[:40], (0, None))[1]
        if d_us is None:
            continue
        g = lambda dd, k: dd.get(k, {}).get('avg', 0.0)      # noqa: E731
        a = acc.setdefault(f, dict(n=0, valu=0.0, lds=0.0, mfma=0.0, dur=0.0, wave=0.0, wait=0.0, nvalu=0.0))
        a['n'] += n
        a['valu'] += n * g(v, 'SQ_ACTIVE_INST_VALU') * 4 / (SIMDS * F_GHZ * 1e3)
        a['lds'] += n * g(v, 'SQ_ACTIVE_INST_LDS') * 4 / (SIMDS * F_GHZ * 1e3)
        a['mfma'] += n * mu.get(name, {}).get('MfmaUtil', {}).get('avg', 0.0) / 100.0 * d_us
        a['dur'] += n * d_us
        a['wave'] += n * g(v, 'SQ_WAVE_CYCLES')
        a['wait'] += n * g(v, 'SQ_WAIT_ANY')
        a['nvalu'] += n * g(p2.get(name, {}), 'SQ_INSTS_VALU')
    res = {}
    for f, a in acc.items():
        n = a['n']
        valu, lds, mfma, d_us = a['valu'] / n, a['lds'] / n, a['mfma'] / n, a['dur'] / n
        floor = max(valu, lds, mfma)
        res[f] = {'avg_launch_us': round(d_us, 2), 'valu_us': round(valu, 2), 'lds_us': round(lds, 2), 'mfma_us': round(mfma, 2),
                  'floor_us': round(floor, 2), 'binding_pipe': 'valu' if floor == valu else ('lds' if floor == lds else 'mfma'),
                  'frac_of_floor': round(floor / d_us, 4) if d_us else None,
                  'waves_waiting_frac': round(a['wait'] / a['wave'], 3) if a['wave'] else None,
                  'valu_insts_per_launch': int(a['nvalu'] / n)}
    out[w] = res
try:      # the kernel-source digest the measurement script wrote on the GPU box (see merge_pmc.measured_digest)
    out['_csrc_digest'] = open(os.path.join(src, f'{sq_tag}_csrc_digest.txt')).read().strip()
except Exception:
    out['_csrc_digest'] = None
dst = os.path.join(root, 'profiles', f'{sq_tag}_issue_floor.json')
json.dump(out, open(dst, 'w'), indent=1)
print(dst)
for w in wls:
    for f, r in sorted(out[w].items(), key=lambda kv: -kv[1]['avg_launch_us']):
        print(w, f, r)

"""Merge the per-counter JSONs written by `profiles/measure_r02.sh TAG pmc` (rocprofv3 --kernel-trace --pmc <counter>, one
counter per pass, averaged per kernel by profiles/pmcstats.py) into the two summaries bench.py reads:
  profiles/<TAG>_pmc_mfma.json   per workload and kernel: MfmaUtil (%), executed MFMA FLOPs per launch
  profiles/<TAG>_traffic.json    per workload and kernel family: FETCH_SIZE_KB, WRITE_SIZE_KB per launch
usage: python profiles/merge_pmc.py TAG [dir=gpurun_out] [workloads, comma separated: B,C,E,C_bf16,R ...]
(round 3: the per-counter files of different gpurun calls may carry different tags: `TAG` may be a comma-separated list, the
first tag names the output; bf16 workloads count MFMA ops with SQ_INSTS_VALU_MFMA_MOPS_BF16; a `_collected` stamp lets
bench.py pick the newest summary)"""
import datetime, json, os, sys

tags = sys.argv[1].split(',')
tag = tags[0]
src = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAM = ('k_edge_attn_fwd', 'k_edge_fwd', 'k_edge_bwd', 'k_attn_fwd', 'k_attn_bwd_kvds', 'k_attn_bwd_qds', 'k_attn_bwd_gather', 'k_attn_bwd', 'k_rowres', 'k_rowwave', 'k_rowchain', 'k_linear', 'k_atb_reduce', 'k_atb',
       'k_node_gather')


def load(w, cnt):
    for tg in tags:
        f = os.path.join(root, src, f'{tg}_pmc_{w}_{cnt}.json')
        try:
            txt = open(f).read()
            d = json.loads(txt[txt.index('{'):])
            when[w] = max(when.get(w, 0.0), os.path.getmtime(f))
            return d
        except Exception:
            continue
    return {}


when = {}      # workload -> newest mtime of the per-counter files it was merged from (= when that gpurun call came back)


def git_head():
    """HEAD of the tree the counters were collected on (the gpurun snapshot is the working tree: commit before measuring);
    '+dirty' when kernel sources differ from HEAD at merge time."""
    import subprocess
    if os.environ.get('EQD_PMC_HEAD'):      # (merging later than measuring: name the measured commit explicitly)
        return os.environ['EQD_PMC_HEAD']
    try:
        h = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=root, capture_output=True, text=True).stdout.strip()
        d = subprocess.run(['git', 'status', '--porcelain', '--', 'equidock_public_amd', 'bench.py'], cwd=root,
                           capture_output=True, text=True).stdout.strip()
        return h + ('+dirty' if d else '')
    except Exception:
        return None


mfma = {'_comment': "rocprofv3 --kernel-trace --pmc <counter> -- python bench.py --eager --workload W --steps 3 --warmup 1 (one counter per "
                    "pass, profiles/measure_r02.sh pmc), averaged over the dispatches of each kernel.  MfmaUtil in %, measured while the "
                    "counters serialise the kernels; executed MFMA FLOP = SQ_INSTS_VALU_MFMA_MOPS_F32 x 512."}
traffic = {'_comment': "FETCH_SIZE / WRITE_SIZE in KB per launch from separate rocprofv3 --pmc passes (summed over counter instances, "
                       "averaged over dispatches, dispatch-weighted over the template variants of a kernel family).  bench.py reports "
                       "traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE counts 128-B requests as 64 B on gfx950 "
                       "(MI355X_MICROARCH.md, HBM section) - an upper bound on the fetch side for mixed access sizes."}
stamp = datetime.datetime.now().isoformat(timespec='seconds')
mfma['_collected'] = traffic['_collected'] = stamp
for w in (sys.argv[3].split(',') if len(sys.argv) > 3 else ('B', 'C', 'E')):
    mu = load(w, 'MfmaUtil')
    mops = load(w, 'SQ_INSTS_VALU_MFMA_MOPS_F32')
    mops16 = load(w, 'SQ_INSTS_VALU_MFMA_MOPS_BF16')
    if mu:
        mfma[w] = {}
        for k, v in mu.items():
            e = {'MfmaUtil': round(v['MfmaUtil']['avg'], 1) if 'MfmaUtil' in v else None}
            mo = mops.get(k, {}).get('SQ_INSTS_VALU_MFMA_MOPS_F32')
            m16 = mops16.get(k, {}).get('SQ_INSTS_VALU_MFMA_MOPS_BF16')
            gf = 0.0
            if mo:
                e['SQ_INSTS_VALU_MFMA_MOPS_F32'] = mo['avg']
                gf += mo['avg'] * 512 / 1e9
            if m16:
                e['SQ_INSTS_VALU_MFMA_MOPS_BF16'] = m16['avg']
                gf += m16['avg'] * 512 / 1e9
            if mo or m16:
                e['executed_mfma_gflop_per_launch'] = round(gf, 3)
            mfma[w][k] = e
    fe, wr = load(w, 'FETCH_SIZE'), load(w, 'WRITE_SIZE')
    if fe:
        traffic[w] = {}
        for fam in FAM:
            tot = {'FETCH_SIZE': [0.0, 0], 'WRITE_SIZE': [0.0, 0]}
            for cnt, d in (('FETCH_SIZE', fe), ('WRITE_SIZE', wr)):
                for k, v in d.items():
                    owner = next((f for f in FAM if f in k), None)      # first family whose name is contained (longest first)
                    if owner == fam and cnt in v:
                        tot[cnt][0] += v[cnt]['avg'] * v[cnt]['n']
                        tot[cnt][1] += v[cnt]['n']
            if tot['FETCH_SIZE'][1]:
                traffic[w][fam] = {'FETCH_SIZE_KB': round(tot['FETCH_SIZE'][0] / tot['FETCH_SIZE'][1], 1),
                                   'WRITE_SIZE_KB': round(tot['WRITE_SIZE'][0] / max(1, tot['WRITE_SIZE'][1]), 1),
                                   'dispatches': tot['FETCH_SIZE'][1]}
# every workload carries its own stamp: nothing is carried over from an older summary, and a reader can tell which
# code state each entry was measured on
head = git_head()


def measured_digest():
    """the kernel-source digest written ON THE GPU BOX by the measurement script (profiles/measure_r06.sh:
    <TAG>_csrc_digest.txt = equidock_public_amd.build.csrc_digest() of the snapshot the counters ran on)"""
    for tg in tags:
        try:
            return open(os.path.join(root, src, f'{tg}_csrc_digest.txt')).read().strip()
        except Exception:
            continue
    return None


digest = measured_digest()
stamps = {w: {'collected': datetime.datetime.fromtimestamp(t).isoformat(timespec='seconds'), 'git_head': head,
              'csrc_digest': digest}
          for w, t in when.items() if w in mfma or w in traffic}
mfma['_workloads'] = traffic['_workloads'] = stamps
json.dump(mfma, open(os.path.join(root, 'profiles', f'{tag}_pmc_mfma.json'), 'w'), indent=1)
json.dump(traffic, open(os.path.join(root, 'profiles', f'{tag}_traffic.json'), 'w'), indent=1)
print('written', [k for k in mfma if not k.startswith('_')], [k for k in traffic if not k.startswith('_')])

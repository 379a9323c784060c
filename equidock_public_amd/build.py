"""Build libequidock_hip.so (gfx950) in-tree with hipcc.  No GPU is needed to compile."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libequidock_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'equidock_hip.h')]
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    build_host(force, verbose)
    return LIB


def csrc_digest():
    """16 hex digits over the contents of every kernel source (csrc/*.hip, csrc/*.h) and the C ABI header: the identity of
    the code a counter summary under profiles/ was collected on.  bench.py compares it with the sources it runs on and
    flags counters of another code state as stale (there is no .git on the GPU box, so a commit hash cannot do that)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h')))
    files.append(os.path.join(os.path.dirname(HERE), 'include', 'equidock_hip.h'))
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0')
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


HOST_LIB = os.path.join(HERE, 'libequidock_host.so')


def build_host(force=False, verbose=True):
    """libequidock_host.so: host-only helpers for DataLoader workers (csrc_host/, plain g++, no HIP runtime)."""
    srcs = sorted(glob.glob(os.path.join(HERE, 'csrc_host', '*.cpp')))
    if force or _stale(HOST_LIB, srcs):
        cmd = [os.environ.get('CXX', 'g++'), '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-o', HOST_LIB] + srcs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed:\n{r.stdout}\n{r.stderr}")
    return HOST_LIB


if __name__ == '__main__':
    if '--digest' in sys.argv:
        print(csrc_digest())
    else:
        print(build(force='--force' in sys.argv))

"""TEST INFRASTRUCTURE: compile the product's kernel sources for x86 against the host
simulator (tests/hostsim/hip/hip_runtime.h) -> tests/hostsim/build/libeqd_hostsim.so."""
import fcntl
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'equidock_public_amd', 'csrc')
OUT = os.path.join(HERE, 'build')
LIB = os.path.join(OUT, 'libeqd_hostsim.so')
CXX = '/opt/rocm/lib/llvm/bin/clang++'
FLAGS = ['-O2', '-g', '-std=c++17', '-fPIC', '-ffp-contract=off', '-I', HERE, '-Wno-unknown-attributes',
         '-Wno-ignored-attributes', '-Wno-unused-function', '-Wno-unused-variable']


def build(force=False):
    """(Re)build the simulator library if a source is newer than its object.  Serialised with a file lock: pytest-xdist
    workers call this at the same time, and two of them compiling into the same objects corrupt the link."""
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(force):
    srcs = [s for s in sorted(glob.glob(os.path.join(CSRC, '*.hip'))) if not s.endswith('eqd_target_gfx950.hip')]
    deps = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, 'hip', 'hip_runtime.h'),
                                                           os.path.join(ROOT, 'include', 'equidock_hip.h')]
    objs, jobs = [], []
    for s in srcs + [os.path.join(HERE, 'hostsim.cpp'), os.path.join(HERE, 'hostsim_abi.cpp')]:
        o = os.path.join(OUT, os.path.basename(s).rsplit('.', 1)[0] + '.o')
        objs.append(o)
        if force or not os.path.exists(o) or any(os.path.getmtime(d) > os.path.getmtime(o) for d in [s] + deps):
            jobs.append([CXX] + FLAGS + ['-x', 'c++', '-c', s, '-o', o])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(' '.join(cmd) + '\n' + r.stdout + r.stderr)
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB):
        run([CXX, '-shared', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force=True))

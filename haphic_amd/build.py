"""Builds libhaphic_hip.so (HIP, gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build()."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libhaphic_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
         '-fno-fast-math', '-Wall', '-Wno-unused-result']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def source_hash():
    """sha256 (16 hex digits) over the kernel sources: ties a rocprofv3 / PMC record to the code it was measured on"""
    import hashlib
    h = hashlib.sha256()
    for f in sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))):
        h.update(os.path.basename(f).encode() + b'\0')
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


OBJ = os.path.join(CSRC, '_obj')


def _stale(obj, src, headers):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + headers)


def build(force=False, verbose=False, jobs=None):
    """every csrc/*.hip -> its own object (in parallel, only the stale ones), then one link: the whole library from scratch in the time of
    its largest file (hhx_expand.hip), an edit of one file in the time of that file"""
    if not force and not needs_build():
        return SO
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(HERE, '..', 'include', '*.h'))
    compile_flags = [f for f in FLAGS if f != '-shared']
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, src, headers):
            todo.append([HIPCC] + compile_flags + ['-c', src, '-o', obj])
    todo.sort(key=lambda c: -os.path.getsize(c[-3]))             # the largest file first

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(jobs or min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, todo))
    run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-lz', '-lpthread', '-o', SO])
    return SO


ASAN_SO = os.path.join(HERE, 'libhaphic_hip_asan.so')


def asan_runtime():
    """the AddressSanitizer runtime to LD_PRELOAD under python: GCC's libasan.  (ROCm clang's own runtime intercepts
    hsa_amd_memory_pool_allocate for xnack+ device sanitizing and aborts the HIP runtime of an xnack- process at its first pool;
    GCC 11's speaks the same instrumentation ABI, the three helpers it lacks are in csrc/asan_shim.cpp.)"""
    out = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    return os.path.realpath(out) if out and os.path.exists(out) else None


def build_asan(verbose=False):
    """The HOST side of the library under AddressSanitizer (the C-ABI layer is hand-written C++ — handle lifetimes, staging
    buffers, host vectors of the BAM / interpret / merge code); the device code is compiled as usual.  Run with
      LD_PRELOAD=$(python -c 'from haphic_amd import build; print(build.asan_runtime())') ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
      HAPHIC_HIP_SO=haphic_amd/libhaphic_hip_asan.so python -m pytest tests -m gpu ...      (tools/gpu_pass.sh asan)"""
    flags = [f for f in FLAGS if f != '-O3'] + ['-O1', '-g', '-fno-omit-frame-pointer', '-fsanitize=address', '-Wno-option-ignored']
    cmd = [HIPCC] + flags + sources() + [os.path.join(CSRC, 'asan_shim.cpp'), '-lz', '-lpthread', '-o', ASAN_SO]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return ASAN_SO


if __name__ == '__main__':
    import sys
    print(build_asan(verbose=True) if 'asan' in sys.argv[1:] else build(force=True, verbose=True))

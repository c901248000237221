"""Builds libyoloret_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m yoloret_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libyoloret_hip.so')
SOURCES = ['runtime.hip', 'pointwise.hip', 'pointwise_lds.hip', 'pointwise_h.hip', 'pointwise_hs.hip', 'pointwise_hq.hip', 'depthwise.hip', 'depthwise_lds.hip', 'stem.hip', 'elementwise.hip', 'postprocess.hip', 'mbconv.hip',
           'preprocess.hip', 'stemblock.hip', 'mblane.hip', 'mbh.hip']
# -ffp-contract=off: decode/NMS must match the oracle's IEEE operation order bit for bit;
# every intended fused multiply-add in the kernels is an explicit fmaf / MFMA.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
         '-Wall', '-Wno-unused-function'] + os.environ.get('YOLORET_HIPCC_FLAGS', '').split()  # e.g. -DPW_BK=64 (experiments)


FLAGS_FILE = os.path.join(CSRC, '_obj', 'flags.txt')   # the flags the library on disk was built with


def _stale():
    if not os.path.exists(LIB):
        return True
    try:   # a library left behind by an experiment (YOLORET_HIPCC_FLAGS=-D...) is rebuilt, not shipped
        if open(FLAGS_FILE).read() != ' '.join(FLAGS):
            return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, '..', 'include', 'yoloret_hip.h'))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'csrc', '_obj')
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(obj)
        cmd = [hipcc] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write('--- %s ---\n%s\n' % (src, out))
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError('hipcc failed')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    with open(FLAGS_FILE, 'w') as f:
        f.write(' '.join(FLAGS))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))

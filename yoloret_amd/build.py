"""Builds libyoloret_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m yoloret_amd.build [--force]

Every compile also asks for the kernel resource report (-Rpass-analysis=kernel-resource-usage); the per-kernel lines
(VGPRs, scratch bytes per lane, waves per SIMD, LDS) are kept as csrc/_obj/<file>.regs.txt - `kernel_resources()` reads
them, tests/test_host_logic.py holds the list of kernels that may spill and fails on any other.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libyoloret_hip.so')
SOURCES = ['runtime.hip', 'pointwise.hip', 'pointwise_lds.hip', 'pointwise_split.hip', 'pointwise_stream.hip', 'pointwise_h.hip', 'pointwise_hs.hip', 'pointwise_hq.hip', 'depthwise.hip', 'depthwise_lds.hip', 'depthwise_walk.hip', 'stem.hip', 'elementwise.hip', 'postprocess.hip',
           'preprocess.hip', 'stemblock.hip', 'stemblock_h.hip', 'mblane.hip', 'mbh.hip', 'mbn_h.hip', 'mbr.hip', 'mbk.hip', 'mbxr_h.hip', 'headblock.hip', 'headwalk.hip', 'headwalk_h.hip', 'headstream.hip']
# -ffp-contract=off: decode/NMS must match the oracle's IEEE operation order bit for bit;
# every intended fused multiply-add in the kernels is an explicit fmaf / MFMA.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
         '-Wall', '-Wno-unused-function'] + os.environ.get('YOLORET_HIPCC_FLAGS', '').split()  # e.g. -DPW_BK=64 (experiments)


FLAGS_FILE = os.path.join(CSRC, '_obj', 'flags.txt')   # the flags the library on disk was built with


def _fingerprint():
    """sha256 over the flags and the CONTENT of every source and header: what the library on disk must have been built from.
    (File times say nothing after a fresh checkout or a copy - round 3's _stale() compared mtimes.)"""
    import hashlib
    h = hashlib.sha256(' '.join(FLAGS).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    deps += [os.path.join(HERE, '..', 'include', 'yoloret_hip.h'), os.path.abspath(__file__)]
    for d in deps:
        h.update(os.path.basename(d).encode())
        h.update(open(d, 'rb').read())
    return h.hexdigest()


STAMP_FILE = os.path.join(CSRC, '_obj', 'fingerprint.txt')


def _stale():
    if not os.path.exists(LIB):
        return True
    try:
        if open(STAMP_FILE).read().strip() == _fingerprint():
            return False        # built from exactly these sources and flags, whatever the file times say
    except OSError:
        pass
    try:   # a library left behind by an experiment (YOLORET_HIPCC_FLAGS=-D...) is rebuilt, not shipped
        if open(FLAGS_FILE).read() != ' '.join(FLAGS):
            return True
    except OSError:
        return True
    if not all(os.path.exists(os.path.join(CSRC, '_obj', f.replace('.hip', '.regs.txt'))) for f in SOURCES):
        return True   # (a library from before the resource reports were kept)
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(HERE, '..', 'include', 'yoloret_hip.h'))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def _split_report(out):
    """hipcc output -> (one line per kernel: name vgprs agprs scratch occupancy lds, everything that is not the report)."""
    import re
    rows, rest, cur = [], [], None
    for line in out.split('\n'):
        m = re.search(r'remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
        if m:
            if m.group(1) == 'Function Name':
                cur = {'name': m.group(2)}
                rows.append(cur)
            elif cur is not None:
                cur[m.group(1).split(' ')[0]] = m.group(2)
        elif 'remark:' not in line and '-Rpass-analysis' not in line:
            rest.append(line)
    text = ''.join('%s vgprs %s agprs %s scratch %s occupancy %s lds %s\n' % (r['name'], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('ScratchSize', '?'),
                                                                             r.get('Occupancy', '?'), r.get('LDS', '?')) for r in rows)
    return text, '\n'.join(rest)


def kernel_resources():
    """{source file: [(mangled kernel, vgprs, scratch bytes per lane, waves per SIMD, lds bytes)]} of the library on disk."""
    res = {}
    objdir = os.path.join(HERE, 'csrc', '_obj')
    for src in SOURCES:
        path = os.path.join(objdir, src.replace('.hip', '.regs.txt'))
        rows = []
        for line in open(path):
            t = line.split()
            rows.append((t[0], int(t[2]), int(t[6]), int(t[8]), int(t[10])))
        res[src] = rows
    return res


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(HERE, 'csrc', '_obj')
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    # incremental: a source is recompiled when it, any header, this file or the flags are newer than its object (or --force)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(HERE, '..', 'include', 'yoloret_hip.h'),
                                                                                       os.path.abspath(__file__)]
    newest_header = max(os.path.getmtime(h) for h in headers)
    try:
        same_flags = open(FLAGS_FILE).read() == ' '.join(FLAGS)
    except OSError:
        same_flags = False
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace('.hip', '.o'))
        objs.append(obj)
        regs = os.path.join(objdir, src.replace('.hip', '.regs.txt'))
        if (not force and same_flags and os.path.exists(obj) and os.path.exists(regs)
                and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(os.path.join(CSRC, src)))):
            continue
        cmd = [hipcc] + FLAGS + ['-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write('--- %s ---\n%s\n' % (src, out))
            continue
        report, rest = _split_report(out)
        with open(os.path.join(objdir, src.replace('.hip', '.regs.txt')), 'w') as f:
            f.write(report)
        if verbose and rest.strip():
            print(rest)
    if failed:
        raise RuntimeError('hipcc failed')
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    with open(FLAGS_FILE, 'w') as f:
        f.write(' '.join(FLAGS))
    with open(STAMP_FILE, 'w') as f:
        f.write(_fingerprint())
    return LIB


def report():
    """The register / scratch / occupancy table of every kernel of the library on disk (profiles/rNN_kernel_regs.txt)."""
    lines = []
    for src, rows in kernel_resources().items():
        spill = [r for r in rows if r[2] > 0]
        lines.append('%-20s %4d kernels, VGPRs %d..%d, %d with scratch%s' % (src, len(rows), min([r[1] for r in rows] or [0]), max([r[1] for r in rows] or [0]),
                                                                            len(spill), (' (max %d bytes per lane)' % max(r[2] for r in spill)) if spill else ''))
        for name, vg, sc, occ, lds in rows:
            lines.append('    %-70s vgprs %3d scratch %4d waves/SIMD %d lds %6d' % (name, vg, sc, occ, lds))
    return '\n'.join(lines)


if __name__ == '__main__':
    if '--report' in sys.argv:
        build()
        print(report())
    else:
        print(build(force='--force' in sys.argv, verbose=True))

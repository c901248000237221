#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2) rocpd SQLite outputs into small text tables for profiles/.

    python tools/rocpd_summary.py stats  gpurun_out/prof/stats_results.db        > profiles/rNN_kernel_stats.txt
    python tools/rocpd_summary.py pmc    gpurun_out/prof_fetch/pmc_results.db     > profiles/rNN_pmc_fetch.txt
"""
import sqlite3
import sys


import re


def demangle_simple(m):
    """'_Z10mbh_kernelIDF16bLi3ELi2ELi4ELi1ELb1EEv7MbhArgs' -> 'mbh_kernel<bf16,3,2,4,1,1>': the template arguments this
    library's kernels take (element types, ints, bools).  rocprofv3's own demangler garbles __bf16 / _Float16 arguments
    ('mbh_kernel<bool _Accum, int, ELi, E, 1, 1, true>'), so the symbol table's mangled names are decoded here."""
    r = re.match(r'^_Z(\d+)', m)
    if not r:
        return None
    n = int(r.group(1))
    pos = r.end() + n
    name = m[r.end():pos]
    if pos >= len(m) or m[pos] != 'I':
        return name
    pos += 1
    args = []
    while pos < len(m) and m[pos] != 'E':
        if m.startswith('DF16b', pos):
            args.append('bf16'); pos += 5
        elif m.startswith('DF16_', pos):
            args.append('f16'); pos += 5
        elif m[pos] == 'f':
            args.append('f32'); pos += 1
        elif m[pos] == 'L':
            q = re.match(r'L([ib])(n?)(\d+)E', m[pos:])
            if not q:
                return None
            args.append(('-' if q.group(2) else '') + q.group(3)); pos += q.end()
        else:
            return None
    return '%s<%s>' % (name, ','.join(args))


_pretty = {}


def load_symbols(c):
    """display name -> decoded name for the kernels whose display name rocprofv3 garbled."""
    try:
        rows = c.execute('select kernel_name, display_name from rocpd_info_kernel_symbol').fetchall()
    except sqlite3.Error:
        return
    for mangled, disp in rows:
        d = demangle_simple(mangled.replace('.kd', ''))
        if d and ('_Accum' in disp or disp.startswith('_Z')):
            _pretty[disp] = d
            _pretty[disp.split('(')[0].replace('void ', '')] = d


def short(name):
    if name in _pretty:
        return _pretty[name]
    base = name.split('(')[0].replace('void ', '')
    if base in _pretty:
        return _pretty[base]
    if base.startswith('_Z'):
        d = demangle_simple(base)
        if d:
            return d
    return base[:64]


def stats(db):
    c = sqlite3.connect(db)
    load_symbols(c)
    rows = c.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) '
                     'from kernels group by name order by sum(duration) desc').fetchall()
    total = sum(r[2] for r in rows) or 1
    print('%-64s %8s %12s %12s %10s %10s %6s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'min_ns', 'max_ns', '%'))
    for n, k, t, a, lo, hi in rows:
        print('%-64s %8d %12d %12.0f %10d %10d %6.2f' % (short(n), k, t, a, lo, hi, 100.0 * t / total))


def pmc(db):
    c = sqlite3.connect(db)
    load_symbols(c)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    rows = c.execute('select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection '
                     'group by kernel_name, counter_name order by sum(value) desc').fetchall() \
        if 'kernel_name' in cols else []
    print('%-64s %-12s %8s %16s %18s' % ('kernel', 'counter', 'calls', 'avg_per_launch', 'sum'))
    for n, cn, k, a, s in rows:
        print('%-64s %-12s %8d %16.1f %18.1f' % (short(n), cn, k, a, s))
    if not rows:
        print('columns:', cols)


if __name__ == '__main__':
    if sys.argv[1] in ('stats', 'pmc'):
        {'stats': stats, 'pmc': pmc}[sys.argv[1]](sys.argv[2])


def traffic(fetch_db, write_db):
    """Per-kernel HBM traffic per launch in bytes: (2*FETCH_SIZE + WRITE_SIZE) KiB.  FETCH_SIZE is doubled
    as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950; WRITE_SIZE is
    taken as is (uncalibrated).  Prints JSON {kernel symbol: {fetch_bytes, write_bytes, traffic_bytes, launches}}."""
    import json
    out = {}
    for db, key in ((fetch_db, 'FETCH_SIZE'), (write_db, 'WRITE_SIZE')):
        c = sqlite3.connect(db)
        load_symbols(c)
        for n, k, a in c.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                                 "group by kernel_name", (key,)):
            d = out.setdefault(short(n).replace(', ', ','), {})
            d[key] = a * 1024.0
            d['launches'] = k
    res = {}
    for n, d in out.items():
        f, w = d.get('FETCH_SIZE', 0.0), d.get('WRITE_SIZE', 0.0)
        res[n] = {'fetch_bytes_corrected': round(2 * f), 'write_bytes': round(w), 'traffic_bytes': round(2 * f + w),
                  'launches_sampled': d['launches']}
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == '__main__' and sys.argv[1] == 'traffic':
    traffic(sys.argv[2], sys.argv[3])

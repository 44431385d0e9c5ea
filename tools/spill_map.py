#!/usr/bin/env python3
"""Where a kernel spills: clusters of scratch_ instructions in the device assembly, with the MFMA count
around them.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math --cuda-device-only -S chain.hip -o /tmp/chain.s
                python tools/spill_map.py /tmp/chain.s chain_kernelILi16ELi1ELb0"""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2:]
funcs = re.split(r'\n(?=_ZN2tc[^\n:]*:)', s)
for f in funcs:
    name = f.split(':')[0]
    if not any(p in name for p in pat):
        continue
    lines = f.split('\n')
    sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    print(name[:70], 'lines', len(lines), 'scratch ops', len(sc), 'mfma', len(mf))
    cl = []
    for i in sc:
        if cl and i - cl[-1][-1] < 40:
            cl[-1].append(i)
        else:
            cl.append([i])
    for c in cl:
        st = sum('scratch_store' in lines[i] for i in c)
        near = sum(1 for m in mf if c[0] - 200 <= m <= c[-1] + 200)
        print('  lines %6d-%6d: %3d stores %3d loads, %3d MFMAs within 200 lines' % (c[0], c[-1], st, len(c) - st, near))

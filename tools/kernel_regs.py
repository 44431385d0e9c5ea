#!/usr/bin/env python3
"""Registers / spills / code size of the kernels of one source file, from its device assembly.
    python tools/kernel_regs.py transcar_amd/csrc/chain.hip [name filter ...]     (extra -D flags through EXTRA=...)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pats = sys.argv[2:]
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, 'k.s')
    cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-fast-math', '-fno-slp-vectorize', '-Xclang',
           '-target-feature', '-Xclang', '-packed-fp32-ops', '--cuda-device-only', '-S', src, '-o', out] + os.environ.get('EXTRA', '').split()
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd=ROOT)
    s = open(out).read()
for m in re.finditer(r'- \.agpr_count:.*?\.wavefront_size', s, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r'\.%s:\s*(\S+)' % k, blk).group(1)
    nm = re.sub(r'_ZN2tc12_GLOBAL__N_1\d+', '', g('name'))[:48]
    if pats and not any(p in nm for p in pats):
        continue
    print('%-48s vgpr %3s spill %3s scratch %4s' % (nm, g('vgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size')))

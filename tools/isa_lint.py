"""ISA lint of the built gfx950 library: instruction forms this code base must not contain.

Rule PK-OPSEL (round 5, profiles/r5_refill_hazard.txt, tools/pk_hazard_probe.hip): on MI355X (gfx950, ROCm 7.2) a
packed-f32 VALU op whose LOW result selects the HIGH half of a source pair through op_sel -- hipcc's SLP
vectoriser forms `v_pk_mul_f32 v[a:a+1], v[a:a+1], v[b:b+1] op_sel:[0,1]` for `x * s, y * s` -- returned 0 in
lanes 48..63 of the low result while ANOTHER wave on the same SIMD had vector-memory loads landing in registers
its matrix-core MFMAs had just read.  The library is built with -fno-slp-vectorize (no v_pk_*_f32 is formed at
all) and this lint fails the CPU test suite if a packed f32 op with a non-zero op_sel entry ever comes back.

Rule PK-MFMA (round 6): every packed-f32 op inside a symbol that also issues MFMAs is REPORTED
(`lint_text(...)[1]['pk_mfma']`: [(symbol, packed ops, MFMAs)]) -- beside MFMAs a packed f32 op costs +22-26 cycles over the
two scalar ops it replaces (MI355X_MICROARCH.md).  Round 5's library held 16 / 40 `v_pk_mul_f32 ... op_sel_hi:[1,0]` in the
two attention cores (hipcc legalises `acc *= alpha` on an ext_vector_type that way, with or without the SLP
vectoriser); the library is now compiled with the packed-f32 ops switched off (`-target-feature -packed-fp32-ops`,
transcar_amd/csrc/Makefile; scalarising them in the source cost the core 29 registers and 11 % -- profiles/r6_attention_pk.txt)
and the CPU suite asserts the list stays empty.

    python tools/isa_lint.py [path/to/libtranscar_hip.so]      (exit code 1 on a finding of either rule)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, 'transcar_amd', 'lib', 'libtranscar_hip.so')
LLVM_BIN = '/opt/rocm/lib/llvm/bin'

PK_F32 = re.compile(r'\b(v_pk_(?:mul|add|fma)_f32)\b(.*)$')
OP_SEL = re.compile(r'\bop_sel:\[([01,]+)\]')


def device_disassembly(lib_path):
    """llvm-objdump of every gfx950 code object bundled in the shared library."""
    with tempfile.TemporaryDirectory() as tmp:
        out = []
        # the fat binary sits in .hip_fatbin; clang-offload-bundler lists and extracts the device images
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([os.path.join(LLVM_BIN, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', lib_path, fat],
                       check=True)
        bundler = os.path.join(LLVM_BIN, 'clang-offload-bundler')
        targets = subprocess.run([bundler, '--list', '--type=o', '--input=' + fat], check=True, capture_output=True,
                                 text=True).stdout.split()
        # a library linked from several objects holds several bundles back to back: split on the magic string
        blob = open(fat, 'rb').read()
        magic = b'__CLANG_OFFLOAD_BUNDLE__'
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for bi, st in enumerate(starts):
            en = starts[bi + 1] if bi + 1 < len(starts) else len(blob)
            part = os.path.join(tmp, 'bundle%d.bin' % bi)
            open(part, 'wb').write(blob[st:en])
            tl = subprocess.run([bundler, '--list', '--type=o', '--input=' + part], check=True, capture_output=True,
                                text=True).stdout.split()
            for t in tl:
                if 'gfx950' not in t:
                    continue
                co = os.path.join(tmp, 'co%d.o' % bi)
                subprocess.run([bundler, '--unbundle', '--type=o', '--targets=' + t, '--input=' + part, '--output=' + co],
                               check=True)
                out.append(subprocess.run([os.path.join(LLVM_BIN, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], check=True,
                                          capture_output=True, text=True).stdout)
        assert out, 'no gfx950 code object found in %s (targets: %s)' % (lib_path, targets)
        return '\n'.join(out)


def lint_text(text):
    """Findings in a disassembly / assembly listing: [(symbol, line, instruction)]."""
    findings = []
    sym = None
    stats = {'pk_f32': 0, 'mfma': 0, 'kernels': 0, 'pk_mfma': []}
    per = {}                                  # symbol -> [packed f32 ops, MFMAs]
    for ln, line in enumerate(text.split('\n')):
        m = re.match(r'^[0-9a-f]* ?<([^>]+)>:$', line) or re.match(r'^(_Z\w+):', line)
        if m:
            sym = m.group(1)
            stats['kernels'] += 1
            continue
        if 'v_mfma_' in line:
            stats['mfma'] += 1
            per.setdefault(sym, [0, 0])[1] += 1
        pk = PK_F32.search(line)
        if not pk:
            continue
        stats['pk_f32'] += 1
        per.setdefault(sym, [0, 0])[0] += 1
        sel = OP_SEL.search(pk.group(2))
        if sel and '1' in sel.group(1):
            findings.append((sym, ln + 1, line.strip()))
    stats['pk_mfma'] = [(s_, n[0], n[1]) for s_, n in per.items() if n[0] and n[1]]      # rule PK-MFMA
    return findings, stats


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    text = open(lib).read() if lib.endswith('.s') else device_disassembly(lib)
    findings, stats = lint_text(text)
    print('%s: %d symbols, %d MFMAs, %d packed f32 ops, %d with a high-half op_sel' % (
        lib, stats['kernels'], stats['mfma'], stats['pk_f32'], len(findings)))
    for sym, ln, ins in findings[:40]:
        print('  PK-OPSEL  %s  line %d: %s' % ((sym or '?')[:80], ln, ins))
    for sym, npk, nm in stats['pk_mfma']:
        print('  PK-MFMA   %s: %d packed f32 ops beside %d MFMAs' % ((sym or '?')[:80], npk, nm))
    return 1 if findings or stats['pk_mfma'] else 0


if __name__ == '__main__':
    sys.exit(main())

"""The built gfx950 library must not contain the instruction form that round 4's flaky rows came from
(tools/isa_lint.py rule PK-OPSEL; profiles/r5_refill_hazard.txt).  CPU only: llvm-objdump of the in-tree .so."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_lint  # noqa: E402


def test_lint_rule_fires_on_the_form_hipcc_emitted():
    listing = '\n'.join([
        '_Zkernel:',
        '\tv_pk_mul_f32 v[62:63], v[42:43], v[60:61]',
        '\tv_pk_mul_f32 v[64:65], v[64:65], v[62:63] op_sel:[0,1]',
        '\tv_pk_fma_f32 v[40:41], v[40:41], v[60:61], v[64:65] op_sel_hi:[1,0,1]',
        '\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]'])
    findings, stats = isa_lint.lint_text(listing)
    assert stats['pk_f32'] == 3 and stats['mfma'] == 1
    assert len(findings) == 1 and 'op_sel:[0,1]' in findings[0][2]
    assert stats['pk_mfma'] == [('_Zkernel', 3, 1)]          # rule PK-MFMA: packed f32 ops inside a kernel that issues MFMAs
    _, clean = isa_lint.lint_text('_Zother:\n\tv_pk_mul_f32 v[2:3], v[2:3], v[4:5]\n_Zmm:\n\tv_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[8:11], v[0:3]')
    assert clean['pk_mfma'] == []                            # (a packed op in a kernel WITHOUT matrix instructions is not a finding)


def test_built_library_has_no_packed_f32_op_with_a_high_half_op_sel():
    lib = isa_lint.DEFAULT_LIB
    if not os.path.isfile(lib):
        pytest.skip('library not built')
    if not os.path.isfile(os.path.join(isa_lint.LLVM_BIN, 'llvm-objdump')):
        pytest.skip('no llvm-objdump')
    findings, stats = isa_lint.lint_text(isa_lint.device_disassembly(lib))
    assert stats['mfma'] > 1000, stats          # the disassembly really is the chains' code
    assert findings == [], findings[:5]
    # rule PK-MFMA (round 6): no packed f32 op in any kernel that issues MFMAs (round 5: 16 / 40 in the attention cores)
    assert stats['pk_mfma'] == [], stats['pk_mfma']

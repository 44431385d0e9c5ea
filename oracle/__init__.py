"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the TransCAR fusion-decoder hot path,
written from the reference's algorithm with file:line citations, plus the
harness that imports the reference's own Python files (only inside the
authoring container, where /root/reference exists) to pin the restatement.

Nothing under this package is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker / the timed CPU baseline -- never as the
thing shipped.  The product path (``transcar_amd``) runs hand-written HIP
through the C-ABI library and fails loudly when that library is missing.

Parity status (see DESIGN.md "Oracle"):
  * reference-owned arithmetic (detr3d_transformer.py, detr3d_head.py,
    nms_free_coder.py, util.py): PINNED -- the restatement is checked against
    the reference itself, imported unmodified through ``ref_harness`` and the
    outputs are committed as fixtures under tests/golden/.
  * arithmetic that lives in un-vendored third-party code (mmcv transformer
    bricks, mmdet DETRHead, nuscenes-devkit radar loader; no version pinned by
    the reference: .gitmodules:1-3 is an empty submodule): "parity unpinned" --
    restated from the published semantics in ``mmcv_bricks.py`` and
    cross-checked against stock torch.nn modules only.
"""

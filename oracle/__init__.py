"""CPU oracle for the Relation-Networks detection hot path.  TEST INFRASTRUCTURE.

This package is a plain numpy (and, for the conv backbone only, torch-CPU fp32)
restatement of the reference's algorithm for every row of SURVEY.md section 8(a).
Each function cites the reference file:line it follows (paths relative to the
reference checkout).  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it; the shipped package
(`relation-networks-for-object-detection_amd/`) never does and fails loudly when its
HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * anchors, box decode/clip, numpy NMS / soft-NMS     -> pinned against the
    reference's own python, imported by tests/golden/gen_golden.py.
  * geometry embedding, relation module, learn-NMS head -> wiring pinned by running
    the reference's own symbols/*.py and operator_py/learn_nms.py on the numpy
    MXNet stand-in (tests/golden/refshim); per-operator MXNet semantics restated.
  * proposal glue, ROIPooling, GPU-NMS bitmask order, backbone layers -> restated
    from the cited lines / MXNet v1.1.0 semantics: PARITY UNPINNED (the reference
    has no tests or vectors and those files cannot execute here).
"""

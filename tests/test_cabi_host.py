"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports exactly what
include/relnet_hip.h declares; host-side logic of the operator mirror and the detector config."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def pkg():
    import __graft_entry__ as ge
    ge.build()
    import relnet_amd  # noqa: F401
    from relnet_amd import lib
    return lib


def _declared():
    hdr = open(os.path.join(ROOT, 'include', 'relnet_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(relnet_\w+|_nms)\s*\(', hdr)))


def test_header_and_library_agree(pkg):
    declared = _declared()
    assert '_nms' in declared and 'relnet_relation_attention' in declared
    lib = ctypes.CDLL(pkg.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), "declared in include/relnet_hip.h but not exported: %s" % sym
    assert sorted(pkg.exported_symbols()) == declared     # the Python binding covers all of it


def test_version_and_error_channel(pkg):
    lib = pkg.load()
    assert lib.relnet_version() == 100
    # argument validation happens before any HIP call -> usable without a GPU
    rc = lib.relnet_gemm_nt(None, 0, 0, None, 0, 0, None, 0, 0, None, 0, None, 0, 1, 1, 64, 1, 1, 1, None)
    assert rc != 0 and b'null operand' in lib.relnet_last_error()
    rc = lib.relnet_nms_mask(None, None, None, 1, 64, 64, 0.7, None)
    assert rc != 0 and b'relnet_nms_mask' in lib.relnet_last_error()


def test_no_cpu_fallback():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, lib
    with pytest.raises(lib.RelnetError):
        ops.gemm_nt(torch.zeros(4, 64), torch.zeros(4, 64))           # CPU tensors are refused


def test_product_does_not_import_oracle():
    pkg_dir = os.path.join(ROOT, 'relation-networks-for-object-detection_amd')
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_proposal_prop_protocol():
    import relnet_amd  # noqa: F401
    from relnet_amd import operator_py
    from relnet_amd.operator_py import proposal as P
    prop = operator_py.get_prop('proposal')(feat_stride='16', scales='(4, 8, 16, 32)', ratios='(0.5, 1, 2)',
                                            output_score='True', rpn_pre_nms_top_n='6000',
                                            rpn_post_nms_top_n='300', threshold='0.7', rpn_min_size='0')
    assert prop.list_arguments() == ['cls_prob', 'bbox_pred', 'im_info']
    assert prop.list_outputs() == ['output', 'score']
    ins, outs = prop.infer_shape([(1, 24, 38, 63), (1, 48, 38, 63), (1, 3)])
    assert outs == [(300, 5), (300, 1)] and ins[2] == (1, 3)
    assert prop.declare_backward_dependency([], [], []) == []
    op = prop.create_operator(None, None, None)
    assert op._num_anchors == 12
    from oracle.boxes import generate_anchors
    assert np.array_equal(P.generate_anchors(16, (0.5, 1, 2), (4, 8, 16, 32)),
                          generate_anchors(16, (0.5, 1, 2), (4, 8, 16, 32)))
    with pytest.raises(KeyError):
        operator_py.get_prop('no_such_op')


def test_backbone_graph_matches_reference_names():
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone, detector
    names = [c for c, _, _, _, _ in backbone.conv_bn_names()]
    assert len(names) == 104 and names[0] == 'conv1' and 'res4b22_branch2c' in names and 'res3b3_branch2a' in names
    units = backbone.unit_names()
    assert [u[5] for u in units if u[7]] == [1, 2, 2, 1]           # strides of res2a/3a/4a/5a
    assert all(u[6] == 2 for u in units if u[0] == 5)               # conv5 dilated
    perm = detector.fc1_channels_last_perm(4, 2, 2)
    # new column (s*C + c) takes reference column (c*S + s)
    assert perm.tolist() == [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15]
    w, b = backbone.fold_bn(torch.ones(2, 1, 1, 1), torch.tensor([2.0, 1.0]), torch.tensor([0.5, 0.0]),
                            torch.tensor([1.0, 0.0]), torch.tensor([3.0, 0.0]))
    assert torch.allclose(w.flatten(), torch.tensor([2 / (3 + 1e-5) ** 0.5, 1 / (1e-5) ** 0.5]), rtol=1e-6)
    assert torch.allclose(b, torch.tensor([0.5 - 2 / (3 + 1e-5) ** 0.5, 0.0]), rtol=1e-6)


def test_oracle_roi_pooling_micro_case():
    from oracle.roi_pooling import roi_pooling
    data = np.arange(2 * 6 * 8, dtype=np.float32).reshape(1, 2, 6, 8)
    rois = np.array([[0, 0, 0, 127, 95], [0, 16, 16, 47, 47]], dtype=np.float32)      # /16 -> (0..8, 0..6), (1..3)
    out, arg = roi_pooling(data, rois, pooled_size=(2, 2), spatial_scale=1 / 16., return_argmax=True)
    # roi 0 spans x 0..8 -> clipped to the 8-wide map; bins are 4.5 x 3.5 -> floor/ceil windows
    assert out[0, 0].tolist() == [[28.0, 31.0], [44.0, 47.0]]
    assert out[1, 0].tolist() == [[18.0, 19.0], [26.0, 27.0]]
    assert arg[1, 0].tolist() == [[18, 19], [26, 27]]


def test_operator_cxx_property_shapes():
    """operator_cxx mirror: parameter defaults, argument lists and InferShape rules of the reference's
    DeformableConvolutionProp / DeformablePSROIPoolingProp (deformable_convolution-inl.h:294-400,
    deformable_psroi_pooling-inl.h:153-230) -- host logic only."""
    import relnet_amd  # noqa: F401
    from relnet_amd import operator_cxx as cxx
    p = cxx.DeformableConvolutionProp(kernel='(3, 3)', num_filter='512', pad='(2, 2)', dilate='(2, 2)',
                                      num_deformable_group='4', no_bias='True')
    assert p.ListArguments() == ['data', 'offset', 'weight'] and p.param_.stride == (1, 1)
    shapes, (out,) = p.InferShape([(1, 512, 38, 63), (1, 72, 38, 63), None])
    assert out == (1, 512, 38, 63) and shapes[2] == (512, 512, 3, 3)
    q = cxx.DeformableConvolutionProp(kernel=(3, 3), num_filter=8)
    assert q.ListArguments() == ['data', 'offset', 'weight', 'bias']
    assert q.InferShape([(2, 4, 9, 9), (2, 18, 7, 7), None, None])[1] == [(2, 8, 7, 7)]
    import pytest
    with pytest.raises(ValueError):
        q.InferShape([(2, 4, 9, 9), (2, 18, 9, 9), None, None])          # offset map size != output size
    with pytest.raises(ValueError):
        q.InferShape([(2, 4, 9, 9), (2, 36, 7, 7), None, None])          # offset channels vs deformable groups
    with pytest.raises(ValueError):
        q.InferShape([(2, 4, 9, 9), (2, 18, 7, 7), None])                # bias missing
    r = cxx.DeformablePSROIPoolingProp(spatial_scale='0.0625', output_dim='256', group_size='1', pooled_size='7',
                                      part_size='7', sample_per_part='4', trans_std='0.1')
    assert r.ListArguments() == ['data', 'rois', 'trans'] and r.NumVisibleOutputs() == 1
    assert r.InferShape([(1, 256, 38, 63), (300, 5), (300, 2, 7, 7)])[1] == [(300, 256, 7, 7)] * 2
    with pytest.raises(ValueError):
        cxx.DeformablePSROIPoolingParam(spatial_scale=2.0, output_dim=1, group_size=1, pooled_size=7)


def test_assign_anchor_host_port():
    """train.assign_anchor (host-side RPN label preparation, lib/rpn/rpn.py:80-244): layouts, sampling limits and the
    definition of positives / negatives; targets invert back to the matched gt box."""
    import numpy as np
    import relnet_amd  # noqa: F401
    from relnet_amd import train
    from relnet_amd.operator_py.proposal import generate_anchors
    cfg = train.TrainConfig()
    H, W, fh, fw = 600, 1000, 38, 63
    rng = np.random.default_rng(0)
    G = 6
    bw, bh = rng.uniform(40, 400, G), rng.uniform(40, 400, G)
    x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
    gt = np.stack([x1, y1, x1 + bw, y1 + bh, rng.integers(1, 81, G)], 1).astype(np.float32)
    L, Tg, Wg = train.assign_anchor((fh, fw), gt, (H, W), cfg, seed=3)
    A = 12
    assert L.shape == (A * fh * fw,) and Tg.shape == (4 * A, fh, fw) and Wg.shape == (4 * A, fh, fw)
    assert set(np.unique(L)) <= {-1.0, 0.0, 1.0}
    nfg, nbg = int((L == 1).sum()), int((L == 0).sum())
    assert 0 < nfg <= cfg.rpn_batch_size // 2 and nfg + nbg == cfg.rpn_batch_size
    # weights mark exactly the positive anchors, all four coordinates
    Lg = L.reshape(A, fh, fw)
    assert np.array_equal(Wg.reshape(A, 4, fh, fw).sum(1) == 4, Lg == 1) and set(np.unique(Wg)) <= {0.0, 1.0}
    # decode the targets of the positive anchors: they reproduce a gt box, and that box overlaps the anchor well
    base = generate_anchors(cfg.feat_stride, cfg.anchor_ratios, cfg.anchor_scales)
    a_idx, ys, xs = np.where(Lg == 1)
    anc = base[a_idx] + np.stack([xs, ys, xs, ys], 1) * cfg.feat_stride
    t = Tg.reshape(A, 4, fh, fw)[a_idx, :, ys, xs]
    aw, ah = anc[:, 2] - anc[:, 0] + 1, anc[:, 3] - anc[:, 1] + 1
    cx, cy = anc[:, 0] + 0.5 * (aw - 1) + t[:, 0] * aw, anc[:, 1] + 0.5 * (ah - 1) + t[:, 1] * ah
    w, h = np.exp(t[:, 2]) * aw, np.exp(t[:, 3]) * ah
    dec = np.stack([cx - 0.5 * (w - 1), cy - 0.5 * (h - 1), cx + 0.5 * (w - 1), cy + 0.5 * (h - 1)], 1)
    err = np.abs(dec[:, None, :] - gt[None, :, :4]).max(2).min(1)
    assert err.max() < 1e-2
    # anchors crossing the image border are never labelled
    sx, sy = np.meshgrid(np.arange(fw) * cfg.feat_stride, np.arange(fh) * cfg.feat_stride)          # [fh, fw]
    ga = base[None, None] + np.stack([sx, sy, sx, sy], -1)[:, :, None, :]                               # [fh, fw, A, 4]
    inside = (ga[..., 0] >= 0) & (ga[..., 1] >= 0) & (ga[..., 2] < W) & (ga[..., 3] < H)            # [fh, fw, A]
    assert (Lg.transpose(1, 2, 0)[~inside] == -1).all()
    # no gt: everything sampled is background
    L0, _, W0 = train.assign_anchor((fh, fw), np.zeros((0, 5), np.float32), (H, W), cfg, seed=1)
    assert (L0 == 1).sum() == 0 and (L0 == 0).sum() == cfg.rpn_batch_size and W0.sum() == 0


def test_assign_anchor_matches_reference_loader():
    """train.assign_anchor vs tests/golden/rpn_targets.npz = the output of the reference's own lib/rpn/rpn.py:assign_anchor
    (gen_golden.py), same numpy seed: identical labels (incl. the random fg / bg subsampling) and weights, targets to 2e-6."""
    import numpy as np
    import relnet_amd  # noqa: F401
    from relnet_amd import train
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rpn_targets.npz'))
    cfg = train.TrainConfig()
    for name, seed in (('six_gt', 3), ('crowded', 4)):
        L, T, W = train.assign_anchor((38, 63), g[name + '/gt'], (600, 1000), cfg, seed=seed)
        assert np.array_equal(L, g[name + '/label'][0]) and np.array_equal(W, g[name + '/bbox_weight'][0])
        assert np.abs(T - g[name + '/bbox_target'][0]).max() <= 2e-6
    assert int((g['crowded/label'] == 1).sum()) == 128          # the fg subsampling branch is exercised


def test_every_registered_reference_op_name_resolves():
    """The five hot-path names the reference registers with mx.operator.register all have a Prop
    here (`monitor`, a debug identity, is out of scope: every use in the reference's symbols is commented out)."""
    import relnet_amd  # noqa: F401
    from relnet_amd import operator_py
    for name in ('proposal', 'proposal_target', 'BoxAnnotatorOHEM', 'learn_nms', 'nms_multi_target'):
        assert operator_py.get_prop(name) is not None



def test_chain_weight_fragment_order_matches_the_header():
    """ops.pack_chain_w1 is the host half of relnet_bottleneck_chain's contract (include/relnet_hip.h): block (rt, ks) = 64 lanes x 8
    values, lane (l31, half) slot t <- W1n[32 rt + l31][16 ks + 8 (t >> 2) + 4 half + (t & 3)].  Checked element by element on CPU;
    and ops.chain_worthwhile keeps the persistent kernels off maps that cannot fill 256 CUs."""
    import torch
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    mid, cout = 64, 256
    w = torch.arange(mid * cout, dtype=torch.float32).reshape(mid, cout).to(torch.bfloat16)   # (values collide in bf16; compare indices)
    idx = torch.arange(mid * cout, dtype=torch.int32).reshape(mid, cout)
    got = ops.pack_chain_w1(w)
    assert got.shape == (mid // 32, cout // 16, 64, 8) and got.is_contiguous()
    for rt, ks, lane, t in ((0, 0, 0, 0), (1, 5, 37, 6), (0, 15, 63, 7), (1, 9, 31, 3), (0, 3, 32, 4)):
        l31, half = lane & 31, lane >> 5
        r, c = 32 * rt + l31, 16 * ks + 8 * (t >> 2) + 4 * half + (t & 3)
        assert got[rt, ks, lane, t] == w[r, c], (rt, ks, lane, t)
    # every source element appears exactly once
    flat = idx[(torch.arange(mid // 32).view(-1, 1, 1, 1) * 32 + (torch.arange(64).view(1, 1, -1, 1) & 31)),
               (torch.arange(cout // 16).view(1, -1, 1, 1) * 16 + 8 * (torch.arange(8).view(1, 1, 1, -1) >> 2)
                + 4 * (torch.arange(64).view(1, 1, -1, 1) >> 5) + (torch.arange(8).view(1, 1, 1, -1) & 3))]
    assert sorted(flat.flatten().tolist()) == list(range(mid * cout))
    assert ops.chain_worthwhile(54 * 150 * 250, 64) and ops.chain_worthwhile(54 * 38 * 63, 256)
    # (res4's role-specialised kernel replaces two launches and pays off from 4 images of 600 x 1000; the lock-step kernels need ~1.5 sets per CU)
    assert not ops.chain_worthwhile(38 * 63, 256) and not ops.chain_worthwhile(2 * 38 * 63, 256) and ops.chain_worthwhile(8 * 38 * 63, 256)
    assert not ops.chain_worthwhile(75 * 125, 128) and ops.chain_worthwhile(8 * 75 * 125, 128)
    assert not ops.chain_worthwhile(8 * 38 * 63, 512) and ops.chain_worthwhile(27 * 38 * 63, 512)


def test_tile_selection_rules_without_a_gpu():
    """relnet_gemm_pick_tile is pure host logic (csrc/gemm.hip:pick_tile): the benchmark's layer shapes land on the tiles DESIGN.md
    section 4 names -- the asm ring tile (19) for the shortcut-free 256-column layers of a 54-image step, 128 x 64 tiles at 8 images,
    64 x 64 tiles with two / four k-slabs per barrier (20 / 21) when a launch has fewer workgroups than CUs (one image per step)."""
    import relnet_amd  # noqa: F401
    from relnet_amd import lib
    L = lib.load()
    assert L.relnet_gemm_tile_count() == 23          # (23, round 6: split-K form of tile 20, chosen at launch time when the work area is registered)
    bf16 = 1
    px = lambda b: b * 38 * 63
    pick = lambda M, N, K: L.relnet_gemm_pick_tile(M, N, K, 1, bf16)
    assert pick(px(54), 256, 2304) == 19 and pick(px(54), 512, 4608) == 19 and pick(px(54), 512, 9216) == 19      # res4 / res5 / RPN 3x3
    assert pick(px(54), 256, 1024) == 19 and pick(px(108), 256, 2304) == 19
    assert pick(px(8), 256, 2304) == 4 and pick(px(8), 256, 1024) == 4                                               # 75 tiles of 256 x 256 would idle most CUs
    assert pick(px(1), 256, 2304) == 21 and pick(px(1), 256, 1024) == 20          # 152 workgroups of 64 x 64: two (K < 2048) / four slabs per barrier
    assert pick(px(1), 512, 9216) == 5                                               # 304 workgroups: more than CUs
    assert pick(px(2), 256, 2304) == 5                                                                               # 300 workgroups: the plain 64 x 64 tile
    assert pick(54 * 75 * 125, 128, 1152) == 22 and pick(54 * 75 * 125, 128, 512) == 22 and pick(54 * 75 * 125, 128, 256) == 3     # res3 3x3 / reduce: 192 x 128 (r05)
    assert pick(3 * 75 * 125, 128, 1152) != 22                                                                        # ... from 200 workgroups of 192 x 128


def test_asm_agpr_guard_ran_for_the_linked_library_and_flags_violations(tmp_path):
    """The hand-scheduled k-loops (gemm.hip tiles 18 / 19) keep their accumulators in literal AGPRs across asm statements; build.py
    checks the device assembly of the SAME sources and flags: no instruction outside the asm blocks of those kernels may touch an
    AGPR.  (a) the report next to the linked library belongs to the current sources and is clean; (b) the scanner does flag a
    compiler-placed accumulator write."""
    import json
    from importlib import import_module
    b = import_module('relation-networks-for-object-detection_amd.build')
    b.build()
    rep = json.load(open(b.GUARD))
    assert rep['digest'] == b._digest() and rep['kernels_checked'] >= 4 and rep['asm_blocks'] > 100 and rep['offenders'] == {}
    k = '_ZN6relnet16gemm_ring_kernelILi256ELi256ELi2ELi4EtLi1ELi64ELi2ELb0ELi6ELi0ELi0EEEvNS_8GemmArgsE'
    other = '_ZN6relnet16gemm_ring_kernelILi256ELi256ELi2ELi4EtLi1ELi64ELi2ELb0ELi0ELi0ELi0EEEvNS_8GemmArgsE'
    asm = '\n'.join([k + ':', '\tv_mov_b32 v1, v2', '\t;;#ASMSTART', '\tv_mfma_f32_32x32x16_bf16 a[0:15], v[72:75], v[80:83], a[0:15]',
                     '\t;;#ASMEND', '\tv_accvgpr_write_b32 a5, v3   ; a spill the compiler parked in an accumulator', '\ts_endpgm',
                     other + ':', '\tv_accvgpr_read_b32 v0, a1', '\ts_endpgm', ''])
    f = tmp_path / 'k.s'
    f.write_text(asm)
    r = b.asm_agpr_guard(str(f))
    assert r['kernels_checked'] == 1 and list(r['offenders']) == [k] and 'a5' in r['offenders'][k][0]


def test_library_binds_to_the_hip_runtime_torch_ships():
    """`build()` before `import torch` in one process (the driver's `build(); smoke()`): librelnet_hip.so must not pull /opt/rocm's libamdhip64 in
    beside the copy the torch wheel ships -- with two HIP runtimes in the process every launch fails with "no ROCm-capable device is
    detected" (round 6).  lib.load() imports torch first; exactly one libamdhip64 may be mapped afterwards."""
    import subprocess
    import sys
    code = (
        "import sys\n"
        "assert 'torch' not in sys.modules\n"
        "import __graft_entry__ as g\n"
        "g.build()\n"
        "import torch\n"
        "paths = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l))\n"
        "print('HIPLIBS', len(paths), paths)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('HIPLIBS')][-1]
    assert line.split()[1] == '1', line

#!/usr/bin/env python
"""Headline benchmark: images/s of the relation-network detector forward pass.

  python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over a batch of `--batch` (default 108) synthetic 600x1000 images per
GPU: ResNet-101 conv1..conv5 + RPN -> proposal (300 rois) -> ROIPooling -> 2FC + 2 relation
modules (16 heads, d=1024) -> cls/bbox -> decode -> per-class soft-NMS -> top-100
(BASELINE.json configs[1], bf16).  Inputs are resident in HBM before the timed region.
Images are independent units: ranks share nothing on the data path (weak scaling, no
collective); the barrier + max-over-ranks timing is the only communication.

The JSON line also carries
  roofline      the relation-attention kernel (north_star's named kernel), stated on the roof that bounds it: HBM.  `achieved` =
                algorithmic bytes per launch (SURVEY.md 8d with the geometry fused: 3.12 MB per image) / launch duration (HIP events on
                the launching stream), `peak` 8000 GB/s, `traffic` = PMC bytes; `mfma_frac_as_written` (3.13 GFLOP per module-image as
                the graph is written / 2.5 PFLOP/s) and `frac_executed` (the FLOPs the re-associated kernel performs) beside it.
  cpu_baseline  the CPU oracle (numpy + torch-CPU fp32 restatement of the same graph): 3 warm-up + 10 timed
                images, median, on this host's cores (N = 1 only).
  parity        the timed detector, on images of the timed batch, checked stage by stage against the oracle
                (oracle/parity.py): identical proposal rows, ROIPooling mismatches, max |cls_prob| error,
                detection-set agreement (N = 1 only).
  batch_sweep   images/s of the same step at 1 (the reference's BATCH_IMAGES), 8 and 54 (the default of rounds 1-4) images per GPU per step.
  train         BASELINE configs[2] on the same N GPUs: training step of relation + learn-NMS end2end with ONE
                summed RCCL all-reduce of the 68.3 M gradients per step (the path the 1 -> 8 GPU scaling target
                is stated on; `value` stays the inference figure so that the N = 1..8 curve is one metric).
"""
import argparse
import contextlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_GFLOP_PER_MODULE_IMAGE = (2 * 16 * 300 * 300 * 64 + 2 * 16 * 300 * 300 * 1024) / 1e9    # 3.1334
EXEC_GFLOP_PER_MODULE_IMAGE = (2 * 16 * 300 * 300 * 64 * 2) / 1e9                           # 0.3686
PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


class KernelTimer(object):
    """HIP-event brackets around every C-ABI launch (events recorded on torch's current
    stream, which is the stream the kernels are launched on)."""

    def __init__(self):
        self.events = {}

    @contextlib.contextmanager
    def __call__(self, name):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        yield
        e.record()
        self.events.setdefault(name, []).append((s, e))

    def summary(self, by_tag=False):
        out = {}
        for name, evs in self.events.items():
            ms = [s.elapsed_time(e) for s, e in evs]
            key = name if by_tag else name.split(':')[0]
            d = out.setdefault(key, dict(calls=0, total_ms=0.0))
            d['calls'] += len(ms); d['total_ms'] += sum(ms)
        for d in out.values():
            d['avg_ms'] = d['total_ms'] / d['calls']
        return out


def cpu_baseline(params, relation=True, soft=True, images=10, warm=3, seed=123, threads=32):
    """Oracle (oracle/network.py) on `images` synthetic 600x1000 images after `warm` warm-up images; median s/image
    (SURVEY 8d protocol)."""
    import numpy as np
    from oracle import network as ON
    # 32 threads: the torch-CPU convolutions stop scaling there on the GPU box's host (0.74 s
    # backbone at 16-32 threads, 1.4 s at 64, ~100 s at all 256 hardware threads)
    cores = max(1, min(os.cpu_count() or 1, threads))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(seed)
    im_info = np.array([[600, 1000, 1.0]], np.float32)
    t_img, t_post = [], []
    for i in range(images + warm):                    # first images = warm-up (thread pools, caches)
        data = torch.randn(1, 3, 600, 1000, generator=g)
        t0 = time.time()
        r = ON.detect(data, im_info, params, relation=relation, soft=soft)
        if i >= warm:
            t_img.append(time.time() - t0)
            t_post.append(r['post_seconds'])
    per = sorted(t_img)[len(t_img) // 2]
    return dict(value=1.0 / per, unit='images/s', cores=cores, kind='port',
                sample='%d synthetic 600x1000 images (after %d warm-up images) through oracle/network.py:detect (torch-CPU fp32 convs on '
                       '%d threads + numpy proposal/ROI/relation/soft-NMS), median %.2f s/image (min %.2f, max %.2f), of which per-class '
                       'soft-NMS + top-100 (1 core, numpy) median %.0f ms (random-init scores keep all 300 rois candidates in all 80 '
                       'classes: the worst case; the reference README reports 59 ms for this stage with trained weights on its own box)'
                       % (images, warm, cores, per, min(t_img), max(t_img), 1e3 * sorted(t_post)[len(t_post) // 2]))


def attention_isolated(batch, n_rois, dtype, launches=100, warm=10):
    """SURVEY 8d protocol for the relation-attention kernel on its own: `launches` back-to-back launches after `warm`
    warm-ups on seeded synthetic operands of the bench shape, one HIP-event pair per launch; returns the median in ms."""
    from relnet_amd import ops, relation
    g = torch.Generator().manual_seed(7)
    B, N, H, Mpad = batch, n_rois, 16, (n_rois + 31) // 32 * 32
    qk = (torch.randn(B, N, 2048, generator=g) * 0.5).cuda().to(dtype)
    vwt = torch.zeros(B, 1024, Mpad, device='cuda', dtype=dtype)
    vwt[:, :, :N] = (torch.randn(B, 1024, N, generator=g) * 0.5).cuda().to(dtype)
    half = dtype == torch.bfloat16
    bias = (torch.randn(B, H, N, Mpad, generator=g) - 3.0).cuda().to(torch.float16 if half else torch.float32)
    resid = torch.randn(B, N, 1024, generator=g).cuda().to(dtype)
    bout = torch.zeros(1024, device='cuda')
    run = lambda: ops.relation_attention(qk[:, :, :1024], qk[:, :, 1024:], vwt, bias, bout=bout, resid=resid, M=N,
                                         want_out=False, want_act=True)
    for _ in range(warm):
        run()
    evs = []
    for _ in range(launches):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); run(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in evs)
    return ms[len(ms) // 2]


def _timed_windows(replay, D, per_window, windows=5, warm_seconds=1.0):
    """The measurement protocol of every side figure in the JSON line: >= `warm_seconds` of untimed replays (clock ramp, caches,
    allocator), then `windows` timed windows of `per_window` replays, each between barrier + synchronize fences.
    -> (median, min, max) seconds per replay."""
    t_end = time.perf_counter() + warm_seconds
    n_warm = 0
    while time.perf_counter() < t_end or n_warm < 3:
        replay(); n_warm += 1
        if n_warm % 8 == 0:
            torch.cuda.synchronize()
    per = []
    for _ in range(windows):
        D.fence(device='cuda')
        t0 = time.perf_counter()
        for _ in range(per_window):
            replay()
        D.fence(device='cuda')
        per.append((time.perf_counter() - t0) / per_window)
    per.sort()
    return per[len(per) // 2], per[0], per[-1]


def _replay_rate(det, bsz, a, D, replays=200, windows=5, step=None, make_det=None, slot_step=None, n_flight=3):
    """images/s of `det` at `bsz` images per step under hipGraph replay: one capture, 1 s of warm replays, then `windows` windows of
    `replays / windows` replays each (>= 200 replays in total: a 136-launch graph of ~3 ms is host-launch and clock-ramp
    sensitive, one 20-replay window is not a measurement); median window with min / max next to it."""
    if step is None:
        g = torch.Generator().manual_seed(4242 + bsz)
        data = torch.randn(bsz, 3, 600, 1000, generator=g).cuda()
        im_info = torch.tensor([[600.0, 1000.0, 1.0]] * bsz).cuda()
        step = lambda: det.forward(data, im_info)
    with torch.no_grad():
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode='thread_local'):      # (thread_local: RCCL's watchdog thread may poll events meanwhile)
            step()
        per_window = max(1, replays // windows)
        med, lo, hi = _timed_windows(graph.replay, D, per_window, windows)
    del graph
    res = {'images_per_s': bsz / med, 'ms_per_step': 1e3 * med, 'ms_min': 1e3 * lo, 'ms_max': 1e3 * hi,
           'images_per_s_min': bsz / hi, 'images_per_s_max': bsz / lo,
           'protocol': '%d windows x %d replays after >= 1 s of warm replays; median window (min / max beside it)' % (windows, per_window)}
    if make_det is not None or slot_step is not None:
        # the same protocol with n_flight batches in flight (detector.InFlight): n_flight FRESH detector instances with the RPN branch in line
        # (one hardware queue per captured step), each on its own resident batch and stream.  slot_step(j) -> the step of slot j.
        from relnet_amd import detector as _det
        if slot_step is None:
            def slot_step(j):
                gj = torch.Generator().manual_seed(977 + bsz + 31 * j)
                dj = torch.randn(bsz, 3, 600, 1000, generator=gj).cuda()
                imj = torch.tensor([[600.0, 1000.0, 1.0]] * bsz).cuda()
                detj = make_det(in_line=True)
                return lambda: detj.forward(dj, imj)
        try:          # (a figure beside the figure: its failure must not take the one-at-a-time rate above with it)
            flight = _det.InFlight([slot_step(j) for j in range(n_flight)])
            med2, lo2, hi2 = _timed_windows(flight.submit, D, per_window, windows)
            res['in_flight'] = {'batches_in_flight': n_flight, 'images_per_s': bsz / med2, 'ms_per_step_by_throughput': 1e3 * med2, 'images_per_s_min': bsz / hi2,
                                'images_per_s_max': bsz / lo2,
                                'note': '%d captured steps (own detector instance and resident batch each, RPN branch in line) on %d streams, submitted round-robin; every step '
                                        'still takes its own ~ms_per_step above (or longer) from launch to result' % (n_flight, n_flight)}
            del flight
        except Exception as ex:      # noqa: BLE001
            res['in_flight'] = {'error': '%s: %s' % (type(ex).__name__, str(ex)[:300])}
            torch.cuda.synchronize()
    return res


def _side_figure(fn, what):
    """Run one SIDE figure of the default line (training rates, other configs).  An exception there -- out of memory, a collective
    error, a diverged 5-step random-init run -- is recorded as {'error': ...} under the figure's key instead of losing the headline
    line.  Every rank calls the same sequence of side figures, so a failure that every rank sees (the reduced
    non-finite check) leaves them in step; a one-rank failure inside a collective cannot be repaired from here."""
    res, err = None, None
    try:
        res = fn()
    except (Exception, SystemExit) as ex:          # noqa: BLE001 -- the line must survive any side figure
        sys.stderr.write('bench.py: side figure %s failed: %s: %s\n' % (what, type(ex).__name__, str(ex)[:300]))
        try:
            torch.cuda.empty_cache()
        except Exception:
            pass
        err = {'error': '%s: %s' % (type(ex).__name__, str(ex)[:300])}
    # every rank learns whether ALL ranks got through: a figure that failed on one rank only is dropped on every rank, so
    # that they enter the next side figure's collectives together; a rank that died INSIDE a collective is caught by the process
    # group's timeout (dist.init: 600 s) instead
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        from relnet_amd import dist as _D
        n_ok = _D.sum_over_ranks(0.0 if err else 1.0, device='cuda')
        if err is None and n_ok != float(torch.distributed.get_world_size()):
            err = {'error': 'side figure %s failed on %d other rank(s)' % (what, int(torch.distributed.get_world_size() - n_ok))}
    return err if err is not None else res


def _fpn_proposals(batch, n_rois, im_h, im_w, g):
    """Given proposals of the FPN graphs (HAS_RPN: false): log-uniform sizes over all pyramid levels."""
    side = torch.exp(torch.empty(batch, n_rois).uniform_(math.log(16), math.log(640), generator=g))
    ar = torch.exp(torch.empty(batch, n_rois).uniform_(-0.7, 0.7, generator=g))
    bw, bh = (side * ar).clamp(max=im_w - 2), (side / ar).clamp(max=im_h - 2)
    x1 = torch.rand(batch, n_rois, generator=g) * (im_w - 1 - bw)
    y1 = torch.rand(batch, n_rois, generator=g) * (im_h - 1 - bh)
    return torch.stack([x1, y1, x1 + bw, y1 + bh], 2).cuda()


#: (dcn, fpn) -> the experiment file whose values the run takes (relnet_amd/config.py:EXPERIMENTS; all three are the relation + learn-NMS rows)
EXPERIMENT_OF = {(False, False): 'rcnn_end2end_relation_learn_nms_8epoch', (True, False): 'rcnn_dcn_end2end_relation_learn_nms_8epoch',
                 (False, True): 'rcnn_fpn_relation_learn_nms_8epoch'}


def other_configs(a, rank, world, D):
    """BASELINE configs[3] / configs[4] AS WORDED (DCN / FPN + relation + learn-NMS), inference graph and training step, timed by
    this process so that the driver's line carries them (rank 0 returns the block; every rank takes part in the training
    all-reduce).  Short runs: they are side figures, each with its own protocol string."""
    from relnet_amd import backbone, detector
    out = {'note': 'BASELINE configs[3] (Deformable Faster-RCNN + relation + learn-NMS) and configs[4] (FPN + relation + learn-NMS, 1000 '
                   'proposals, 800x1024) with the hyper-parameters of their experiment files (config.EXPERIMENTS -> Config.from_experiment: '
                   'configs[4] = FIRST_N 150, LEARN_NMS_CLASS_SCORE_TH 0.05, BATCH_ROIS_OHEM 512, ..._rcnn_fpn_relation_learn_nms_8epoch.yaml:92,141,166-167; '
                   'configs[3] = FIRST_N 100, 0.01, OHEM 128); configs[4] says "fp16 MFMA stress": run here with bf16 operands -- MEASURED (r05, '
                   'profiles/r05_notes/fp16_vs_bf16.txt, relnet_gemm_nt_f16): the same kernel with fp16 operands (v_mfma_f32_32x32x16_f16, same 8-pass '
                   'instruction) runs 8-13 % slower on dense random data (power-limited pipes, 10 toggling mantissa bits against 7), so bf16 is the faster '
                   '16-bit format here; fp16 is used for the geometry-bias operand of the attention kernel only',
           'operand_dtype': 'bf16 (fp16 twin of the GEMM kernel measured 8-13 % slower: profiles/r05_notes/fp16_vs_bf16.txt)'}
    for key, dcn, fpn, bsz in (('configs3_dcn_relation_learn_nms_inference', True, False, 27),
                               ('configs4_fpn_relation_learn_nms_inference', False, True, 8)):
        params = backbone.init_params(seed=1, dcn_offset_std=0.01 if dcn else 0.0, fpn=fpn)
        cfg = detector.Config.from_experiment(EXPERIMENT_OF[(dcn, fpn)])
        assert cfg.learn_nms and cfg.dcn == dcn
        im_h, im_w = (800, 1024) if fpn else (600, 1000)
        im_info = torch.tensor([[float(im_h), float(im_w), 1.0]] * bsz).cuda()

        def slot(seed, in_line=False):            # one detector instance + its resident batch -> (detector, step)
            gs = torch.Generator().manual_seed(seed)
            d_ = torch.randn(bsz, 3, im_h, im_w, generator=gs).cuda()
            if fpn:
                dt = detector.FPNDetector(params, dtype=torch.bfloat16, device='cuda', cfg=cfg)
                pr = _fpn_proposals(bsz, 1000, im_h, im_w, gs)
                return dt, (lambda: dt.forward(d_, pr, im_info))
            dt = detector.Detector(params, dtype=torch.bfloat16, device='cuda', cfg=cfg)
            if in_line:
                dt.overlap_rpn = False
            return dt, (lambda: dt.forward(d_, im_info))
        det, step = slot(77 + rank)
        nfl = getattr(a, 'in_flight', 1)
        r = _side_figure(lambda: _replay_rate(det, bsz, a, D, replays=30, windows=3, step=step, n_flight=nfl,
                                              slot_step=(lambda j: slot(1077 + rank + 13 * j, True)[1]) if nfl > 1 else None), key)
        if 'error' not in r:
            r.update(images_per_gpu_per_step=bsz, n_gpus=world, images_per_s_all_gpus=r['images_per_s'] * world,
                     experiment=cfg.experiment, first_n=cfg.first_n, class_thresh=cfg.learn_nms_class_thresh)
        out[key] = r
        del det, step
        torch.cuda.empty_cache()
    # training steps: 8 images per GPU and step like the headline training figure (configs[2]); round 6: also for the FPN graph, whose
    # 2-image step of rounds 2-5 (6 400-pixel res4 maps: every convolution a launch of < 1 workgroup per CU) stays beside it as `at_2_images_per_gpu`
    # (round 6) + the reference README's fourth experiment: the learn-NMS-only step (detector fixed by FIXED_PARAMS, plain 2FC head, no OHEM)
    for key, dcn, fpn, bsz, also in (('configs3_dcn_relation_learn_nms_training', True, False, 8, ()),
                                     ('configs4_fpn_relation_learn_nms_training', False, True, 8, (2,)),
                                     ('rcnn_end2end_learn_nms_3epoch_training', False, False, 8, (1,))):
        def one(bsz_):
            ta = argparse.Namespace(**vars(a))
            ta.batch, ta.learn_nms, ta.dcn, ta.fpn, ta.steps, ta.warmup, ta.no_graph = bsz_, True, dcn, fpn, 5, 2, False
            ta.experiment = 'rcnn_end2end_learn_nms_3epoch' if key.startswith('rcnn_end2end_learn_nms_3epoch') else None
            tr = _side_figure(lambda: bench_train(ta, rank, world, D, emit=False, fatal=False), key)
            if rank != 0:
                return None
            r = tr if (tr is None or 'error' in tr) else dict({k: tr[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'n_gpus', 'weights_finite_on_all_ranks')},
                                                               **{k: tr['config'][k] for k in ('experiment', 'first_n', 'ohem', 'lr', 'lr_rule', 'relation_bwd_of_the_learn_nms_head')})
            if r is not None:
                r['images_per_gpu_per_step'] = bsz_
            return r
        main = one(bsz)
        extra = {('at_%d_image%s_per_gpu' % (b_, '' if b_ == 1 else 's')): one(b_) for b_ in also}
        if rank == 0:
            out[key] = main
            if main is not None:
                for k_, v_ in extra.items():
                    main[k_] = v_ if (v_ is None or 'error' in v_) else {k: v_[k] for k in ('value', 'ms_per_step', 'steps', 'lr')}
    return out if rank == 0 else None


def bench_lr(yaml_lr, images_per_gpu, world):
    """Learning rate of the training benchmark.  The gradient is SUMMED over every image of the step (rescale_grad 1.0,
    train_end2end.py:167) and the yaml's lr is quoted for 4 images per step on ImageNet-initialised weights.  The benchmark's
    weights are random-init: beyond 16 summed images (the largest step measured finite at the yaml's lr; 4 ranks x 8 images were
    not) the rate is scaled down linearly so that B x world images move the weights as far as 16 do.  The arithmetic of a step
    does not depend on the value."""
    return yaml_lr * min(1.0, 16.0 / float(images_per_gpu * world))


def bench_train(a, rank, world, D, emit=True, fatal=True):
    """Training throughput of the relation end2end graph (reference config ..._end2end_relation_8epoch.yaml): one step =
    forward + backward over `batch` images per GPU, ONE summed all-reduce of the 67.7 M trainable gradients, SGD."""
    import numpy as np
    from relnet_amd import backbone, train
    H, W, G = (800, 1024, 8) if a.fpn else (600, 1000, 8)
    B = a.batch
    params = backbone.init_params(seed=1, dcn_offset_std=0.005 if a.dcn else 0.0, fpn=a.fpn)
    if getattr(a, 'experiment', None):     # any shipped experiment by its name (config.EXPERIMENTS), e.g. rcnn_end2end_learn_nms_3epoch = the learn-NMS-only step
        cfg = train.TrainConfig.from_experiment(a.experiment, train=True)
        assert cfg.dcn == bool(a.dcn) and cfg.fpn == bool(a.fpn), '--experiment %s needs --dcn %s --fpn %s' % (a.experiment, cfg.dcn, cfg.fpn)
    elif a.learn_nms:     # the experiment file's own values: FPN = FIRST_N 150 / OHEM 512 / lr 0.00125, C4 and DCN = 100 / 128 / 0.0005
        cfg = train.TrainConfig.from_experiment(EXPERIMENT_OF[(bool(a.dcn), bool(a.fpn))], train=True)
        assert cfg.learn_nms and cfg.dcn == bool(a.dcn)
    else:
        cfg = train.TrainConfig()
        cfg.experiment = 'rcnn_end2end_relation_8epoch'
        cfg.dcn = a.dcn
    yaml_lr = cfg.lr
    cfg.lr = bench_lr(cfg.lr, a.batch, world)
    tr = train.FPNTrainer(params, cfg) if a.fpn else train.Trainer(params, cfg, im_hw=(H, W))
    g = torch.Generator().manual_seed(1000 + rank)
    data = torch.randn(B, 3, H, W, generator=g).cuda()
    im_info = torch.tensor([[float(H), float(W), 1.0]] * B).cuda()
    rng = np.random.default_rng(2 + rank)
    gt = np.zeros((B, G, 5), np.float32)
    for b in range(B):                      # 8 gt boxes, w,h in [32,400], classes uniform in 1..80
        bw, bh = rng.uniform(32, 400, G), rng.uniform(32, 400, G)
        x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
        gt[b] = np.stack([x1, y1, x1 + bw, y1 + bh, rng.integers(1, 81, G)], 1)
    if a.fpn:       # proposals are an input of the FPN graphs (HAS_RPN: false, TOP_ROIS 1000): log-uniform sizes over all levels
        n_rois = 1000
        side = torch.exp(torch.empty(B, n_rois).uniform_(math.log(16), math.log(640), generator=g))
        ar = torch.exp(torch.empty(B, n_rois).uniform_(-0.7, 0.7, generator=g))
        bw_, bh_ = (side * ar).clamp(max=W - 2), (side / ar).clamp(max=H - 2)
        x1_ = torch.rand(B, n_rois, generator=g) * (W - 1 - bw_); y1_ = torch.rand(B, n_rois, generator=g) * (H - 1 - bh_)
        batch = (data, im_info, torch.as_tensor(gt).cuda(), torch.stack([x1_, y1_, x1_ + bw_, y1_ + bh_], 2).cuda())
    else:       # RPN anchor labels / targets (lib/rpn/rpn.py:assign_anchor, host numpy in the reference's loader) are computed
        batch = (data, im_info, torch.as_tensor(gt).cuda())       # on the device INSIDE the step (relnet_assign_anchor)

    def fence():
        D.fence(device='cuda')

    with torch.no_grad():
        for _ in range(a.warmup):
            out = tr.step(*batch)
        graph = None
        if not a.no_graph:
            # forward + backward as a chain of hipGraphs cut at the gradient buckets (train.CapturedStep): every bucket's
            # all-reduce is issued between two graph launches and overlaps the rest of the backward pass; SGD stays eager
            graph = train.CapturedStep(tr, batch)
            out = graph.out
        def one_step():
            nonlocal out
            if graph is not None:
                graph.replay()
            else:
                out = tr.forward_backward(*batch)
            tr.all_reduce(wait=False)      # launches the buckets the backward pass has not announced; update() waits bucket by bucket
            tr.update()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            one_step()
        fence()
        elapsed = time.perf_counter() - t0
        comm = None
        if world > 1:       # self-diagnosis of the N > 1 line (verdict r05 item 8), OUTSIDE the timed region: the same steps again with HIP
            tr.comm_timing = []               # events around every bucket's wait -> exposed communication per step and per bucket
            n_c = min(a.steps, 5)
            for _ in range(n_c):
                one_step()
            fence()
            comm = tr.comm_report(n_c)
            tr.comm_timing = None
        nprobe = int(os.environ.get('RELNET_BENCH_RECAPTURE', '0'))
        if nprobe and graph is not None:          # diagnostic, outside the timed region: does the replay time depend on the CAPTURE?
            for i in range(nprobe):
                g2 = train.CapturedStep(tr, batch)
                for _ in range(3):
                    g2.replay()
                fence(); t1 = time.perf_counter()
                for _ in range(20):
                    g2.replay()
                fence()
                print('recapture %d: %.3f ms per forward+backward replay' % (i, (time.perf_counter() - t1) / 20 * 1e3), file=sys.stderr)
                del g2
    elapsed = D.max_over_ranks(elapsed, device='cuda')
    if comm is not None:
        comm['exposed_comm_ms_per_step_max_over_ranks'] = D.max_over_ranks(comm['exposed_comm_ms_per_step'], device='cuda')
    ok = bool(torch.isfinite(tr.W.master).all())
    if not ok:
        sys.stderr.write('bench.py: rank %d: non-finite weights after the training steps\n' % rank)
    ok = D.sum_over_ranks(float(ok), device='cuda') == float(world)
    if not ok and not os.environ.get('RELNET_BENCH_ONE_DEVICE'):
        # a diverged run must not be recorded as a throughput number (every rank sees the same `ok`: they leave together).  The
        # stand-alone --train run exits non-zero; a SIDE figure of the default line (fatal=False) is dropped with the reason recorded,
        # so that it cannot take the headline line down with it
        msg = 'non-finite weights on at least one rank after %d training steps: refusing to report a rate' % a.steps
        if fatal:
            raise SystemExit('bench.py: ' + msg)
        del tr
        torch.cuda.empty_cache()
        return {'error': msg} if rank == 0 else None
    res = None
    if rank == 0:
        images = world * B * a.steps
        res = {
            'metric': 'images/sec (1000x600, 300 ROIs)', 'value': images / elapsed, 'unit': 'images/s', 'n_gpus': world,
            'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': (('experiment %s (%s): ' % (a.experiment, 'learn-NMS head only: detector fixed by FIXED_PARAMS, JOINT_TRAINING false' if getattr(tr, 'lnms_only', False) else 'its own hyper-parameters')) if getattr(a, 'experiment', None) else '') +
                                   ('BASELINE configs[4] (FPN, 800x1024 images, 1000 given proposals + 8 gt rows): ' if a.fpn else '') +
                                   ('BASELINE configs[3] (deformable res5 + deformable PSROI pooling): ' if a.dcn else '') +
                                   ('BASELINE configs[2]: TRAINING step of ResNet-101 Faster-RCNN + 2 relation modules + learn-NMS '
                                    'head end2end (..._rcnn_end2end_relation_learn_nms_8epoch.yaml)' if a.learn_nms else
                                    'TRAINING step of ResNet-101 Faster-RCNN + 2 relation modules end2end '
                                    '(..._rcnn_end2end_relation_8epoch.yaml)') +
                                   ': forward + backward + summed all-reduce of %d gradients + SGD, %dx%d images, '
                                   '%d proposals + 8 gt rows, OHEM %d, learn-NMS first_n %d, random-init weights'
                                   % (tr.num_trainable(), H, W, 1000 if a.fpn else 300, cfg.batch_rois_ohem, cfg.first_n),
                       'experiment': cfg.experiment, 'first_n': cfg.first_n, 'ohem': cfg.batch_rois_ohem,
                       'relation_bwd_of_the_learn_nms_head': ('one workgroup per (image, class, head), S / dL in LDS (relation_attention_bwd_small_kernel: first_n <= 128)'
                                                             if cfg.first_n <= 128 else
                                                             'two-kernel form with fp32 S / dL maps in HBM (first_n %d -> Mpad %d is past the small kernel\'s 128)' % (cfg.first_n, (cfg.first_n + 31) // 32 * 32))
                                                            if cfg.learn_nms else None,
                       'fixed_params': list(getattr(cfg, 'fixed_params', None) or []), 'learn_nms_only_step': bool(getattr(tr, 'lnms_only', False)),
                       'images_per_gpu_per_step': B, 'launch': 'eager' if a.no_graph else ('hipGraph replay (forward+backward in %d segments cut at the gradient buckets)' % len(graph.segments) if len(graph.segments) > 1 else 'hipGraph replay (forward+backward as one graph: one rank, no bucket exchange to cut for)'),
                       'parallelism': 'dp%d (RCCL all-reduce SUM)' % world, 'lr': cfg.lr,
                       'lr_rule': 'yaml lr %g x min(1, 16 / images summed per step over all ranks)' % yaml_lr},
            'communication': comm if comm is not None else 'single rank: no collective',
            'losses': {k: float(out[k]) for k in ('bbox_loss', 'rpn_bbox_loss', 'nms_pos_loss', 'nms_neg_loss') if k in out},
            'weights_finite_on_all_ranks': bool(ok)}
        if emit:
            print(json.dumps(res))
    del tr
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='images per GPU per step (default 108 for the headline graph, 54 for the other inference graphs; 8 with --train)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-overlap', action='store_true', help='A/B: RPN head + proposal on the main stream instead of beside res5')
    ap.add_argument('--no-chain', action='store_true', help='A/B: res2 block boundaries as two convolution launches instead of relnet_bottleneck_chain')
    ap.add_argument('--no-relation', action='store_true', help='plain 2FC head (config 1 graph)')
    ap.add_argument('--learn-nms', action='store_true', help='learned duplicate removal instead of soft-NMS (config 3 graph, inference)')
    ap.add_argument('--dcn', action='store_true', help='deformable res5 + deformable PSROI pooling (config 4 graph, inference)')
    ap.add_argument('--fpn', action='store_true', help='FPN graph, 800x1024 images, 1000 given proposals (inference graph of config 5)')
    ap.add_argument('--train', action='store_true', help='training step (relation end2end graph): forward + backward + summed all-reduce + SGD')
    ap.add_argument('--experiment', default=None, help='with --train: build the step from this shipped experiment (config.EXPERIMENTS key, e.g. rcnn_end2end_learn_nms_3epoch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-images', type=int, default=10)
    ap.add_argument('--parity-images', type=int, default=2, help='images of the batch checked stage by stage against the oracle')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-batch-sweep', action='store_true')
    ap.add_argument('--no-train-line', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the configs[3] / configs[4] (+ learn-NMS) side figures of the default line')
    ap.add_argument('--cpu-threads', type=int, default=32)
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--stem', default='hip', choices=['hip', 'hip3', 'miopen'], help="stem: 'hip' one fused conv1 + ReLU + pool1 kernel, 'hip3' the three-launch form, 'miopen' library 7x7")
    ap.add_argument('--no-graph', action='store_true', help='time eager launches instead of hipGraph replays')
    ap.add_argument('--in-flight', type=int, default=3, help='side figures of the inference line: batches in flight (captured steps with the RPN branch in line, each on its own stream, submitted round-robin; 1 = none). `value` is always the one-at-a-time rate')
    ap.add_argument('--shapes', action='store_true', help='print per-shape GEMM/conv times to stderr')
    ap.add_argument('--head-init-std', type=float, default=0.05,
                    help='std of the random cls_score / bbox_pred weights (reference init: 0.01, which makes every class posterior '
                         '~1/81 and the parity block blind to the head; the cost of the step does not depend on it)')
    ap.add_argument('--stub', action='store_true', help='launcher dry run: gloo ranks on CPU, a stub step instead of the detector')
    a = ap.parse_args()
    if a.batch is None:
        # 108 images: 1010 row tiles of 256 on the res4 maps = 3.95 rounds over 256 CUs (54: 505 = 1.97 rounds) and half the per-launch
        # ramps per image: +2 % images/s over 54 on the same box (r04); the other graphs keep 54 (27 / 8 in other_configs)
        a.batch = 8 if a.train else (108 if not (a.dcn or a.fpn or a.learn_nms or a.no_relation) else 54)

    # `python bench.py --gpus N` with no torchrun environment: start the N ranks ourselves (one process per GPU, RCCL over
    # xGMI) -- the reference trains over len(ctx) devices from one command too (train_end2end.py:69-71)
    from importlib import import_module
    launcher = import_module('relation-networks-for-object-detection_amd.launch')
    if a.gpus > 1 and not launcher.under_torchrun():
        sys.exit(launcher.respawn(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    if a.stub:
        return launcher.stub_bench(a)

    import __graft_entry__ as ge
    ge.build()
    import relnet_amd  # noqa: F401
    from relnet_amd import lib, backbone, detector
    from relnet_amd import dist as D

    local = int(os.environ.get('LOCAL_RANK', '0'))
    # test aid (one-GPU boxes): RELNET_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and exchanges over gloo, so that the whole N > 1
    # code path -- launcher, rank count, per-rank graphs, fences, segmented training step with bucketed all-reduces, one JSON line
    # from rank 0 -- runs without a second GPU; the numbers it prints are not N-GPU numbers and say so
    one_dev = bool(os.environ.get('RELNET_BENCH_ONE_DEVICE'))
    torch.cuda.set_device(0 if one_dev else local)
    rank, world, local = D.init(backend='gloo' if one_dev else 'nccl')          # 'nccl' = RCCL over xGMI
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) came up (WORLD_SIZE): refusing to report a %d-GPU number"
                         % (a.gpus, world, a.gpus))
    ranks_seen = int(D.sum_over_ranks(1, device='cuda'))   # counted THROUGH the collective library, not read from the environment
    assert ranks_seen == a.gpus, (ranks_seen, a.gpus)

    if a.train:
        bench_train(a, rank, world, D)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return
    tdt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
    params = backbone.init_params(seed=1, dcn_offset_std=0.01 if a.dcn else 0.0, fpn=a.fpn)
    if a.head_init_std != 0.01:
        gh = torch.Generator().manual_seed(5)
        for k in ('cls_score_weight', 'bbox_pred_weight'):
            params[k] = torch.randn(params[k].shape, generator=gh) * a.head_init_std
    if a.learn_nms:
        cfg = detector.Config.from_experiment(EXPERIMENT_OF[(bool(a.dcn), bool(a.fpn))])
    else:
        cfg = detector.Config()
        cfg.dcn = a.dcn
    def make_det(in_line=False):
        if a.fpn:
            det = detector.FPNDetector(params, dtype=tdt, device='cuda', relation=not a.no_relation, cfg=cfg, stem=a.stem)
        else:
            det = detector.Detector(params, dtype=tdt, device='cuda', relation=not a.no_relation, cfg=cfg, stem=a.stem)
        if (a.no_overlap or in_line) and hasattr(det, 'overlap_rpn'):
            det.overlap_rpn = False          # (in_line: a step in flight beside others keeps to ONE hardware queue)
        if a.no_chain:
            from relnet_amd import ops as _ops
            bb = det.backbone
            bb.chain, bb.halo3, bb.chain_proj = {}, {}, {}
            for name, (w, _, k) in bb.wp.items():          # res4 expand layers back on the row-panel kernel (the pre-fusion state)
                if name.endswith('_branch2c') and k == 1 and w.shape[1] == 256 and w.shape[0] % 256 == 0:
                    bb.wf[name] = _ops.pack_w_frag(w)
        return det
    det = make_det()
    g = torch.Generator().manual_seed(1000 + rank)
    # unit-variance synthetic pixels: with random-init weights (no checkpoints offline) this gives
    # O(1) RPN logits/deltas, i.e. several hundred distinct proposals survive NMS per image; N(0,50)
    # pixels would push every delta past exp overflow and degenerate all rois to the full image.
    # the step starts from the raw fp32 NCHW image batch (dtype/layout conversion is part of the step)
    im_h, im_w, n_rois = (800, 1024, 1000) if a.fpn else (600, 1000, 300)
    data = torch.randn(a.batch, 3, im_h, im_w, generator=g).cuda()
    im_info = torch.tensor([[float(im_h), float(im_w), 1.0]] * a.batch).cuda()
    torch.backends.cudnn.benchmark = True
    if a.fpn:      # proposals are an input of the FPN graphs (HAS_RPN: false): log-uniform sizes over all levels
        side = torch.exp(torch.empty(a.batch, n_rois).uniform_(math.log(16), math.log(640), generator=g))
        ar = torch.exp(torch.empty(a.batch, n_rois).uniform_(-0.7, 0.7, generator=g))
        bw, bh = (side * ar).clamp(max=im_w - 2), (side / ar).clamp(max=im_h - 2)
        x1 = torch.rand(a.batch, n_rois, generator=g) * (im_w - 1 - bw)
        y1 = torch.rand(a.batch, n_rois, generator=g) * (im_h - 1 - bh)
        proposals = torch.stack([x1, y1, x1 + bw, y1 + bh], 2).cuda()

    def step():
        if a.fpn:
            return det.forward(data, proposals, im_info)
        return det.forward(data, im_info)

    def fence():
        D.fence(device='cuda')

    with torch.no_grad():
        for _ in range(a.warmup):
            out = step()
        graph = None
        if not a.no_graph:
            # the whole step (~190 launches, no host synchronisation inside) as ONE hipGraph:
            # the Python/ctypes launch path costs ~50 us per kernel, i.e. ~10 ms per step eager
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                out = step()
            graph.replay()
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            if graph is not None:
                graph.replay()
            else:
                out = step()
        fence()
        elapsed = time.perf_counter() - t0
        # throughput mode (detector.InFlight), a SIDE figure of the line (`in_flight`): the same K steps with two batches in flight -- a second
        # detector instance (own buffers), a second resident batch, each step captured on its own stream, steps submitted round-robin.
        # `value` stays the one-at-a-time rate of rounds 1-5
        elapsed_fl, n_flight, flight_err = None, 1, None
        if graph is not None and a.in_flight > 1 and world == 1:      # (one rank only: a side figure must not be able to leave other ranks in a barrier)
            del graph                 # (its executable graph holds runtime streams: with it alive the graphs below measured no overlap at all)
            graph = True
            torch.cuda.synchronize()

            def extra_step(j):
                gj = torch.Generator().manual_seed(1000 + rank + 7919 * (j + 1))
                dj, detj = torch.randn(a.batch, 3, im_h, im_w, generator=gj).cuda(), make_det(in_line=True)
                return (lambda: detj.forward(dj, proposals, im_info)) if a.fpn else (lambda: detj.forward(dj, im_info))
            try:                      # a side figure: whatever happens here, the headline line above survives
                flight = detector.InFlight([extra_step(j) for j in range(a.in_flight)])
                for _ in range(2 * len(flight)):
                    flight.submit()
                fence()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    flight.submit()
                fence()
                elapsed_fl, n_flight = time.perf_counter() - t0, len(flight)
                n_fl = int(flight.result(0)['num_detections'].sum().item())
                assert n_fl > 0 or a.learn_nms
                del flight
            except Exception as ex:   # noqa: BLE001
                sys.stderr.write('bench.py: in-flight side figure failed: %s: %s\n' % (type(ex).__name__, str(ex)[:300]))
                elapsed_fl, flight_err = None, '%s: %s' % (type(ex).__name__, str(ex)[:300])
                torch.cuda.synchronize()
        # per-kernel durations: the same step launched eagerly with HIP events around every C-ABI
        # launch (events cannot be timed inside a captured graph); same inputs, same stream
        timer = None if a.no_kernel_timing else KernelTimer()
        if timer is not None:
            lib.timing_hook = timer
            for _ in range(min(a.steps, 5)):
                step()
            torch.cuda.synchronize()
            lib.timing_hook = None
    elapsed = D.max_over_ranks(elapsed, device='cuda')
    if elapsed_fl is not None:
        elapsed_fl = D.max_over_ranks(elapsed_fl, device='cuda')
    n_det = int(out['num_detections'].sum().item())
    # (random-init learn-NMS logits start at sigmoid(-3) x ~1/81 < 1e-3: zero detections is expected there)
    assert (n_det > 0 or a.learn_nms) and bool(torch.isfinite(out['cls_score']).all())

    plain_graph = not (a.dcn or a.fpn or a.learn_nms or a.no_relation) and a.dtype == 'bf16'
    res = None
    if rank == 0:
        images = world * a.batch * a.steps
        res = {
            'metric': 'images/sec (1000x600, 300 ROIs)', 'value': images / elapsed, 'unit': 'images/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype,
            'data': 'synthetic',
            'config': {'workload': '%sResNet-101 Faster-RCNN + %s + %s + top-100, 600x1000 images, '
                                   '300 proposals, random-init weights'
                                   % ('BASELINE configs[1]: ' if not (a.dcn or a.learn_nms or a.no_relation or a.fpn) else
                                      ('inference graph of BASELINE configs[3] (DCN): deformable ' if a.dcn else
                                       'inference graph of BASELINE configs[4] (FPN, 800x1024 images, 1000 given proposals): ' if a.fpn else ''),
                                      '2 relation modules (N=300, 16 heads, d=1024)' if not a.no_relation else 'plain 2FC head',
                                      'learn-NMS (first_n %d, class_thresh %g, 80 classes)' % (cfg.first_n, cfg.learn_nms_class_thresh) if a.learn_nms else 'soft-NMS(0.6)'),
                       'images_per_gpu_per_step': a.batch,
                       'launch': 'eager' if a.no_graph else 'hipGraph replay', 'batches_in_flight': 1, 'parallelism': 'replicas x%d (no data-path collective)' % world,
                       'cross_round_figure': "`value` is quoted at %d images per GPU per step (the default since the end of round 4); rounds 1-3 quoted 54: compare those with batch_sweep['54']" % a.batch,
                       'ranks_seen_by_rccl': ranks_seen, 'head_init_std': a.head_init_std,
                       'precision': 'bf16 operands, fp32 accumulation (fp16 only for the log2 geometry bias read by the attention kernel); BASELINE '
                                    'configs[4] words its FPN run as "fp16": it is run with bf16 operands here -- the fp16 twin of the GEMM kernel was '
                                    'measured 8-13 % SLOWER on gfx950 (same 8-pass MFMA, power-limited pipes; profiles/r05_notes/fp16_vs_bf16.txt)',
                       'scaling_figure': ('`value` is the replica-inference rate (no data-path collective: linear by construction). The 1 -> N GPU '
                                          'scaling north_star targets is the TRAINING step with its RCCL gradient all-reduce: read `train.value` '
                                          '(and other_configs.*_training.value) across N') if world > 1 else
                                         'single GPU; at N > 1 the data-parallel scaling figure is train.value (training step incl. the RCCL all-reduce)',
                       **({'one_device_test': 'all %d ranks share cuda:0 and exchange over gloo (RELNET_BENCH_ONE_DEVICE): a code-path check, NOT a multi-GPU number' % world} if one_dev else {})},
        }
        if flight_err is not None:
            res['in_flight'] = {'error': flight_err}
        if elapsed_fl is not None:
            res['in_flight'] = {'batches_in_flight': n_flight, 'value': images / elapsed_fl, 'unit': 'images/s', 'ms_per_step_by_throughput': 1e3 * elapsed_fl / a.steps,
                                'steps': a.steps, 'vs_one_at_a_time': elapsed / elapsed_fl,
                                'note': 'the same K steps with %d batches in flight (detector.InFlight: %d captured steps -- own detector instance and resident batch each, RPN '
                                        'branch in line = one hardware queue per step -- submitted round-robin on %d streams); a step then takes longer from launch to result. '
                                        "Pays most where one step leaves the chip idle: see batch_sweep[*]['in_flight']" % (n_flight, n_flight, n_flight)}
        if timer is not None:
            ks = timer.summary()
            if a.shapes:
                for k, v in sorted(timer.summary(by_tag=True).items(), key=lambda kv: -kv[1]['total_ms']):
                    sys.stderr.write('%-60s calls/step %5.1f avg %8.1f us  per-step %8.3f ms\n' % (
                        k, v['calls'] / min(a.steps, 5), v['avg_ms'] * 1e3, v['total_ms'] / min(a.steps, 5)))
            res['kernels_ms'] = {k: round(v['avg_ms'], 5) for k, v in sorted(ks.items())}
            # the kernels that decide images/s: per convolution shape of the step, the measured launch time against both roofs
            # (2 M N K FLOPs on the 2.5 PFLOP/s bf16 MFMA peak; input + output + weight bytes once on 8 TB/s)
            conv = []
            n_timed = max(1, min(a.steps, 5))
            for k, v in timer.summary(by_tag=True).items():
                nm, _, tag = k.partition(':')
                if not nm.startswith('relnet_conv2d_nhwc') or not tag.startswith('M'):
                    continue
                try:
                    M_, N_, K_, ks_ = [int(x[1:]) for x in tag.split('_')]
                except ValueError:
                    continue
                sec = v['avg_ms'] * 1e-3
                byts = 2.0 * (M_ * (K_ // (ks_ * ks_)) + M_ * N_ + N_ * K_)
                conv.append({'shape': tag, 'launches_per_step': round(v['calls'] / n_timed, 1), 'avg_us': round(v['avg_ms'] * 1e3, 1),
                             'ms_per_step': round(v['total_ms'] / n_timed, 3), 'tflops': round(2.0 * M_ * N_ * K_ / sec / 1e12, 1),
                             'mfma_frac': round(2.0 * M_ * N_ * K_ / sec / 1e12 / PEAK_TFLOPS[a.dtype], 3),
                             'hbm_frac': round(byts / sec / 8e12, 3)})
            conv.sort(key=lambda c: -c['ms_per_step'])
            if conv:
                res['conv_roofline'] = {'note': 'implicit-GEMM convolution launches of the step by shape (M = pixels, N = Cout, K = taps x Cin, k = '
                                                'kernel size), HIP-event time of an eager re-run; mfma_frac on 2.5 PFLOP/s dense bf16, hbm_frac = '
                                                '(input + output + weights once) / time / 8 TB/s; the expand / block-boundary layers run in '
                                                'relnet_bottleneck_chain and are not listed; the RPN head (N512_K9216) runs on a side stream BESIDE res5a / res5 / '
                                                'conv_new_1, so those rows are timed while two full-GPU kernels share the CUs', 'top': conv[:8]}
            att = ks.get('relnet_relation_attention_kc') or ks.get('relnet_relation_attention')
            if att:
                sec = att['avg_ms'] * 1e-3
                rs = (n_rois / 300.0) ** 2                                         # N = M = n_rois keys and queries
                algo = ALGO_GFLOP_PER_MODULE_IMAGE * rs * a.batch / 1e3 / sec     # TFLOP/s
                peak = PEAK_TFLOPS[a.dtype]
                traffic = None
                pmc = os.path.join(ROOT, 'profiles', 'attention_pmc.json')
                if os.path.exists(pmc):
                    traffic = json.load(open(pmc)).get('hbm_bytes_per_launch_at_batch', {}).get(str(a.batch))
                # The kernel is an HBM stream, not an MFMA-bound loop (PMC: MFMA-busy 9.6 %, 3.4 TB/s; profiles/attention_pmc.json), so the
                # contract's roofline object is stated on the HBM roof: `achieved` = ALGORITHMIC bytes per launch (SURVEY 8d with the geometry
                # fused: boxes + Q|K + VW^T + shortcut in + activation out = 3.12 MB per image at N = 300 -- NOT counting the materialised
                # fp16 geometry bias the kernel also reads today) / launch duration; `traffic` = the bytes it really moves (PMC).  The MFMA
                # pricing the earlier rounds led with stays beside it: `mfma_frac_as_written` prices the graph as written (softmax.V over
                # 1024-d values, then the grouped linear_out: 3.13 GFLOP per module-image), `frac_executed` the FLOPs the re-associated
                # kernel S.(F_K Wout^T) really performs (8.5x fewer).
                mp = (n_rois + 31) // 32 * 32
                algo_bytes = a.batch * (n_rois * 16 + n_rois * 2048 * 2 + 1024 * mp * 2 + 2 * n_rois * 1024 * 2)
                bias_bytes = a.batch * 16 * n_rois * mp * 2
                gbs = algo_bytes / sec / 1e9
                res['roofline'] = {
                    'kernel': 'relation_attention_lds_kernel' if a.dtype == 'bf16' else 'relation_attention_kernel<float>', 'bound': 'hbm',
                    'achieved': gbs, 'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0, 'traffic': traffic,
                    'launch_ms': att['avg_ms'], 'launches': att['calls'],
                    'algorithmic_bytes_per_launch': algo_bytes,
                    'bytes_per_launch_with_the_materialised_geometry_bias': algo_bytes + bias_bytes,
                    'hbm_frac_of_bytes_moved': (algo_bytes + bias_bytes) / sec / 8e12,
                    'traffic_over_algorithmic': (traffic / algo_bytes) if traffic else None,
                    'mfma_tflops_as_written': algo, 'mfma_peak_tflops': peak, 'mfma_frac_as_written': algo / peak,
                    'executed_tflops': EXEC_GFLOP_PER_MODULE_IMAGE * rs * a.batch / 1e3 / sec,
                    'frac_executed': EXEC_GFLOP_PER_MODULE_IMAGE * rs * a.batch / 1e3 / sec / peak,
                    'algorithmic_gflop_per_launch': ALGO_GFLOP_PER_MODULE_IMAGE * rs * a.batch,
                    'frac_basis': 'algorithmic HBM bytes with the geometry fused (SURVEY 8d) / launch time / 8 TB/s; the kernel reads a materialised fp16 '
                                  'geometry bias on top (2x the algorithmic bytes): that is `traffic`, written by geometry_bias_mfma_kernel for both modules',
                    'traffic_source': ('profiles/attention_pmc.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE of an earlier run of this kernel at this '
                                       'batch; not re-measured by this process)') if traffic else None,
                }
                iso = attention_isolated(a.batch, n_rois, tdt)           # kernel alone: median of 100 launches
                res['roofline'].update(isolated_median_ms=iso, isolated_frac=algo_bytes / (iso * 1e-3) / 8e12)
        plain = not (a.dcn or a.fpn or a.learn_nms)
        if world == 1 and plain and not a.no_parity and a.dtype == 'bf16':
            # the timed configuration itself (same detector object, same batch) checked stage by stage against the oracle
            from oracle import parity as OPAR
            res['parity'] = OPAR.stagewise(det, data, im_info, params, images=range(min(a.parity_images, a.batch)),
                                           relation=not a.no_relation)
        if world == 1 and plain and not a.no_batch_sweep:
            res['batch_sweep'] = {'note': 'images/s of the same step at other images-per-GPU-per-step settings (hipGraph replay); '
                                          '1 = the reference protocol (BATCH_IMAGES: 1, SURVEY 8d)'}
            for bsz in (1, 8, 54):
                if bsz == a.batch:
                    continue
                res['batch_sweep'][str(bsz)] = _replay_rate(det, bsz, a, D, make_det=make_det if a.in_flight > 1 else None, n_flight=a.in_flight)
        if world == 1 and not a.no_cpu_baseline and not a.dcn and not a.fpn:      # the CPU port of the DCN graph is parity-only (slow)
            res['cpu_baseline'] = cpu_baseline(params, relation=not a.no_relation, images=a.cpu_images, threads=a.cpu_threads)
    if plain_graph and not a.no_train_line:
        # BASELINE configs[2] (relation + learn-NMS end2end TRAINING, one summed RCCL all-reduce per step) on the same N GPUs:
        # the data-parallel path north_star's scaling target is stated on; every rank takes part in the collective
        del det
        torch.cuda.empty_cache()
        ta = argparse.Namespace(**vars(a))
        keys = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'n_gpus', 'config', 'communication', 'losses', 'weights_finite_on_all_ranks')
        sub = ('value', 'ms_per_step', 'steps', 'communication')

        def train_at(bsz, steps):
            ta.batch, ta.learn_nms, ta.steps, ta.warmup = bsz, True, steps, 2
            return _side_figure(lambda: bench_train(ta, rank, world, D, emit=False, fatal=False), 'train@%d' % bsz)
        tr_res = train_at(8, min(a.steps, 10))
        # the same step at 16 images per GPU (larger GEMMs fill the chip better): a single-GPU side figure -- an N > 1 run carries the
        # scaling workload only (8 and 1 images per GPU), every extra collective-bearing figure is one more way to lose the line
        tr16 = train_at(16, min(a.steps, 6)) if world == 1 else None
        # ... and at ONE image per GPU: the reference's own training protocol (BATCH_IMAGES: 1 per device,
        # cfgs/resnet_v1_101_coco_trainvalminus_rcnn_end2end_relation_learn_nms_8epoch.yaml:80, train_end2end.py:70-71)
        tr1 = train_at(1, min(a.steps, 10))
        if rank == 0:
            pick = lambda r, ks: r if (r is None or 'error' in r) else {k: r[k] for k in ks}
            res['train'] = pick(tr_res, keys) or {}
            res['train']['at_16_images_per_gpu'] = pick(tr16, sub)
            res['train']['at_1_image_per_gpu'] = pick(tr1, sub)
            if tr1 is not None and 'error' not in tr1:
                res['train']['at_1_image_per_gpu']['note'] = ("the reference's own protocol: BATCH_IMAGES 1 per device (cfgs/..._rcnn_end2end_relation_"
                                                               "learn_nms_8epoch.yaml:80); ~380 launches of 4 - 37 us on 2 394-pixel maps, latency bound")
    if plain_graph and not a.no_other_configs and world > 1 and rank == 0:
        res['other_configs'] = 'single-GPU side figures (configs[3] / configs[4] inference and training rates): run `python bench.py` with --gpus 1'
    if plain_graph and not a.no_other_configs and world == 1:
        if 'det' in locals():
            del det
        torch.cuda.empty_cache()
        oc = _side_figure(lambda: other_configs(a, rank, world, D), 'other_configs')
        if rank == 0:
            res['other_configs'] = oc
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

"""Fold the raw counter files of `gpu_runs.sh pmc_trunk` / `pmc_attn` (tools/pmc_collect.py output + kernel-trace durations) into the
tracked evidence files, with the derived figures spelled out:
    python tools/pmc_trunk_fold.py trunk gpurun_out/r05_pmc_trunk profiles/r05_trunk_pmc.json
    python tools/pmc_trunk_fold.py attn  gpurun_out/r05_pmc_attn  profiles/attention_pmc.json
Units and corrections (MI355X_MICROARCH.md, HBM / PMC sections): FETCH_SIZE, WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports half the bytes
of wide coalesced reads -> doubled; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs;
SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles summed over the chip; TCP_TCC_*_REQ in 128-byte requests."""
import json, os, sys

KERNELS = {
    'res4_3x3': dict(match='gemm_ring_kernel', what='res4 3x3 / 256 -> 256 convolution, 54 images, asm ring tile 19 (SCHED 6)', flop=2.0 * 129276 * 256 * 2304,
                     algo_read=129276 * 256 * 2 + 256 * 2304 * 2, algo_write=129276 * 256 * 2),
    'res5_3x3': dict(match='gemm_ring_kernel', what='res5 3x3 dilated / 512 -> 512 convolution, 54 images, asm ring tile 19', flop=2.0 * 129276 * 512 * 4608,
                     algo_read=129276 * 512 * 2 + 512 * 4608 * 2, algo_write=129276 * 512 * 2),
    'chain256': dict(match='chain256_roles_kernel', what='res4 expand 256 -> 1024 + shortcut + ReLU and next reduce 1024 -> 256 + ReLU, 54 images (chain256_roles_kernel, out of place)',
                     flop=2.0 * 2 * 129276 * 256 * 1024, algo_read=129276 * (256 + 1024) * 2 + 2 * 1024 * 256 * 2, algo_write=129276 * (1024 + 256) * 2),
}


def derive(c, flop, algo_read, algo_write):
    a = lambda k: c[k]['avg']
    dur = c['duration_ns']['median'] * 1e-9
    cyc_xcd = a('GRBM_GUI_ACTIVE') / 8.0
    d = {'duration_us_median': dur * 1e6, 'duration_us_min_max': [c['duration_ns']['min'] / 1e3, c['duration_ns']['max'] / 1e3],
         'effective_clock_GHz': cyc_xcd / dur / 1e9,
         'mfma_busy_frac_of_launch': a('SQ_VALU_MFMA_BUSY_CYCLES') / 1024.0 / cyc_xcd,
         'mfma_instructions': a('SQ_INSTS_MFMA'), 'valu_per_mfma': a('SQ_INSTS_VALU') / a('SQ_INSTS_MFMA'),
         'tflops': flop / dur / 1e12, 'frac_of_2500_TFLOPS_nominal': flop / dur / 2.5e15,
         'frac_of_peak_at_effective_clock': flop / dur / (2.5e15 * (cyc_xcd / dur) / 2.4e9),
         'wave_cycles_parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)': a('SQ_WAIT_ANY') / a('SQ_WAVE_CYCLES'),
         'wave_cycles_issue_stalled_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)': a('SQ_WAIT_INST_ANY') / a('SQ_WAVE_CYCLES'),
         'hbm_read_bytes (2 x FETCH_SIZE KiB)': 2 * 1024 * a('FETCH_SIZE'), 'hbm_write_bytes (WRITE_SIZE KiB)': 1024 * a('WRITE_SIZE'),
         'algorithmic_read_bytes': algo_read, 'algorithmic_write_bytes': algo_write,
         'hbm_traffic_over_algorithmic': (2 * 1024 * a('FETCH_SIZE') + 1024 * a('WRITE_SIZE')) / float(algo_read + algo_write),
         'hbm_TBps': (2 * 1024 * a('FETCH_SIZE') + 1024 * a('WRITE_SIZE')) / dur / 1e12,
         'l2_hit_rate': a('TCC_HIT_sum') / (a('TCC_HIT_sum') + a('TCC_MISS_sum')),
         'l2_to_cu_read_bytes (TCP_TCC_READ_REQ x 128 B)': 128 * a('TCP_TCC_READ_REQ_sum'),
         'l2_to_cu_read_TBps': 128 * a('TCP_TCC_READ_REQ_sum') / dur / 1e12,
         'fill_over_hbm_read': 128 * a('TCP_TCC_READ_REQ_sum') / (2 * 1024 * a('FETCH_SIZE'))}
    return d


def kernels_at(images, tile4):
    """The same three kernels at another images-per-launch setting (round 5: 8 = the training step, where pick_tile runs the 3x3 on 128 x 64 tiles)."""
    P = images * 2394
    k = {key: dict(v) for key, v in KERNELS.items()}
    k['res4_3x3'].update(flop=2.0 * P * 256 * 2304, algo_read=P * 256 * 2 + 256 * 2304 * 2, algo_write=P * 256 * 2,
                         what='res4 3x3 / 256 -> 256 convolution, %d images, %s' % (images, 'gemm_nt_bf16_kernel<128, 64> (tile 4: what pick_tile runs at this size)' if tile4 else 'asm ring tile 19'))
    if tile4:
        k['res4_3x3']['match'] = 'gemm_nt_bf16_kernel<128, 64'
    k['res5_3x3'].update(flop=2.0 * P * 512 * 4608, algo_read=P * 512 * 2 + 512 * 4608 * 2, algo_write=P * 512 * 2, what='res5 3x3 dilated / 512 -> 512 convolution, %d images, asm ring tile 19' % images)
    k['chain256'].update(flop=2.0 * 2 * P * 256 * 1024, algo_read=P * (256 + 1024) * 2 + 2 * 1024 * 256 * 2, algo_write=P * (1024 + 256) * 2,
                         what='res4 expand 256 -> 1024 + shortcut + ReLU and next reduce 1024 -> 256 + ReLU, %d images (chain256_roles_kernel, out of place)' % images)
    return k


def trunk(src, out, kernels=None, images=54):
    kernels = kernels or KERNELS
    res = {'round': 5, 'images_per_launch': images,
           'command': 'cd /tmp && export TMPDIR=/tmp; rocprofv3 --pmc <GROUP> --kernel-trace --output-format csv -- python tools/kernel_pmc.py <kernel> 54 4   '
                      '(six passes per kernel: GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_BUSY_CYCLES | SQ_WAIT_ANY SQ_WAIT_INST_ANY '
                      'SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU | FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum | TCP_TCC_READ_REQ_sum '
                      'TCP_TCC_WRITE_REQ_sum; `bash tools/scripts/gpu_runs.sh pmc_trunk`, folded by tools/pmc_collect.py + tools/pmc_trunk_fold.py; MI355X, ROCm 7.2, '
                      'the round-5 binary); per-launch averages over 4 isolated launches',
           'note': 'effective_clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration; mfma_busy_frac_of_launch = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / those cycles '
                   '(= 32 cycles x MFMA instructions per SIMD).  Four isolated launches run at 1.8 - 2.15 GHz; inside the 54-image step the same kernels were '
                   'timestamped at 1.48 - 1.75 GHz (profiles/r04_notes/tile_phase_probe_b54.txt): the counters are the less throttled case.',
           'kernels': {}}
    for key, meta in kernels.items():
        f = os.path.join(src, key + '_pmc_raw.json')
        if not os.path.exists(f):
            continue
        raw = json.load(open(f))
        name = [k for k in raw if meta['match'] in k and 'duration_ns' in raw[k]][0]
        res['kernels'][key] = {'kernel': name, 'what': meta['what'], 'derived': derive(raw[name], meta['flop'], meta['algo_read'], meta['algo_write']),
                               'counters_avg_per_launch': {c: v['avg'] for c, v in raw[name].items() if c != 'duration_ns'}}
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    for k, v in res['kernels'].items():
        d = v['derived']
        print('%-9s %.1f us  clock %.2f GHz  MFMA busy %.3f  %.0f TFLOP/s (%.3f nominal, %.3f at clock)  HBM %.2f TB/s (x%.2f algorithmic)  L2 hit %.2f  fill %.2f TB/s' % (
            k, d['duration_us_median'], d['effective_clock_GHz'], d['mfma_busy_frac_of_launch'], d['tflops'], d['frac_of_2500_TFLOPS_nominal'],
            d['frac_of_peak_at_effective_clock'], d['hbm_TBps'], d['hbm_traffic_over_algorithmic'], d['l2_hit_rate'], d['l2_to_cu_read_TBps']))


def attn(src, out):
    K = 'relnet::relation_attention_lds_kernel(relnet::AttnArgs, int, int)'
    res = {'round': int(os.environ.get('RELNET_ROUND', '6')), 'kernel': K,
           'command': 'cd /tmp && export TMPDIR=/tmp; rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python tools/attn_only.py <108|54> 6   (one pass per '
                      'counter group: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_ANY GRBM_GUI_ACTIVE; `bash tools/scripts/gpu_runs.sh '
                      'pmc_attn`, folded by tools/pmc_collect.py + tools/pmc_trunk_fold.py; MI355X, ROCm 7.2, the binary of the round named above)',
           'note': 'FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so the '
                   'read side is doubled; WRITE_SIZE is taken as is. The launch is the pipeline\'s: one output, ReLU(out + shortcut). bench.py reads '
                   'hbm_bytes_per_launch_at_batch[<images per step>] as roofline.traffic.',
           'hbm_bytes_per_launch_at_batch': {}, 'algorithmic_bytes_per_launch_at_batch': {}, 'per_batch': {}}
    for b in (108, 54):
        f = os.path.join(src, 'attention_pmc_raw_b%d.json' % b)
        if not os.path.exists(f):
            continue
        raw = json.load(open(f))
        c = raw[K]
        mp = 320
        algo = b * (300 * 2048 * 2 + 1024 * mp * 2 + 16 * 300 * mp * 2 + 2 * 300 * 1024 * 2)
        hbm = int(round(2 * 1024 * c['FETCH_SIZE']['avg'] + 1024 * c['WRITE_SIZE']['avg']))
        res['hbm_bytes_per_launch_at_batch'][str(b)] = hbm
        res['algorithmic_bytes_per_launch_at_batch'][str(b)] = algo
        res['per_batch'][str(b)] = {'FETCH_SIZE_KiB_raw': c['FETCH_SIZE']['avg'], 'WRITE_SIZE_KiB_raw': c['WRITE_SIZE']['avg'], 'traffic_over_algorithmic': hbm / float(algo),
                                    'mfma_instructions': c['SQ_INSTS_MFMA']['avg'], 'valu_per_mfma': c['SQ_INSTS_VALU']['avg'] / c['SQ_INSTS_MFMA']['avg'],
                                    'mfma_busy_cycles_per_simd': c['SQ_VALU_MFMA_BUSY_CYCLES']['avg'] / 1024.0, 'gui_active_cycles_per_xcd': c['GRBM_GUI_ACTIVE']['avg'] / 8.0,
                                    'mfma_busy_frac_of_launch': c['SQ_VALU_MFMA_BUSY_CYCLES']['avg'] / 1024.0 / (c['GRBM_GUI_ACTIVE']['avg'] / 8.0),
                                    'counters': {k: {c2: v['avg'] for c2, v in cs.items()} for k, cs in raw.items() if 'relnet' in k}}
        print('attention b%d: HBM %d B / launch = %.3f x algorithmic %d; MFMA busy %.3f; VALU per MFMA %.1f' % (
            b, hbm, hbm / float(algo), algo, res['per_batch'][str(b)]['mfma_busy_frac_of_launch'], res['per_batch'][str(b)]['valu_per_mfma']))
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    if sys.argv[1] == 'trunk8':          # python tools/pmc_trunk_fold.py trunk8 gpurun_out/r05_pmc_trunk8 profiles/r05_train_shapes_pmc.json  (IMAGES=8 TILE=4 runs)
        trunk(sys.argv[2], sys.argv[3], kernels_at(8, True), images=8)
    else:
        {'trunk': trunk, 'attn': attn}[sys.argv[1]](sys.argv[2], sys.argv[3])

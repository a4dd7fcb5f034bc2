import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import relnet_amd
    from relnet_amd import ops, lib
    cfg, M, N, K, od = [int(x) for x in sys.argv[1:6]]
    lib.load().relnet_gemm_force_tile(cfg)
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda')
    out = ops.gemm_nt(a, w, b, relu=False, out_dtype=torch.float32 if od else torch.bfloat16)
    torch.cuda.synchronize()
    ref = a.double() @ w.double().t() + b.double()
    print('cfg', cfg, M, N, K, od, 'err', ((out.double() - ref).abs().max() / ref.abs().max()).item())
else:
    for cfg in (1, 2, 3, 4, 5):
        for (M, N, K, od) in ((300, 300, 128, 0), (1000, 512, 256, 1), (77, 89, 64, 1)):
            r = subprocess.run([sys.executable, __file__, str(cfg), str(M), str(N), str(K), str(od)], capture_output=True, text=True)
            print((r.stdout.strip() or 'CRASH cfg %d %s' % (cfg, (M, N, K, od))), '|', r.stderr.strip()[-120:].replace('\n', ' ') if r.returncode else '')

import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libgbench.so'))
lib.g_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
N, H, W = 32, 256, 256
st = torch.cuda.current_stream().cuda_stream
for amp in (0.3, 8.0, 60.0):
    low = torch.randn(N, 2, 16, 16, device='cuda')
    disp = torch.nn.functional.interpolate(low, size=(H, W), mode='bilinear', align_corners=True)
    disp = disp / disp.abs().max() * amp
    planar = disp.contiguous()
    inter = disp.permute(0, 2, 3, 1).contiguous()
    out = torch.empty_like(planar)
    for mode, src in ((0, planar), (1, inter)):
        for blocks in (64, 128, 256):
            for _ in range(3): lib.g_run(mode, src.data_ptr(), out.data_ptr(), N, H, W, blocks, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): lib.g_run(mode, src.data_ptr(), out.data_ptr(), N, H, W, blocks, st)
            e1.record(); torch.cuda.synchronize()
            print("amp %5.1f  %s  blocks/sample %3d : %.1f us" % (amp, "planar 8 x dword " if mode == 0 else "interl 4 x dwordx2", blocks, e0.elapsed_time(e1) / 20 * 1e3), flush=True)

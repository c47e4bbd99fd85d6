import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdispbench.so'))
lib.disp_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(1 << 20, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def t(blocks, threads, lds, work=1, reps=20):
    for _ in range(3): lib.disp_run(out.data_ptr(), blocks, threads, lds, work, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): lib.disp_run(out.data_ptr(), blocks, threads, lds, work, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for threads in (64, 256, 512, 1024):
    for lds in (0, 8192, 32768, 49152, 65536):
        print("threads %4d lds %6d : " % (threads, lds) + "  ".join("%d WGs %.1f us" % (b, t(b, threads, lds)) for b in (1024, 4096, 16384, 65536)), flush=True)

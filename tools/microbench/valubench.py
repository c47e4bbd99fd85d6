import torch, ctypes, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libvalubench.so'))
lib.valu_run.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.empty(256 * 4096, device='cuda')
st = torch.cuda.current_stream().cuda_stream
names = {0: "v_fma_f32", 1: "v_pk_fma_f32 (2 floats)", 2: "tent (3 ops)", 3: "v_med3_f32", 4: "floor+mul (2 ops)", 5: "int mul+add (mad?)"}
iters = 4000
for wgs_per_cu in (1, 2, 4, 8):
    blocks = 256 * wgs_per_cu          # 4 waves per WG = 1 per SIMD
    for mode in range(6):
        for _ in range(2):
            lib.valu_run(mode, out.data_ptr(), blocks, iters, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.valu_run(mode, out.data_ptr(), blocks, iters, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        stmts = 8 * iters                    # statements per wave
        waves_per_simd = wgs_per_cu
        clk = ms * 1e-3 * 2.4e9
        print("waves/SIMD %d  %-26s %.3f ms  -> %.2f clk per statement per SIMD (at 2.4 GHz)" % (waves_per_simd, names[mode], ms, clk / (stmts * waves_per_simd)))

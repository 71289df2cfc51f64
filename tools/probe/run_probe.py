import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libprobe.so'))
P.probe_run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda', 0)
M, T = 1 << 16, 1 << 21
z64 = torch.empty(M, dtype=torch.int64, device=dev); z32 = torch.empty(M, dtype=torch.int32, device=dev); cnt = torch.zeros(1, dtype=torch.int32, device=dev)
# expected: slot s gets writers t = s, s+M, ...: min key = largest t -> (T - t_max) << 32 | t_max ; z32 = T - t_max
tmax = torch.arange(M, device=dev, dtype=torch.int64) + (T // M - 1) * M
exp64 = ((T - tmax) << 32) | tmax
exp32 = (T - tmax).to(torch.int32)
x = torch.randn(4, 256, 128, 128, device=dev); w = torch.randn(256, 256, 3, 3, device=dev) / 48
wt = cg.prep_weight_bf16x3(w)
side = torch.cuda.Stream()
for mode in ('alone', 'with_conv'):
    bad64 = bad32 = badc = 0
    for it in range(20):
        if mode == 'with_conv':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    cg.conv_launch(x, wt, 3, 0, 256, bf16x3=True)
        P.probe_run(z64.data_ptr(), z32.data_ptr(), cnt.data_ptr(), M, T, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        bad64 += int((z64 != exp64).sum()); bad32 += int((z32 != exp32).sum()); badc += int(cnt.item() != T)
    print(mode, 'wrong 64-bit min slots:', bad64, ' wrong 32-bit min slots:', bad32, ' wrong thread counts:', badc, 'of 20 runs')

P.probe_math_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
T2 = 1 << 20
out = torch.empty(T2, device=dev)
P.probe_math_run(out.data_ptr(), T2, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
refm = out.clone()
for mode in ('alone', 'with_conv'):
    bad = 0
    for it in range(20):
        if mode == 'with_conv':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    cg.conv_launch(x, wt, 3, 0, 256, bf16x3=True)
        out.zero_()
        P.probe_math_run(out.data_ptr(), T2, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        bad += int((out != refm).sum())
    print('math probe', mode, 'elements differing (20 runs):', bad)

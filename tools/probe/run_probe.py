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

P.probe_gather_run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
V, F, T3 = 5023 * 8, 9976, 9976 * 8
g = torch.Generator(device=dev).manual_seed(3)
data = torch.randn(V * 3, device=dev, generator=g); idx = torch.randint(0, V, (F * 3,), device=dev, generator=g, dtype=torch.int32)
outg = torch.empty(T3, device=dev)
P.probe_gather_run(data.data_ptr(), idx.data_ptr(), outg.data_ptr(), F, T3, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
refg = outg.clone()
for mode in ('alone', 'with_conv'):
    bad = 0
    for it in range(30):
        if mode == 'with_conv':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    cg.conv_launch(x, wt, 3, 0, 256, bf16x3=True)
        outg.fill_(-1.0)
        data2 = data.clone()                     # freshly written source, like the transformed vertices
        P.probe_gather_run(data2.data_ptr(), idx.data_ptr(), outg.data_ptr(), F, T3, torch.cuda.current_stream().cuda_stream)
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        bad += int((outg != refg).sum())
    print('gather probe', mode, 'elements differing (30 runs):', bad)

P.probe_many_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
M2, T4, PER = 8 * 65536, 8 * 9976, 40
zz = torch.empty(M2, dtype=torch.int64, device=dev)
def many():
    zz.fill_(-1)
    P.probe_many_run(zz.data_ptr(), M2, T4, PER, torch.cuda.current_stream().cuda_stream)
many(); torch.cuda.synchronize(); refz = zz.clone()
for mode in ('alone', 'with_conv'):
    bad = 0
    for it in range(30):
        if mode == 'with_conv':
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    cg.conv_launch(x, wt, 3, 0, 256, bf16x3=True)
        many()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        bad += int((zz != refz).sum())
    print('many-atomics probe', mode, 'slots differing (30 runs):', bad)

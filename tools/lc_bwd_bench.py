import sys, time, torch
sys.path.insert(0, '/root/repo')
from ag2video_amd import _lib, ops
torch.manual_seed(0)
NB, J, K, H, W, R = 120, 22, 512, 256, 256, 3
dense = torch.rand(NB, J, 4, device='cuda') * 0.5
dense[:, :, 2:] = dense[:, :, 2:] * 0.6 + 0.1
mask = torch.zeros(J, device='cuda'); mask[[0, 1, 2, 11, 12, 13]] = 1
boxes = dense * mask.view(1, J, 1)
keep = torch.ones(NB, J, dtype=torch.int32, device='cuda')
dout = torch.randn(NB, K, H, W, device='cuda')
dU = torch.empty(NB, J, K, 9, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def f(): _lib.call('ag2v_layout_conv_bwd', dout.data_ptr(), boxes.data_ptr(), keep.data_ptr(), dU.data_ptr(), NB, J, K, H, W, R, 1, 1, 0, st)
f(); torch.cuda.synchronize(); t = time.time()
for _ in range(5): f()
torch.cuda.synchronize(); print('layout_conv_bwd %.2f ms' % ((time.time() - t) / 5 * 1e3), float(dU.abs().sum()))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ag2video_amd import graph_program as gp
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (M, N, K, masked) in [(136, 512, 1152, False), (136, 512, 1152, True), (136, 512, 1536, True), (88, 512, 512, True), (88, 516, 512, True),
                          (88, 512, 4, False), (136, 1536, 512, True), (88, 128, 512, True), (136, 512, 1088, False), (136, 512, 1024, True)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); m = torch.randn(M, K, device=dev)
    P = gp.Program(dev)
    out = P.new(M, N)
    P.gemm(P.ext(a), P.ext(w), out, mask=P.ext(m) if masked else None)
    P.finalize().run()
    torch.cuda.synchronize()
    ref = ((a * (m > 0)) if masked else a).double() @ w.double().t()
    err = (P.tensor(out).double() - ref).abs().max().item() / ref.abs().max().item()
    print(M, N, K, masked, 'err %.2e' % err)
for (M, N, K, masked) in [(88, 512, 516, False), (88, 512, 132, False), (88, 512, 516, True), (136, 512, 384, False), (88, 4, 512, False), (88, 512, 640, False)]:
    a = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); m = torch.randn(M, K, device=dev)
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev)
    P = gp.Program(dev)
    out = P.new(M, N)
    P.gemm(P.ext(a), P.ext(w), out, mask=P.ext(m) if masked else None, bias=bias, res=P.ext(res), relu=True)
    P.finalize().run()
    torch.cuda.synchronize()
    ref = torch.relu(((a * (m > 0)) if masked else a).double() @ w.double().t() + bias.double()) + res.double()
    err = (P.tensor(out).double() - ref).abs().max().item() / ref.abs().max().item()
    print(M, N, K, masked, 'bias+relu+res err %.2e' % err)

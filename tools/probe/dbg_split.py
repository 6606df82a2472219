import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from doc2tex_amd import _lib
lib = _lib.require_device()
def run(B,H,W,Cin,Cout,k,pad):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B,Cin,H,W,generator=g); w = torch.randn(Cout,Cin,k,k,generator=g)*(2/(Cin*k*k))**0.5
    xd = x.permute(0,2,3,1).contiguous().cuda(); wd = w.permute(0,2,3,1).contiguous().cuda()
    OH, OW = H+2*pad-k+1, W+2*pad-k+1
    y = torch.full((B,OH,OW,Cout), float('nan'), device='cuda')
    rc = lib.d2t_op_conv2d_bf16x3_split(_lib.ptr(xd), _lib.ptr(wd), None, None, _lib.ptr(y), B,H,W,Cin,Cout,k,k,1,1,pad,pad,0,_lib.stream_of(xd))
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), w.double(), None, 1, pad).float().permute(0,2,3,1)
    d = (y.cpu()-ref).abs()
    bad_rows = (d.reshape(-1,Cout).max(1).values > 1e-3).nonzero().flatten().tolist()
    bad_cols = (d.reshape(-1,Cout).max(0).values > 1e-3).nonzero().flatten().tolist()
    print(f"B{B} {H}x{W} Cin{Cin} Cout{Cout} k{k}: rc={rc} maxerr={float(d.max()):.3e} nan={int(torch.isnan(y).sum())} bad_rows={len(bad_rows)} {bad_rows[:12]} bad_cols={len(bad_cols)} {bad_cols[:12]}")
run(1,8,16,32,128,1,0)    # M=128, KT=1
run(1,8,16,64,128,1,0)    # KT=2
run(1,8,16,128,128,1,0)   # KT=4
run(1,8,16,32,128,3,1)    # KT=9, taps
run(2,16,20,64,256,3,1)

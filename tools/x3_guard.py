"""Out-of-bounds write check of mvs_conv3d_x3_fwd: the output lives in the middle of a sentinel-filled buffer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvsformer_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
cases = [(8, 16, (1, 2), 8, 576, 768), (16, 32, (1, 2), 8, 288, 384), (32, 64, (1, 2), 8, 144, 192), (8, 16, (1, 2), 4, 1152, 1536), (16, 32, (1, 2), 4, 576, 768),
         (32, 64, (1, 2), 4, 288, 384), (16, 16, (1, 1), 4, 576, 768), (32, 32, (1, 1), 8, 144, 192), (64, 64, (1, 1), 4, 144, 192)]
for (cin, cout, stride, d, h, w) in cases:
    x = torch.randn(1, cin, d, h, w, device=dev)
    wt = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    px = ops.conv3d_x3_pack(wt, stride)
    ho, wo = (h - 1) // stride[1] + 1, (w - 1) // stride[1] + 1
    n = cout * d * ho * wo
    guard = 1 << 22
    buf = torch.full((n + 2 * guard,), 12345.0, device=dev)
    y = buf[guard:guard + n]
    rc = lib.mvs_conv3d_x3_fwd(x.data_ptr(), px.data_ptr(), None, None, None, y.data_ptr(), 1, cin, cout, d, h, w, stride[0], stride[1], 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lo_bad = int((buf[:guard] != 12345.0).sum()); hi_bad = int((buf[guard + n:] != 12345.0).sum())
    ref = ops.conv3d_x3(x, px, cin, cout, stride)
    print(cin, cout, stride, d, h, w, "rc", rc, "guard violations lo/hi:", lo_bad, hi_bad, "unwritten (sentinel left):", int((y == 12345.0).sum()), "equal ref:", torch.equal(y.view_as(ref), ref))

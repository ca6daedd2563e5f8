import sys
sys.path.insert(0, '/root/repo')
import torch, gan2shape_amd
from gan2shape_amd import modconv as mc, lib as _lib
for (B, cin, cout, H) in [(8,512,512,32),(8,128,128,128),(8,256,256,64)]:
    x = torch.randn(B, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") / 68
    y = torch.empty(B, cout, H, H, device="cuda")
    dbg = torch.zeros(4, device="cuda")
    U = mc.wino_weights(w, 0)
    L = _lib.load()
    for _ in range(20):
        _lib.check(L.g2s_conv3x3_wino(_lib.ptr(x), _lib.ptr(U), None, None, _lib.ptr(dbg), _lib.ptr(y), B, cin, cout, H, H, 0, 0.0, 1.0, 1, _lib.stream()))
    torch.cuda.synchronize()
    c, wl, n = dbg[:3].tolist()
    print((B,cin,cout,H), "cycles/ktile", c / n, "wall us/ktile", wl / 100.0 / n, "=> clock GHz", c / (wl / 100.0) / 1e3)

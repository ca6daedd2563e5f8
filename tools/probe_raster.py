import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from raster_cases import scene
so = os.path.join(ROOT, "probe_tmp", "probe.so")
L = C.CDLL(so)
L.g2s_raster_workspace_bytes.restype = C.c_size_t
S=128; N=S*S; F=2*(S-1)**2
for name, seed, rot in (("hard",1,60.0),("easy",1,5.0)):
    B=1
    geo, verts, _ = scene(S, B=B, seed=seed, rot=rot)
    K = (C.c_float * 9)(*np.asarray(geo.K[0], np.float32).reshape(9).tolist())
    v = torch.tensor(verts, device="cuda").contiguous()
    wsb = L.g2s_raster_workspace_bytes(B, N, F, S)
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    d = torch.empty(B, S, S, device="cuda"); fi = torch.empty(B, 2*S, 2*S, dtype=torch.int32, device="cuda"); ba = torch.empty(B, 2*S, 2*S, 3, device="cuda")
    for _ in range(3):
        rc = L.g2s_raster_depth_fwd(C.c_void_p(v.data_ptr()), None, B, N, F, S, K, C.c_float(S), 2, 1, C.c_float(0.1), C.c_float(100.0),
            C.c_void_p(d.data_ptr()), C.c_void_p(fi.data_ptr()), C.c_void_p(ba.data_ptr()), C.c_void_p(ws.data_ptr()), C.c_size_t(wsb), None)
        assert rc == 0
    torch.cuda.synchronize()
    b = ba[0].cpu().numpy()  # [256,256,3]
    tot = b[0::8, 0::8, 0]; walk = b[0::8, 0::8, 1]; ev = b[0::8, 0::8, 2]
    fin = b[0::8, 1::8, 0]; ncand = b[0::8, 1::8, 1]; nev = b[0::8, 2::8, 0]
    i = np.unravel_index(np.argmax(tot), tot.shape)
    print(name, "cycles: max tot %.0f (walk %.0f eval %.0f fin %.0f ncand %.0f nevalrounds %.0f) | mean tot %.0f walk %.0f eval %.0f fin %.0f ncand %.1f nev %.1f" % (
        tot[i], walk[i], ev[i], fin[i], ncand[i], nev[i], tot.mean(), walk.mean(), ev.mean(), fin.mean(), ncand.mean(), nev.mean()))
    srt = np.sort(tot.ravel())[::-1]
    print("  top tot:", srt[:8], "sum tot", tot.sum())

import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import O, fill_module_, pkg
from test_gpu_rdb_fused import _plans
shape = (1, 20, 28)
ref, fus = _plans(shape, True, num_blocks=1)
x = O.hash_input("fr.x", (shape[0], 3, shape[1], shape[2])).cuda()
ref.forward(x); fus.forward(x); torch.cuda.synchronize()
for r in range(2):
    a = ref.R[r].t.float().view(shape[0], 24, shape[1], shape[2], 8)
    b = fus.R[r].t.float().view(shape[0], 24, shape[1], shape[2], 8)
    for name, lo, hi in (("x", 0, 8), ("x1", 8, 12), ("x2", 12, 16), ("x3", 16, 20), ("x4", 20, 24)):
        d = (a[:, lo:hi] - b[:, lo:hi]).abs()
        nz = (d > 0).nonzero()
        print("R[%d] %s: max diff %.4g  differing %d of %d; ref max %.3g" % (r, name, float(d.max()), nz.shape[0], d.numel(), float(a[:, lo:hi].abs().max())))
        if nz.shape[0]:
            ys, xs = nz[:, 2], nz[:, 3]
            print("   rows", sorted(set(ys.tolist()))[:40], "cols", sorted(set(xs.tolist()))[:40])

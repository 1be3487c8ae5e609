"""srx_sn_power_iter_noalloc (the entry point that allocates nothing: its W v scratch rows live in
a static device array of the library) against srx_sn_power_iter (caller-provided scratch), through
the C ABI: same u, v and 1/sigma bit for bit, on two calls in a row, the second with a smaller
problem; sizes beyond the static array are rejected."""
import ctypes as C

import pytest
import torch

from helpers import O, pkg

pytestmark = pytest.mark.gpu

# (rows, cols) of W per layer: the discriminator's ten spectral-norm layers (esrgan/discriminator.py:
# 18-25,30,32), then a smaller problem
BIG = [(64, 27), (64, 576), (128, 576), (128, 1152), (256, 1152), (256, 2304), (512, 2304), (512, 4608),
       (1024, 512), (1, 1024)]
SMALL = [(48, 100), (7, 33), (96, 640)]


def _problem(shapes, salt):
    dev = torch.device("cuda")
    ws = [O.hash_uniform(f"na.w{i}", s, 0.05, salt).to(dev) for i, s in enumerate(shapes)]
    us = [torch.nn.functional.normalize(O.hash_uniform(f"na.u{i}", (s[0],), 1.0, salt), dim=0).to(dev)
          for i, s in enumerate(shapes)]
    vs = [torch.nn.functional.normalize(O.hash_uniform(f"na.v{i}", (s[1],), 1.0, salt), dim=0).to(dev)
          for i, s in enumerate(shapes)]
    return ws, us, vs


def _run(entry, shapes, salt, do_pi, give_wv, stream=None):
    L = pkg("_lib")
    lib = L.lib()
    dev = torch.device("cuda")
    ws, us, vs = _problem(shapes, salt)
    n = len(shapes)
    inv_sigma = torch.zeros(n, dtype=torch.float32, device=dev)
    wv = [torch.full((s[0],), float("nan"), dtype=torch.float32, device=dev) for s in shapes]
    arr = (L.SnLayer * n)()
    for i, s in enumerate(shapes):
        a = arr[i]
        a.w, a.u, a.v = ws[i].data_ptr(), us[i].data_ptr(), vs[i].data_ptr()
        a.inv_sigma = inv_sigma.data_ptr() + 4 * i
        a.wv = wv[i].data_ptr() if give_wv else None
        a.rows, a.cols = s
    table = torch.frombuffer(bytearray(C.string_at(C.addressof(arr), C.sizeof(arr))), dtype=torch.uint8).to(dev)
    torch.cuda.synchronize()          # the table and the operands are in place before another stream reads them
    st = stream if stream is not None else torch.cuda.current_stream()
    rc = getattr(lib, entry)(table.data_ptr(), n, do_pi, max(s[0] for s in shapes), max(s[1] for s in shapes),
                             st.cuda_stream)
    assert rc == 0, lib.srx_last_error()
    torch.cuda.synchronize()
    return [u.cpu() for u in us], [v.cpu() for v in vs], inv_sigma.cpu()


def _same(a, b):
    (ua, va, sa), (ub, vb, sb) = a, b
    assert all(torch.equal(x, y) for x, y in zip(ua, ub))
    assert all(torch.equal(x, y) for x, y in zip(va, vb))
    assert torch.equal(sa, sb)
    assert torch.isfinite(sa).all() and float(sa.abs().min()) > 0


def test_noalloc_power_iteration_equals_the_allocating_entry_point():
    L = pkg("_lib")
    lib = L.lib()
    # the set-up call: size of the static array; also resolves its address on this device, so that
    # the calls below (the first of them on a stream of its own) find it ready
    assert lib.srx_sn_scratch_floats() == 65536
    # first call: the discriminator's layers, training mode (power iteration), on a non-default stream ...
    _same(_run("srx_sn_power_iter_noalloc", BIG, 1, 1, False, stream=torch.cuda.Stream()),
          _run("srx_sn_power_iter", BIG, 1, 1, True))
    # ... second call, smaller problem: whatever the first left in the static array must not matter
    _same(_run("srx_sn_power_iter_noalloc", SMALL, 2, 1, False), _run("srx_sn_power_iter", SMALL, 2, 1, True))
    # eval mode (sigma from the stored u, v; no update)
    _same(_run("srx_sn_power_iter_noalloc", SMALL, 3, 0, False), _run("srx_sn_power_iter", SMALL, 3, 0, True))
    # against the oracle's restatement of the hook too (torch spectral_norm.py:92-114)
    us, vs, inv_sigma = _run("srx_sn_power_iter_noalloc", SMALL, 2, 1, False)
    ws, u0, v0 = _problem(SMALL, 2)
    for i, w in enumerate(ws):
        _, u_ref, v_ref = O.sn_weight(w.cpu(), u0[i].cpu(), v0[i].cpu(), True)
        sigma = torch.dot(u_ref, w.cpu() @ v_ref)
        assert (us[i] - u_ref).abs().max() < 1e-5 and (vs[i] - v_ref).abs().max() < 1e-5
        assert abs(float(inv_sigma[i]) * float(sigma) - 1.0) < 1e-4


def test_noalloc_rejects_sizes_beyond_the_static_scratch():
    L = pkg("_lib")
    lib = L.lib()
    dummy = torch.zeros(64, dtype=torch.uint8, device="cuda")
    rc = lib.srx_sn_power_iter_noalloc(dummy.data_ptr(), 65, 1, 1024, 16, torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    assert b"static" in lib.srx_last_error() and b"66560" in lib.srx_last_error()

"""CPU: the C-ABI library loads and exports every symbol include/srx.h declares; the drop-in
modules keep the reference's state_dict layout, load its checkpoints, and refuse to run
without the HIP path."""
import ctypes
import os
import re

import pytest
import torch

from helpers import O, ROOT, golden, pkg


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "srx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(srx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = pkg("_lib")
    lib = L.lib()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} is declared in include/srx.h but not exported by libsrx.so"
    assert set(syms) == set(L.EXPORTS), set(syms) ^ set(L.EXPORTS)
    assert lib.srx_version() >= 100


def test_binding_struct_sizes_match_library():
    L = pkg("_lib")
    lib = L.lib()
    structs = (L.ConvDesc, L.WgradTask, L.WgradDesc, L.PackSeg, L.PackLayer, L.UnpackItem, L.SnLayer,
               L.SnBwdLayer, L.GenericArgs, L.Op)
    for i, st in enumerate(structs):
        assert lib.srx_sizeof(i) == ctypes.sizeof(st), st.__name__
    assert lib.srx_sizeof(99) == -1


def test_kernel_attributes_are_kept_per_device():
    """The dynamic-LDS attribute of a kernel is set once for every device it is launched on (a
    process that drives two GPUs must not skip the second), again after a failed attempt, and
    never indexed out of range: the library's own host-side check of that bookkeeping."""
    assert pkg("_lib").lib().srx_selftest_per_device_once() == 0


def test_argument_errors_are_reported_not_thrown():
    L = pkg("_lib")
    lib = L.lib()
    assert lib.srx_conv2d(None, None) == 1
    assert b"null descriptor" in lib.srx_last_error()
    d = L.ConvDesc()
    d.dtype, d.ksize, d.stride = 7, 3, 1
    assert lib.srx_conv2d(ctypes.byref(d), None) == 1
    assert b"dtype" in lib.srx_last_error()
    with pytest.raises(RuntimeError):
        L.check(1, "probe")
    # one image's input beyond what a 32-bit buffer descriptor addresses: refused, never mis-addressed
    d = L.ConvDesc(dtype=0, ksize=3, stride=1, B=1, H=8192, W=8192, cin=64, cout=64)
    d.x = d.y = d.w = 16
    assert lib.srx_conv2d(ctypes.byref(d), None) == 1
    assert b"tile the image" in lib.srx_last_error()
    w = L.WgradDesc(dtype=1, ksize=3, stride=1, B=1, H=4096, W=4096, ntasks=1, nslices=1)
    w.tasks = 16
    assert lib.srx_conv2d_wgrad(ctypes.byref(w), None) == 1
    assert b"too large" in lib.srx_last_error()
    assert lib.srx_stream_create(0, None) == 1
    assert b"null" in lib.srx_last_error()


def test_generator_state_dict_and_checkpoint_compat():
    man, arr = golden()
    G = pkg("esrgan.generator")
    g = G.GeneratorESRGAN()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in g.state_dict().items()] == man["g_state_dict"]
    assert [n for n, _ in g.named_parameters()] == [k for k, _, _ in man["g_state_dict"]]
    # a checkpoint written by the REAL reference (esrgan/train_esrgan.py:193 format) loads strictly
    sd = torch.load(os.path.join(ROOT, "tests", "golden", "ref_generator_nf16_nb1.pth"),
                    map_location="cpu", weights_only=True)
    gs = G.GeneratorESRGAN(num_features=16, num_blocks=1)
    res = gs.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # and the oracle on those weights reproduces the reference's output
    out = O.generator_forward(O.hash_input("ckpt.x", (1, 3, 8, 8)), {k: v for k, v in gs.state_dict().items()})
    assert torch.equal(out.detach(), torch.from_numpy(arr["ckpt_g_out"]))


def test_discriminator_state_dict_and_constructor_rng():
    man, arr = golden()
    D = pkg("esrgan.discriminator")
    torch.manual_seed(0)
    d = D.Discriminator()
    sd = d.state_dict()
    assert [[k, list(v.shape), str(v.dtype)] for k, v in sd.items()] == man["d_state_dict"]
    assert {k: v for k, v in sd._metadata.items()} == man["d_metadata"]
    assert [k for k, _ in d.named_parameters()] == man["d_named_parameters"]
    assert [k for k, _ in d.named_buffers()] == man["d_named_buffers"]
    # same RNG consumption as the reference constructor under the same seed
    assert torch.equal(sd["features.0.0.weight_u"], torch.from_numpy(arr["d_ctor_u0"]))
    assert torch.equal(sd["classifier.3.weight_v"], torch.from_numpy(arr["d_ctor_v_last"]))
    assert torch.equal(sd["features.0.0.weight_orig"].flatten()[:16], torch.from_numpy(arr["d_ctor_w0_head"]))
    # reference-format full checkpoint dict round trip (esrgan/train_esrgan.py:199-207)
    d2 = D.Discriminator()
    d2.load_state_dict(sd, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(d.state_dict().values(), d2.state_dict().values()))


def test_modules_fail_loudly_without_gpu():
    G = pkg("esrgan.generator")
    D = pkg("esrgan.discriminator")
    g = G.GeneratorESRGAN(num_blocks=1)
    with pytest.raises(RuntimeError, match="ROCm GPU only"):
        g(torch.zeros(1, 3, 8, 8))
    d = D.Discriminator()
    with pytest.raises(RuntimeError, match="ROCm GPU only"):
        d(torch.zeros(1, 3, 32, 32))


def test_wgrad_task_tables():
    E = pkg("engine")
    # one dense block: 26 valid 32x32 sub-blocks over 8 merged 64x64 tasks
    tasks = []
    for k in range(1, 6):
        tasks += E.Plan.wgrad_tasks_for_layer(64 + 32 * (k - 1), 32 if k < 5 else 64, 0, (k - 1) * 4, 1000 * k,
                                              bias_ptr=7000 + k)
    merged = E.Plan.merge_tasks(tasks)
    assert len(merged) == 8
    assert sum(len(t["sub"]) for t in merged) == 26
    pairs = sum(min(32, e["ld"] - e["ci0"]) * min(32, e["co_n"] - e["co0"]) for t in merged for e in t["sub"].values())
    assert pairs == 64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64
    # each conv's bias handled exactly once per 32-wide cout half
    biases = [(e["out"], e["co0"]) for t in merged for e in t["bias"].values()]
    assert sorted(biases) == sorted([(7001, 0), (7002, 0), (7003, 0), (7004, 0), (7005, 0), (7005, 32)])
    # the two tasks with a single live sub-block (y1 x gy2 of conv2, y3 x gy4 of conv4) pair up
    paired = E.Plan.pair_sparse_tasks(merged)
    assert len(paired) == 7 and sum(len(t["sub"]) for t in paired) == 26
    sp = [t for t in paired if t.get("x_skip") or t.get("gy_skip")]
    assert len(sp) == 1 and sorted(sp[0]["sub"]) == [0, 3]
    t = sp[0]      # x: y1 = octets 8..11, y3 = 16..19 ; gy: gy2 = 4..7, gy4 = 12..15
    assert (t["x_oct"], t["x_skip"], t["x_noct"]) == (8, 4, 8) and (t["gy_oct"], t["gy_skip"], t["gy_noct"]) == (4, 4, 8)
    assert (t["sub"][0]["out"], t["sub"][0]["ci0"]) == (2000, 64) and (t["sub"][3]["out"], t["sub"][3]["ci0"]) == (4000, 128)
    # ragged layer (Cin=3, Cout=64): one task, two sub-blocks (cin half 0 only)
    t = E.Plan.wgrad_tasks_for_layer(3, 64, 0, 0, 1)
    assert len(t) == 1 and sorted(t[0]["sub"]) == [0, 2] and t[0]["x_noct"] == 1


def test_row_permutation_is_a_bijection():
    # mirror of srx_row_to_chan in csrc/srx_common.h
    def row_to_chan(row, perm):
        h = (row >> 2) & 1
        i = (row & 3) | ((row >> 3) << 2)
        return 8 * (2 * (i >> 3) + h) + (i & 7) if perm == 0 else 4 * (i & 7) + 2 * h + (i >> 3)
    for perm in (0, 1):
        assert sorted(row_to_chan(r, perm) for r in range(32)) == list(range(32))


def test_roctx_ranges_are_off_by_default_and_switchable():
    """utils.trace: ranges cost nothing unless enabled; enabling loads the ROCm marker library."""
    T = pkg("utils.trace")
    assert not T.enabled()
    with T.range_("nothing"):        # no-op
        pass
    T.push("x"); T.pop()
    T.enable(True)
    try:
        assert T.enabled()
        with T.range_("phase"):
            T.push("inner"); T.pop()
    finally:
        T.enable(False)
    assert not T.enabled()

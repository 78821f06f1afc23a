"""CPU-only tests of the host logic: the C ABI library loads and exports every symbol the
header declares, config parsing, the nn.Module surface (state_dict keys, errors), flat
parameter layout, and the 2-rank gradient exchange over gloo.  No compute calls (no GPU here)."""
import json
import os
import subprocess
import sys
import numpy as np
import pytest
import torch

import detdata
from make_golden_cfg import NETS


def test_library_exports_every_declared_symbol():
    from fplx import _lib
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 29
    for n in names:
        assert hasattr(lib, n), n
    assert lib.fplx_version() >= 1
    # every exported fplx_ symbol is declared (nm view)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if " T " in l)      # every exported function, whatever its name
    assert exported == names, set(exported) ^ set(names)
    # error path works without a GPU: bad arguments are rejected before any launch
    assert lib.fplx_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, 1.0, None) == -5
    assert "adam_step" in _lib.last_error()
    assert lib.fplx_num_partials(1) == 1 and lib.fplx_num_partials(10 ** 9) == 512


def test_tuning_table_round_trip_and_env_translation():
    """fplx_set_tuning / fplx_get_tuning / fplx_tuning_key, and the FPLX_<KEY> environment translation of fplx/_lib.py"""
    from fplx import _lib
    keys = _lib.tuning_keys()
    assert "brick" in keys and "march32_v2" in keys and "xcd" in keys and len(set(keys)) == len(keys) >= 20
    assert _lib.get_tuning("brick") == 1 and _lib.get_tuning("march32_v2") == 4
    e0 = _lib.tuning_epoch
    _lib.set_tuning("brick", 3)
    try:
        assert _lib.get_tuning("brick") == 3
    finally:
        _lib.set_tuning("brick", 1)
    assert _lib.tuning_epoch == e0 + 2           # what Engine.forward compares its caches' epoch with (packs, folds, plans)
    with pytest.raises(ValueError):
        _lib.set_tuning("no_such_knob", 1)
    assert _lib.tuning_epoch == e0 + 2           # a refused knob changes nothing
    code = "import sys; sys.path.insert(0, %r); from fplx import _lib; print(_lib.get_tuning('brick'), _lib.get_tuning('tile_ks'))" % (
        os.path.dirname(os.path.dirname(_lib.__file__)),)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, FPLX_BRICK="0", FPLX_TILE_KS="9"))
    assert out.stdout.split() == ["0", "9"], out.stderr[-500:]


def test_brick_kernel_plan_for_the_benchmark_layers():
    """host-side dispatch of conv_fwd_brick (conv_brick.hip): geometry, Cin split and statistics rows for the benchmark's
    levels - pure host code, no launch (fplx_conv3d_plan_query)"""
    import ctypes
    from fplx import _lib
    lib = _lib.lib()

    def plan(n, d, h, w, cin, cout):
        """(brick kernel?, geometry, Cin split, statistics rows) through the declared plan query"""
        kern, g, k, r = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert lib.fplx_conv3d_plan_query(n, d, h, w, cin, cout, 3, 3, 3, _lib.BF16, _lib.BF16, ctypes.byref(kern), ctypes.byref(g),
                                          ctypes.byref(k), ctypes.byref(r)) == 0
        return int(kern.value == BRICK), g.value, k.value, r.value, kern.value

    GENERIC, DIRECT, TILE, STREAM, MARCH, BRICK, STEM, OUTCONV = range(8)           # FPLX_KERNEL_* of include/fplx.h
    assert plan(2, 40, 80, 80, 64, 64)[:4] == (1, 0, 1, 2000)      # level 1: 4 x 8 x 8 bricks, 64 channels per block: rows = bricks
    assert plan(2, 20, 40, 40, 128, 128)[:4] == (1, 0, 1, 250)     # level 2
    assert plan(2, 10, 20, 20, 256, 256)[:4] == (1, 1, 2, 512)     # level 3: 5 x 4 x 8 bricks, Cin split in two: the finish kernel's rows
    assert plan(2, 10, 20, 20, 256, 512)[:4] == (1, 1, 1, 60)
    assert plan(2, 5, 10, 10, 512, 512)[:4] == (1, 1, 4, 125)      # level 4 pads 1.9x as 5 x 4 x 8 bricks and still beats the tile kernel (round 4)
    assert plan(2, 5, 10, 10, 512, 256)[:4] == (1, 1, 8, 125)
    assert plan(2, 8, 8, 8, 512, 512)[4] == TILE                   # too few blocks even with the deepest Cin split
    assert plan(2, 80, 160, 160, 32, 32)[4] == MARCH and plan(2, 80, 160, 160, 64, 32)[4] == MARCH   # level 0: the march kernels'
    assert plan(2, 80, 160, 160, 48, 48)[4] == GENERIC
    # the rows the dispatcher promises: bricks, or the split-K finish kernel's blocks
    assert lib.fplx_conv3d_stats_rows(2, 20, 40, 40, 128, 128, 3, 3, 3, 1, 1) == 250
    assert lib.fplx_conv3d_stats_rows(2, 10, 20, 20, 256, 256, 3, 3, 3, 1, 1) == 512
    assert lib.fplx_conv3d_fwd_ws_bytes(2, 10, 20, 20, 256, 256, 3, 3, 3, 1, 1) == 2 * 8000 * 256 * 4
    assert lib.fplx_conv3d_fwd_ws_bytes(2, 20, 40, 40, 128, 128, 3, 3, 3, 1, 1) == 0


def test_parse_config_matches_reference_parser(golden_dir):
    import fplx
    cfg = fplx.parse_config(os.path.join(golden_dir, "sample_vs.cfg"))
    ref = json.load(open(os.path.join(golden_dir, "sample_vs.cfg.json")))
    assert json.loads(json.dumps(cfg)) == ref
    assert cfg["network"]["feature_chns"] == [32, 64, 128, 256, 512]
    assert cfg["testing"]["domian_label"] == 1 and cfg["training"]["learning_rate"] == 1e-4
    assert cfg["dataset"]["normalizewithmeanstd_mean"] is None
    cfg = fplx.synchronize_config(cfg)
    assert cfg["dataset"]["labeltoprobability_class_num"] == 2


def test_net_module_surface():
    import fplx
    p = dict(NETS["tiny"])
    net = fplx.SegNetDict["UNet2D5_dsbn"](p)
    want = detdata.state_dict_3d(p)
    sd = net.state_dict()
    assert sorted(sd.keys()) == sorted(want.keys())
    for k, v in want.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    # reference checkpoints carry dead 2D twins: accepted and ignored
    ref_like = {k: torch.from_numpy(v) for k, v in want.items()}
    ref_like["block0.conv.conv2d_1.weight"] = torch.zeros(8, 1, 3, 3)
    ref_like["block0.conv.bn2d1.bns.0.weight"] = torch.zeros(8)
    ref_like["up1.trans2d.weight"] = torch.zeros(128, 64, 2, 2)
    ref_like["up1.conv3d.weight"] = torch.zeros(64, 128, 1, 1, 1)
    net.load_state_dict(ref_like)
    np.testing.assert_array_equal(net.out_conv.weight.detach().numpy(), want["out_conv.weight"])
    for bad in ({"conv_dims": [2, 2, 3, 3]}, {"conv_dims": [1, 2, 3, 3, 3]}, {"precision": "fp8"}):
        q = dict(p)
        q.update(bad)
        with pytest.raises(ValueError):
            fplx.UNet2D5_dsbn(q)
    with pytest.raises(RuntimeError):          # no CPU path
        net(torch.zeros(1, 1, 16, 32, 32), domain_label=torch.zeros(1, dtype=torch.long))
    # the shipped configs' 2.5D pattern: the live members of the 2D levels carry the reference's 2D key names / shapes
    q = dict(p)
    q["conv_dims"] = [2, 2, 3, 3, 3]
    n25 = fplx.UNet2D5_dsbn(q)
    sd25 = n25.state_dict()
    assert len(sd25) == len(sd) and tuple(sd25["block0.conv.conv2d_1.weight"].shape) == (8, 1, 3, 3)
    assert "up4.trans2d.weight" in sd25 and "up3.trans2d.weight" in sd25 and "up2.trans3d.weight" in sd25
    assert "block1.conv.bn2d2.bns.1.running_var" in sd25 and "block2.conv.bn3d1.bns.0.weight" in sd25
    # test-time dropout switch of the reference (agent_seg.py:845-852) reaches our Dropout children
    q = dict(p)
    q["dropout"] = [0, 0, 0.3, 0.4, 0.5]
    net = fplx.UNet2D5_dsbn(q)
    net.eval()
    assert net.dropout_active() == [False] * 9
    net.apply(lambda m: m.train() if type(m) == torch.nn.Dropout else None)
    assert net.dropout_active() == [False, False, True, True, True, True, True, False, False]


def test_flat_layout_and_buckets():
    import fplx
    p = dict(NETS["tiny"])
    net = fplx.UNet2D5_dsbn(p)
    before = {k: v.detach().clone() for k, v in net.named_parameters()}
    net._ensure_flat()
    total = sum((v.numel() + 3) // 4 * 4 for v in before.values())      # every parameter starts on a 16-byte boundary
    assert net.flat_params.numel() == total
    for k, v in net.named_parameters():          # values preserved, storage shared
        assert torch.equal(v, before[k])
        o, n, shp = net._layout[k]
        assert o % 4 == 0 and v.data_ptr() == net.flat_params.data_ptr() + 4 * o
    used = torch.zeros(total, dtype=torch.bool)
    for o, n, _ in net._layout.values():
        used[o:o + n] = True
    assert float(net.flat_params[~used].abs().sum()) == 0.0              # the padding is zero
    shared, doms = net.segments()
    assert shared[0] == 0 and doms[-1][1] == total and len(doms) == 2
    assert doms[0][1] - doms[0][0] == doms[1][1] - doms[1][0] == 2 * 2 * sum([8, 16, 32, 64, 128, 64, 32, 16, 8])
    for d in (0, 1):
        names = net.active_param_names(d)
        assert all((".bns.%d." % d) in k for k in names if ".bns." in k)
    b = net.bucket_ranges(1 << 16)
    assert b[0][0] == 0 and b[-1][1] == shared[1] and all(x[1] == y[0] for x, y in zip(b, b[1:]))
    # production order: the head of the flat buffer is the decoder end of the net
    assert net._order[0] == "out_conv.weight" and net._order[net._n_shared_names - 1] == "block0.conv.relu_1.weight"


def test_loss_registry_and_errors_cpu():
    import fplx
    assert set(fplx.SegLossDict) == {"DiceLoss", "CrossEntropyLoss", "DiceLoss_weight"}
    c = fplx.CombinedLoss({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.6, 0.4]}, fplx.SegLossDict)
    assert c.terms == (0.6, 0.4, 0.0, 0.0)
    with pytest.raises(ValueError):
        fplx.make_loss({"loss_type": "FocalDiceLoss"})
    with pytest.raises(RuntimeError):          # CPU tensors are refused, not silently computed
        fplx.DiceLoss()({"prediction": torch.zeros(1, 2, 2, 2, 2), "ground_truth": torch.zeros(1, 2, 2, 2, 2)})
    with pytest.raises(ValueError):
        fplx.get_optimizer("SGD", fplx.UNet2D5_dsbn(dict(NETS["tiny"])), {"learning_rate": 1e-3, "weight_decay": 0})


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(sys.argv[3], "fpl-plus_amd"))
from fplx.ddp import GradAllReducer
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 1000
buckets = [(0, 300), (300, 700), (700, 900)]
doms = [(900, 950), (950, 1000)]
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
red = GradAllReducer(buckets, doms)
assert red.enabled and red.world == world
red.begin(g)
red.ready(250)            # nothing complete yet
assert red._next == 0
red.ready(700)            # buckets 0 and 1
assert red._next == 2
red.finish([1])           # last bucket + domain 1 only
exp = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
own = torch.arange(n, dtype=torch.float32) * (rank + 1)
assert torch.equal(g[:900], exp[:900]) and torch.equal(g[950:], exp[950:])
assert torch.equal(g[900:950], own[900:950])          # inactive domain untouched
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_grad_allreduce_two_ranks_gloo(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", root, port], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_reference_key_order_rule_matches_reference_dump(golden_dir):
    """fplx.checkpoint.reference_state_keys / reference_param_names vs the key lists dumped from the reference network
    (tests/golden/make_golden_ckpt.py): 484 state_dict keys, 268 parameters, same order."""
    import json
    import os
    from fplx import checkpoint as C
    k = json.load(open(os.path.join(golden_dir, "ref_state_keys.json")))
    assert C.reference_state_keys(2) == k["state_dict"] and len(k["state_dict"]) == 484
    assert C.reference_param_names(2) == k["named_parameters"] and len(k["named_parameters"]) == 268


def test_filter_cut_points_are_numpys():
    """The FPL boundary count (agent_seg.py:922-924) is decided on the float32 mean with two cut points compiled into
    csrc/loss_filter.hip: they must be exactly where numpy's float32 u(m) = -m*log(m+1e-6) crosses 0.01."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "fpl-plus_amd", "csrc", "loss_filter.hip")).read()
    m = re.search(r"FPL_CUT_LO = (0x[0-9A-Fa-f]+)u, FPL_CUT_HI = (0x[0-9A-Fa-f]+)u", src)
    lo, hi = int(m.group(1), 16), int(m.group(2), 16)

    def pred(bits):
        v = np.asarray(bits, np.uint32).view(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            return (-1.0 * (v * np.log(v + 1e-6))) > 0.01
    for cut in (lo, hi):
        w = np.arange(cut - 4096, cut + 4097, dtype=np.uint32)
        assert np.array_equal(pred(w), (w >= lo) & (w <= hi))
    coarse = np.arange(0, int(np.float32(1.0).view(np.uint32)) + 1, 4099, dtype=np.uint32)
    assert np.array_equal(pred(coarse), (coarse >= lo) & (coarse <= hi))


def test_bench_refuses_a_world_size_that_is_not_its_gpus_argument():
    """`bench.py --gpus N` must never print a line for another world size (--gpus was parsed and ignored):
    under a launcher with WORLD_SIZE != N it exits 2 before touching any GPU; without a launcher and N > 1 it starts the
    ranks itself - here, without GPUs, the child fails and the parent relays the failure instead of a result line."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and "{" not in r.stdout
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and '"metric"' not in r.stdout and "2-rank child" in r.stderr


def test_partial_rows_contract():
    """fplx_num_partials (what callers size the reductions' partial-row buffers from, round 5: one row per 16 voxels up to 32768
    voxels, one per 64 above, at most 512, monotonic) - no launch"""
    from fplx import _lib
    lib = _lib.lib()
    rows = [lib.fplx_num_partials(v) for v in (1, 16, 17, 1000, 8000, 8192, 32768, 32769, 64000, 4096000)]
    assert rows == [1, 1, 2, 63, 500, 512, 512, 512, 512, 512]
    assert all(a <= b for a, b in zip(rows, rows[1:]))
    _lib.set_tuning("rows_small_div", 64)                 # A/B knob: round 4's rows
    try:
        assert [lib.fplx_num_partials(v) for v in (1000, 8000, 64000)] == [16, 125, 512]
    finally:
        _lib.set_tuning("rows_small_div", 16)



def test_late_import_warns_about_the_runtime_environment():
    """fplx/_lib.py: GPU_MAX_HW_QUEUES / HIP_FORCE_DEV_KERNARG only reach the HIP runtime if fplx is imported before the first GPU
    call.  A host program that initialised the GPU first (PyMIC imports torch, builds its device, reaches
    fplx through the registry) gets a RuntimeWarning naming what it lost; a preset variable or an early import stays silent."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    prog = ("import sys, warnings, torch\n"
            "sys.path.insert(0, %r)\n"
            "late = sys.argv[1] == '1'\n"
            "torch.cuda.is_initialized = lambda: late\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    from fplx import _lib\n"
            "msgs = [str(x.message) for x in w if issubclass(x.category, RuntimeWarning)]\n"
            "print(len(msgs), sorted(_lib.runtime_env_report().items()))\n"
            "print('|'.join(msgs))\n") % os.path.join(root, "fpl-plus_amd")
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG")}

    def run(late, extra):
        r = subprocess.run([sys.executable, "-c", prog, "1" if late else "0"], env=dict(base, **extra), capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        return r.stdout
    out = run(True, {})
    assert out.startswith("1 ") and "GPU_MAX_HW_QUEUES / HIP_FORCE_DEV_KERNARG" in out and "'too late'" in out
    out = run(True, {"GPU_MAX_HW_QUEUES": "8"})                      # one preset: only the other is named
    assert out.startswith("1 ") and "HIP_FORCE_DEV_KERNARG was not" in out and "('8', 'preset')" in out
    out = run(True, {"GPU_MAX_HW_QUEUES": "4", "HIP_FORCE_DEV_KERNARG": "0"})     # the user's own settings win, silently
    assert out.startswith("0 ") and "('4', 'preset')" in out and "('0', 'preset')" in out
    out = run(False, {})                                             # imported in time: set by fplx, no warning
    assert out.startswith("0 ") and "('8', 'fplx')" in out and "('1', 'fplx')" in out

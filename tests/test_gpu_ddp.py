"""Data parallelism through the HIP path with TWO ranks: this box has one GPU, so both processes use cuda:0 and talk over
gloo (FPLX_DDP_BACKEND=gloo; RCCL refuses two ranks on one device) - the collectives, bucketing, full-batch loss and the
agent's sharding are the code that runs over RCCL on a node.  The 2-rank results are compared with a single-process
emulation of the reference's nn.DataParallel (per-chunk BatchNorm, one loss over the gathered batch, gradients added)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import detdata
from make_golden_cfg import NETS
from util import load_det_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
root = sys.argv[1]
for p in (root, os.path.join(root, "fpl-plus_amd"), os.path.join(root, "tests", "golden"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import fplx, detdata
from fplx import ddp
from make_golden_cfg import NETS
from util import load_det_weights
assert ddp.init_from_env()
rank, world = ddp.rank(), ddp.world_size()
torch.cuda.set_device(0)
p = dict(NETS["tiny"])
net = fplx.UNet2D5_dsbn(p)
load_det_weights(net, p, "cuda")
net._ensure_flat()
ddp.broadcast_params_from_rank0(net)
ts = fplx.TrainStep(net, (1.0, 0.5, 0.0, 0.0), True, lr=1e-3, weight_decay=1e-5, bucket_elems=1 << 14)
assert ts.reducer.enabled and ts.reducer.world == 2 and len(ts.reducer.buckets) > 2
N = 4
xs = torch.from_numpy(detdata.normal("ddp.x", (N, 1, 16, 32, 32)))
ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)]))
per = N // world
x, y = xs[rank * per:(rank + 1) * per].cuda(), ys[rank * per:(rank + 1) * per].cuda()
losses = []
for it in range(2):
    out = ts.step(x, y, it % 2)
    losses.append(float(out[0].item()))
torch.cuda.synchronize()
np.savez(sys.argv[2] + ".%d.npz" % rank, flat=net.flat_params.cpu().numpy(), losses=np.array(losses))
ddp.barrier()
print("OK", rank)
'''


def _launch(tmp_path, script_text, extra=(), nproc=2, timeout=900):
    script = tmp_path / "worker.py"
    script.write_text(script_text)
    env = dict(os.environ, FPLX_DDP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = str(24500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", port, str(script), ROOT] + list(extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    return r


def test_two_rank_train_step_equals_dataparallel_emulation(tmp_path):
    import fplx
    from fplx import ops
    out = str(tmp_path / "res")
    _launch(tmp_path, _WORKER, [out])
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["flat"], r1["flat"])                       # replicas stay identical
    assert np.array_equal(r0["losses"], r1["losses"])                   # every rank reports the full-batch loss
    # ---- single process: chunk forwards (per-chunk BN statistics), ONE loss over both chunks, gradients added, Adam
    p = dict(NETS["tiny"])
    net = fplx.UNet2D5_dsbn(p)
    load_det_weights(net, p, "cuda")
    net._ensure_flat()
    net.train()
    opt = fplx.FusedAdam(net, 1e-3, weight_decay=1e-5)
    N = 4
    xs = torch.from_numpy(detdata.normal("ddp.x", (N, 1, 16, 32, 32))).cuda()
    ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)])).cuda()
    terms, one = (1.0, 0.5, 0.0, 0.0), torch.ones(1, device="cuda")
    losses = []
    for it in range(2):
        dom = it % 2
        fw = []
        for c in range(2):
            x, y = xs[2 * c:2 * c + 2].contiguous(), ys[2 * c:2 * c + 2].contiguous()
            logits, sv = net.engine.forward(x, dom, True, net.dropout_active(), net.dropout_seed, 10 * it + c, keep=True)
            n, cc = logits.shape[0], logits.shape[1]
            v = logits[0, 0].numel()
            part = torch.empty((n, ops.loss_rows(v), ops.loss_k(cc)), device="cuda")
            sums = torch.empty((n + 1, ops.loss_k(cc)), dtype=torch.float64, device="cuda")
            ops.call("fplx_seg_loss_sums", ops.ptr(logits), ops.ptr(y), 0, n, cc, v, 1, ops.ptr(part), ops.ptr(sums[:n]),
                     ops.ptr(sums[n]), ops.stream())
            fw.append((logits, sv, y, sums))
        tot = fw[0][3][2] + fw[1][3][2]
        g = torch.zeros_like(net.flat_params)
        for logits, sv, y, sums in fw:
            outv = torch.empty(4 + 2, device="cuda")
            coef = torch.empty(2 * 2 * 2 + 2, device="cuda")
            ops.call("fplx_seg_loss_from_sums", ops.ptr(sums[:2]), ops.ptr(tot), 0, 2, 4, 2, logits[0, 0].numel(), 0, terms[0],
                     terms[1], terms[2], terms[3], ops.ptr(outv), ops.ptr(coef), ops.stream())
            dl = torch.empty_like(logits)
            ops.seg_loss_bwd(logits, y, None, coef, one, terms, True, dl)
            gi = torch.empty_like(net.flat_params)
            net.engine.backward(sv, dl, gi)
            g += gi
        losses.append(float(outv[0].item()))
        opt.step_flat(g, [dom])
        net.engine.invalidate()
    torch.cuda.synchronize()
    np.testing.assert_allclose(r0["losses"], losses, rtol=0, atol=2e-6)
    ref = net.flat_params.cpu().numpy()
    # the same kernels on the same chunks; what differs is where the two chunks' gradients are added (gloo's sum against
    # `g += gi`) and the loss sums' all-reduce: Adam normalises the step, so the comparison is on the update, scaled by lr
    # (2 steps of at most lr = 1e-3 each)
    d_self = np.abs(r0["flat"] - ref)
    print("two ranks vs fplx emulation: max %.3g, share above 1e-6: %.3g" % (d_self.max(), (d_self > 1e-6).mean()))
    assert d_self.max() <= 1e-7, d_self.max()           # (measured: 0 - two ranks' sums commute; 1e-7 leaves room for a reordered all-reduce)
    # ---- the ORACLE's DataParallel emulation of the same two steps (not only fplx against itself): replica
    # forwards on the chunks with per-chunk BatchNorm statistics, gather, ONE loss over the full batch, backward, Adam
    # (agent_seg.py:692-698 + 336-357; the numbers tests/test_ddp_equivalence_cpu.py pins for the gradient exchange on CPU)
    from oracle import torch_ref as R
    sd, prm = R.split_state(detdata.state_dict_3d(p))
    opt_r = R.AdamRef(prm, 1e-3, 1e-5)
    xs_c, ys_c = xs.cpu(), ys.cpu()
    o_losses = []
    for it in range(2):
        dom = it % 2
        opt_r.zero_grad()
        lg = torch.cat([R.unet_forward(sd, p, xs_c[i:i + 2], dom, True) for i in (0, 2)], 0)
        l = 1.0 * R.dice_loss(lg, ys_c) + 0.5 * R.ce_loss(lg, ys_c)
        l.backward()
        opt_r.step()
        o_losses.append(float(l.item()))
    np.testing.assert_allclose(r0["losses"], o_losses, rtol=0, atol=5e-5)
    flat = r0["flat"]
    checked = 0
    for k in net._order:
        o, n, shp = net._layout[k]
        if k not in prm or (k.endswith("bias") and "conv3d" in k):
            continue          # conv bias under train-mode BN: the oracle random-walks on fp noise (Adam), fplx decays it
        diff = np.abs(flat[o:o + n].reshape(shp) - prm[k].detach().numpy())
        refv = np.abs(prm[k].detach().numpy())
        assert diff.max() <= 1e-3 * 2 + 1e-6, (k, diff.max())
        # (Adam normalises the step: an element whose gradient is fp32 noise around zero moves by up to lr either way - one such
        # element per small tensor is allowed, 1 % of a large one)
        out = int((diff > 5e-5 * 2 + 1e-4 * refv).sum())
        assert out <= max(1, diff.size // 100), (k, out, diff.size)
        checked += 1
    assert checked > 40


_WORKER_ALL = r"""
import os, sys
root, out, mode = sys.argv[1], sys.argv[2], sys.argv[3]
for p in (root, os.path.join(root, "fpl-plus_amd"), os.path.join(root, "tests", "golden"), os.path.join(root, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import fplx, detdata
from fplx import ddp
from make_golden_cfg import NETS
from util import load_det_weights
p = dict(NETS["tiny"])
N = 4
data = []
for dom in (0, 1):
    xs = torch.from_numpy(detdata.normal("ddp.all.x%d" % dom, (N, 1, 16, 32, 32)))
    ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)][dom:] +
                                             [(0, 1, -2)][:dom]))
    data.append((xs, ys))
terms = (0.5, 0.5, 0.0, 0.0)
launched_before_finish, n_buckets = [], 0
if mode == "ts":
    assert ddp.init_from_env()
    rank, world = ddp.rank(), ddp.world_size()
    torch.cuda.set_device(0)
    net = fplx.UNet2D5_dsbn(p)
    load_det_weights(net, p, "cuda")
    net._ensure_flat()
    ddp.broadcast_params_from_rank0(net)
    ts = fplx.TrainStep(net, terms, True, lr=1e-3, weight_decay=1e-5, bucket_elems=1 << 14)
else:
    tcfg = {"dis": False, "train_fpl_uda": True, "loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.5, 0.5],
            "optimizer": "Adam", "learning_rate": 1e-3, "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR",
            "lr_gamma": 0.5, "lr_milestones": [100], "iter_valid": 1, "gpus": [0, 0]}
    cfg = {"dataset": {"tensor_type": "float"}, "network": dict(p), "training": tcfg, "testing": {}}
    agent = fplx.SegmentationAgent(cfg, "train")
    assert agent.distributed and agent.world == 2
    rank, world = agent.rank, agent.world
    agent.create_network()
    load_det_weights(agent.net, cfg["network"], "cuda")
    agent.net._ensure_flat()
    ddp.broadcast_params_from_rank0(agent.net)
    agent.create_optimizer()
    agent.create_loss_calculator()
    agent.engine_mode = mode == "agent_engine"
    ts = agent._engine_step()
    assert (ts is not None) == (mode == "agent_engine")       # the distributed agent runs the engine step (round 5)
    net = agent.net
per = N // world
b = [{"image": data[d][0][rank * per:(rank + 1) * per].cuda(), "label_prob": data[d][1][rank * per:(rank + 1) * per].cuda()}
     for d in (0, 1)]
if ts is not None:
    assert ts.reducer.enabled and ts.reducer.world == 2
    n_buckets = len(ts.reducer.buckets)
    inner = ts.reducer.finish
    def finish(active, _inner=inner):
        launched_before_finish.append(list(ts.reducer.launched))
        return _inner(active)
    ts.reducer.finish = finish
losses = []
for it in range(2):
    if mode == "ts":
        outs = ts.step_all(b)
        losses.append(0.5 * (float(outs[0][0].item()) + float(outs[1][0].item())))
    else:
        agent.set_loaders([b[0]], [b[1]])
        losses.append(agent.training_all()["loss"] * 2)       # the agent reports loss / num_domains (train_avg_loss)
torch.cuda.synchronize()
np.savez(out + ".%d.npz" % rank, flat=net.flat_params.detach().cpu().numpy(), losses=np.array(losses),
         before=np.array([len(l) for l in launched_before_finish]), n_buckets=n_buckets,
         first=np.array(launched_before_finish[0][0] if launched_before_finish and launched_before_finish[0] else (-1, -1)),
         dom0=np.array(ts.reducer.domain_ranges[0] if ts is not None else (-1, -1)))
ddp.barrier()
print("OK", rank)
"""


def _emulate_step_all(terms, iters=2):
    """single process = the reference's nn.DataParallel on a training_all iteration (agent_seg.py:459-495, 692-698): per domain
    the replica forwards on the chunks (per-chunk BatchNorm statistics), ONE loss over the gathered batch, loss = (l0 + l1) / 2,
    the replicas' gradients added, one Adam step"""
    import fplx
    from fplx import ops
    p = dict(NETS["tiny"])
    net = fplx.UNet2D5_dsbn(p)
    load_det_weights(net, p, "cuda")
    net._ensure_flat()
    net.train()
    opt = fplx.FusedAdam(net, 1e-3, weight_decay=1e-5)
    N = 4
    data = []
    for dom in (0, 1):
        xs = torch.from_numpy(detdata.normal("ddp.all.x%d" % dom, (N, 1, 16, 32, 32))).cuda()
        ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)][dom:] +
                                                 [(0, 1, -2)][:dom])).cuda()
        data.append((xs, ys))
    half = torch.full((1,), 0.5, device="cuda")
    losses = []
    step = 0
    for it in range(iters):
        g = torch.zeros_like(net.flat_params)
        tot_loss = 0.0
        for dom in (0, 1):
            xs, ys = data[dom]
            fw = []
            for c in range(2):
                x, y = xs[2 * c:2 * c + 2].contiguous(), ys[2 * c:2 * c + 2].contiguous()
                logits, sv = net.engine.forward(x, dom, True, net.dropout_active(), net.dropout_seed, step, keep=True)
                step += 1
                n, cc = logits.shape[0], logits.shape[1]
                v = logits[0, 0].numel()
                part = torch.empty((n, ops.loss_rows(v), ops.loss_k(cc)), device="cuda")
                sums = torch.empty((n + 1, ops.loss_k(cc)), dtype=torch.float64, device="cuda")
                ops.call("fplx_seg_loss_sums", ops.ptr(logits), ops.ptr(y), 0, n, cc, v, 1, ops.ptr(part), ops.ptr(sums[:n]),
                         ops.ptr(sums[n]), ops.stream())
                fw.append((logits, sv, y, sums))
            tot = fw[0][3][2] + fw[1][3][2]
            for logits, sv, y, sums in fw:
                outv = torch.empty(4 + 2, device="cuda")
                coef = torch.empty(2 * 2 * 2 + 2, device="cuda")
                ops.call("fplx_seg_loss_from_sums", ops.ptr(sums[:2]), ops.ptr(tot), 0, 2, 4, 2, logits[0, 0].numel(), 0, terms[0],
                         terms[1], terms[2], terms[3], ops.ptr(outv), ops.ptr(coef), ops.stream())
                dl = torch.empty_like(logits)
                ops.seg_loss_bwd(logits, y, None, coef, half, terms, True, dl)
                gi = torch.empty_like(net.flat_params)
                net.engine.backward(sv, dl, gi)
                g += gi
            tot_loss += 0.5 * float(outv[0].item())
        losses.append(tot_loss)
        opt.step_flat(g, [0, 1])
        net.engine.invalidate()
    torch.cuda.synchronize()
    return net.flat_params.cpu().numpy(), losses


def _same_after_adam(flat, ref, lr=1e-3, steps=2):
    """parameters after `steps` Adam steps of two runs that add the same fp32 gradient terms in another order: Adam normalises
    the step (the first one is lr * sign(g)), so an element whose gradient is round-off around zero (|g| < 1e-7) may move
    by up to lr per step either way - a hard bound for every element, and all but a sliver of them equal to 5 % of a step"""
    diff = np.abs(flat - ref)
    assert diff.max() <= 2.0 * lr * steps * 1.05, diff.max()
    assert float(np.mean(diff > 0.05 * lr * steps)) < 2e-3, float(np.mean(diff > 0.05 * lr * steps))
    assert float(np.mean(diff)) < 0.002 * lr * steps, float(np.mean(diff))


def test_two_rank_step_all_overlaps_the_exchange_and_equals_dataparallel(tmp_path):
    """TrainStep.step_all (one training_all iteration, agent_seg.py:459-495) with two ranks - the buckets are
    folded and all-reduced DURING the last domain's backward (domain 0's BatchNorm segment first, every bucket but the tail
    before the flush), the result is bit-identical to the run that exchanges everything behind the last kernel
    (FPLX_DDP_OVERLAP_ALL=0) and equals the single-process DataParallel emulation."""
    res = {}
    for ov in ("1", "0"):
        out = str(tmp_path / ("res" + ov))
        os.environ["FPLX_DDP_OVERLAP_ALL"] = ov
        try:
            _launch(tmp_path, _WORKER_ALL, [out, "ts"])
        finally:
            os.environ.pop("FPLX_DDP_OVERLAP_ALL", None)
        r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
        assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["losses"], r1["losses"])
        res[ov] = r0
    on, off = res["1"], res["0"]
    nb = int(on["n_buckets"])
    assert nb > 3
    # overlapped: domain 0's BatchNorm segment + all buckets were launched from inside the backward (the engine's last block
    # boundary completes the last bucket); not overlapped: nothing before the flush
    assert list(on["before"]) == [nb + 1, nb + 1] and list(off["before"]) == [0, 0]
    assert tuple(on["first"]) == tuple(on["dom0"])
    assert np.array_equal(on["flat"], off["flat"]) and np.array_equal(on["losses"], off["losses"])
    ref, losses = _emulate_step_all((0.5, 0.5, 0.0, 0.0))
    np.testing.assert_allclose(on["losses"], losses, rtol=0, atol=2e-6)
    _same_after_adam(on["flat"], ref)


@pytest.mark.parametrize("mode", ["agent_engine", "agent_autograd"])
def test_two_rank_agent_training_all_equals_dataparallel(tmp_path, mode):
    """SegmentationAgent.training_all under torch.distributed.run with two ranks: the engine route (the default since round 5:
    full-batch loss, bucketed exchange during backward) and the autograd route (all-reduce inside FusedAdam.step) against the
    single-process DataParallel emulation; the engine route gives the bits of TrainStep.step_all."""
    out = str(tmp_path / "res")
    _launch(tmp_path, _WORKER_ALL, [out, mode])
    r0, r1 = np.load(out + ".0.npz"), np.load(out + ".1.npz")
    assert np.array_equal(r0["flat"], r1["flat"])
    ref, losses = _emulate_step_all((0.5, 0.5, 0.0, 0.0))
    np.testing.assert_allclose(r0["losses"], losses, rtol=0, atol=5e-6)
    _same_after_adam(r0["flat"], ref)
    if mode == "agent_engine":
        assert list(r0["before"]) == [int(r0["n_buckets"]) + 1] * 2
        out2 = str(tmp_path / "res_ts")
        _launch(tmp_path, _WORKER_ALL, [out2, "ts"])
        t0 = np.load(out2 + ".0.npz")
        # the agent's TrainStep uses the default bucket size, the bare one 1 << 14: other ranges, the same sums (two ranks: one add)
        assert np.array_equal(r0["flat"], t0["flat"])


_AGENT_WORKER = r'''
import os, sys
root, cfg = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, "fpl-plus_amd"))
import random, numpy as np, torch
random.seed(3)
from fplx import net_run, ddp
res = net_run.main(["fplx.net_run", sys.argv[3], cfg])
from fplx import ddp
print("OK", ddp.rank(), ddp.world_size())
'''


def test_agent_train_and_sharded_fpl_inference_with_two_ranks(tmp_path):
    """`python -m torch.distributed.run ... fplx.net_run train cfg` with two ranks: global batch 2 = one crop per rank, rank 0
    writes the checkpoints, both ranks test; then the FPL stage (testing.fpl) shards the volumes over the ranks and rank 0
    writes the sorted (uncertainty, name) list with every volume in it."""
    from fplx import nifti
    from test_gpu_25d import SHIPPED_STYLE_CFG
    root = tmp_path
    rs = np.random.RandomState(9)
    (root / "img").mkdir()
    (root / "lab").mkdir()
    names = []
    for i in range(5):
        shp = (12, 30 + 2 * (i % 2), 44)
        lab = np.zeros(shp, np.uint8)
        lab[3:9, 8 + i:20 + i, 10:30] = 1
        img = rs.randn(*shp) * 15 + 90 + 70.0 * lab
        nifti.write_nifti(str(root / "img" / ("c%d.nii.gz" % i)), img.astype(np.float32), (0.5, 0.5, 1.5))
        nifti.write_nifti(str(root / "lab" / ("c%d.nii.gz" % i)), lab, (0.5, 0.5, 1.5))
        names.append("c%d.nii.gz" % i)
    rows = ["img/%s,lab/%s" % (n, n) for n in names]
    (root / "train_1.csv").write_text("image,label\n" + "\n".join(rows[:4]) + "\n")
    (root / "train_2.csv").write_text("image,label\n" + "\n".join(rows[1:5]) + "\n")
    (root / "valid_1.csv").write_text("image,label\n" + rows[0] + "\n")
    (root / "valid_2.csv").write_text("image,label\n" + rows[4] + "\n")
    (root / "test.csv").write_text("image\n" + "\n".join("img/" + n for n in names[2:]) + "\n")
    (root / "pair.csv").write_text("ground_truth,segmentation\n" + "\n".join("%s,%s" % (n, n) for n in names[2:]) + "\n")
    text = SHIPPED_STYLE_CFG.format(root=str(root)).replace("gpus       = [0]", "gpus       = [0, 0]")
    cfg = root / "vs_like.cfg"
    cfg.write_text(text)
    _launch(tmp_path, _AGENT_WORKER, [str(cfg), "train"], timeout=1500)
    ck = root / "model_dual" / "vs_t1s_g"
    files = sorted(os.listdir(str(ck)))
    assert "vs_t1s_g_latest.txt" in files and "vs_t1s_g_best.txt" in files and any(f.endswith(".pt") for f in files)
    out = root / "results_dual" / "vs_t1s_g_test"
    for n in names[2:]:                                     # predictions of BOTH ranks' shares are there
        assert (out / n).exists(), n
    assert (out / "test_tumor_dice_all.csv").exists()
    # ---- FPL stage, sharded over the ranks
    srt = root / "uncertainty_sorted.npy"
    cfg.write_text(text.replace("fpl = False", "fpl = True\nfpl_uncertainty_sorted = %s" % str(srt)))
    _launch(tmp_path, _AGENT_WORKER, [str(cfg), "test"], timeout=1500)
    lst = np.load(str(srt), allow_pickle=True)
    assert sorted(str(e[1]) for e in lst) == sorted("img/" + n for n in names[2:])
    u = [float(e[0][0]) for e in lst]
    assert u == sorted(u)

"""SURVEY section 4 property of the data-parallel scheme, checked on CPU with the oracle over gloo (world size 2):

    reference (nn.DataParallel, agent_seg.py:692-698): the batch is cut into per-replica chunks, every replica normalises
    with ITS chunk's BatchNorm statistics, the logits are gathered, ONE loss over the full batch, replica gradients are added;
    fplx (one process per GPU): rank r forwards chunk r (fplx.dataset.BatchLoader hands it out), the loss sums are
    all-reduced so that every rank differentiates the full-batch loss, fplx.ddp.GradAllReducer adds the gradients.

The single-process emulation of the reference and the 2-rank run must give the same gradient (fp32 round-off)."""
import os
import subprocess
import sys

import numpy as np

_WORKER = r'''
import os, sys
root = sys.argv[3]
for p in (root, os.path.join(root, "fpl-plus_amd"), os.path.join(root, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import detdata
from make_golden_cfg import NETS
from oracle import torch_ref as R
from fplx.ddp import GradAllReducer
from fplx.dataset import BatchLoader
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(2)
p = dict(NETS["tiny"])
N = 4
xs = torch.from_numpy(detdata.normal("ddp.x", (N, 1, 16, 32, 32)))
ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)]))


class DS(object):
    def __len__(self): return N
    def __getitem__(self, i): return {"image": xs[i], "label_prob": ys[i]}


# rank r's chunk of the (only) global batch, as nn.DataParallel's scatter would cut it
batch = next(iter(BatchLoader(DS(), N, False, None, rank, world)))
per = N // world
assert torch.equal(batch["image"], xs[rank * per:(rank + 1) * per])
sd, prm = R.split_state(detdata.state_dict_3d(p))
logits = R.unet_forward(sd, p, batch["image"], 1, True)                   # per-rank BatchNorm statistics
# full-batch Dice from all-reduced sums (what fplx_seg_loss_sums / _from_sums do on the GPU)
prob = torch.softmax(logits, 1)
p2, y2 = R.to_2d(prob), R.to_2d(batch["label_prob"])
loc = torch.stack([y2.sum(0), p2.sum(0), (y2 * p2).sum(0)])
tot = loc.detach().clone()
dist.all_reduce(tot)
glob = loc + (tot - loc.detach())                                        # value = global sums, gradient = local part
loss = 1.0 - ((2.0 * glob[2] + 1e-5) / (glob[0] + glob[1] + 1e-5)).mean()
loss.backward()
names = [k for k in prm if prm[k].grad is not None]
sizes = [prm[k].numel() for k in names]
flat = torch.cat([prm[k].grad.reshape(-1) for k in names])
cut = int(sum(sizes) * 0.6)
red = GradAllReducer([(0, cut), (cut, flat.numel())], [], None)
red.begin(flat); red.ready(cut); red.finish([])
if rank == 0:
    np.savez(sys.argv[5], loss=float(loss.item()), grad=flat.numpy(), names=np.array(names), sizes=np.array(sizes))
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_two_rank_scheme_equals_dataparallel_semantics(tmp_path):
    import torch
    import detdata
    from make_golden_cfg import NETS
    from oracle import torch_ref as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    out = str(tmp_path / "r0.npz")
    port = str(23500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", root, port, out], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o
    got = np.load(out)
    # single process: the reference's DataParallel - replica forwards on the chunks, gather, ONE Dice loss, backward
    p = dict(NETS["tiny"])
    N = 4
    xs = torch.from_numpy(detdata.normal("ddp.x", (N, 1, 16, 32, 32)))
    ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)]))
    sd, prm = R.split_state(detdata.state_dict_3d(p))
    chunks = [R.unet_forward(sd, p, xs[i:i + 2], 1, True, ) for i in (0, 2)]
    loss = R.dice_loss(torch.cat(chunks, 0), ys)
    loss.backward()
    assert abs(float(loss.item()) - float(got["loss"])) < 1e-6
    off = 0
    for k, n in zip(got["names"], got["sizes"]):
        ref = prm[str(k)].grad.numpy().reshape(-1)
        g = got["grad"][off:off + int(n)]
        off += int(n)
        if str(k).endswith("bias") and "conv3d_" in str(k):
            continue                                               # cancellation noise around an exact zero
        assert np.abs(g - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-12) + 1e-9, k


_WORKER_ALL = r'''
import os, sys
root = sys.argv[3]
for p in (root, os.path.join(root, "fpl-plus_amd"), os.path.join(root, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import detdata
from make_golden_cfg import NETS
from oracle import torch_ref as R
from fplx.ddp import GradAllReducer
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[4]
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.set_num_threads(2)
p = dict(NETS["tiny"])
N = 4
per = N // world
sd, prm = R.split_state(detdata.state_dict_3d(p))
names = sorted(prm.keys(), key=lambda k: (2 if ".bns.1." in k else 1 if ".bns.0." in k else 0, k))
sizes = [prm[k].numel() for k in names]
offs = np.concatenate([[0], np.cumsum(sizes)])
shared_end = int(sum(n for k, n in zip(names, sizes) if ".bns." not in k))
d0_end = shared_end + int(sum(n for k, n in zip(names, sizes) if ".bns.0." in k))
total = int(offs[-1])
flats, losses = [], []
for dom in (0, 1):
    xs = torch.from_numpy(detdata.normal("ddp.all.x%d" % dom, (N, 1, 16, 32, 32)))
    ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)]))
    x, y = xs[rank * per:(rank + 1) * per], ys[rank * per:(rank + 1) * per]
    for v in prm.values():
        v.grad = None
    logits = R.unet_forward(sd, p, x, dom, True)                              # per-rank BatchNorm statistics
    prob = torch.softmax(logits, 1)
    p2, y2 = R.to_2d(prob), R.to_2d(y)
    loc = torch.stack([y2.sum(0), p2.sum(0), (y2 * p2).sum(0)])
    tot = loc.detach().clone()
    dist.all_reduce(tot)
    glob = loc + (tot - loc.detach())                                         # value = global sums, gradient = local part
    loss = 1.0 - ((2.0 * glob[2] + 1e-5) / (glob[0] + glob[1] + 1e-5)).mean()
    (0.5 * loss).backward()                                                   # loss = (l0 + l1) / 2, agent_seg.py:482
    losses.append(float(loss.item()))
    flats.append(torch.cat([(prm[k].grad if prm[k].grad is not None else torch.zeros_like(prm[k])).reshape(-1) for k in names]))
g0, g1 = flats
assert float(g0[d0_end:].abs().max()) == 0.0 and float(g1[shared_end:d0_end].abs().max()) == 0.0     # DSBN: the other set is untouched
# TrainStep.step_all's sequence: domain 0's BatchNorm segment as soon as its pass is over; the last domain's buffer folded
# and all-reduced bucket by bucket; tail + the last domain's segment at the flush
cut = int(shared_end * 0.6)
red = GradAllReducer([(0, cut), (cut, shared_end)], [(shared_end, d0_end), (d0_end, total)], None)
red.begin(g0)
red.reduce_domain(0)
red.set_acc(g1)
assert red.pending(cut) and not red.pending(cut - 1)
red.ready(cut)
assert red.launched == [(shared_end, d0_end), (0, cut)]
red.finish([1])
assert red.launched == [(shared_end, d0_end), (0, cut), (cut, shared_end), (d0_end, total)]
if rank == 0:
    np.savez(sys.argv[5], loss=np.array(losses), grad=g0.numpy(), names=np.array(names), sizes=np.array(sizes))
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_two_rank_training_all_exchange_folds_the_second_domain_bucket_by_bucket(tmp_path):
    """TrainStep.step_all's exchange (GradAllReducer with `acc`: gflat[s:e] += gacc[s:e], then all-reduce, per bucket while the last
    domain's backward runs; reduce_domain for the finished domain's BatchNorm segment) on the oracle over gloo, world 2 ==
    the single-process DataParallel emulation of a training_all iteration: replica forwards on the chunks, one loss per
    domain over the gathered batch, loss = (l0 + l1) / 2 (agent_seg.py:459-495, 692-698)."""
    import torch
    import detdata
    from make_golden_cfg import NETS
    from oracle import torch_ref as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w_all.py"
    script.write_text(_WORKER_ALL)
    out = str(tmp_path / "r0_all.npz")
    port = str(25500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", root, port, out], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o
    got = np.load(out)
    p = dict(NETS["tiny"])
    N = 4
    sd, prm = R.split_state(detdata.state_dict_3d(p))
    total, ls = None, []
    for dom in (0, 1):
        xs = torch.from_numpy(detdata.normal("ddp.all.x%d" % dom, (N, 1, 16, 32, 32)))
        ys = torch.from_numpy(detdata.ball_label((16, 32, 32), 6.0, n=N, offsets=[(0, 1, -2), (1, -3, 2), (-2, 0, 3), (2, 2, -1)]))
        chunks = [R.unet_forward(sd, p, xs[i:i + 2], dom, True) for i in (0, 2)]
        l = R.dice_loss(torch.cat(chunks, 0), ys)
        ls.append(float(l.item()))
        total = 0.5 * l if total is None else total + 0.5 * l
    total.backward()
    np.testing.assert_allclose(got["loss"], ls, rtol=0, atol=1e-6)
    off = 0
    for k, n in zip(got["names"], got["sizes"]):
        g = got["grad"][off:off + int(n)]
        off += int(n)
        ref = prm[str(k)].grad.numpy().reshape(-1)
        if str(k).endswith("bias") and "conv3d_" in str(k):
            continue                                               # cancellation noise around an exact zero
        assert np.abs(g - ref).max() <= 2e-5 * max(np.abs(ref).max(), 1e-12) + 1e-9, k

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE
(HiLab-git/FPL-plus mounted read-only at /root/reference) on deterministic inputs.

Run in the build container only:   python tests/golden/make_golden.py
The reference never travels to the GPU box; the emitted .npz/.json files do.

What is executed from the reference (file:line are reference paths):
  * pymic/net/net3d/unet2d5_dsbn.py:239-309   UNet2D5_dsbn (conv_dims all 3)
  * pymic/net_run_dsbn/dsbn.py:35-64          DomainSpecificBatchNorm3d
  * pymic/loss/seg/{dice,ce,combined}.py      DiceLoss / CrossEntropyLoss / DiceLoss_weight / CombinedLoss
  * pymic/net_run_dsbn/agent_seg.py:415-508   SegmentationAgent.training_all  (CPU, list loaders)
  * pymic/net_run_dsbn/agent_seg.py:834-964   SegmentationAgent.infer, FPL branch (torch.device patched to cpu)
  * pymic/net_run_dsbn/infer_func.py:188-222  Inferer.run (sliding window + TTA)
  * pymic/net_run_dsbn/get_optimizer.py       Adam + MultiStepLR
  * pymic/io/nifty_dataset.py:165-168         NiftyDataset.set_weight_
  * data/get_pixel_weight.py                  executed verbatim with an in-memory SimpleITK stand-in
  * pymic/util/parse_config.py:86-100         parse_config
  * dataset/weight/cyc121_vst1s-gan.npy + config_dual/data_vs/train_vs_t1s_wi+wp.csv  (known-answer DATA pair)
"""
import os
import sys
import io
import json
import copy
import csv
import contextlib
import tempfile
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import  # noqa: E402
import detdata  # noqa: E402

_ref_import.install()
torch.set_num_threads(8)

from pymic.net.net3d.unet2d5_dsbn import UNet2D5_dsbn  # noqa: E402
from pymic.loss.seg.dice import DiceLoss, DiceLoss_weight  # noqa: E402
from pymic.loss.seg.ce import CrossEntropyLoss  # noqa: E402
from pymic.loss.seg.combined import CombinedLoss  # noqa: E402
from pymic.loss.loss_dict_seg import SegLossDict  # noqa: E402
from pymic.net_run_dsbn.infer_func import Inferer  # noqa: E402
from pymic.net_run_dsbn import agent_seg as ref_agent_seg  # noqa: E402
from pymic.io.nifty_dataset import NiftyDataset  # noqa: E402
from pymic.util.parse_config import parse_config  # noqa: E402

from make_golden_cfg import NETS, SHAPES, label_for  # noqa: E402

GRAD_KEYS = ["out_conv.weight", "out_conv.bias", "block0.conv.conv3d_1.weight", "block0.conv.conv3d_1.bias",
             "block1.conv.conv3d_2.weight", "block4.conv.conv3d_2.weight", "up1.trans3d.weight",
             "up4.trans3d.weight", "up4.trans3d.bias", "up4.conv.conv3d_1.weight",
             "block0.conv.relu_1.weight", "block0.conv.relu_2.weight", "up2.conv.relu_1.weight"]


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def build_ref_net(name):
    params = copy.deepcopy(NETS[name])
    torch.manual_seed(1)
    net = UNet2D5_dsbn(params).float()
    sd = detdata.state_dict_3d(params)
    missing, unexpected = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    # every missing key must belong to a 2D twin / the bilinear 1x1 conv (dead when conv_dims are all 3)
    for k in missing:
        assert ("2d" in k) or (".conv3d." in k and k.startswith("up")), k
    return net, params


def gen_net(name):
    net, params = build_ref_net(name)
    x = torch.from_numpy(detdata.normal("x." + name, SHAPES[name]))
    y = torch.from_numpy(label_for(name))
    n = x.shape[0]
    out = {}
    # eval-mode logits first (running stats still the deterministic ones)
    net.eval()
    with torch.no_grad():
        for d in (0, 1):
            out["logits_eval_d%d" % d] = net(x, domain_label=d * torch.ones(n, dtype=torch.long)).numpy()
    # train-mode forward/backward per domain, fresh net each (running stats update once)
    for d in (0, 1):
        net, _ = build_ref_net(name)
        net.train()
        logits = net(x, domain_label=d * torch.ones(n, dtype=torch.long))
        out["logits_train_d%d" % d] = logits.detach().numpy()
        loss = DiceLoss()({"prediction": logits, "ground_truth": y})
        out["loss_dice_d%d" % d] = np.float32(loss.item())
        loss.backward()
        sd = net.state_dict()
        for bn in ("block0.conv.bn3d1", "block4.conv.bn3d2", "up4.conv.bn3d2"):
            for dd in (0, 1):
                for s in ("running_mean", "running_var", "num_batches_tracked"):
                    k = "%s.bns.%d.%s" % (bn, dd, s)
                    out["d%d.%s" % (d, k)] = sd[k].numpy().copy()
        named = dict(net.named_parameters())
        norms = {}
        for k, p in named.items():
            if p.grad is not None:
                norms[k] = float(p.grad.norm())
        out["gradnorm_keys_d%d" % d] = np.array(sorted(norms.keys()))
        out["gradnorm_vals_d%d" % d] = np.array([norms[k] for k in sorted(norms.keys())], np.float64)
        for k in GRAD_KEYS:
            g = named[k].grad.numpy()
            if g.size > 100000:  # keep fixtures small: strided sample of big tensors
                stride = g.size // 50000
                out["gradsub%d_d%d.%s" % (stride, d, k)] = g.reshape(-1)[::stride].copy()
            else:
                out["grad_d%d.%s" % (d, k)] = g.copy()
        for bn in ("block0.conv.bn3d1", "block2.conv.bn3d2", "up4.conv.bn3d2"):
            for s in ("weight", "bias"):
                k = "%s.bns.%d.%s" % (bn, d, s)
                out["grad_d%d.%s" % (d, k)] = named[k].grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "net_%s.npz" % name), **out)
    print("net", name, "ok:", {k: v.shape for k, v in out.items() if k.startswith("logits")})


def gen_losses():
    shape = (2, 2, 8, 16, 16)
    logits = detdata.normal("loss.logits", shape, scale=2.0)
    lab = detdata.ball_label(shape[2:], 3.0, n=2, offsets=[(0, 0, 0), (1, 2, -1)])
    pw = (detdata.uniform("loss.pw", (2, 1) + shape[2:]) > 0.3).astype(np.float32)
    iw = np.array([0.37, 0.93], np.float32)
    pw_iw = pw * iw[:, None, None, None, None]
    out = {"logits": logits, "label": lab, "pixel_weight": pw_iw, "image_weight": iw}

    def run(tag, loss_mod, d):
        lg = torch.from_numpy(logits).clone().requires_grad_(True)
        dd = {"prediction": lg, "ground_truth": torch.from_numpy(lab)}
        dd.update({k: torch.from_numpy(v) for k, v in d.items()})
        with quiet():
            val = loss_mod(dd)
        val.backward()
        out[tag + ".loss"] = np.float32(val.item())
        out[tag + ".dlogits"] = lg.grad.numpy().copy()

    run("dice", DiceLoss(), {})
    run("dice_pw", DiceLoss(), {"pixel_weight": pw_iw})
    run("ce", CrossEntropyLoss(), {})
    run("ce_pw", CrossEntropyLoss(), {"pixel_weight": pw_iw})
    run("dice_weight", DiceLoss_weight(), {"pixel_weight": torch.from_numpy(pw).numpy(), "image_weight": iw})
    comb = CombinedLoss({"loss_type": ["DiceLoss", "CrossEntropyLoss"], "loss_weight": [0.6, 0.4]}, SegLossDict)
    run("combined", comb, {})
    run("combined_pw", comb, {"pixel_weight": pw_iw})
    # entropy regulariser, agent_seg.py:352-354 (divisor = shape[0]*[2]*[3]*[4] under the mislabelled unpack)
    lg = torch.from_numpy(logits).clone().requires_grad_(True)
    D, B, C, W, H = lg.shape
    ent = -(lg.softmax(1) * torch.log2(lg.softmax(1) + 1e-10)).sum() / (W * H * C * D)
    ent.backward()
    out["entropy.loss"] = np.float32(ent.item())
    out["entropy.dlogits"] = lg.grad.numpy().copy()
    # 3-class variant for Dice / CE
    shape3 = (1, 3, 4, 8, 8)
    lg3 = detdata.normal("loss.logits3", shape3, scale=1.5)
    idx = (detdata.uniform("loss.lab3", (1,) + shape3[2:]) * 3).astype(np.int64).clip(0, 2)
    lab3 = np.eye(3, dtype=np.float32)[idx].transpose(0, 4, 1, 2, 3).copy()
    out["logits3"], out["label3"] = lg3, lab3
    for tag, mod in (("dice3", DiceLoss()), ("ce3", CrossEntropyLoss())):
        t = torch.from_numpy(lg3).clone().requires_grad_(True)
        v = mod({"prediction": t, "ground_truth": torch.from_numpy(lab3)})
        v.backward()
        out[tag + ".loss"] = np.float32(v.item())
        out[tag + ".dlogits"] = t.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses ok", {k: float(v) for k, v in out.items() if k.endswith(".loss")})


def make_agent(net_name, training_extra=None, testing=None):
    cfg = {
        "dataset": {"tensor_type": "float", "task_type": "seg", "root_dir": "/", "train_batch_size": 2},
        "network": copy.deepcopy(NETS[net_name]),
        "training": {"dis": False, "train_fpl_uda": True, "loss_type": "DiceLoss", "optimizer": "Adam",
                     "learning_rate": 1e-3, "momentum": 0.9, "weight_decay": 1e-5,
                     "lr_scheduler": "MultiStepLR", "lr_gamma": 0.5, "lr_milestones": [2, 4],
                     "iter_valid": 1, "ckpt_save_dir": "/tmp/fplx_golden_ckpt", "gpus": [0]},
        "testing": testing or {},
    }
    if training_extra:
        cfg["training"].update(training_extra)
    with quiet():
        agent = ref_agent_seg.SegmentationAgent(cfg, "train")
        net, _ = build_ref_net(net_name)
        agent.set_network(net)
        agent.create_network()
    agent.device = torch.device("cpu")
    agent.checkpoint = None
    agent.fpl_uda = True
    return agent, cfg


def gen_train_step():
    name = "tiny"
    out = {}
    for variant, extra, with_w in (("dice", {}, False), ("dice_pw", {}, True),
                                   ("combined", {"loss_type": ["DiceLoss", "CrossEntropyLoss"],
                                                 "loss_weight": [0.5, 0.5]}, True)):
        agent, cfg = make_agent(name, extra)
        with quiet():
            agent.create_optimizer(agent.get_parameters_to_update())
            agent.create_loss_calculator()
        n, _, D, H, W = SHAPES[name]
        batches = []
        for dom in (0, 1):
            x = detdata.normal("ts.x.d%d" % dom, SHAPES[name])
            lab = detdata.ball_label((D, H, W), 5.0, n=n, offsets=[(dom, 1, -2), (1, -3, 2 + dom)])
            b = {"image": torch.from_numpy(x), "label_prob": torch.from_numpy(lab)}
            if with_w and dom == 1:
                pw = (detdata.uniform("ts.pw", (n, 1, D, H, W)) > 0.25).astype(np.float32)
                iw = np.array([0.8, 0.45], np.float32)
                b["pixel_weight"] = torch.from_numpy(pw * iw[:, None, None, None, None])
                b["image_weight"] = torch.from_numpy(iw)
            batches.append(b)
        agent.train_loader_1 = [batches[0]]
        agent.train_loader_2 = [batches[1]]
        watch = ["out_conv.weight", "block0.conv.conv3d_1.weight", "block4.conv.conv3d_2.bias",
                 "up1.trans3d.weight", "block0.conv.bn3d1.bns.0.weight", "block0.conv.bn3d1.bns.1.bias",
                 "block0.conv.relu_1.weight", "up4.conv.bn3d2.bns.1.running_var"]
        lrs = []
        for step in range(1, 6):
            with quiet():
                sc = agent.training_all()
            out["%s.step%d.loss" % (variant, step)] = np.float64(sc["loss"])
            out["%s.step%d.avg_dice" % (variant, step)] = np.float64(sc["avg_dice"])
            out["%s.step%d.class_dice" % (variant, step)] = np.asarray(sc["class_dice"], np.float64)
            lrs.append(agent.optimizer.param_groups[0]["lr"])
            if step in (1, 3):
                sd = agent.net.state_dict()
                for k in watch:
                    out["%s.step%d.%s" % (variant, step, k)] = sd[k].numpy().copy()
        out["%s.lrs" % variant] = np.array(lrs, np.float64)
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **out)
    print("train_step ok", {k: v for k, v in out.items() if k.endswith("loss")})


def gen_inferer():
    net, _ = build_ref_net("tiny")
    net.eval()
    x = torch.from_numpy(detdata.normal("inf.x", (1, 1, 40, 72, 72)))
    dl = torch.ones(1, dtype=torch.long)
    out = {}
    cfgs = {"sw_nooverlap_tta0": dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32],
                                     sliding_window_stride=[16, 32, 32], tta_mode=0, class_num=2),
            "sw_overlap_tta1": dict(sliding_window_enable=True, sliding_window_size=[16, 32, 32],
                                    sliding_window_stride=[8, 24, 16], tta_mode=1, class_num=2),
            "full_tta1": dict(sliding_window_enable=False, tta_mode=1, class_num=2)}
    with torch.no_grad():
        for tag, c in cfgs.items():
            xx = x if tag != "full_tta1" else x[:, :, :32, :64, :64]
            out[tag] = Inferer(c).run(net, xx, dl).numpy()
    np.savez_compressed(os.path.join(HERE, "inferer.npz"), **out)
    print("inferer ok", {k: v.shape for k, v in out.items()})


def gen_fpl_filter():
    """Run the reference's infer() FPL branch on CPU and record what went in and out."""
    name = "tiny"
    netp = copy.deepcopy(NETS[name])
    netp["dropout"] = [0, 0, 0.3, 0.4, 0.5]
    tmp = tempfile.mkdtemp(prefix="fplx_golden_")
    testing = {"domian_label": 1, "gpus": [0], "fpl": True, "ae": None, "ckpt_mode": 2,
               "ckpt_name": os.path.join(tmp, "ck.pt"), "evaluation_mode": True,
               "test_time_dropout": False, "tta_mode": 1, "sliding_window_enable": True,
               "sliding_window_size": [16, 32, 32], "sliding_window_stride": [16, 32, 32],
               "fpl_uncertainty_sorted": os.path.join(tmp, "unc.npy"), "output_dir": tmp}
    out = {}
    agent, cfg = make_agent(name, testing=testing)
    cfg["network"].update(netp)
    torch.manual_seed(1)
    with quiet():
        net = UNet2D5_dsbn(copy.deepcopy(netp)).float()
    sd = detdata.state_dict_3d(netp)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    agent.net = net
    torch.save({"model_state_dict": net.state_dict()}, testing["ckpt_name"])
    agent.transform_list = []
    vols = []
    for i in range(3):
        x = detdata.normal("fpl.x%d" % i, (1, 1, 16, 64, 32), scale=1.0 + i)
        vols.append({"image": torch.from_numpy(x), "names": ["./dataset/hrT2_train/img/vol%d.nii.gz" % i]})
    agent.test_loader = vols
    recorded = []

    class RecInferer(Inferer):
        def run(self, model, image, domain_label):
            r = Inferer.run(self, model, image, domain_label)
            recorded.append(r.cpu().numpy().copy())
            return r

    icfg = dict(testing)
    icfg["class_num"] = 2
    agent.set_inferer(RecInferer(icfg))
    real_device = torch.device

    class _Meta(type):
        def __instancecheck__(cls, inst):
            return isinstance(inst, real_device)

        def __call__(cls, *a, **k):
            return real_device("cpu")

    class _CpuDevice(metaclass=_Meta):
        pass

    torch.manual_seed(1234)
    torch.device = _CpuDevice
    # numpy>=2 refuses np.save of the ragged [( [u], name ), ...] list the reference builds
    # (agent_seg.py:957-960); capture the list instead of writing it.
    saved = {}
    real_save = np.save
    np.save = lambda path, arr: saved.__setitem__(path, arr)
    try:
        with quiet():
            agent.infer()
    finally:
        torch.device = real_device
        np.save = real_save
    res = saved[testing["fpl_uncertainty_sorted"]]
    assert len(recorded) == 18
    for i in range(3):
        out["vol%d.logits" % i] = np.concatenate(recorded[6 * i:6 * i + 6], axis=0)  # [6,2,D,H,W]
    out["sorted_uncertainty"] = np.array([float(r[0][0]) for r in res], np.float64)
    out["sorted_names"] = np.array([str(r[1]) for r in res])
    # a synthetic confident-everywhere case -> boundary < 50 -> uncer_one = 1 (agent_seg.py:926-927)
    np.savez_compressed(os.path.join(HERE, "fpl_filter.npz"), **out)
    print("fpl filter ok", list(zip(out["sorted_names"], out["sorted_uncertainty"])))


def gen_pixel_weight():
    """Execute data/get_pixel_weight.py verbatim with an in-memory SimpleITK stand-in."""
    a = (detdata.uniform("pw.a", (12, 40, 36)) > 0.6).astype(np.uint8)
    b = a.copy()
    flip = detdata.uniform("pw.flip", a.shape) > 0.9
    b[flip] = 1 - b[flip]
    store = {"./results_dual/vs_t1s_g_i-train_hrT2/v0.nii.gz": a,
             "./results_dual/vs_t1s_g_i-train_hrT2-ceT1_cyc/v0.nii.gz": b}
    written = {}
    import types
    fake = types.ModuleType("SimpleITK")
    fake.ReadImage = lambda p: store[p]
    fake.GetArrayFromImage = lambda im: np.array(im)
    fake.GetImageFromArray = lambda arr: arr
    fake.WriteImage = lambda im, p: written.__setitem__(p, np.array(im))
    real_listdir = os.listdir
    os.listdir = lambda p: ["v0.nii.gz"] if "results_dual" in p else real_listdir(p)
    old = sys.modules.get("SimpleITK")
    sys.modules["SimpleITK"] = fake
    try:
        src = open(os.path.join(_ref_import.REF_ROOT, "data/get_pixel_weight.py")).read()
        exec(compile(src, "get_pixel_weight.py", "exec"), {"__name__": "__main__"})
    finally:
        os.listdir = real_listdir
        sys.modules["SimpleITK"] = old
    (wname, w), = written.items()
    out = {"mask_a": a, "mask_b": b, "weight": w}
    # NiftyDataset.set_weight_ (nifty_dataset.py:165-168) on the float32 view the dataset loads (line 157)
    for iw in (0.37, 1.0):
        w32 = np.asarray(w, np.float32).copy()
        out["set_weight_%s" % iw] = NiftyDataset.set_weight_(None, np.float32(iw), w32)
        out["set_weight_py_%s" % iw] = NiftyDataset.set_weight_(None, iw, np.asarray(w, np.float32).copy())
    np.savez_compressed(os.path.join(HERE, "pixel_weight.npz"), **out)
    print("pixel weight ok", w.dtype, np.unique(w), wname)


def gen_image_weight_kat():
    src = np.load(os.path.join(_ref_import.REF_ROOT, "dataset/weight/cyc121_vst1s-gan.npy"), allow_pickle=True)
    rows = [[float(r[0][0]), str(r[1])] for r in src]
    exp = []
    with open(os.path.join(_ref_import.REF_ROOT, "config_dual/data_vs/train_vs_t1s_wi+wp.csv")) as f:
        rd = csv.reader(f)
        header = next(rd)
        for r in rd:
            if r:  # the published csv has blank lines between records (\r\r\n line ends)
                exp.append(r)
    json.dump({"input": rows, "csv_header": header, "csv_rows": exp},
              open(os.path.join(HERE, "image_weight_kat.json"), "w"), indent=0)
    print("image weight KAT ok", len(rows), len(exp))


def gen_parse_config():
    cfg_path = os.path.join(HERE, "sample_vs.cfg")
    with quiet():
        d = parse_config(cfg_path)
    json.dump(d, open(os.path.join(HERE, "sample_vs.cfg.json"), "w"), indent=1, sort_keys=True)
    print("parse_config ok", list(d.keys()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["net", "losses", "train", "inferer", "fpl", "pw", "kat", "cfg"]
    if "net" in which:
        for n in ("tiny", "c4", "cfg1"):
            gen_net(n)
    if "losses" in which:
        gen_losses()
    if "train" in which:
        gen_train_step()
    if "inferer" in which:
        gen_inferer()
    if "fpl" in which:
        gen_fpl_filter()
    if "pw" in which:
        gen_pixel_weight()
    if "kat" in which:
        gen_image_weight_kat()
    if "cfg" in which:
        gen_parse_config()

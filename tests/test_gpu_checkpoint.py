"""-m gpu: checkpoint interop with the reference (fplx/checkpoint.py).
tests/golden/ref_ckpt_3.pt is a checkpoint dictionary the REFERENCE wrote after 3 training_all iterations
(tests/golden/make_golden_ckpt.py); ref_ckpt.npz holds the numbers of its iterations 4 and 5."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NET = dict(in_chns=1, feature_chns=[4, 4, 8, 8, 8], dropout=[0, 0, 0, 0, 0], conv_dims=[3, 3, 3, 3, 3], class_num=2,
           bilinear=False, num_domains=2, net_type="UNet2D5_dsbn")


def _cfg(tmp_path, **training):
    tr = {"dis": False, "train_fpl_uda": True, "loss_type": "DiceLoss", "optimizer": "Adam", "learning_rate": 1e-3,
          "momentum": 0.9, "weight_decay": 1e-5, "lr_scheduler": "MultiStepLR", "lr_gamma": 0.5, "lr_milestones": [2, 4],
          "iter_valid": 1, "gpus": [0], "ckpt_save_dir": str(tmp_path / "model" / "vs_t1s_g")}
    tr.update(training)
    return {"dataset": {"tensor_type": "float"}, "network": dict(NET), "training": tr, "testing": {"gpus": [0]}}


def _batches(g):
    return [{"image": torch.from_numpy(g["x%d" % d]), "label_prob": torch.from_numpy(g["lab%d" % d])} for d in (0, 1)]


def _check(sd, g, step):
    pre = "step%d." % step
    for k in g.files:
        if not k.startswith(pre) or k[len(pre):] not in sd:
            continue
        kk = k[len(pre):]
        if kk.endswith("bias") and "conv3d" in kk:
            continue                      # conv bias under BN: gradient is fp noise (see test_gpu_loss_filter_parity)
        ref, got = g[k], sd[kk].cpu().numpy()
        if kk.endswith("num_batches_tracked"):
            assert int(got) == int(ref), kk
            continue
        diff = np.abs(got - ref)
        assert diff.max() <= 1e-3 * step + 1e-6, (k, diff.max())


def test_resume_from_reference_checkpoint_lands_on_reference_numbers(golden_dir, tmp_path):
    import fplx
    g = np.load(os.path.join(golden_dir, "ref_ckpt.npz"))
    ck = torch.load(os.path.join(golden_dir, "ref_ckpt_3.pt"), map_location="cuda:0", weights_only=False)
    assert ck["iteration"] == 3 and len(ck["model_state_dict"]) == 484
    agent = fplx.SegmentationAgent(_cfg(tmp_path), "train")
    agent.create_network()
    agent.checkpoint = ck
    agent.net.load_state_dict(ck["model_state_dict"])                 # 484 keys, dead 2D twins kept aside
    _check(agent.net.state_dict(), g, 3)
    agent.create_optimizer()                                           # Adam moments + steps, MultiStepLR at last_iter = 2
    agent.create_loss_calculator()
    assert agent.optimizer.seg_steps == [3, 3, 3]
    assert agent.optimizer.param_groups[0]["lr"] == float(g["step3.lr"])
    b = _batches(g)
    agent.set_loaders([b[0]], [b[1]])
    for step in (4, 5):
        sc = agent.training_all()
        assert abs(sc["loss"] - float(g["step%d.loss" % step])) < 5e-5, (step, sc["loss"])
        assert agent.optimizer.param_groups[0]["lr"] == pytest.approx(float(g["step%d.lr" % step]), rel=1e-12)
    _check(agent.net.state_dict(), g, 5)

    # what we save is what the reference would have saved: key order, dead twins verbatim, Adam layout
    from fplx import checkpoint as C
    keys = json.load(open(os.path.join(golden_dir, "ref_state_keys.json")))
    msd = C.reference_model_state_dict(agent.net)
    assert list(msd.keys()) == keys["state_dict"]
    for k, v in msd.items():
        assert list(v.shape) == keys["shapes"][k], k
    dead = [k for k in msd if k not in agent.net.state_dict()]
    assert len(dead) == 240 and all(torch.equal(msd[k], ck["model_state_dict"][k].cpu()) for k in dead)
    osd = agent.optimizer.state_dict()
    assert sorted(osd["state"].keys()) == sorted(ck["optimizer_state_dict"]["state"].keys())      # the 136 live params
    assert osd["param_groups"][0]["params"] == list(range(268))
    assert set(ck["optimizer_state_dict"]["param_groups"][0].keys()) <= set(osd["param_groups"][0].keys())
    assert all(float(s["step"]) == 5.0 for s in osd["state"].values())
    # torch itself accepts it: a torch.optim.Adam over 268 dummy parameters of the reference's shapes
    dummies = [torch.nn.Parameter(torch.zeros(keys["shapes"][k])) for k in keys["named_parameters"]]
    ref_opt = torch.optim.Adam(dummies, lr=1e-3, weight_decay=1e-5)
    ref_opt.load_state_dict(osd)
    i = keys["named_parameters"].index("out_conv.weight")
    assert torch.equal(ref_opt.state[dummies[i]]["exp_avg"].cpu(), osd["state"][i]["exp_avg"].cpu())


def test_state_round_trip_is_bit_exact_and_fresh_net_emits_all_keys(golden_dir, tmp_path):
    import fplx
    from fplx import checkpoint as C
    g = np.load(os.path.join(golden_dir, "ref_ckpt.npz"))
    torch.manual_seed(5)
    a = fplx.SegmentationAgent(_cfg(tmp_path), "train")
    a.create_network()
    a.create_optimizer()
    a.create_loss_calculator()
    b = _batches(g)
    a.set_loaders([b[0]], [b[1]])
    a.training_all()
    # domain 1 only for one extra step -> the three segments carry different step counts
    x, y = b[1]["image"].cuda(), b[1]["label_prob"].cuda()
    a.optimizer.zero_grad()
    out = a.net(x, domain_label=torch.ones(2, dtype=torch.long))
    a.get_loss_value(b[1], out, y).backward()
    a.optimizer.step()
    assert a.optimizer.seg_steps == [2, 1, 2]
    msd = C.reference_model_state_dict(a.net)
    assert len(msd) == 484                                             # fresh net: neutral dead twins of the right shape
    keys = json.load(open(os.path.join(golden_dir, "ref_state_keys.json")))
    assert all(list(v.shape) == keys["shapes"][k] for k, v in msd.items())
    C.save_checkpoint(a.config, 2, 0.5, msd, a.optimizer, "latest")
    ck = torch.load(C.checkpoint_file(a.config, 2), map_location="cuda:0", weights_only=False)
    steps = sorted(set(float(s["step"]) for s in ck["optimizer_state_dict"]["state"].values()))
    assert steps == [1.0, 2.0]
    c = fplx.SegmentationAgent(_cfg(tmp_path), "train")
    c.create_network()
    c.checkpoint = ck
    c.net.load_state_dict(ck["model_state_dict"])
    c.create_optimizer()
    assert c.optimizer.seg_steps == [2, 1, 2]
    assert torch.equal(c.optimizer.exp_avg, a.optimizer.exp_avg) and torch.equal(c.optimizer.exp_avg_sq, a.optimizer.exp_avg_sq)
    c.net._ensure_flat()
    assert torch.equal(c.net.flat_params, a.net.flat_params)
    bad = {"state": {}, "param_groups": [{"params": list(range(10)), "lr": 1e-3}]}
    with pytest.raises(ValueError):
        c.optimizer.load_state_dict(bad)


def _write_cases(root, n):
    from fplx import nifti
    rs = np.random.RandomState(4)
    rows = []
    os.makedirs(os.path.join(root, "img"), exist_ok=True)
    for i in range(n):
        shp = (16, 32, 32)
        lab = np.zeros(shp, np.uint8)
        lab[5:11, 8 + i:20 + i, 10:24] = 1
        img = rs.randn(*shp) * 20 + 100 + 60.0 * lab
        nifti.write_nifti(os.path.join(root, "img", "i%d.nii.gz" % i), img.astype(np.float32))
        nifti.write_nifti(os.path.join(root, "img", "l%d.nii.gz" % i), lab)
        rows.append("img/i%d.nii.gz,img/l%d.nii.gz" % (i, i))
    return rows


def test_train_valid_writes_the_reference_checkpoint_protocol_and_resumes(tmp_path):
    """run() = create_dataset + create_network + train_valid: files `<prefix>_<it>.pt`, `_latest.txt`, `_best.txt`
    (agent_seg.py:786-830), resume from iter_start, get_checkpoint_name for inference (agent_abstract.py:136-153)."""
    import fplx
    root = str(tmp_path / "data")
    rows = _write_cases(root, 4)
    for name, sel in (("tr1", rows[:2]), ("tr2", rows[2:]), ("va1", rows[:1]), ("va2", rows[3:])):
        (tmp_path / (name + ".csv")).write_text("image,label\n" + "\n".join(sel) + "\n")
    cfg = _cfg(tmp_path, iter_start=0, iter_max=4, iter_valid=2, iter_save=2, dual=True, val_t2=True, random_seed=3)
    cfg["dataset"].update({
        "root_dir": root, "1_train_csv": str(tmp_path / "tr1.csv"), "2_train_csv": str(tmp_path / "tr2.csv"),
        "1_valid_csv": str(tmp_path / "va1.csv"), "2_valid_csv": str(tmp_path / "va2.csv"), "train_batch_size": 2,
        "train_transform": ["NormalizeWithMeanStd", "RandomFlip", "LabelToProbability"],
        "valid_transform": ["NormalizeWithMeanStd", "LabelToProbability"],
        "normalizewithmeanstd_channels": [0], "randomflip_flip_depth": False, "randomflip_flip_height": True,
        "randomflip_flip_width": True, "labeltoprobability_class_num": 2})
    cfg["training"]["learning_rate"] = 1e-2
    cfg["training"]["lr_milestones"] = [100]
    import random
    random.seed(1)
    agent = fplx.SegmentationAgent(cfg, "train")
    hist = agent.run()
    d = tmp_path / "model" / "vs_t1s_g"
    files = sorted(os.listdir(str(d)))
    assert "vs_t1s_g_latest.txt" in files and "vs_t1s_g_best.txt" in files
    assert (d / "vs_t1s_g_latest.txt").read_text() == "4"
    best_it = int((d / "vs_t1s_g_best.txt").read_text())
    assert best_it in (2, 4) and "vs_t1s_g_%d.pt" % best_it in files and "vs_t1s_g_2.pt" in files
    assert [h[0] for h in hist] == [2, 4] and all(np.isfinite(h[2]["loss"]) for h in hist)
    ck = torch.load(str(d / "vs_t1s_g_4.pt"), map_location="cpu", weights_only=False)
    assert sorted(ck.keys()) == ["iteration", "model_state_dict", "optimizer_state_dict", "valid_pred"]
    assert len(ck["model_state_dict"]) == 484 and ck["iteration"] in (4, best_it)
    # resume from iteration 2 for two more iterations
    cfg2 = _cfg(tmp_path, iter_start=2, iter_max=4, iter_valid=2, iter_save=2, dual=True, val_t2=True, random_seed=3)
    cfg2["dataset"] = dict(cfg["dataset"])
    cfg2["training"]["learning_rate"] = 1e-2
    cfg2["training"]["lr_milestones"] = [100]
    a2 = fplx.SegmentationAgent(cfg2, "train")
    h2 = a2.run()
    # if iteration 2 was the best one its file was re-written at the end with the optimizer of iteration 4, as the
    # reference does (agent_seg.py:806-826 saves the CURRENT optimizer next to the best weights)
    want = 6 if best_it == 2 else 4
    assert [h[0] for h in h2] == [4] and a2.optimizer.seg_steps == [want] * 3
    # inference picks the file through the txt protocol
    cfg2["testing"].update({"ckpt_mode": 1})
    from fplx import checkpoint as C
    assert C.get_checkpoint_name(cfg2) == str(d / ("vs_t1s_g_%s.pt" % (d / "vs_t1s_g_best.txt").read_text()))
    cfg2["testing"].update({"ckpt_mode": 2, "ckpt_name": "some/file.pt"})
    assert C.get_checkpoint_name(cfg2) == "some/file.pt"

"""Checkpoint interop with the reference.

The reference saves {'iteration', 'valid_pred', 'model_state_dict', 'optimizer_state_dict'} with torch.save
(PyMIC/pymic/net_run_dsbn/agent_seg.py:786-799, 806-826) next to `<prefix>_latest.txt` / `<prefix>_best.txt`, and picks a
file for inference with get_checkpoint_name (agent_abstract.py:136-153).  Its UNet2D5_dsbn carries a dead 2D twin of
every layer (unet2d5_dsbn.py:48-64, 129-150), so a model_state_dict has 484 keys and Adam's param list 268 entries, in
module-definition order; this build keeps only the live 3D members in a flat, gradient-ordered buffer.  This module
translates both ways:
  * reference_state_keys / reference_param_names: the reference's key order, by rule (checked against a key list dumped
    from the reference, tests/golden/ref_state_keys.json);
  * reference_model_state_dict(net): all 484 keys - the dead twins are re-emitted verbatim if they came in with a loaded
    checkpoint, else filled with neutral values of the right shape - so the reference's strict load_state_dict accepts it;
  * optimizer_to_reference / optimizer_from_reference: torch.optim.Adam's {'state': {index: {step, exp_avg, exp_avg_sq}},
    'param_groups': [...]} with indices in the reference's parameter order.  torch gives a parameter state only once it has
    seen a gradient and counts steps per parameter; here steps are counted per flat segment (shared | BN of domain d), which
    is the same thing because a domain's BN parameters always step together (dsbn.py:56).
"""
import collections
import os

import torch

_BUFFERS = ("running_mean", "running_var", "num_batches_tracked")


def _conv_block_keys(pre, num_domains):
    ks = []
    for dim in ("2d", "3d"):
        for i in (1, 2):
            ks += ["%sconv%s_%d.weight" % (pre, dim, i), "%sconv%s_%d.bias" % (pre, dim, i)]
    for dim in ("2d", "3d"):
        for i in (1, 2):
            for d in range(num_domains):
                for t in ("weight", "bias") + _BUFFERS:
                    ks.append("%sbn%s%d.bns.%d.%s" % (pre, dim, i, d, t))
    ks += [pre + "relu_1.weight", pre + "relu_2.weight"]
    return ks


def reference_state_keys(num_domains=2):
    """ordered state_dict keys of the reference UNet2D5_dsbn (bilinear False): block0-4, up1-4, out_conv"""
    ks = []
    for i in range(5):
        ks += _conv_block_keys("block%d.conv." % i, num_domains)
    for j in range(1, 5):
        for m in ("conv2d", "conv3d", "trans2d", "trans3d"):
            ks += ["up%d.%s.weight" % (j, m), "up%d.%s.bias" % (j, m)]
        ks += _conv_block_keys("up%d.conv." % j, num_domains)
    return ks + ["out_conv.weight", "out_conv.bias"]


def reference_param_names(num_domains=2):
    """net.parameters() order of the reference = state_dict order without the buffers"""
    return [k for k in reference_state_keys(num_domains) if k.rsplit(".", 1)[1] not in _BUFFERS]


def _dead_default(key, live):
    """neutral tensor for a key that is dead in this configuration, shaped after its live sibling of the other
    dimensionality (a dim-3 level carries dead 2D twins and the other way round; the bilinear 1x1 convs are always dead)"""
    if ".conv2d_" in key or ".conv3d_" in key:                 # ConvBlockND convolutions
        to3 = ".conv2d_" in key                                # the live twin is the 3D one
        w = live[key.replace("conv2d_", "conv3d_") if to3 else key.replace("conv3d_", "conv2d_")]
        if key.endswith("bias"):
            return torch.zeros(w.shape, dtype=w.dtype)
        return torch.zeros(tuple(w.shape[:4]) if to3 else tuple(w.shape) + (3,), dtype=w.dtype)
    if ".bn2d" in key or ".bn3d" in key:
        w = live[key.replace("bn2d", "bn3d") if ".bn2d" in key else key.replace("bn3d", "bn2d")]
        if key.endswith("weight") or key.endswith("running_var"):
            return torch.ones_like(w, device="cpu")
        return torch.zeros_like(w, device="cpu")
    up = key.split(".")[0]
    t = live.get(up + ".trans3d.weight")
    if t is None:
        t = live.get(up + ".trans2d.weight")
    if t is not None:
        ci, co = t.shape[0], t.shape[1]
    else:                                                      # bilinear = True: the live member is the kernel-1 convolution
        c1 = live.get(up + ".conv3d.weight")
        if c1 is None:
            c1 = live[up + ".conv2d.weight"]
        co, ci = c1.shape[0], c1.shape[1]
    if key.endswith("bias"):
        return torch.zeros(co)
    if ".trans2d." in key:
        return torch.zeros(ci, co, 2, 2)
    if ".trans3d." in key:
        return torch.zeros(ci, co, 2, 2, 2)
    if ".conv2d." in key:
        return torch.zeros(co, ci, 1, 1)
    return torch.zeros(co, ci, 1, 1, 1)                        # up.conv3d: the bilinear branch's 1x1x1 conv


def reference_model_state_dict(net):
    live = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    dead = getattr(net, "_dead_state", None) or {}
    out = collections.OrderedDict()
    for k in reference_state_keys(net.num_domains):
        if k in live:
            out[k] = live[k]
        elif k in dead:
            out[k] = dead[k]
        else:
            out[k] = _dead_default(k, live)
    assert len(out) == len(live) + sum(1 for k in out if k not in live)
    return out


def optimizer_to_reference(opt):
    """FusedAdam -> torch.optim.Adam state_dict over the reference's 268-entry parameter list"""
    net = opt.net
    names = reference_param_names(net.num_domains)
    index = {k: i for i, k in enumerate(names)}
    state = {}
    for si, (start, end) in enumerate(opt.seg_ranges):
        if opt.seg_steps[si] == 0:
            continue                                           # never stepped: torch holds no state for these
        for k in net._order:
            o, n, shp = net._layout[k]
            if start <= o < end:
                state[index[k]] = {"step": torch.tensor(float(opt.seg_steps[si])),
                                   "exp_avg": opt.exp_avg[o:o + n].view(shp).clone(),
                                   "exp_avg_sq": opt.exp_avg_sq[o:o + n].view(shp).clone()}
    g = {k: v for k, v in opt.param_groups[0].items() if k != "params"}
    for k, v in (("amsgrad", False), ("maximize", False), ("foreach", None), ("capturable", False),
                 ("differentiable", False), ("fused", None), ("decoupled_weight_decay", False)):
        g.setdefault(k, v)
    g["params"] = list(range(len(names)))
    return {"state": dict(sorted(state.items())), "param_groups": [g]}


def optimizer_from_reference(opt, sd):
    net = opt.net
    names = reference_param_names(net.num_domains)
    if len(sd["param_groups"]) != 1 or len(sd["param_groups"][0]["params"]) != len(names):
        raise ValueError("fplx: optimizer state has {0:} parameters, the reference network has {1:}".format(
            sum(len(g["params"]) for g in sd["param_groups"]), len(names)))
    ids = sd["param_groups"][0]["params"]
    by_name = {names[pos]: sd["state"][pid] for pos, pid in enumerate(ids) if pid in sd["state"]}
    stray = [k for k in by_name if k not in net._layout]
    if stray:
        raise ValueError("fplx: optimizer state for parameters this build does not train: {0:}".format(stray[:4]))
    opt.exp_avg.zero_()
    opt.exp_avg_sq.zero_()
    steps = []
    for si, (start, end) in enumerate(opt.seg_ranges):
        seen = set()
        for k in net._order:
            o, n, shp = net._layout[k]
            if not (start <= o < end):
                continue
            st = by_name.get(k)
            seen.add(None if st is None else int(round(float(st["step"]))))
            if st is not None:
                opt.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
                opt.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
        if len(seen) != 1:
            raise ValueError("fplx: parameters of one segment carry different Adam step counts {0:}".format(seen))
        s = seen.pop()
        steps.append(0 if s is None else s)
    opt.seg_steps = steps
    for k, v in sd["param_groups"][0].items():
        if k != "params":
            opt.param_groups[0][k] = v


# ---- files (agent_seg.py:701-704, 786-830; agent_abstract.py:136-153)
def checkpoint_prefix(config):
    ckpt_dir = config['training']['ckpt_save_dir']
    prefix = config['training'].get('ckpt_prefix', None)       # NB: the cfg key `ckpt_save_prefix` is not read
    return ckpt_dir, (ckpt_dir.split('/')[-1] if prefix is None else prefix)


def checkpoint_file(config, iteration):
    ckpt_dir, prefix = checkpoint_prefix(config)
    return "{0:}/{1:}_{2:}.pt".format(ckpt_dir, prefix, iteration)


def save_checkpoint(config, iteration, valid_pred, model_state_dict, optimizer, which="latest"):
    ckpt_dir, prefix = checkpoint_prefix(config)
    os.makedirs(ckpt_dir, exist_ok=True)
    save_dict = {'iteration': iteration, 'valid_pred': valid_pred, 'model_state_dict': model_state_dict,
                 'optimizer_state_dict': optimizer.state_dict()}
    torch.save(save_dict, checkpoint_file(config, iteration))
    with open("{0:}/{1:}_{2:}.txt".format(ckpt_dir, prefix, which), 'wt') as f:
        f.write(str(iteration))


def get_checkpoint_name(config):
    ckpt_mode = config['testing']['ckpt_mode']
    if ckpt_mode == 0 or ckpt_mode == 1:
        ckpt_dir, prefix = checkpoint_prefix(config)
        txt_name = ckpt_dir + '/' + prefix + ("_latest.txt" if ckpt_mode == 0 else "_best.txt")
        with open(txt_name, 'r') as f:
            it_num = f.read().replace('\n', '')
        return "{0:}/{1:}_{2:}.pt".format(ckpt_dir, prefix, it_num)
    return config['testing']['ckpt_name']

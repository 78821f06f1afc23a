"""CPU ORACLE (test infrastructure, NOT product code) - PyTorch-CPU restatement of the
FPL+ hot path: 3D U-Net with domain-specific BatchNorm, segmentation losses, the
`training_all` step and the sliding-window/TTA inferer.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (fpl-plus_amd/fplx) never does; it fails loudly without libfplx.so.

The reference's arithmetic lives in a third-party dependency (torch>=1.7.1, pinned only
by PyMIC/requirements.txt:10); this restatement therefore uses the SAME ATen CPU ops the
reference calls, arranged by our own code, and is PINNED against fixtures produced by
running the reference itself (tests/golden/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_golden.py).  All file:line citations are reference paths under
/root/reference/PyMIC/pymic/.

Extras the GPU tests need that the reference does not have:
  * `dropout_masks`: keep-masks supplied by the caller (the GPU path uses a counter-based
    Philox stream, oracle/np_ref.py:philox_keep_mask reproduces it) - the reference draws
    from torch's global RNG (net3d/unet2d5_dsbn.py:60-61,78) which no other device can replay.
  * `act_dtype=torch.bfloat16`: rounds activations to bf16 at exactly the points where the
    HIP pipeline stores bf16 tensors, so bf16 kernels can be checked tightly.
"""
import math
import torch
import torch.nn.functional as F

BN_EPS = 1e-5       # torch.nn.BatchNorm3d default, net_run_dsbn/dsbn.py:39-41
BN_MOMENTUM = 0.1


def _q(t, act_dtype):
    """Round-trip through the activation storage dtype (identity for fp32)."""
    if act_dtype is None or act_dtype == torch.float32:
        return t
    return t.to(act_dtype).to(torch.float32)


class _QuantSTE(torch.autograd.Function):
    """bf16 storage emulation with a straight-through gradient that is ALSO stored in bf16."""

    @staticmethod
    def forward(ctx, t, act_dtype):
        ctx.act_dtype = act_dtype
        return _q(t, act_dtype)

    @staticmethod
    def backward(ctx, g):
        return _q(g, ctx.act_dtype), None


def quant(t, act_dtype):
    if act_dtype is None or act_dtype == torch.float32:
        return t
    return _QuantSTE.apply(t, act_dtype)


def dsbn(x, sd, key, domain, train, update_stats=True):
    """DomainSpecificBatchNorm3d.forward (net_run_dsbn/dsbn.py:54-57): ONE BatchNorm3d,
    bns[domain_label[0]], for the whole batch.  Train: batch statistics (biased var for
    normalisation, unbiased for the running update, momentum 0.1, eps 1e-5)."""
    if x.dim() != 5:
        raise ValueError('expected 5D input (got {}D input)'.format(x.dim()))  # dsbn.py:61-64
    k = "%s.bns.%d." % (key, domain)
    w, b = sd[k + "weight"], sd[k + "bias"]
    rm, rv = sd[k + "running_mean"], sd[k + "running_var"]
    if train:
        dims = (0, 2, 3, 4)
        mean = x.mean(dims)
        var = x.var(dims, unbiased=False)
        if update_stats:
            n = x.numel() / x.shape[1]
            with torch.no_grad():
                rm.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
                rv.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * n / (n - 1))
                sd[k + "num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    sh = (1, -1, 1, 1, 1)
    return (x - mean.view(sh)) * torch.rsqrt(var.view(sh) + BN_EPS) * w.view(sh) + b.view(sh)


PRELU_TAPS = None      # tests set a list: every prelu() call appends (slope tensor, min(x, 0)) with the gradient retained


def prelu(x, slope):
    """nn.PReLU(), one shared slope (unet2d5_dsbn.py:62-63).  With PRELU_TAPS set the same values and gradients are formed
    as max(x, 0) + slope * min(x, 0) (bit-identical: one of the two terms is an exact zero) and min(x, 0) keeps its gradient,
    so that a test can form the slope gradient's cancellation scale sum |d out * min(x, 0)|."""
    if PRELU_TAPS is None:
        return torch.where(x > 0, x, x * slope)
    zero = torch.zeros((), dtype=x.dtype)
    neg = torch.where(x > 0, zero, x)
    if neg.requires_grad:
        neg.retain_grad()
    PRELU_TAPS.append((slope, neg))
    return torch.where(x > 0, x, zero) + neg * slope


def fold_depth(x):
    """DownBlock / UpBlock with dim == 2 (unet2d5_dsbn.py:110-115, 160-169): [N,C,D,H,W] -> [N*D,C,H,W]"""
    n, c, d, h, w = x.shape
    return torch.transpose(x, 1, 2).reshape(n * d, c, h, w)


def unfold_depth(x, n):
    """the inverse (unet2d5_dsbn.py:121-127, 185-188): [N*D,C,H,W] -> [N,C,D,H,W]"""
    nd, c, h, w = x.shape
    return torch.transpose(x.reshape(n, nd // n, c, h, w), 1, 2)


def dsbn2d(x, sd, key, domain, train):
    """DomainSpecificBatchNorm2d on [N*D,C,H,W]: the statistics over (0,2,3) are those of the 5-D form over (0,2,3,4)"""
    return dsbn(x.unsqueeze(2), sd, key, domain, train).squeeze(2)


def conv_block2d(x, sd, key, domain, train, p, keep_mask, act_dtype, dropout_on=None):
    """ConvBlockND.forward, 2D branch (net3d/unet2d5_dsbn.py:66-73) on the depth-folded tensor.  keep_mask (if any) is
    given in the 5-D layout [N,C,D,H,W] like the 3D branch's and folded here."""
    if dropout_on is None:
        dropout_on = train
    x = F.conv2d(x, quant(sd[key + ".conv2d_1.weight"], act_dtype), sd[key + ".conv2d_1.bias"], padding=1)
    x = quant(x, act_dtype)
    x = dsbn2d(x, sd, key + ".bn2d1", domain, train)
    x = prelu(x, sd[key + ".relu_1.weight"])
    if p > 0 and dropout_on:
        if keep_mask is None:
            x = F.dropout(x, p, True)
        else:
            p32 = float(torch.tensor(p, dtype=torch.float32))
            x = x * fold_depth(keep_mask).to(x.dtype) * float(torch.tensor(1.0 / (1.0 - p32), dtype=torch.float32))
    x = quant(x, act_dtype)
    x = F.conv2d(x, quant(sd[key + ".conv2d_2.weight"], act_dtype), sd[key + ".conv2d_2.bias"], padding=1)
    x = quant(x, act_dtype)
    x = dsbn2d(x, sd, key + ".bn2d2", domain, train)
    x = prelu(x, sd[key + ".relu_2.weight"])
    return quant(x, act_dtype)


def conv_block(x, sd, key, domain, train, p, keep_mask, act_dtype, dropout_on=None):
    """ConvBlockND.forward, 3D branch (net3d/unet2d5_dsbn.py:74-81)."""
    if dropout_on is None:
        dropout_on = train
    # the HIP path packs convolution weights into the activation dtype (bf16 operands, fp32 accumulate)
    x = F.conv3d(x, quant(sd[key + ".conv3d_1.weight"], act_dtype), sd[key + ".conv3d_1.bias"], padding=1)
    x = quant(x, act_dtype)
    x = dsbn(x, sd, key + ".bn3d1", domain, train)
    x = prelu(x, sd[key + ".relu_1.weight"])
    if p > 0 and dropout_on:
        if keep_mask is None:
            x = F.dropout(x, p, True)
        else:
            # fplx passes p as fp32 and scales by float(1 / (1 - double(p)))
            p32 = float(torch.tensor(p, dtype=torch.float32))
            x = x * keep_mask.to(x.dtype) * float(torch.tensor(1.0 / (1.0 - p32), dtype=torch.float32))
    x = quant(x, act_dtype)
    x = F.conv3d(x, quant(sd[key + ".conv3d_2.weight"], act_dtype), sd[key + ".conv3d_2.bias"], padding=1)
    x = quant(x, act_dtype)
    x = dsbn(x, sd, key + ".bn3d2", domain, train)
    x = prelu(x, sd[key + ".relu_2.weight"])
    return quant(x, act_dtype)


# layer ids used to key the dropout Philox stream; shared with fplx (one id per ConvBlockND)
BLOCK_KEYS = ["block0.conv", "block1.conv", "block2.conv", "block3.conv", "block4.conv",
              "up1.conv", "up2.conv", "up3.conv", "up4.conv"]


def block_dropout_p(net_params):
    d = net_params["dropout"]
    # UNet2D5_dsbn.__init__ (unet2d5_dsbn.py:279-291): blocks 0-4 use dropout[0..4], up1..up4 use [3],[2],[1],[0]
    return [d[0], d[1], d[2], d[3], d[4], d[3], d[2], d[1], d[0]]


def unet_forward(sd, net_params, x, domain, train=True, dropout_masks=None, act_dtype=None,
                 dropout_on=None):
    """UNet2D5_dsbn.forward (unet2d5_dsbn.py:296-309), both UpBlock forms (bilinear False / True); conv_dims[l] = 2 levels fold the depth axis
    into the batch and use the 2D members, exactly like DownBlock / UpBlock (108-129, 156-188).

    sd: dict key -> tensor (reference state_dict names, live members only).
    domain: python int (= domain_label[0], dsbn.py:56).
    dropout_masks: optional list of 9 keep-masks (or None entries), one per ConvBlockND.
    dropout_on: dropout active (defaults to `train`); eval-mode BN + active dropout is the
        FPL test-time setting (net_run_dsbn/agent_seg.py:843-852)."""
    ps = block_dropout_p(net_params)
    masks = dropout_masks or [None] * 9
    dims = list(net_params.get("conv_dims", [3] * 5))
    x = quant(x, act_dtype)
    n = x.shape[0]
    skips = []
    h = x
    for i in range(5):
        if dims[i] == 2:
            h2 = conv_block2d(fold_depth(h), sd, BLOCK_KEYS[i], domain, train, ps[i], masks[i], act_dtype, dropout_on)
            h = unfold_depth(h2, n)
            if i < 4:
                skips.append(h)
                h = unfold_depth(F.max_pool2d(h2, 2, 2), n)              # DownBlock, line 104/117
        else:
            h = conv_block(h, sd, BLOCK_KEYS[i], domain, train, ps[i], masks[i], act_dtype, dropout_on)
            if i < 4:
                skips.append(h)
                h = F.max_pool3d(h, 2, 2)                               # DownBlock, line 106/117
    bilinear = bool(net_params.get("bilinear", False))
    for j in range(4):
        key = "up%d" % (j + 1)
        if bilinear:
            # UpBlock with bilinear = True (unet2d5_dsbn.py:148-150, 172-176): Conv(kernel 1) then nn.Upsample(scale 2,
            # 'bilinear' / 'trilinear', align_corners=True)
            if dims[3 - j] == 2:
                up = F.conv2d(fold_depth(h), quant(sd[key + ".conv2d.weight"], act_dtype), sd[key + ".conv2d.bias"])
                up = quant(up, act_dtype)
                up = quant(F.interpolate(up, scale_factor=2, mode="bilinear", align_corners=True), act_dtype)
                hc = torch.cat([fold_depth(skips[3 - j]), up], dim=1)
                h = unfold_depth(conv_block2d(hc, sd, BLOCK_KEYS[5 + j], domain, train, ps[5 + j], masks[5 + j], act_dtype,
                                              dropout_on), n)
            else:
                up = F.conv3d(h, quant(sd[key + ".conv3d.weight"], act_dtype), sd[key + ".conv3d.bias"])
                up = quant(up, act_dtype)
                up = quant(F.interpolate(up, scale_factor=2, mode="trilinear", align_corners=True), act_dtype)
                h = torch.cat([skips[3 - j], up], dim=1)
                h = conv_block(h, sd, BLOCK_KEYS[5 + j], domain, train, ps[5 + j], masks[5 + j], act_dtype, dropout_on)
            continue
        if dims[3 - j] == 2:
            up = F.conv_transpose2d(fold_depth(h), quant(sd[key + ".trans2d.weight"], act_dtype),
                                    sd[key + ".trans2d.bias"], stride=2)   # line 179
            up = quant(up, act_dtype)
            hc = torch.cat([fold_depth(skips[3 - j]), up], dim=1)       # line 182
            h = unfold_depth(conv_block2d(hc, sd, BLOCK_KEYS[5 + j], domain, train, ps[5 + j], masks[5 + j], act_dtype,
                                          dropout_on), n)
            continue
        up = F.conv_transpose3d(h, quant(sd[key + ".trans3d.weight"], act_dtype), sd[key + ".trans3d.bias"],
                                stride=2)                                  # line 181
        up = quant(up, act_dtype)
        h = torch.cat([skips[3 - j], up], dim=1)                        # line 182
        h = conv_block(h, sd, BLOCK_KEYS[5 + j], domain, train, ps[5 + j], masks[5 + j], act_dtype, dropout_on)
    return F.conv3d(h, sd["out_conv.weight"], sd["out_conv.bias"], padding=(0, 1, 1))  # lines 293-294, 307


# ----------------------------------------------------------------------------- losses
def to_2d(x):
    """reshape_tensor_to_2D (loss/seg/util.py:36-50): [N,C,D,H,W] -> [N*D*H*W, C]."""
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1])


def classwise_dice(p2, y2, w2=None):
    """get_classwise_dice (loss/seg/util.py:85-107)."""
    if w2 is None:
        yv, pv, it = y2.sum(0), p2.sum(0), (y2 * p2).sum(0)
    else:
        yv, pv, it = (y2 * w2).sum(0), (p2 * w2).sum(0), (y2 * p2 * w2).sum(0)
    return (2.0 * it + 1e-5) / (yv + pv + 1e-5)


def dice_loss(logits, soft_y, pixel_weight=None, softmax=True):
    """DiceLoss.forward (loss/seg/dice.py:20-57)."""
    p = torch.softmax(logits, 1) if softmax else logits
    w2 = None if pixel_weight is None else to_2d(pixel_weight)
    return 1.0 - classwise_dice(to_2d(p), to_2d(soft_y), w2).mean()


def ce_loss(logits, soft_y, pixel_weight=None, softmax=True):
    """CrossEntropyLoss.forward (loss/seg/ce.py:23-44)."""
    p = torch.softmax(logits, 1) if softmax else logits
    p2 = to_2d(p) * 0.999 + 5e-4
    ce = -(to_2d(soft_y) * torch.log(p2)).sum(1)
    if pixel_weight is None:
        return ce.mean()
    w = to_2d(pixel_weight).squeeze()
    return (w * ce).sum() / (w.sum() + 1e-5)


def dice_loss_image_weighted(logits, soft_y, pixel_weight, image_weight, softmax=True):
    """DiceLoss_weight.forward (loss/seg/dice.py:106-128)."""
    p = torch.softmax(logits, 1) if softmax else logits
    tot = 0.0
    for i in range(p.shape[0]):
        d = classwise_dice(to_2d(p[i:i + 1]), to_2d(soft_y[i:i + 1]), to_2d(pixel_weight[i:i + 1]))
        tot = tot + (1.0 - d.mean()) * image_weight[i]
    return tot / p.shape[0]


def entropy_term(logits):
    """agent_seg.py:352-354: -(p*log2(p+1e-10)).sum()/(s0*s2*s3*s4) (the unpack `D,B,C,W,H`
    mislabels the axes; the divisor is N*D*H*W)."""
    p = torch.softmax(logits, 1)
    s = logits.shape
    return -(p * torch.log2(p + 1e-10)).sum() / (s[0] * s[2] * s[3] * s[4])


def loss_from_config(training_cfg):
    """create_loss_calculator (agent_seg.py:113-132) for the in-scope losses.
    Returns f(loss_input_dict) -> scalar; dict keys as loss/seg/abstract.py:23-37."""
    names = training_cfg["loss_type"]
    softmax = training_cfg.get("loss_softmax", True)

    def one(name):
        if name == "DiceLoss":
            return lambda d: dice_loss(d["prediction"], d["ground_truth"], d.get("pixel_weight"), softmax)
        if name == "CrossEntropyLoss":
            return lambda d: ce_loss(d["prediction"], d["ground_truth"], d.get("pixel_weight"), softmax)
        if name == "DiceLoss_weight":
            return lambda d: dice_loss_image_weighted(d["prediction"], d["ground_truth"], d["pixel_weight"],
                                                      d["image_weight"], softmax)
        raise ValueError("Undefined loss function {0:}".format(name))   # agent_seg.py:120-121

    if isinstance(names, (list, tuple)):
        fs = [one(n) for n in names]
        ws = training_cfg["loss_weight"]
        assert len(fs) == len(ws)                                        # combined.py:24
        return lambda d: sum(w * f(d) for w, f in zip(ws, fs))           # combined.py:34-37
    return one(names)


def hard_dice_metric(logits, soft_y):
    """agent_seg.py:472-476: argmax -> one-hot -> classwise Dice (no grad)."""
    with torch.no_grad():
        c = logits.shape[1]
        hard = F.one_hot(logits.argmax(1), c).permute(0, 4, 1, 2, 3).float()
        return classwise_dice(to_2d(hard), to_2d(soft_y))


# ----------------------------------------------------------------------------- optimiser
class AdamRef(object):
    """torch.optim.Adam(lr, weight_decay) as get_optimizer builds it (get_optimizer.py:17):
    betas (0.9, 0.999), eps 1e-8, L2 weight decay folded into the gradient; parameters whose
    grad is None are skipped entirely.  MultiStepLR(milestones, gamma) (get_optimizer.py:49-54)."""

    def __init__(self, params, lr, weight_decay, milestones=None, gamma=0.5):
        self.params = params           # dict name -> tensor (requires_grad leaf)
        self.lr0, self.wd = lr, weight_decay
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = {k: 0 for k in params}
        self.sched_steps = 0
        self.milestones, self.gamma = list(milestones or []), gamma

    @property
    def lr(self):
        return self.lr0 * self.gamma ** sum(1 for m in self.milestones if self.sched_steps >= m)

    def step(self):
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr = self.lr
        with torch.no_grad():
            for k, p in self.params.items():
                if p.grad is None:
                    continue
                g = p.grad + self.wd * p
                self.t[k] += 1
                t = self.t[k]
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = self.v[k].sqrt() / math.sqrt(1 - b2 ** t) + eps
                p.addcdiv_(self.m[k], denom, value=-lr / (1 - b1 ** t))

    def sched_step(self):
        self.sched_steps += 1

    def zero_grad(self):
        for p in self.params.values():
            p.grad = None


PARAM_SUFFIXES = (".weight", ".bias")


def split_state(sd_np, requires_grad=True):
    """numpy state dict -> (sd tensors, trainable params dict).  BN running stats are buffers."""
    sd, params = {}, {}
    for k, v in sd_np.items():
        t = torch.from_numpy(v.copy()) if not torch.is_tensor(v) else v.clone()
        if k.endswith(PARAM_SUFFIXES) and requires_grad:
            t.requires_grad_(True)
            params[k] = t
        sd[k] = t
    return sd, params


def training_all_step(sd, params, opt, net_params, batches, loss_fn, act_dtype=None, fpl_uda=True,
                      dropout_masks=None):
    """One iteration of SegmentationAgent.training_all (agent_seg.py:459-495): zero_grad, one
    forward per domain with domain_label = k, loss = (l0 + l1)/2 (l0 alone for one domain),
    backward, Adam step, scheduler step.  Returns (loss value, [class dice per domain])."""
    opt.zero_grad()
    loss, dices = None, []
    for k, b in enumerate(batches):
        logits = unet_forward(sd, net_params, b["image"], k, True,
                              None if dropout_masks is None else dropout_masks[k], act_dtype)
        d = {"prediction": logits, "ground_truth": b["label_prob"]}
        if fpl_uda and b.get("pixel_weight") is not None:          # get_loss_value, agent_seg.py:134-142
            d["pixel_weight"] = b["pixel_weight"]
            if b.get("image_weight") is not None:
                d["image_weight"] = b["image_weight"]
        lk = loss_fn(d)
        loss = lk if loss is None else (loss + lk) / 2
        dices.append(hard_dice_metric(logits, b["label_prob"]).numpy())
    loss.backward()
    opt.step()
    opt.sched_step()
    return float(loss.item()), dices


# ----------------------------------------------------------------------------- inferer
def sliding_window_infer(model_fn, image, class_num, window, stride):
    """Inferer.__infer_with_sliding_window, single-output branch (infer_func.py:50-112)."""
    shp = list(image.shape[2:])
    window, stride = list(window), list(stride)
    for d in range(3):
        if window[d] is None or window[d] > shp[d]:
            window[d] = shp[d]
        if stride[d] is None or stride[d] > window[d]:
            stride[d] = window[d]
    if all(window[d] >= shp[d] for d in range(3)):
        return model_fn(image)
    starts = []
    for w in range(0, shp[2], stride[2]):           # loop nest order of lines 75-83
        w0 = min(w, shp[2] - window[2])
        for h in range(0, shp[1], stride[1]):
            h0 = min(h, shp[1] - window[1])
            for d in range(0, shp[0], stride[0]):
                starts.append((min(d, shp[0] - window[0]), h0, w0))
    out = torch.zeros([image.shape[0], class_num] + shp)
    cnt = torch.zeros_like(out)
    for (d0, h0, w0) in starts:
        sl = (slice(None), slice(None), slice(d0, d0 + window[0]), slice(h0, h0 + window[1]),
              slice(w0, w0 + window[2]))
        out[sl] += model_fn(image[sl])
        cnt[sl] += 1.0
    return out / cnt


def inferer_run(model_fn, image, cfg):
    """Inferer.run (infer_func.py:188-222): optional 4-flip TTA (H, W, HW) averaged on logits."""
    def infer(img):
        if not cfg.get("sliding_window_enable", False):
            return model_fn(img)
        return sliding_window_infer(model_fn, img, cfg["class_num"], cfg["sliding_window_size"],
                                    cfg["sliding_window_stride"])
    tta = cfg.get("tta_mode", 0)
    if tta == 0:
        return infer(image)
    if tta != 1:
        raise ValueError("Undefined tta_mode {0:}".format(tta))
    o1 = infer(image)
    o2 = torch.flip(infer(torch.flip(image, [-2])), [-2])
    o3 = torch.flip(infer(torch.flip(image, [-1])), [-1])
    o4 = torch.flip(infer(torch.flip(image, [-2, -1])), [-2, -1])
    return (o1 + o2 + o3 + o4) / 4

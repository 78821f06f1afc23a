"""SegmentationAgent - host-side mirror of the reference's DSBN agent for the hot path
(reference: PyMIC/pymic/net_run_dsbn/agent_seg.py:34-1083 and agent_abstract.py:28-357).

Kept: the plugin surface (config dict with sections dataset / network / training / testing;
set_network / set_net_dict / set_loss_dict / set_optimizer / set_scheduler / set_inferer /
set_datasets-style loader injection), create_network, create_optimizer, create_loss_calculator,
get_loss_value, training_all (dual=True), training (dual=False), infer with its FPL branch, and
get_stage_dataset_from_config / create_dataset over device-resident NiftyDatasets with GPU transforms, the inverse
transforms of the prediction and save_outputs (uint8 masks as .nii.gz with the input's geometry), and (8f #4) validation,
train_valid with the reference's checkpoint files (`<prefix>_<it>.pt`, `_latest.txt`, `_best.txt`), get_checkpoint_name, run.
Not here (out of the hot-path tier): tensorboard, the `dis` adversarial branch.
Several GPUs: the reference wraps the network in nn.DataParallel over config[...]['gpus'] (agent_seg.py:692-698); here the
agent runs as ONE PROCESS PER GPU under `python -m torch.distributed.run` (fplx.ddp.init_from_env, RCCL): every rank
collates its chunk of each global batch (DataParallel's scatter), BatchNorm statistics stay per rank (per replica there),
the loss is evaluated over the full batch of all ranks (DataParallel's gather + one loss), gradients are all-reduced
before the optimizer step (its reduce_add), rank 0's running statistics and checkpoints win (replica 0's there), and
pseudo-label inference shards the volumes over the ranks with one gather of the (uncertainty, name) pairs to rank 0.
Loaders are any iterables of batch dicts ('image', 'label_prob', optional 'pixel_weight',
'image_weight', 'names'); create_dataset builds them from the config, set_loaders injects them.
"""
import copy
import logging
import os
import numpy as np
import torch
import torch.nn as nn
from torch.optim import lr_scheduler

from . import checkpoint as ckpt_mod
from . import ddp
from . import filter as fpl_filter_mod
from . import ops
from .dataset import NiftyDataset, BatchLoader
from .infer import Inferer
from .nifti import save_array_as_nifty_volume
from .transform import TransformDict, Compose
from .loss import SegLossDict, make_loss
from .net import UNet2D5_dsbn
from .optim import get_optimizer, get_lr_scheduler

SegNetDict = {
    'UNet2D5_dsbn': UNet2D5_dsbn,      # registry entry of the reference, net/net_dict_seg.py:44
    'UNet3D_dsbn': UNet2D5_dsbn,       # the all-3D configuration under its descriptive name
}


class SegmentationAgent(object):
    def __init__(self, config, stage='train'):
        assert (stage in ['train', 'inference', 'test'])                    # agent_abstract.py:44
        self.config = config
        self.stage = 'test' if stage == 'inference' else stage
        self.net = None
        self.optimizer = None
        self.scheduler = None
        self.net_dict = SegNetDict
        self.loss_dict = None
        self.inferer = None
        self.checkpoint = None
        self.train_loader_1 = self.train_loader_2 = self.test_loader = None
        self.engine_mode = True            # training loops through fplx.TrainStep where network, loss and optimiser are fplx's own
        self._ts = None
        self.tensor_type = config['dataset'].get('tensor_type', 'float')
        self.fpl_uda = config['training'].get('train_fpl_uda', False) if 'training' in config else False
        # one process per GPU: the process group comes up BEFORE anything touches the device
        # (the device is resolved ONCE, from the config's gpus list, and the group is bound to it: ADVICE r02)
        gpus = self._gpus()
        lr_ = ddp.local_rank()
        under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
        index = (gpus[lr_] if lr_ < len(gpus) else lr_) if under_launcher else gpus[0]
        self.distributed = ddp.init_from_env(device=index)
        self.rank, self.world = ddp.rank(), (ddp.world_size() if self.distributed else 1)
        self.device = torch.device("cuda:{0:}".format(index if self.distributed else gpus[0]))
        self.transform_list = []
        self.transform_dict = TransformDict
        self.test_set = None
        self.random_seed = config.get('training', {}).get('random_seed', 1)
        seed = config.get('training', {}).get('random_seed', 1)
        if config.get('training', {}).get('deterministic', True):
            torch.manual_seed(seed)                                          # agent_abstract.py:61-65

    def _gpus(self):
        sec = 'training' if self.stage == 'train' else 'testing'
        return self.config.get(sec, {}).get('gpus', [0])

    # ---- plugin setters (agent_abstract.py:67-134)
    def set_network(self, net):
        self.net = net

    def set_net_dict(self, net_dict):
        self.net_dict = net_dict

    def set_loss_dict(self, loss_dict):
        self.loss_dict = loss_dict

    def set_optimizer(self, optimizer):
        self.optimizer = optimizer

    def set_scheduler(self, scheduler):
        self.scheduler = scheduler

    def set_inferer(self, inferer):
        self.inferer = inferer

    def set_loaders(self, train_loader_1=None, train_loader_2=None, test_loader=None):
        self.train_loader_1, self.train_loader_2, self.test_loader = train_loader_1, train_loader_2, test_loader

    def set_transform_dict(self, custom_transform_dict):
        self.transform_dict = custom_transform_dict                          # agent_abstract.py:82-88

    def set_datasets(self, train_set_1=None, train_set_2=None, test_set=None):
        self.train_set_1, self.train_set_2, self.test_set = train_set_1, train_set_2, test_set

    # ---- datasets (agent_seg.py:42-77, agent_abstract.py:241-320)
    def get_stage_dataset_from_config(self, stage):
        assert (stage in ['1_train', '1_valid', '1_test', '2_train', '2_valid', '2_test', 'test'])
        ds = self.config['dataset']
        real_stage = stage.split('_')[-1]
        transform_key = real_stage + '_transform'
        if real_stage == "valid" and transform_key not in ds:
            transform_key = "train_transform"
        transform_names = ds[transform_key]
        self.transform_list = []
        data_transform = None
        if transform_names is not None and len(transform_names) > 0:
            transform_param = ds
            transform_param['task'] = 'segmentation'
            for name in transform_names:
                if name not in self.transform_dict:
                    raise ValueError("Undefined transform {0:}".format(name))
                self.transform_list.append(self.transform_dict[name](transform_param))
            data_transform = Compose(self.transform_list)
        return NiftyDataset(root_dir=ds['root_dir'], csv_file=ds.get(stage + '_csv', None),
                            modal_num=ds.get('modal_num', 1), with_label=not (stage == 'test'),
                            transform=data_transform, device=self.device, cache=ds.get('cache_on_device', True))

    def create_dataset(self):
        ds = self.config['dataset']
        if self.stage == 'train':
            bn_train = ds['train_batch_size']
            g_train = torch.Generator()
            g_train.manual_seed(self.random_seed)
            self.train_set_1 = self.get_stage_dataset_from_config('1_train')
            self.train_loader_1 = BatchLoader(self.train_set_1, bn_train, True, g_train, self.rank, self.world)
            if self.config['network']['num_domains'] == 2:
                self.train_set_2 = self.get_stage_dataset_from_config('2_train')
                self.train_loader_2 = BatchLoader(self.train_set_2, bn_train, True, g_train, self.rank, self.world)
            bn_valid = ds.get('valid_batch_size', 1)
            self.valid_loader_1 = self.valid_loader_2 = None
            if ds.get('1_valid_csv', None) is not None:
                self.valid_loader_1 = BatchLoader(self.get_stage_dataset_from_config('1_valid'), bn_valid, False)
            if self.config['network']['num_domains'] == 2 and ds.get('2_valid_csv', None) is not None:
                self.valid_loader_2 = BatchLoader(self.get_stage_dataset_from_config('2_valid'), bn_valid, False)
        else:
            if self.test_set is None:
                self.test_set = self.get_stage_dataset_from_config('test')
            self.test_loader = BatchLoader(self.test_set, ds.get('test_batch_size', 1), False)

    # ---- construction (agent_seg.py:82-132, agent_abstract.py:320-337)
    def create_network(self):
        if self.net is None:
            net_name = self.config['network']['net_type']
            if net_name not in self.net_dict:
                raise ValueError("Undefined network {0:}".format(net_name))  # agent_seg.py:86-87
            self.net = self.net_dict[net_name](self.config['network'])
        if self.tensor_type != 'float':
            raise ValueError("fplx: tensor_type must be float (fp32 parameters)")
        self.net.float()
        self.net.to(self.device)
        n = sum(p.numel() for p in self.net.parameters() if p.requires_grad)
        logging.info('parameter number {0:}'.format(n))
        if self.distributed and isinstance(self.net, UNet2D5_dsbn):
            ddp.broadcast_params_from_rank0(self.net)            # DataParallel replicates the master module every forward

    def get_parameters_to_update(self):
        return self.net.parameters()

    def create_optimizer(self, params=None):
        opt_params = self.config['training']
        if self.optimizer is None:
            self.optimizer = get_optimizer(opt_params['optimizer'], self.net, opt_params)
        last_iter = -1
        if self.checkpoint is not None:
            self.optimizer.load_state_dict(self.checkpoint['optimizer_state_dict'])
            last_iter = self.checkpoint['iteration'] - 1
        if self.scheduler is None:
            opt_params["last_iter"] = last_iter
            self.scheduler = get_lr_scheduler(self.optimizer, opt_params)
        if self.distributed and hasattr(self.optimizer, 'dist_sync'):
            self.optimizer.dist_sync = True                       # all-reduce(sum) of the gradients inside step()

    def create_loss_calculator(self, entropy_weight=0.0):
        if self.loss_dict is None:
            self.loss_dict = SegLossDict
        self.loss_calculator = make_loss(self.config['training'], self.loss_dict, entropy_weight)
        if self.distributed:
            self.loss_calculator.dist_sync = True                 # ONE loss over the full batch of all ranks

    def convert_tensor_type(self, t):
        return t.float()

    def get_loss_value(self, data, pred, gt, fpl_uda=False):
        """agent_seg.py:134-142"""
        d = {'prediction': pred, 'ground_truth': gt}
        if fpl_uda and data.get('pixel_weight', None) is not None:
            d['pixel_weight'] = data['pixel_weight'].to(pred.device)
            if data.get('image_weight', None) is not None:
                d['image_weight'] = data['image_weight'].to(pred.device)
        return self.loss_calculator(d)

    # ---- training loops
    def _next(self, it, loader):
        try:
            return next(it), it
        except StopIteration:
            it = iter(loader)
            return next(it), it

    def _engine_step(self):
        """fplx.TrainStep over THIS agent's network, loss terms and optimiser when all three are fplx's own: the training
        loops below then run the engine step - flat gradient buffer, fused loss, one Adam launch per segment, no autograd
        bookkeeping - instead of net(x) / loss.backward() / optimizer.step().  Same kernels, same numbers
        (tests/test_gpu_loss_filter_parity.py runs the reference's training_all fixtures through both); measured at the
        benchmark shape the autograd route was 9 % slower (bench.py `secondary.plugin_path`, round 4).  Under data
        parallelism (one process per GPU, round 5) the engine step is the route as well: the full-batch loss through
        ops.seg_loss_fwd_dist, the gradients through GradAllReducer on the default group, bucket by bucket DURING backward
        (training_all: during the last domain's backward, TrainStep.step_all) - the autograd route all-reduces the whole
        90 MB inside FusedAdam.step(), behind the last kernel.  Anything foreign - a network / loss / optimiser registered
        through set_network / set_loss_dict / set_optimizer, more than two domains, engine_mode = False - keeps the
        autograd route."""
        from .loss import AbstractSegLoss
        from .optim import FusedAdam
        from .train import TrainStep
        lc, opt, net = self.loss_calculator, self.optimizer, self.net
        ok = (self.engine_mode and type(net) is UNet2D5_dsbn and
              isinstance(lc, AbstractSegLoss) and type(lc).forward is AbstractSegLoss.forward and type(lc)._run is AbstractSegLoss._run
              and type(opt) is FusedAdam and opt.net is net and int(self.config['network']['num_domains']) <= 2)
        if not ok:
            self._ts = None
            return None
        key = (id(net), id(opt), tuple(lc.terms), bool(lc.softmax))
        if self._ts is None or self._ts_key != key:
            self._ts = TrainStep(net, lc.terms, lc.softmax, optimizer=opt)
            self._ts_key = key
        return self._ts

    def _train_loop(self, dual):
        class_num = self.config['network']['class_num']
        iter_valid = self.config['training']['iter_valid']
        nd = int(self.config['network']['num_domains'])
        loaders = [self.train_loader_1, self.train_loader_2][:nd]
        iters = [iter(l) for l in loaders]
        self.net.train()
        train_loss = None
        dice_lists = [[] for _ in range(nd)]
        ts = self._engine_step()
        for _ in range(iter_valid):
            datas = []
            for k in range(nd):
                d, iters[k] = self._next(iters[k], loaders[k])
                datas.append(d)
            if ts is not None:
                # engine route: the same iteration (agent_seg.py:459-495 / 336-357) on flat buffers
                bl = []
                for k in range(nd):
                    b = {'image': self.convert_tensor_type(datas[k]['image']).to(self.device),
                         'label_prob': self.convert_tensor_type(datas[k]['label_prob']).to(self.device)}
                    if self.fpl_uda and datas[k].get('pixel_weight', None) is not None:      # get_loss_value, agent_seg.py:134-142
                        b['pixel_weight'] = datas[k]['pixel_weight'].to(self.device)
                        if datas[k].get('image_weight', None) is not None:
                            b['image_weight'] = datas[k]['image_weight'].to(self.device)
                    bl.append(b)
                step_sched = self.scheduler is not None and not isinstance(self.scheduler, lr_scheduler.ReduceLROnPlateau)
                if dual:
                    outs = ts.step_all(bl)
                    loss = outs[0][0] if nd == 1 else (outs[0][0] + outs[1][0]) / 2           # agent_seg.py:468,482
                    if step_sched:
                        self.scheduler.step()
                    train_loss = loss if train_loss is None else train_loss + loss
                else:
                    outs = []
                    for k in range(nd):
                        o = ts.step(bl[k]['image'], bl[k]['label_prob'], k, bl[k].get('pixel_weight'), bl[k].get('image_weight'))
                        outs.append(o)
                        if step_sched:
                            self.scheduler.step()                                              # agent_seg.py:355-357
                        train_loss = o[0] if train_loss is None else train_loss + o[0]
                for k in range(nd):
                    dice_lists[k].append(outs[k][4:4 + class_num])
                    self.loss_calculator.last_out = outs[k]
                continue
            if dual:
                self.optimizer.zero_grad()
            loss = None
            for k in range(nd):
                x = self.convert_tensor_type(datas[k]['image']).to(self.device)
                y = self.convert_tensor_type(datas[k]['label_prob']).to(self.device)
                if not dual:
                    self.optimizer.zero_grad()
                if dual and k > 0 and isinstance(self.net, UNet2D5_dsbn):
                    self.net.parameters_unchanged_since_last_forward()     # one optimiser step per iteration: same weight packs
                out = self.net(x, domain_label=k * torch.ones(x.shape[0], dtype=torch.long))
                lk = self.get_loss_value(datas[k], out, y, self.fpl_uda)
                dice_lists[k].append(self.loss_calculator.last_out[4:4 + class_num])   # hard Dice of this pass
                if dual:
                    loss = lk if loss is None else (loss + lk) / 2                      # agent_seg.py:468,482
                else:
                    lk.backward()
                    self.optimizer.step()
                    if self.scheduler is not None and not isinstance(self.scheduler, lr_scheduler.ReduceLROnPlateau):
                        self.scheduler.step()                                           # agent_seg.py:355-357
                    train_loss = lk.detach() if train_loss is None else train_loss + lk.detach()
            if dual:
                loss.backward()                                                         # agent_seg.py:490-494
                self.optimizer.step()
                if self.scheduler is not None and not isinstance(self.scheduler, lr_scheduler.ReduceLROnPlateau):
                    self.scheduler.step()
                train_loss = loss.detach() if train_loss is None else train_loss + loss.detach()
        # one host sync per round, not per iteration
        train_avg_loss = float(train_loss.item()) / iter_valid / nd
        cls = [torch.stack(dl).mean(0).cpu().numpy().astype(np.float64) for dl in dice_lists]
        train_cls_dice = sum(cls) / nd
        return {'loss': train_avg_loss, 'avg_dice': float(train_cls_dice.mean()), 'class_dice': train_cls_dice}

    def training_all(self):
        """agent_seg.py:415-508 (selected by [training] dual = True, 748-749)"""
        return self._train_loop(True)

    def training(self):
        """agent_seg.py:291-414 with the backward/step the published loop lacks; the entropy
        regulariser of lines 352-354 is part of the loss here (create_loss_calculator(1.0))."""
        return self._train_loop(False)

    # ---- validation (agent_seg.py:509-606)
    def _valid_domain(self, loader, domain, class_num):
        losses, dices = [], []
        for data in loader:
            inputs = self.convert_tensor_type(data['image']).to(self.device)
            labels_prob = self.convert_tensor_type(data['label_prob']).to(self.device)
            outputs = self.inferer.run(self.net, inputs, domain * torch.ones(inputs.shape[0], dtype=torch.long))
            losses.append(float(self.get_loss_value(data, outputs, labels_prob).item()))
            hard = fpl_filter_mod.hard_label(outputs)                         # argmax on the device, uint8 [N,D,H,W]
            truth = fpl_filter_mod.hard_label(labels_prob)
            for i in range(inputs.shape[0]):                                  # get_classwise_dice of one-hot maps, fp32
                cnt = np.asarray(ops.overlap_counts(hard[i], truth[i], list(range(class_num))), np.float32)
                dices.append((np.float32(2.0) * cnt[:, 0] + np.float32(1e-5)) / (cnt[:, 2] + cnt[:, 1] + np.float32(1e-5)))
        return np.asarray(losses).mean(), np.asarray(dices).mean(axis=0)

    def validation(self):
        nd = int(self.config['network']['num_domains'])
        class_num = self.config['network']['class_num']
        if self.inferer is None:
            infer_cfg = dict(self.config.get('testing', {}))
            infer_cfg['class_num'] = class_num
            self.inferer = Inferer(infer_cfg)
        if self.distributed and isinstance(self.net, UNet2D5_dsbn):
            ddp.broadcast_buffers_from_rank0(self.net)           # replica 0's running statistics are the module's
        sync, self.loss_calculator.dist_sync = getattr(self.loss_calculator, 'dist_sync', False), False
        with torch.no_grad():                                    # every rank validates the whole set: same numbers everywhere
            self.net.eval()
            loss_0, cls_0 = self._valid_domain(self.valid_loader_1, 0, class_num)
            if nd == 2:
                loss_1, cls_1 = self._valid_domain(self.valid_loader_2, 1, class_num)
        self.loss_calculator.dist_sync = sync
        avg_0 = cls_0.mean()
        if nd == 2:
            avg_1 = cls_1.mean()
            loss, cls, avg = (loss_0 + loss_1) / 2, (cls_0 + cls_1) / 2, (avg_0 + avg_1) / 2
        else:
            loss, cls, avg = loss_0, cls_0, avg_0
        if isinstance(self.scheduler, lr_scheduler.ReduceLROnPlateau):
            self.scheduler.step(avg)
        tr = self.config['training']
        if nd == 2 and tr.get('val_t2', False):
            return {'loss': loss_1, 'avg_dice': avg_1, 'class_dice': cls_1}
        if tr.get('val_t1', False):
            return {'loss': loss_0, 'avg_dice': avg_0, 'class_dice': cls_0}
        return {'loss': loss, 'avg_dice': avg, 'class_dice': cls}

    # ---- train / validate / checkpoint loop (agent_seg.py:690-830)
    def get_checkpoint_name(self):
        return ckpt_mod.get_checkpoint_name(self.config)

    def train_valid(self):
        tr = self.config['training']
        if tr.get('dis', False):
            raise ValueError("fplx: the `dis` adversarial branch (agent_seg.py:249-275) is not built")
        self.dual = tr.get('dual', True)
        self.net.to(self.device)
        iter_start, iter_max, iter_valid = tr['iter_start'], tr['iter_max'], tr['iter_valid']
        iter_save = tr.get('iter_save', None)
        early_stop_it = tr.get('early_stop_patience', None)
        if iter_save is None:
            iter_save_list = [iter_max]
        elif isinstance(iter_save, (tuple, list)):
            iter_save_list = iter_save
        else:
            iter_save_list = range(0, iter_max + 1, iter_save)
        self.max_val_dice, self.max_val_it, self.best_model_wts, self.checkpoint = 0.0, 0, None, None
        if iter_start > 0:
            self.checkpoint = torch.load(ckpt_mod.checkpoint_file(self.config, iter_start), map_location=self.device,
                                         weights_only=False)
            self.checkpoint['valid_pred'] = 0                                 # agent_seg.py:723: resume restarts "best"
            self.net.load_state_dict(self.checkpoint['model_state_dict'])
            self.max_val_dice = self.checkpoint.get('valid_pred', 0)
            self.max_val_it = iter_start
            self.best_model_wts = self.checkpoint['model_state_dict']
        self.create_optimizer(self.get_parameters_to_update())
        # training() (dual = False) adds the prediction-entropy term to each domain's loss (agent_seg.py:352-354)
        self.create_loss_calculator(0.0 if self.dual else 1.0)
        self.glob_it = iter_start
        history = []
        for it in range(iter_start, iter_max, iter_valid):
            lr_value = self.optimizer.param_groups[0]['lr']
            train_scalars = self.training_all() if self.dual else self.training()
            valid_scalars = self.validation()
            self.glob_it = it + iter_valid
            logging.info("it {0:} lr {1:} train loss {2:.4f} valid dice {3:.4f}".format(
                self.glob_it, lr_value, train_scalars['loss'], valid_scalars['avg_dice']))
            history.append((self.glob_it, lr_value, train_scalars, valid_scalars))
            if valid_scalars['avg_dice'] > self.max_val_dice:
                self.max_val_dice = valid_scalars['avg_dice']
                self.max_val_it = self.glob_it
                self.best_model_wts = copy.deepcopy(ckpt_mod.reference_model_state_dict(self.net))
            stop_now = early_stop_it is not None and self.glob_it - self.max_val_it > early_stop_it
            if ((self.glob_it in iter_save_list) or stop_now) and self.rank == 0:
                ckpt_mod.save_checkpoint(self.config, self.glob_it, valid_scalars['avg_dice'],
                                         ckpt_mod.reference_model_state_dict(self.net), self.optimizer, "latest")
            if stop_now:
                logging.info("The training is early stopped")
                break
        # the best performing checkpoint (agent_seg.py:806-826)
        if self.best_model_wts is None:
            self.best_model_wts = ckpt_mod.reference_model_state_dict(self.net)
        if self.rank == 0:
            ckpt_mod.save_checkpoint(self.config, self.max_val_it, self.max_val_dice, self.best_model_wts, self.optimizer, "best")
        logging.info('The best performing iter is {0:}, valid dice {1:}'.format(self.max_val_it, self.max_val_dice))
        return history

    def run(self):
        """agent_abstract.py:348-357"""
        self.create_dataset()
        self.create_network()
        return self.train_valid() if self.stage == 'train' else self.infer()

    # ---- inference (agent_seg.py:834-964)
    def infer(self, mc_passes=6, return_outputs=False):
        cfg = self.config['testing']
        ckpt_names = None
        if 'ckpt_mode' in cfg and self.checkpoint is None:                  # agent_seg.py:854-866
            ckpt_name = self.get_checkpoint_name()
            if cfg['ckpt_mode'] == 3:
                assert (isinstance(ckpt_name, (tuple, list)))
                ckpt_names = list(ckpt_name)                                  # ensemble of checkpoints (966-1019)
            elif isinstance(ckpt_name, (tuple, list)):
                raise ValueError("ckpt_mode should be 3 if ckpt_name is a list")
            else:
                self.checkpoint = torch.load(ckpt_name, map_location=self.device, weights_only=False)
        domian_label = cfg['domian_label']
        self.FPL = cfg.get('fpl', False)
        self.net.to(self.device)
        if cfg.get('evaluation_mode', True):
            self.net.eval()
            if cfg.get('test_time_dropout', False) or self.FPL:
                def test_time_dropout(m):
                    if type(m) == nn.Dropout:
                        m.train()
                self.net.apply(test_time_dropout)                            # agent_seg.py:845-852
        if self.checkpoint is not None:
            self.net.load_state_dict(self.checkpoint['model_state_dict'])
        if self.inferer is None:
            infer_cfg = dict(cfg)
            infer_cfg['class_num'] = self.config['network']['class_num']
            self.inferer = Inferer(infer_cfg)
        uncertainty_list, outputs = {}, {}
        if ckpt_names is not None:
            return self.infer_with_multiple_checkpoints(ckpt_names, domian_label)
        # an injected inferer that overrides run() is called pass by pass, as the reference does
        batched_mc = isinstance(self.inferer, Inferer) and type(self.inferer).run is Inferer.run
        with torch.no_grad():
            for case_no, data in enumerate(self.test_loader):
                if self.distributed and case_no % self.world != self.rank:
                    continue                                             # volumes are sharded over the ranks
                images = self.convert_tensor_type(data['image']).to(self.device)
                names = data['names']
                dl = domian_label * torch.ones(images.shape[0], dtype=torch.long)
                if self.FPL:
                    if batched_mc:                                           # all passes x flips x tiles in one batch
                        stack = self.inferer.run_mc(self.net, images, dl, mc_passes)[:, 0]
                    else:
                        stack = torch.empty((mc_passes, self.config['network']['class_num']) + tuple(images.shape[2:]),
                                            dtype=torch.float32, device=self.device)
                        for i in range(mc_passes):                           # agent_seg.py:898-899 (6 passes)
                            stack[i] = self.inferer.run(self.net, images, dl)[0]
                    r = fpl_filter_mod.fpl_uncertainty(stack)
                    uncertainty_list[names[0]] = r['uncer_one']
                    if return_outputs:
                        outputs[names[0]] = r
                else:
                    pred = self.inferer.run(self.net, images, dl)
                    outputs.update(self._finish_prediction(data, pred, names))
        if self.FPL:
            if self.distributed:                                         # the only exchange: (uncertainty, name) pairs to rank 0
                parts = ddp.gather_objects_to_rank0(uncertainty_list)
                if self.rank == 0:
                    uncertainty_list = {}
                    for part in parts:
                        uncertainty_list.update(part)
            srt = fpl_filter_mod.sort_uncertainty(uncertainty_list)     # agent_seg.py:957-959
            path = cfg.get('fpl_uncertainty_sorted', None)
            if path and self.rank == 0:
                np.save(path, np.array(srt, dtype=object), allow_pickle=True)
            return (srt, outputs) if return_outputs else srt
        return outputs

    def _finish_prediction(self, data, pred, names):
        """agent_seg.py:944-953: inverse transforms of the prediction, hard labels, optional files"""
        cfg = self.config['testing']
        data['predict'] = pred
        for transform in self.transform_list[::-1]:                          # agent_seg.py:944-947
            if transform.inverse:
                data = transform.inverse_transform_for_prediction(data)
        pr = data['predict']
        hard = fpl_filter_mod.hard_label(pr[0] if isinstance(pr, (list, tuple)) else pr)   # save_outputs, 1049-1050
        if cfg.get('output_dir', None) is not None:
            self.save_outputs(data, hard)
        return {name: hard[i] for i, name in enumerate(names)}

    def infer_with_multiple_checkpoints(self, ckpt_names, domian_label):
        """agent_seg.py:966-1019 (ckpt_mode = 3): the prediction is the mean over the checkpoints' logits - numpy's float32
        mean over the list axis = additions in list order, one division - formed on the device."""
        states = [torch.load(n, map_location=self.device, weights_only=False)['model_state_dict'] for n in ckpt_names]
        outputs = {}
        with torch.no_grad():
            for case_no, data in enumerate(self.test_loader):
                if self.distributed and case_no % self.world != self.rank:
                    continue                 # volumes are sharded over the ranks, as in infer(): one writer per output file
                images = self.convert_tensor_type(data['image']).to(self.device)
                dl = domian_label * torch.ones(images.shape[0], dtype=torch.long)
                acc = None
                for sd in states:
                    self.net.load_state_dict(sd)
                    pred = self.inferer.run(self.net, images, dl)
                    if isinstance(pred, (tuple, list)):
                        acc = [p.clone() for p in pred] if acc is None else [a + p for a, p in zip(acc, pred)]
                    else:
                        acc = pred.clone() if acc is None else acc + pred
                k = float(len(states))
                pred = [a / k for a in acc] if isinstance(acc, list) else acc / k
                outputs.update(self._finish_prediction(data, pred, data['names']))
        return outputs

    def save_outputs(self, data, hard=None):
        """agent_seg.py:1022-1083: uint8 argmax masks (softmax is monotone, the argmax is taken on the device) written to
        <output_dir>/<basename(ckpt_save_dir)>_<test csv stem>/<name> with the geometry of the input image."""
        cfg = self.config['testing']
        ignore_dir = cfg.get('filename_ignore_dir', True)
        src, dst = cfg.get('filename_replace_source', None), cfg.get('filename_replace_target', None)
        ckpt_dir = self.config.get('training', {}).get('ckpt_save_dir', 'model').split('/')[-1]
        subset = self.config['dataset'].get('test_csv', 'test.csv').split('/')[-1][:-4]
        output_dir = os.path.join(cfg['output_dir'], ckpt_dir + '_' + subset)
        self.output_dir = output_dir
        os.makedirs(output_dir, exist_ok=True)
        names, pred = data['names'], data['predict']
        if isinstance(pred, (list, tuple)):
            pred = pred[0]
        output = (fpl_filter_mod.hard_label(pred) if hard is None else hard).cpu().numpy()
        ls, lt = cfg.get('label_source', None), cfg.get('label_target', None)
        if ls is not None and lt is not None:                                 # util/image_process.convert_label
            conv = np.zeros_like(output)
            for a, b in zip(ls, lt):
                conv[output == a] = b
            output = conv
        root_dir = self.config['dataset']['root_dir']
        for i in range(len(names)):
            save_name = names[i].split('/')[-1] if ignore_dir else names[i].replace('/', '_')
            if src is not None and dst is not None:
                save_name = save_name.replace(src, dst)
            save_array_as_nifty_volume(output[i], "{0:}/{1:}".format(output_dir, save_name), root_dir + '/' + names[i])

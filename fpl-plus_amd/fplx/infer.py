"""Inferer - mirror of reference PyMIC/pymic/net_run_dsbn/infer_func.py:7-222 (single-output
networks): sliding window with overlap averaging and 4-flip test-time augmentation, every
forward passes `domain_label`.  Same config keys (`sliding_window_enable`, `sliding_window_size`,
`sliding_window_stride`, `tta_mode`, `class_num`) and the same `run(model, image, domain_label)`.
Tensors stay on the GPU; accumulation is fp32 in the reference's tile order.
"""
import torch


class Inferer(object):
    def __init__(self, config):
        self.config = config

    def _infer(self, image, domain_label):
        if not self.config.get('sliding_window_enable', False):
            return self.model(image, domain_label=domain_label)
        return self._infer_with_sliding_window(image, domain_label)

    def _infer_with_sliding_window(self, image, domain_label):
        window_size = [x for x in self.config['sliding_window_size']]
        window_stride = [x for x in self.config['sliding_window_stride']]
        class_num = self.config['class_num']
        img_full_shape = list(image.shape)
        img_shape = img_full_shape[2:]
        img_dim = len(img_shape)
        if img_dim != 3:
            raise ValueError("Inference using sliding window only supports 2D and 3D images")   # infer_func.py:63-64
        for d in range(img_dim):                                        # infer_func.py:66-70
            if (window_size[d] is None) or window_size[d] > img_shape[d]:
                window_size[d] = img_shape[d]
            if (window_stride[d] is None) or window_stride[d] > window_size[d]:
                window_stride[d] = window_size[d]
        if all([window_size[d] >= img_shape[d] for d in range(img_dim)]):
            return self.model(image, domain_label)
        crop_start_list = []                                            # same nesting as infer_func.py:75-84
        for w in range(0, img_shape[-1], window_stride[-1]):
            w_min = min(w, img_shape[-1] - window_size[-1])
            for h in range(0, img_shape[-2], window_stride[-2]):
                h_min = min(h, img_shape[-2] - window_size[-2])
                for d in range(0, img_shape[0], window_stride[0]):
                    d_min = min(d, img_shape[0] - window_size[0])
                    crop_start_list.append([d_min, h_min, w_min])
        output_shape = [img_full_shape[0], class_num] + img_shape
        output = torch.zeros(output_shape, device=image.device)
        counter = torch.zeros(output_shape, device=image.device)
        for c0 in crop_start_list:
            c1 = [c0[d] + window_size[d] for d in range(img_dim)]
            patch_in = image[:, :, c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]]
            patch_out = self.model(patch_in, domain_label=domain_label)
            if isinstance(patch_out, (tuple, list)):
                patch_out = patch_out[0]
            output[:, :, c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]] += patch_out
            counter[:, :, c0[0]:c1[0], c0[1]:c1[1], c0[2]:c1[2]] += 1.0
        return output / counter

    def run(self, model, image, domain_label):
        """infer_func.py:188-222"""
        self.model = model
        tta_mode = self.config.get('tta_mode', 0)
        if tta_mode == 0:
            return self._infer(image, domain_label)
        if tta_mode == 1:
            outputs1 = self._infer(image, domain_label)
            outputs2 = torch.flip(self._infer(torch.flip(image, [-2]), domain_label), [-2])
            outputs3 = torch.flip(self._infer(torch.flip(image, [-1]), domain_label), [-1])
            outputs4 = torch.flip(self._infer(torch.flip(image, [-2, -1]), domain_label), [-2, -1])
            return (outputs1 + outputs2 + outputs3 + outputs4) / 4
        raise ValueError("Undefined tta_mode {0:}".format(tta_mode))

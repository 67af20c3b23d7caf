"""mirrors det3d/models/bbox_heads/mg_head_sessd.py: Head (:195-230), MultiGroupHead ctor (:379-493),
forward (:518-523), predict / get_task_detections (:893-1057), and the training losses: loss (:706-808),
get_model_ema_loss (:810-890), consistency_loss (:618-704), nn_distance (:573-607), prepare_loss_weights (:525-571) with the
module-level helpers (:27-77). The geometric pieces of the loss run on the HIP kernels (ODIoU: sessd_odiou3d; IoU targets and
teacher-student matching: the iou3d operators); the rest is elementwise torch on the same device."""
import logging

import numpy as np
import torch
from torch import nn

from sessd_hip import ops

from ..builder import build_loss
from ..registry import HEADS


@HEADS.register_module
class Head(nn.Module):
    def __init__(self, num_input, num_pred, num_cls, use_dir=False, num_dir=0, header=True, name="",
                 focal_loss_init=False, **kwargs):
        super().__init__(**kwargs)
        self.use_dir = use_dir
        self.conv_box = nn.Conv2d(num_input, num_pred, 1)
        self.conv_cls = nn.Conv2d(num_input, num_cls, 1)
        self.conv_iou = nn.Conv2d(num_input, 2, 1)
        self.trans_conv = None
        if self.use_dir:
            self.conv_dir = nn.Conv2d(num_input, num_dir, 1)
        self._packed = None
        self.fused_train = True  # train mode: the heads as one conv on the HIP kernels (False: four torch / MIOpen convs)

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, None if k == "_packed" else copy.deepcopy(v, memo))
        return new

    def planar(self, x):
        """(B, 14+2+4+2, H, W): the four 1x1 convs as ONE fused launch, channels [box | cls | dir | iou]."""
        convs = [self.conv_box, self.conv_cls] + ([self.conv_dir] if self.use_dir else []) + [self.conv_iou]
        key = (ops.param_generation(),) + tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in convs)
        if self._packed is None or self._packed[0] != key:
            w = torch.cat([c.weight.detach() for c in convs], 0)
            b = torch.cat([c.bias.detach() for c in convs], 0).float().contiguous()
            self._packed = (key, ops.pack_conv2d(w), b, [c.out_channels for c in convs])
        _, pc, b, split = self._packed
        return ops.conv2d(x.float().contiguous(), pc, None, b, False), split

    def forward(self, x):
        if self.training and self.fused_train and x.is_cuda and x.shape[1] % 2 == 0:
            # autograd path on the HIP kernels: the four 1x1 convs (mg_head_sessd.py:217-230) as ONE 22-channel conv through
            # ops.Conv2dFunction (forward sessd_conv2d_mfma, data gradient = the adjoint 1x1 conv, weight gradient
            # sessd_conv2d_wgrad); torch.cat / split route the gradients back to the four modules' parameters. Round 2 ran them
            # as four MIOpen convs forward and backward.
            convs = [self.conv_box, self.conv_cls] + ([self.conv_dir] if self.use_dir else []) + [self.conv_iou]
            w = torch.cat([c.weight for c in convs], 0)
            b = torch.cat([c.bias for c in convs], 0)
            y = ops.Conv2dFunction.apply(x, w, b, False, 1)
            names = ["box_preds", "cls_preds"] + (["dir_cls_preds"] if self.use_dir else []) + ["iou_preds"]
            # the parts in NHWC (each conv's `.permute(0, 2, 3, 1).contiguous()` of the reference): one launch each way
            return dict(zip(names, ops.split_nhwc(y, [c.out_channels for c in convs])))
        if self.training:  # the four 1x1 convs as torch modules
            ret = dict(box_preds=self.conv_box(x).permute(0, 2, 3, 1).contiguous(),
                       cls_preds=self.conv_cls(x).permute(0, 2, 3, 1).contiguous())
            if self.use_dir:
                ret["dir_cls_preds"] = self.conv_dir(x).permute(0, 2, 3, 1).contiguous()
            ret["iou_preds"] = self.conv_iou(x).permute(0, 2, 3, 1).contiguous()
            return ret
        y, split = self.planar(x)
        parts = torch.split(y, split, dim=1)
        names = ["box_preds", "cls_preds"] + (["dir_cls_preds"] if self.use_dir else []) + ["iou_preds"]
        ret = {n: p.permute(0, 2, 3, 1).contiguous() for n, p in zip(names, parts)}
        ret["_planar"] = y
        return ret


def one_hot_f(tensor, depth, dim=-1, on_value=1.0, dtype=torch.float32):
    out = torch.zeros(*tensor.shape, depth, dtype=dtype, device=tensor.device)
    return out.scatter_(dim, tensor.unsqueeze(dim).long(), on_value)


def add_sin_difference(boxes1, boxes2):
    """sin(a - b) = sin a cos b - cos a sin b: replace the yaw of the prediction by sin(a)cos(b) and of the target by
    cos(a)sin(b), so that their smooth-L1 difference is sin(a - b) (mg_head_sessd.py:39-44)."""
    a, b = boxes1[..., -1:], boxes2[..., -1:]
    return (torch.cat([boxes1[..., :-1], torch.sin(a) * torch.cos(b)], dim=-1),
            torch.cat([boxes2[..., :-1], torch.cos(a) * torch.sin(b)], dim=-1))


def get_direction_target(anchors, reg_targets, one_hot=True, dir_offset=0.0):
    """Direction class = (target yaw + anchor yaw - offset) > 0 (mg_head_sessd.py:62-76)."""
    B = reg_targets.shape[0]
    rot_gt = reg_targets[..., -1] + anchors.view(B, -1, anchors.shape[-1])[..., -1]
    t = ((rot_gt - dir_offset) > 0).long()
    return one_hot_f(t, 2, dtype=anchors.dtype) if one_hot else t


def _get_pos_neg_loss(cls_loss, labels):
    """Sum of the classification loss over the positive / the negative anchors, per batch sample (mg_head_sessd.py:47-59)."""
    B = cls_loss.shape[0]
    if cls_loss.shape[-1] == 1 or cls_loss.dim() == 2:
        flat = cls_loss.view(B, -1)
        return ((labels > 0).type_as(flat) * flat).sum() / B, ((labels == 0).type_as(flat) * flat).sum() / B
    return cls_loss[..., 1:].sum() / B, cls_loss[..., 0].sum() / B


@HEADS.register_module
class MultiGroupHead(nn.Module):
    def __init__(self, mode="3d", in_channels=[128, ], norm_cfg=None, tasks=[], weights=[], num_classes=[1, ],
                 box_coder=None, with_cls=True, with_reg=True, reg_class_agnostic=False,
                 encode_background_as_zeros=True,
                 loss_norm=dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=1.0),
                 loss_cls=dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0),
                 use_sigmoid_score=True,
                 loss_bbox=dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=[1.0] * 7, codewise=True, loss_weight=2.0),
                 encode_rad_error_by_sin=True,
                 loss_aux=dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier", loss_weight=0.2),
                 direction_offset=0.0, name="rpn", logger=None):
        super().__init__()
        assert with_cls or with_reg
        num_classes = [len(t["class_names"]) for t in tasks]
        self.class_names = [t["class_names"] for t in tasks]
        self.num_anchor_per_locs = [2 * n for n in num_classes]
        self.box_coder = box_coder
        box_code_sizes = [box_coder.n_dim] * len(num_classes)
        self.with_cls, self.with_reg = with_cls, with_reg
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.reg_class_agnostic = reg_class_agnostic
        self.encode_rad_error_by_sin = encode_rad_error_by_sin
        self.encode_background_as_zeros = encode_background_as_zeros
        self.use_sigmoid_score = use_sigmoid_score
        self.box_n_dim = self.box_coder.n_dim
        self.loss_cls = build_loss(loss_cls)
        self.loss_reg = build_loss(loss_bbox)
        if loss_aux is not None:
            self.loss_aux = build_loss(loss_aux)
        self.loss_norm = loss_norm
        self.logger = logger or logging.getLogger("MultiGroupHead")
        self.use_direction_classifier = loss_aux is not None
        if loss_aux:
            self.direction_offset = direction_offset
        self.bev_only = mode == "bev"
        num_clss, num_preds, num_dirs = [], [], []
        for num_c, num_a, box_cs in zip(num_classes, self.num_anchor_per_locs, box_code_sizes):
            num_clss.append(num_a * num_c if self.encode_background_as_zeros else num_a * (num_c + 1))
            num_preds.append(num_a * (box_cs - 2) if self.bev_only else num_a * box_cs)
            if self.use_direction_classifier:
                num_dirs.append(num_a * 2)
        self.logger.info(f"num_classes: {num_classes}, num_preds: {num_preds}, num_dirs: {num_dirs}")
        self.tasks = nn.ModuleList()
        for task_id, (num_pred, num_cls) in enumerate(zip(num_preds, num_clss)):
            self.tasks.append(Head(in_channels, num_pred, num_cls, use_dir=self.use_direction_classifier,
                                   num_dir=num_dirs[task_id] if self.use_direction_classifier else None, header=False))
        self.logger.info("Finish MultiGroupHead Initialization")
        # the reference hard-codes these and calls .cuda() in the ctor (mg_head_sessd.py:484-487); kept as plain
        # attributes so that the module can be built without a device
        self.post_center_range = [0, -40.0, -5.0, 70.4, 40.0, 5.0]
        self.thresh = 0.3
        # training-only members (mg_head_sessd.py:430-431, 488-491); the geometric loss is the device op
        sl1 = dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=None, codewise=True, loss_weight=1.0)
        self.loss_iou_pred = build_loss(dict(sl1))
        self.loss_iou_consistency = build_loss(dict(sl1))
        self.loss_score_consistency = build_loss(dict(sl1))
        self.loss_dir_consistency = nn.MSELoss(reduction="mean")

    def init_weights(self, pretrained=None):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return [task(x) for task in self.tasks]

    # ------------------------------------------------------------------ training losses
    def prepare_loss_weights(self, labels, loss_norm=None, dtype=torch.float32):
        """labels (B,A) in {-1 ignore, 0 negative, >0 positive} -> cls_weights, reg_weights, cared (mg_head_sessd.py:525-571)."""
        loss_norm = loss_norm or self.loss_norm
        pos, neg, cared = labels > 0, labels == 0, labels >= 0
        cls_w = neg.type(dtype) * loss_norm["neg_cls_weight"] + pos.type(dtype) * loss_norm["pos_cls_weight"]
        reg_w = pos.type(dtype)
        n_pos = torch.clamp(pos.sum(1, keepdim=True).type(dtype), min=1.0)
        kind = loss_norm["type"]
        if kind == "NormByNumPositives":
            reg_w, cls_w = reg_w / n_pos, cls_w / n_pos
        elif kind == "NormByNumExamples":
            cls_w = cls_w / torch.clamp(cared.type(dtype).sum(1, keepdim=True), min=1.0)
            reg_w = reg_w / n_pos
        elif kind == "DontNorm":
            reg_w = reg_w / n_pos
        elif kind == "NormByNumPosNeg":
            pn = torch.stack([pos, neg], dim=-1).type(dtype)
            norm = pn.sum(1, keepdim=True)
            cls_w = cls_w / torch.clamp((pn * norm).sum(-1), min=1.0)
            reg_w = reg_w / torch.clamp(norm, min=1.0)[:, 0:1, 0]
        else:
            raise ValueError("unknown loss norm type %r" % (kind,))
        return cls_w, reg_w, cared

    def _supervised(self, preds, labels, reg_targets, anchors, with_odiou):
        """The per-task supervised terms shared by loss() (:715-780) and get_model_ema_loss() (:815-868):
        focal classification, (logged) smooth-L1 localisation on sin-encoded yaw, direction softmax on the positives, smooth-L1
        regression of the predicted IoU towards 2*IoU3D(decoded prediction, decoded target) - 1, and (student only) ODIoU."""
        from det3d.core.iou3d import iou3d_utils
        B = anchors.shape[0]
        cls_w, reg_w, cared = self.prepare_loss_weights(labels)
        cls_targets = (labels * cared.type_as(labels)).unsqueeze(-1)
        box = preds["box_preds"].view(B, -1, self.box_n_dim)
        cls = preds["cls_preds"].view(B, -1, self.num_classes[0])
        enc_p, enc_t = add_sin_difference(box, reg_targets) if self.encode_rad_error_by_sin else (box, reg_targets)
        loc = self.loss_reg(enc_p, enc_t, weights=reg_w)
        focal = self.loss_cls(cls, cls_targets, weights=cls_w)
        out = dict(loc_loss_reduced=self.loss_reg._loss_weight * loc.sum() / B, cls_loss_reduced=self.loss_cls._loss_weight * focal.sum() / B,
                   loc_loss_elem=[loc[:, :, i].sum() / B for i in range(loc.shape[-1])])
        pos_l, neg_l = _get_pos_neg_loss(focal, labels)
        out["cls_pos_loss"], out["cls_neg_loss"] = pos_l / self.loss_norm["pos_cls_weight"], neg_l / self.loss_norm["neg_cls_weight"]
        dir_loss = box.new_zeros(())
        if self.use_direction_classifier:
            dir_targets = get_direction_target(anchors, reg_targets, dir_offset=self.direction_offset)
            w = (labels > 0).type_as(box)
            w = w / torch.clamp(w.sum(-1, keepdim=True), min=1.0)
            dir_loss = self.loss_aux._loss_weight * self.loss_aux(preds["dir_cls_preds"].view(B, -1, 2), dir_targets, weights=w).sum() / B
        out["dir_loss"] = dir_loss
        pos = reg_w > 0
        qboxes = self.box_coder.decode_torch(box[pos], anchors[pos])
        gboxes = self.box_coder.decode_torch(reg_targets[pos], anchors[pos])
        if int(pos.sum()) > 0:
            iou_t = 2 * iou3d_utils.boxes_aligned_iou3d_gpu(qboxes.detach(), gboxes).detach() - 1
        else:
            iou_t = box.new_zeros((0, 1))
        out["iou_pred_loss"] = self.loss_iou_pred(preds["iou_preds"].view(B, -1, 1)[pos], iou_t, reg_w[pos]).sum() / B
        if with_odiou:
            out["ious_loss"] = ops.odiou_3d_loss(gboxes, qboxes, reg_w[pos], B) if int(pos.sum()) > 0 else box.new_zeros(())
        out["num_pos"], out["num_neg"] = (labels > 0)[0].sum(), (labels == 0)[0].sum()
        return out

    def nn_distance(self, box1, box2, iou_thres=0.7):
        """Mutual matching of student (box1) and teacher (box2) boxes by BEV IoU > iou_thres (mg_head_sessd.py:573-607, return
        mode '10'): smooth-L1 (yaw sin-encoded) between every kept student box and its best teacher box, averaged."""
        from det3d.core.iou3d import iou3d_utils
        iou = iou3d_utils.boxes_iou_bev_gpu(box1.detach().contiguous(), box2.detach().contiguous())
        m1, m2 = iou.max(dim=1)[0] > iou_thres, iou.max(dim=0)[0] > iou_thres
        iou = iou[m1][:, m2]
        if iou.shape[0] == 0 or iou.shape[1] == 0:
            return [None] * 5
        idx1, idx2 = iou.max(dim=1)[1], iou.max(dim=0)[1]
        enc_s, enc_t = add_sin_difference(box1[m1], box2[m2][idx1])
        per_box = self.loss_reg(enc_s, enc_t).sum(-1) / 7.0
        return per_box.sum() / per_box.shape[0], idx1, idx2, m1, m2

    def consistency_loss(self, preds_stu, preds_tea, example):
        """Teacher-student consistency (mg_head_sessd.py:618-704): per sample, boxes with sigmoid score >= 0.3 inside the
        post-centre range; teacher boxes mapped into the student's frame with the recorded global augmentation (flip, rotation,
        scale); box / score / IoU-prediction terms over the mutually matched pairs. (box + cls + iou) / batch_size."""
        from det3d.core.bbox import box_torch_ops
        B = preds_stu[0]["box_preds"].shape[0]
        anchors0 = example["anchors"][0][0]   # :649-650: every sample is decoded with the anchors of sample 0
        dev = anchors0.device
        lo, hi = (torch.tensor(self.post_center_range[:3], device=dev), torch.tensor(self.post_center_range[3:], device=dev))
        V = lambda d, k, c: d[0][k].view(B, -1, c)
        box_l = cls_l = iou_l = torch.zeros((1,), dtype=torch.float32, device=dev)
        for b in range(B):
            trans = example["transformation"][b]
            bs = self.box_coder.decode_torch(V(preds_stu, "box_preds", 7)[b], anchors0)
            bt = self.box_coder.decode_torch(V(preds_tea, "box_preds", 7)[b], anchors0)
            cs, ct = V(preds_stu, "cls_preds", 1)[b], V(preds_tea, "cls_preds", 1)[b]
            ms = (bs[:, :3] >= lo).all(1) & (bs[:, :3] <= hi).all(1) & (torch.sigmoid(cs).squeeze(-1) >= 0.3)
            mt = (bt[:, :3] >= lo).all(1) & (bt[:, :3] <= hi).all(1) & (torch.sigmoid(ct).squeeze(-1) >= 0.3)
            if int(ms.sum()) == 0 or int(mt.sum()) == 0:
                continue
            top_s, top_t = bs[ms], bt[mt].clone()
            if trans["flipped"]:
                top_t[:, 1] = -top_t[:, 1]
                top_t[:, -1] = -top_t[:, -1] + np.pi
            top_t[:, :3] = box_torch_ops.rotation_points_single_angle(top_t[:, :3], trans["noise_rotation"], axis=2)
            top_t[:, -1] += trans["noise_rotation"]
            top_t[:, :-1] *= trans["noise_scale"]
            box_c, idx1, idx2, m1, m2 = self.nn_distance(top_s, top_t)
            if box_c is None:
                continue
            box_l = box_l + box_c
            score_s, score_t = torch.sigmoid(cs[ms][m1]), torch.sigmoid(ct[mt][m2][idx1])
            cls_l = cls_l + self.loss_score_consistency(score_s, score_t).mean()
            iou_s = (V(preds_stu, "iou_preds", 1)[b][ms][m1] + 1) * 0.5
            iou_t = (V(preds_tea, "iou_preds", 1)[b][mt][m2][idx1] + 1) * 0.5
            iou_l = iou_l + self.loss_iou_consistency(iou_s, iou_t).mean()
        return (box_l + cls_l + iou_l) / B

    def get_model_ema_loss(self, example, preds_dicts):
        """The teacher's own supervised terms on the un-augmented targets, for logging only (mg_head_sessd.py:810-890)."""
        t = self._supervised(preds_dicts[0], example["labels_raw"][0], example["reg_targets_raw"][0], example["anchors_raw"][0], False)
        loss = t["cls_loss_reduced"] + t["dir_loss"] + t["iou_pred_loss"]
        d = lambda v: v.detach().cpu()
        return {"loss_ema": [d(loss)], "cls_loss_reduced_ema": [d(t["cls_loss_reduced"]).mean()],
                "loc_loss_reduced_ema": [d(t["loc_loss_reduced"]).mean()], "dir_loss_reduced_ema": [d(t["dir_loss"])],
                "iou_pred_loss_ema": [d(t["iou_pred_loss"])], "loc_loss_elem_ema": [[d(e) for e in t["loc_loss_elem"]]],
                "cls_pos_loss_ema": [d(t["cls_pos_loss"])], "cls_neg_loss_ema": [d(t["cls_neg_loss"])],
                "num_pos_ema": [t["num_pos"]], "num_neg_ema": [t["num_neg"]]}

    def loss(self, example, preds_dicts, preds_ema, **kwargs):
        """MultiGroupHead.loss (mg_head_sessd.py:706-808) for the single-task car head: dict of one-element lists with
        loss = focal + ODIoU + direction + IoU-prediction (the smooth-L1 localisation loss is computed for the log only),
        consistency_loss to be weighted by the trainer (trainer_sessd.py:267), and the teacher's *_ema log terms."""
        assert len(preds_dicts) == 1, "single-task head (config.py tasks = [Car])"
        consistency = self.consistency_loss(preds_dicts, preds_ema, example)
        ema = self.get_model_ema_loss(example, preds_ema)
        t = self._supervised(preds_dicts[0], example["labels"][0], example["reg_targets"][0], example["anchors"][0], True)
        d = lambda v: v.detach().cpu()
        ret = {"loss": [t["cls_loss_reduced"] + t["ious_loss"] + t["dir_loss"] + t["iou_pred_loss"]],
               "cls_loss_reduced": [d(t["cls_loss_reduced"]).mean()], "loc_loss_reduced": [d(t["loc_loss_reduced"]).mean()],
               "dir_loss_reduced": [d(t["dir_loss"])], "iou_pred_loss": [d(t["iou_pred_loss"])], "consistency_loss": [consistency],
               "loc_loss_elem": [[d(e) for e in t["loc_loss_elem"]]], "cls_pos_loss": [d(t["cls_pos_loss"])],
               "cls_neg_loss": [d(t["cls_neg_loss"])], "ious_loss": [d(t["ious_loss"])], "num_pos": [t["num_pos"]], "num_neg": [t["num_neg"]]}
        ret.update(ema)
        return ret

    # ------------------------------------------------------------------ the same loss as ONE capacity-form device op
    def device_loss_covers(self, example, preds_dicts):
        """sessd_head_loss (csrc/head_loss.hip) covers the configuration of examples/second/configs/config.py: one task, one
        class, NormByNumPositives, sigmoid focal loss, codewise smooth-L1 without code weights, sin-encoded yaw, softmax
        direction classifier, head outputs on the HIP device."""
        p = preds_dicts[0]
        return (len(preds_dicts) == 1 and len(self.num_classes) == 1 and self.num_classes[0] == 1 and self.encode_background_as_zeros
                and self.loss_norm["type"] == "NormByNumPositives" and self.encode_rad_error_by_sin and self.use_direction_classifier
                and type(self.loss_cls).__name__ == "SigmoidFocalLoss" and self.loss_cls._alpha is not None
                and type(self.loss_reg).__name__ == "WeightedSmoothL1Loss" and self.loss_reg._codewise
                and getattr(self.loss_aux, "_logit_scale", 1.0) == 1.0 and self.box_n_dim == 7 and p["box_preds"].is_cuda
                and all(k in example for k in ("labels", "reg_targets", "anchors", "labels_raw", "reg_targets_raw", "anchors_raw")))

    @staticmethod
    def transformation_tensor(example, device):
        """(B, 5) float32 [flipped, cos, sin, noise_rotation, noise_scale] of example['transformation'] (the global augmentation
        recorded by Preprocess, mg_head_sessd.py:670-674); a captured iteration carries it as example['transformation_dev']."""
        import math
        rows = [[1.0 if t["flipped"] else 0.0, math.cos(float(t["noise_rotation"])), math.sin(float(t["noise_rotation"])),
                 float(t["noise_rotation"]), float(t["noise_scale"])] for t in example["transformation"]]
        return torch.tensor(rows, dtype=torch.float32).to(device)

    def loss_device(self, example, preds_dicts, preds_ema, consistency_weight=1.0, unit_grad=False, pos_capacity=None,
                    cons_capacity=2048):
        """MultiGroupHead.loss + trainer_sessd.py:267 as one device op: returns (total, record) -- total = loss + consistency_weight
        * consistency_loss as an autograd scalar of the student's head outputs (its backward hands over the kernel's gradients),
        record = the 64-float device log (ops.HEAD_LOSS_RECORD names its entries; nothing is read back here).
        consistency_weight: float or a device float32 scalar tensor (a captured iteration passes the tensor and refills it)."""
        p, q = preds_dicts[0], preds_ema[0]
        B = p["box_preds"].shape[0]
        dev = p["box_preds"].device
        A = p["box_preds"].numel() // (B * 7)
        labels = example["labels"][0]
        key = (B, A, str(dev), labels.dtype, pos_capacity, cons_capacity)
        run = getattr(self, "_device_loss", None)
        if run is None or run[0] != key:
            cfg = dict(pos_cls_weight=self.loss_norm["pos_cls_weight"], neg_cls_weight=self.loss_norm["neg_cls_weight"],
                       focal_alpha=self.loss_cls._alpha, focal_gamma=self.loss_cls._gamma or 0.0, smooth_l1_sigma=self.loss_reg._sigma,
                       cls_loss_weight=self.loss_cls._loss_weight, loc_loss_weight=self.loss_reg._loss_weight,
                       dir_loss_weight=self.loss_aux._loss_weight, direction_offset=self.direction_offset, score_thresh=self.thresh,
                       match_iou_thresh=0.7, center_range=self.post_center_range)
            run = (key, ops.HeadLoss(B, A, dev, labels.dtype == torch.int64, cfg, pos_capacity, cons_capacity))
            self._device_loss = run
        run = run[1]
        C = lambda t: t if t.is_contiguous() else t.contiguous()
        F = lambda t: C(t if t.dtype == torch.float32 else t.float())
        lab = lambda t: C(t if t.dtype in (torch.int32, torch.int64) else t.long())
        stu = dict(labels=lab(labels), reg_targets=F(example["reg_targets"][0]), anchors=F(example["anchors"][0]))
        tea = dict(box=F(q["box_preds"].detach()), cls=F(q["cls_preds"].detach()), dir=F(q["dir_cls_preds"].detach()),
                   iou=F(q["iou_preds"].detach()), labels=lab(example["labels_raw"][0]), reg_targets=F(example["reg_targets_raw"][0]),
                   anchors=F(example["anchors_raw"][0]))
        if tea["labels"].dtype != stu["labels"].dtype:
            tea["labels"] = tea["labels"].to(stu["labels"].dtype)
        anchors0 = F(example["anchors"][0][0])
        trans = example.get("transformation_dev")
        if trans is None:
            trans = self.transformation_tensor(example, dev)
        if not torch.is_tensor(consistency_weight):
            consistency_weight = torch.tensor(float(consistency_weight), dtype=torch.float32, device=dev)
        total, rec = ops.HeadLossFunction.apply(F(p["box_preds"]), F(p["cls_preds"]), F(p["dir_cls_preds"]), F(p["iou_preds"]), run, stu,
                                               tea, anchors0, F(trans), consistency_weight.reshape(()), unit_grad)
        return total, rec

    def record_to_dict(self, rec):
        """The device log record as the dict MultiGroupHead.loss returns (one host read of 64 floats; every N iterations)."""
        r = rec.detach().cpu()
        R = ops.HEAD_LOSS_RECORD
        T = lambda k: r[R[k]].clone()
        out = {k: [T(k)] for k in ("loss", "cls_loss_reduced", "loc_loss_reduced", "dir_loss_reduced", "iou_pred_loss", "cls_pos_loss",
                                   "cls_neg_loss", "ious_loss", "loss_ema", "cls_loss_reduced_ema", "loc_loss_reduced_ema",
                                   "dir_loss_reduced_ema", "iou_pred_loss_ema", "cls_pos_loss_ema", "cls_neg_loss_ema")}
        out["consistency_loss"] = [T("consistency_loss").reshape(1)]
        out["loc_loss_elem"] = [[v for v in r[R["loc_loss_elem"]]]]
        out["loc_loss_elem_ema"] = [[v for v in r[R["loc_loss_elem_ema"]]]]
        for k in ("num_pos", "num_neg", "num_pos_ema", "num_neg_ema"):
            out[k] = [T(k).long()]
        out["total"] = [T("total")]
        out["overflow"] = int(r[R["overflow"]])
        return out

    @torch.no_grad()
    def predict(self, example, preds_dicts, test_cfg, **kwargs):
        """Same contract as the reference: list (per sample) of dict(box3d_lidar, scores, label_preds, metadata).
        Decode, score filter, rotated NMS, frustum / direction / range filters all run on the device in ONE call."""
        assert len(preds_dicts) == 1 and len(self.num_classes) == 1, "multi-task heads are outside the SE-SSD car config"
        preds = preds_dicts[0]
        anchors = example["anchors"][0]
        B = anchors.shape[0]
        planar = preds.get("_planar")
        if planar is None:  # rebuilt from the NHWC tensors of the reference contract
            parts = [preds["box_preds"], preds["cls_preds"], preds["dir_cls_preds"], preds["iou_preds"]]
            planar = torch.cat([p.permute(0, 3, 1, 2) for p in parts], 1).contiguous()
        Bc, C, H, W = planar.shape
        head = planar.reshape(Bc, C, H * W).float().contiguous()
        anc = anchors.reshape(B, -1, self.box_n_dim).float().contiguous()
        frustum = None
        calib = example.get("calib")
        if calib is not None and "frustum" in calib:
            frustum = calib["frustum"].to(head.device).double().contiguous()
        nms = test_cfg["nms"] if isinstance(test_cfg, dict) else test_cfg.nms
        out = ops.predict(head, anc, frustum, float(self.thresh), int(nms["nms_pre_max_size"]),
                          int(nms["nms_post_max_size"]), float(nms["nms_iou_threshold"]), self.post_center_range,
                          float(self.direction_offset))
        counts = out["count"].cpu().tolist()  # the single host read of the whole predict path
        meta = example.get("metadata", [None] * B)
        ret = []
        for b in range(B):
            n = int(counts[b])
            ret.append({"box3d_lidar": out["box"][b, :n], "scores": out["score"][b, :n],
                        "label_preds": out["label"][b, :n].long(), "metadata": meta[b]})
        return ret

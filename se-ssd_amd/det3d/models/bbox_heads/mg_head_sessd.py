"""mirrors det3d/models/bbox_heads/mg_head_sessd.py: Head (:195-230), MultiGroupHead ctor (:379-493),
forward (:518-523), predict / get_task_detections (:893-1057) -- inference surface only."""
import logging

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
        key = tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in convs)
        if self._packed is None or self._packed[0] != key:
            w = torch.cat([c.weight.detach() for c in convs], 0)
            b = torch.cat([c.bias.detach() for c in convs], 0).float().contiguous()
            self._packed = (key, ops.pack_conv2d(w), b, [c.out_channels for c in convs])
        _, pc, b, split = self._packed
        return ops.conv2d(x.float().contiguous(), pc, None, b, False), split

    def forward(self, x):
        if self.training:  # autograd path: the four 1x1 convs as torch modules (mg_head_sessd.py:217-230)
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

    def init_weights(self, pretrained=None):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        return [task(x) for task in self.tasks]

    def loss(self, example, preds_dicts, preds_ema=None, **kwargs):
        raise NotImplementedError("SE-SSD training step: SURVEY.md section 8f row 1 (not part of the inference hot path)")

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

"""Generates tests/golden/head_loss_ref.npz by running the REFERENCE's MultiGroupHead.loss from source on CPU:
    det3d/models/bbox_heads/mg_head_sessd.py  loss :706-808, get_model_ema_loss :810-890, consistency_loss :618-704,
                                             nn_distance :573-607, prepare_loss_weights :525-571 and the module-level
                                             helpers :27-77
together with the reference's own losses.py, odious.py, box_torch_ops.py, box_coders.py, iou3d_utils.py / utils.py.
What cannot run here is substituted, and only that:
  * `iou3d_cuda` (CUDA extension)      -> a stub whose entry points fill the caller's output tensor from oracle/iou3d.c
                                          (bit-equal to the compiled reference iou3d_cpu.cpp, tests/test_oracle_golden.py);
                                          the reference's Python wrappers in iou3d_utils.py run unchanged on top of it
  * `.cuda()` / `torch.cuda.FloatTensor` -> identity / CPU FloatTensor (there is no GPU in the build container)
  * `MultiGroupHead.__init__`           -> bypassed (it builds conv layers and calls .cuda()); the attributes loss() reads are
                                          set from the same constructor arguments config.py passes
  * registry decorators, logging / checkpoint helpers, matplotlib, numba -> inert stubs (make_golden.install_stubs)
Nothing from the reference is copied into the repository; only input / output vectors are stored.
Run in the build container only:  python tests/golden/make_golden_head_loss.py"""
import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd"))
sys.path.insert(0, HERE)


def mod(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        sys.modules[name] = m
    if not hasattr(m, "__path__"):
        m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def load_as(relpath, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def install(oracle_capi):
    import make_golden as MG
    MG.install_stubs()

    class _Reg:
        @staticmethod
        def register_module(obj):
            return obj

    # ---- CUDA extension stub: same calling convention (fills the caller's tensor, returns 1)
    def _np(t):
        return np.ascontiguousarray(t.detach().cpu().numpy(), np.float32)

    def boxes_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(oracle_capi.boxes_overlap_bev(_np(a), _np(b))))
        return 1

    def boxes_iou_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(oracle_capi.boxes_iou_bev(_np(a), _np(b))))
        return 1

    def boxes_aligned_overlap_bev_gpu(a, b, out):
        out.copy_(torch.from_numpy(oracle_capi.boxes_aligned_overlap_bev(_np(a), _np(b))).view(-1, 1))
        return 1

    mod("iou3d_cuda", boxes_overlap_bev_gpu=boxes_overlap_bev_gpu, boxes_iou_bev_gpu=boxes_iou_bev_gpu,
        boxes_aligned_overlap_bev_gpu=boxes_aligned_overlap_bev_gpu)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    # ---- reference sources that are plain torch / numpy
    mod("det3d"); mod("det3d.core"); mod("det3d.core.bbox"); mod("det3d.core.iou3d"); mod("det3d.models"); mod("det3d.torchie")
    load_as("det3d/core/bbox/geometry.py", "det3d.core.bbox.geometry")
    bnp = load_as("det3d/core/bbox/box_np_ops.py", "det3d.core.bbox.box_np_ops")
    bto = load_as("det3d/core/bbox/box_torch_ops.py", "det3d.core.bbox.box_torch_ops")
    mod("det3d.core.bbox", box_np_ops=bnp, box_torch_ops=bto)
    utils = load_as("det3d/core/iou3d/utils.py", "det3d.core.iou3d.utils")
    mod("det3d.core.iou3d", utils=utils)
    iu = load_as("det3d/core/iou3d/iou3d_utils.py", "det3d.core.iou3d.iou3d_utils")
    mod("det3d.core.iou3d", iou3d_utils=iu)
    # losses under a stub package (relative imports)
    for name in ("refpkg", "refpkg.models", "refpkg.models.losses", "refpkg.models.bbox_heads"):
        mod(name)
    mod("refpkg.models.registry", LOSSES=_Reg(), HEADS=_Reg())
    lu = load_as("det3d/models/losses/utils.py", "refpkg.models.losses.utils")
    ll = load_as("det3d/models/losses/losses.py", "refpkg.models.losses.losses")
    od = load_as("det3d/models/losses/odious.py", "det3d.models.losses.odious")
    mod("det3d.models.losses", metrics=types.ModuleType("metrics"), odious=od)
    mod("refpkg.models.losses", accuracy=None)

    def build_loss(cfg):
        cfg = dict(cfg)
        return getattr(ll, cfg.pop("type"))(**cfg)

    mod("det3d.models.builder", build_loss=build_loss)
    mod("refpkg.models.builder", build_loss=build_loss)
    mod("refpkg.models", builder=sys.modules["refpkg.models.builder"])
    mod("det3d.torchie.cnn", constant_init=None, kaiming_init=None)
    mod("det3d.torchie.trainer", load_checkpoint=None)
    mod("det3d.core.sampler"); mod("det3d.core.sampler.preprocess")
    head = load_as("det3d/models/bbox_heads/mg_head_sessd.py", "refpkg.models.bbox_heads.mg_head_sessd")
    return head, ll, od, bto, build_loss


def make_case(seed=0, B=2, A=1408):
    """Synthetic head outputs / targets of one iteration: A anchors per sample (the loss does not depend on the grid shape),
    a handful of positives with small regression targets, teacher outputs close to the student's."""
    rng = np.random.RandomState(seed)
    anchors = np.zeros((B, A, 7), np.float32)
    anchors[..., 0] = rng.uniform(0, 70, (B, A)); anchors[..., 1] = rng.uniform(-40, 40, (B, A)); anchors[..., 2] = -1.0
    anchors[..., 3:6] = [1.6, 3.9, 1.56]
    anchors[..., 6] = rng.choice([0.0, 1.57], (B, A))
    anchors[1] = anchors[0]  # consistency_loss decodes every sample with the anchors of sample 0 (:649-650)
    labels = np.zeros((B, A), np.int64)
    for b in range(B):
        pos = rng.choice(A, 14, replace=False)
        labels[b, pos] = 1
        labels[b, rng.choice(A, 30, replace=False)] = -1
        labels[b, pos] = 1
    reg_targets = np.zeros((B, A, 7), np.float32)
    reg_targets[labels > 0] = rng.normal(0, 0.15, (int((labels > 0).sum()), 7)).astype(np.float32)
    out = dict(anchors=anchors, labels=labels, reg_targets=reg_targets)
    for who, noise in (("stu", 0.0), ("tea", 0.04)):
        box = rng.normal(0, 0.05, (B, A, 7)).astype(np.float32)
        box[labels > 0] = reg_targets[labels > 0] + rng.normal(0, 0.08, (int((labels > 0).sum()), 7)).astype(np.float32)
        cls = rng.normal(-3.0, 1.0, (B, A, 1)).astype(np.float32)
        cls[labels > 0] = rng.normal(1.5, 0.7, (int((labels > 0).sum()), 1)).astype(np.float32)
        if who == "tea":
            box = out["box_stu"] + rng.normal(0, noise, box.shape).astype(np.float32)
            cls = out["cls_stu"] + rng.normal(0, 0.2, cls.shape).astype(np.float32)
        out["box_" + who] = box
        out["cls_" + who] = cls
        out["dir_" + who] = rng.normal(0, 1, (B, A, 2)).astype(np.float32)
        out["iou_" + who] = rng.uniform(-1, 1, (B, A, 1)).astype(np.float32)
    # teacher-side (raw) targets: same anchors, slightly different assignment
    out["labels_raw"] = labels.copy()
    out["reg_targets_raw"] = (reg_targets + (labels > 0)[..., None] * rng.normal(0, 0.02, reg_targets.shape)).astype(np.float32)
    # sample 0: (almost) identity augmentation, so teacher and student boxes match; sample 1: flipped + rotated (no matches:
    # exercises the early `continue` of consistency_loss)
    out["trans"] = [dict(flipped=False, noise_rotation=0.002, noise_scale=1.001)] + \
                   [dict(flipped=True, noise_rotation=float(rng.uniform(-0.3, 0.3)), noise_scale=float(rng.uniform(0.95, 1.05)))
                    for b in range(1, B)]
    return out


def main():
    assert os.path.isdir(REF)
    warnings.filterwarnings("ignore")
    from oracle import capi
    head, ll, od, bto, build_loss = install(capi)
    H = head.MultiGroupHead
    h = object.__new__(H)
    torch.nn.Module.__init__(h)

    class Coder:  # what config.py:60 builds: GroundBox3dCoderTorch(linear_dim=False, encode_angle_vector=False)
        n_dim = 7
        code_size = 7

        @staticmethod
        def decode_torch(enc, anchors):
            return bto.second_box_decode(enc, anchors, False, False)

    h.box_coder = Coder()
    h.num_classes = [1]
    h.box_n_dim = 7
    h.encode_rad_error_by_sin = True
    h.use_direction_classifier = True
    h.direction_offset = 0.0
    h.loss_norm = dict(type="NormByNumPositives", pos_cls_weight=1.0, neg_cls_weight=1.0)
    h.loss_cls = build_loss(dict(type="SigmoidFocalLoss", alpha=0.25, gamma=2.0, loss_weight=1.0))
    h.loss_reg = build_loss(dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=[1.0] * 7, codewise=True, loss_weight=2.0))
    h.loss_aux = build_loss(dict(type="WeightedSoftmaxClassificationLoss", name="direction_classifier", loss_weight=0.2))
    h.loss_iou_pred = build_loss(dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=None, codewise=True, loss_weight=1.0))
    h.loss_iou_consistency = build_loss(dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=None, codewise=True, loss_weight=1.0))
    h.loss_score_consistency = build_loss(dict(type="WeightedSmoothL1Loss", sigma=3.0, code_weights=None, codewise=True, loss_weight=1.0))
    h.loss_size_consistency = torch.nn.MSELoss(reduction="mean")
    h.loss_dir_consistency = torch.nn.MSELoss(reduction="mean")
    h.odiou_3d_loss = od.odiou_3D()
    h.post_center_range = torch.tensor([0, -40.0, -5.0, 70.4, 40.0, 5.0])

    c = make_case()
    T = torch.from_numpy
    B = c["labels"].shape[0]
    example = dict(anchors=[T(c["anchors"])], anchors_raw=[T(c["anchors"])], labels=[T(c["labels"])], reg_targets=[T(c["reg_targets"])],
                   labels_raw=[T(c["labels_raw"])], reg_targets_raw=[T(c["reg_targets_raw"])], metadata=[{}] * B,
                   transformation=c["trans"], annos_raw=[None] * B)
    stu = {k: T(c[k + "_stu"]).clone().requires_grad_(True) for k in ("box", "cls", "dir", "iou")}
    preds = [dict(box_preds=stu["box"], cls_preds=stu["cls"], dir_cls_preds=stu["dir"], iou_preds=stu["iou"])]
    preds_ema = [dict(box_preds=T(c["box_tea"]), cls_preds=T(c["cls_tea"]), dir_cls_preds=T(c["dir_tea"]), iou_preds=T(c["iou_tea"]))]
    ret = h.loss(example, preds, preds_ema)
    loss = ret["loss"][0]
    cons = ret["consistency_loss"][0]
    total = loss + 1.0 * cons.sum()  # trainer_sessd.py:267: loss + consistency_weight * consistency_loss
    total.backward()
    out = {k: v for k, v in c.items() if k != "trans"}
    out["trans_flipped"] = np.array([t["flipped"] for t in c["trans"]])
    out["trans_rot"] = np.array([t["noise_rotation"] for t in c["trans"]])
    out["trans_scale"] = np.array([t["noise_scale"] for t in c["trans"]])
    for k in ("loss", "cls_loss_reduced", "loc_loss_reduced", "dir_loss_reduced", "iou_pred_loss", "ious_loss", "cls_pos_loss",
              "cls_neg_loss", "loss_ema", "cls_loss_reduced_ema", "iou_pred_loss_ema", "dir_loss_reduced_ema"):
        out["ret_" + k] = np.array(float(ret[k][0].detach() if torch.is_tensor(ret[k][0]) else ret[k][0]))
    out["ret_consistency_loss"] = cons.detach().numpy()
    out["ret_num_pos"] = np.array(int(ret["num_pos"][0]))
    for k in stu:
        out["grad_" + k] = stu[k].grad.numpy() if stu[k].grad is not None else np.zeros_like(c[k + "_stu"])
    np.savez_compressed(os.path.join(HERE, "head_loss_ref.npz"), **out)
    print("head loss golden written:", {k: (float(v) if v.ndim == 0 else v.shape) for k, v in out.items() if k.startswith("ret_")})


if __name__ == "__main__":
    main()

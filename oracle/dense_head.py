"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the dense BEV neck and heads in plain torch float32.

Follows det3d/models/necks/rpn_v1.py:135-235 (SSFA layers + forward) and
det3d/models/bbox_heads/mg_head_sessd.py:202-230 (Head: four 1x1 convs, NHWC outputs), operating on a state_dict
with the reference's key names (neck.bottom_up_block_0.1.weight, ..., bbox_head.tasks.0.conv_box.weight).
These are the very torch ops the reference calls (F.conv2d / conv_transpose2d / batch_norm / softmax), run on CPU."""
import torch
import torch.nn.functional as F

EPS = 1e-3  # rpn_v1.py:131 norm_cfg = dict(type="BN", eps=1e-3, momentum=0.01)


_TRAIN = [False]  # set by ssfa_forward(training=...): batch statistics instead of the running ones


def _bn(x, sd, p):
    if _TRAIN[0]:
        return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.0, EPS)
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, EPS)


def _cbr(x, sd, p, ci, bi, stride=1, pad=1, relu=True):
    x = F.conv2d(x, sd["%s.%d.weight" % (p, ci)], None, stride=stride, padding=pad)
    x = _bn(x, sd, "%s.%d" % (p, bi))
    return torch.relu(x) if relu else x


def ssfa_forward(x, sd, prefix="neck.", training=False):
    try:
        _TRAIN[0] = bool(training)
        return _ssfa_forward(x, sd, prefix)
    finally:
        _TRAIN[0] = False


def _ssfa_forward(x, sd, prefix):
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    x0 = _cbr(x, sd, "bottom_up_block_0", 1, 2)      # ZeroPad2d(1) + unpadded 3x3 == padding 1
    x0 = _cbr(x0, sd, "bottom_up_block_0", 4, 5)
    x0 = _cbr(x0, sd, "bottom_up_block_0", 7, 8)
    x1 = _cbr(x0, sd, "bottom_up_block_1", 0, 1, stride=2)
    x1 = _cbr(x1, sd, "bottom_up_block_1", 3, 4)
    x1 = _cbr(x1, sd, "bottom_up_block_1", 6, 7)
    t0 = _cbr(x0, sd, "trans_0", 0, 1, pad=0)
    t1 = _cbr(x1, sd, "trans_1", 0, 1, pad=0)

    def deconv(p):
        y = F.conv_transpose2d(t1, sd[p + ".0.weight"], None, stride=2, padding=1, output_padding=1)
        return torch.relu(_bn(y, sd, p + ".1"))

    m0 = deconv("deconv_block_0") + t0
    m1 = deconv("deconv_block_1")
    o0 = _cbr(m0, sd, "conv_0", 0, 1)
    o1 = _cbr(m1, sd, "conv_1", 0, 1)
    w0 = _cbr(o0, sd, "w_0", 0, 1, pad=0, relu=False)
    w1 = _cbr(o1, sd, "w_1", 0, 1, pad=0, relu=False)
    w = torch.softmax(torch.cat([w0, w1], dim=1), dim=1)
    return o0 * w[:, 0:1] + o1 * w[:, 1:]


def head_forward(x, sd, prefix="bbox_head.tasks.0."):
    out = {}
    for name, key in (("box_preds", "conv_box"), ("cls_preds", "conv_cls"), ("dir_cls_preds", "conv_dir"), ("iou_preds", "conv_iou")):
        y = F.conv2d(x, sd[prefix + key + ".weight"], sd[prefix + key + ".bias"])
        out[name] = y.permute(0, 2, 3, 1).contiguous()
    return out

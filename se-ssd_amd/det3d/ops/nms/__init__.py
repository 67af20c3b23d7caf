from det3d.ops.nms.nms_gpu import nms_gpu, rotate_iou_gpu, rotate_iou_gpu_eval, rotate_nms_gpu  # noqa: F401

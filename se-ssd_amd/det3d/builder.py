"""mirrors det3d/builder.py:409-442 (build_box_coder is called by config.py at import time)."""
from det3d.core.bbox.box_coders import GroundBox3dCoderTorch


def build_box_coder(box_coder_config):
    box_coder_type = box_coder_config["type"]
    cfg = box_coder_config
    if box_coder_type == "ground_box3d_coder":
        return GroundBox3dCoderTorch(cfg["linear_dim"], cfg["encode_angle_vector"], n_dim=cfg.get("n_dim", 9),
                                     norm_velo=cfg.get("norm_velo", False))
    raise ValueError("unknown box_coder type (only ground_box3d_coder is on the SE-SSD hot path)")

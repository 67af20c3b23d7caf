"""mirrors det3d/builder.py:409-442 (build_box_coder is called by config.py at import time)."""
from det3d.core.bbox.box_coders import GroundBox3dCoderTorch


def build_box_coder(box_coder_config):
    box_coder_type = box_coder_config["type"]
    cfg = box_coder_config
    if box_coder_type == "ground_box3d_coder":
        return GroundBox3dCoderTorch(cfg["linear_dim"], cfg["encode_angle_vector"], n_dim=cfg.get("n_dim", 9),
                                     norm_velo=cfg.get("norm_velo", False))
    raise ValueError("unknown box_coder type (only ground_box3d_coder is on the SE-SSD hot path)")


def _cfg_get(cfg, key, default=None):
    return cfg[key] if key in cfg else default


def build_db_preprocess(db_prep_config, logger=None):
    """one database filter from dict(filter_by_difficulty=[...]) / dict(filter_by_min_num_points={cls: n}) (builder.py:67-77)."""
    from det3d.core.sampler import preprocess as prep
    if "filter_by_difficulty" in db_prep_config:
        return prep.DBFilterByDifficulty(db_prep_config["filter_by_difficulty"], logger=logger)
    if "filter_by_min_num_points" in db_prep_config:
        return prep.DBFilterByMinNumPoint(db_prep_config["filter_by_min_num_points"], logger=logger)
    raise ValueError("unknown database prep type")


def build_dbsampler(cfg, logger=None, db_infos=None):
    """the GT-AUG sampler of config.py:128-143 (builder.py:378-406). `db_infos` may be handed in instead of being unpickled
    from cfg.db_info_path (with context enlargement the reference switches to the *_enlarged_train database file)."""
    import pickle
    from det3d.core.sampler import preprocess as prep
    from det3d.core.sampler.sample_ops_v2 import DataBaseSamplerV2
    prepor = prep.DataBasePreprocessor([build_db_preprocess(c, logger=logger) for c in cfg["db_prep_steps"]])
    grot = list(cfg["global_random_rotation_range_per_object"])
    ctx = _cfg_get(cfg, "gt_aug_with_context", -1.0)
    if db_infos is None:
        path = cfg["db_info_path"]
        if ctx > 0.0:
            path = path[:-17] + "dbinfos_enlarged_train.pkl"
        with open(path, "rb") as f:
            db_infos = pickle.load(f)
    return DataBaseSamplerV2(db_infos, cfg["sample_groups"], prepor, cfg["rate"], grot if grot else None, logger=logger,
                             gt_random_drop=_cfg_get(cfg, "gt_random_drop", -1.0), gt_aug_with_context=ctx,
                             gt_aug_similar_type=_cfg_get(cfg, "gt_aug_similar_type", False))

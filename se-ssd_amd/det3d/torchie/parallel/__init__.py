from .collate import collate_kitti, example_to_device  # noqa: F401

from det3d.utils import Registry

DATASETS = Registry("dataset")
PIPELINES = Registry("pipeline")

from .compose import Compose  # noqa: F401
from .preprocess import AssignTarget, Preprocess, Voxelization  # noqa: F401
from .formating import Reformat  # noqa: F401
from .loading import LoadPointCloudAnnotations, LoadPointCloudFromFile  # noqa: F401

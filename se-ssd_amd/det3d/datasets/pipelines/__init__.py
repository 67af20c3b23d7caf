from .preprocess import AssignTarget, Voxelization  # noqa: F401
from .formating import Reformat  # noqa: F401

from .preprocess import Voxelization  # noqa: F401
from .formating import Reformat  # noqa: F401

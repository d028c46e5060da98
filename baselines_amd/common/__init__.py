from .math_util import explained_variance   # noqa: F401
from .misc_util import set_global_seeds     # noqa: F401

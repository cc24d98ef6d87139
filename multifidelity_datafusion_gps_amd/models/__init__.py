from .GPDF import GPDF
from .GPDFC import GPDFC
from .NARGP import NARGP

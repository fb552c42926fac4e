"""wan2gp_amd -- MI355X-native Wan 2.1/2.2 denoise hot path (drop-in behind the reference's
WanModel.forward / pay_attention / scheduler / generate() surfaces).  See DESIGN.md."""
from .lib import WanHipError, load  # noqa: F401

__all__ = ["WanHipError", "load"]

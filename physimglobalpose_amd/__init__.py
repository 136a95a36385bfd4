"""physimglobalpose_amd -- MI355X-native pose-hypothesis scoring (the LCP hot path of
cmitash/PhysimGlobalPose) behind the C ABI of include/pgp.h.  See DESIGN.md."""
from .scorer import LcpScorer, MultiGpuScorer, PGP_MODE_PLAIN, PGP_MODE_WEIGHTED  # noqa: F401

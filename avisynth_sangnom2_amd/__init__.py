"""MI355X-native SangNom2 edge-directed interpolation hot path behind the reference's plugin
interface.  The kernels live in csrc/ (HIP, gfx950) and are reached through the C ABI declared in
include/sangnom_hip.h; this package is the thin host-side mirror used by tests and bench.py."""
from .capi import LIB_PATH, build, load  # noqa: F401
from .filter import ClipFormat, SangNom, SangNom2, SangNomAA, SangNomAAHost, SangNomError, clip_format, pin_host_array, unpin_host_array  # noqa: F401

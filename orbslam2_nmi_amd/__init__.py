"""orbslam2_nmi_amd -- MI355X (gfx950) implementation of the NMI pose-candidate scoring path of
gsanya/orbslam2_NMI behind the C ABI of include/nmi_hip.h.  See DESIGN.md / INTEGRATION.md."""
from .capi import (MODE_ENMI, MODE_SUC, NmiContext, NmiError, NmiLevel, NmiStream, NmiTexture, key_pack, key_unpack, library_path,  # noqa: F401
                   load_library)

__all__ = ["MODE_ENMI", "MODE_SUC", "NmiContext", "NmiError", "NmiLevel", "NmiStream", "NmiTexture", "key_pack", "key_unpack", "library_path", "load_library"]

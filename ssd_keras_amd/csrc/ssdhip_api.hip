// ssdhip_api.hip -- ABI version and error strings of libssdhip.so.
#include "ssdhip.h"

extern "C" int ssdhip_abi_version(void) { return SSDHIP_ABI_VERSION; }

extern "C" const char* ssdhip_strerror(int rc) {
    switch (rc) {
        case SSDHIP_OK: return "ok";
        case SSDHIP_E_BADARG: return "bad argument (inconsistent sizes or unsupported option combination)";
        case SSDHIP_E_WORKSPACE: return "workspace missing or too small (see *_workspace_bytes)";
        case SSDHIP_E_LAUNCH: return "HIP launch failed (hipGetLastError)";
        default: return "unknown ssdhip error code";
    }
}

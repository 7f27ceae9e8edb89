// Version / error-string entry points of libdenet_hip.so.
#include "denet_common.hpp"

extern "C" const char* denet_version(void) { return "denet_hip 0.1 (gfx950)"; }

extern "C" const char* denet_strerror(int code) {
    switch (code) {
        case DENET_OK: return "ok";
        case DENET_E_BADARG: return "bad argument (null pointer or non-positive dimension)";
        case DENET_E_UNSUPPORTED: return "shape or dtype outside what the kernels support";
        case DENET_E_LAUNCH: return "kernel launch failed";
        case DENET_E_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

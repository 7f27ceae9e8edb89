// Version / error-string entry points of libdenet_hip.so.
#include "denet_common.hpp"
#include <mutex>
#include <unordered_map>

int denet_ensure_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return DENET_OK;
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> seen;
    std::lock_guard<std::mutex> lock(mu);
    auto it = seen.find(kernel);
    if (it != seen.end() && it->second >= bytes) return DENET_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return DENET_E_LAUNCH;
    seen[kernel] = bytes;
    return DENET_OK;
}

extern "C" const char* denet_version(void) { return "denet_hip 0.1 (gfx950)"; }

extern "C" const char* denet_strerror(int code) {
    switch (code) {
        case DENET_OK: return "ok";
        case DENET_E_BADARG: return "bad argument (null pointer or non-positive dimension)";
        case DENET_E_UNSUPPORTED: return "shape or dtype outside what the kernels support";
        case DENET_E_LAUNCH: return "kernel launch failed";
        case DENET_E_WORKSPACE: return "workspace too small";
        case DENET_E_STATIC_LIMIT:
            return "problem needs more scratch than the static buffer of the no-allocation entry point holds "
                   "(denet_proto_extend_bwd_static_limit); use the workspace variant";
        default: return "unknown error";
    }
}
